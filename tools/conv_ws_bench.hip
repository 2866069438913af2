// EXPERIMENT (end of round 2, groundwork for the next one): weights-stationary form of the fp16-input vocoder convolution.
// One persistent workgroup per CU keeps the weights of its 64-channel tile for ALL input channels and taps in LDS (115 KB for
// k = 7 at 128 channels) and streams activation tiles: the 16-channel chunks of the NEXT tile are requested into the registers
// that the current tile's chunk has just left for LDS, i.e. a whole tile (8 chunks) of lookahead instead of none.
// Compared bit for bit and timed against conv1d_mfma_f16_kernel on the 128-channel stage of a 64-utterance batch.
// Build + run on the GPU box:  bash tools/conv_ws_bench.sh
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../auralis_amd/csrc/vocoder_kernels.hip"

using namespace aur;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int KS, int DIL, int NCH, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_ws_kernel(ConvArgs a, int tiles_t, int n_mtiles, int wgs_per_mtile) {
    constexpr int MT = 64, WM = 2, WN = 2, NTW = 64, NT = 256, HALO = (KS - 1) * DIL, XROW = NT + HALO, RS = 24;
    constexpr int XI = (XROW + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16* ws = lds;                                        // [NCH][KS][2 halves of k][MT][8]
    _Float16* xs = lds + (size_t)NCH * KS * 2 * MT * 8;        // [2][XROW][RS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mtile = blockIdx.x % n_mtiles, wg = blockIdx.x / n_mtiles;
    {   // the tile's weights, once: packed source [chunk][tap][co][16] -> planes [chunk][tap][k half][co][8]
        const uint4* src = reinterpret_cast<const uint4*>(a.wp16) + (long)mtile * NCH * KS * MT * 2;
        uint4* dst = reinterpret_cast<uint4*>(ws);
        for (int p = tid; p < NCH * KS * MT * 2; p += 256) {
            const int h = p & 1, co = (p >> 1) % MT, cj = (p >> 1) / MT;
            dst[(cj * 2 + h) * MT + co] = src[p];
        }
    }
    const int total = a.B * tiles_t;
    h16x8 xr[NCH][XI][2];
    auto load_chunk = [&](int tile, auto cc) {
        constexpr int c = decltype(cc)::value;
        const int b = tile / tiles_t, q0 = (tile - b * tiles_t) * NT;
        const int len_in = a.base_len[b] * a.len_mul;
        const _Float16* xhb = reinterpret_cast<const _Float16*>(a.x) + (long)b * a.x_bstride;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int t = q0 - a.padl + tid + it * 256;
            const int tc = min(max(t, 0), len_in - 1);
            const _Float16* p = xhb + ((long)c * a.x_stride + tc) * 16;
            xr[c][it][0] = *reinterpret_cast<const h16x8*>(p);
            xr[c][it][1] = *reinterpret_cast<const h16x8*>(p + 8);
        }
    };
    int tile = wg;
    if (tile < total) static_for<0, NCH>([&](auto C) { load_chunk(tile, C); });
    __syncthreads();
    for (; tile < total; tile += wgs_per_mtile) {
        const int b = tile / tiles_t, q0 = (tile - b * tiles_t) * NT;
        const int len_in = a.base_len[b] * a.len_mul;
        const int n_q = len_in;
        const int next = tile + wgs_per_mtile;
        f32x16 acc[WM][WN];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        static_for<0, NCH>([&](auto C) {
            constexpr int c = decltype(C)::value;
            _Float16* xb = xs + (c & 1) * XROW * RS;
#pragma unroll
            for (int it = 0; it < XI; ++it) {
                const int i = tid + it * 256;
                const int t = q0 - a.padl + i;
                const bool ok = (t >= 0 && t < len_in);
                const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                if (i < XROW) {
                    *reinterpret_cast<h16x8*>(&xb[i * RS]) = ok ? xr[c][it][0] : zero;
                    *reinterpret_cast<h16x8*>(&xb[i * RS + 8]) = ok ? xr[c][it][1] : zero;
                }
            }
            if (next < total) load_chunk(next, C);   // the registers of chunk c are free again: a whole tile of lookahead
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            const _Float16* xbase = &xb[(wv * NTW + l31) * RS + 8 * hi];
            const _Float16* wbase = &ws[((c * KS * 2 + hi) * MT + l31) * 8];
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                h16x8 av[WM], bv[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) av[m] = *reinterpret_cast<const h16x8*>(wbase + (j * 2 * MT + m * 32) * 8);
#pragma unroll
                for (int n = 0; n < WN; ++n) bv[n] = *reinterpret_cast<const h16x8*>(xbase + (j * DIL + n * 32) * RS);
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        });
        if (q0 < n_q) conv_epilogue<WM, WN, MT, NTW>(a, acc, b, mtile, q0, wv, l31, hi, len_in, n_q);
    }
}

static void* dalloc_bytes(size_t n) {
    void* p;
    HIP_CHECK(hipMalloc(&p, n));
    return p;
}

template <int KS, int DIL>
static void run(const char* what, int B, int C, int len_mul, int base) {
    hipStream_t st = 0;
    const long L = (long)base * len_mul;
    const size_t n_elem = (size_t)B * C * L;
    std::vector<_Float16> hx(n_elem);
    unsigned s = 12345u;
    for (size_t i = 0; i < n_elem; ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = ((float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f);
        hx[i] = (_Float16)(v > 0 ? v : 0.1f * v);
    }
    _Float16* x = (_Float16*)dalloc_bytes(n_elem * 2);
    HIP_CHECK(hipMemcpy(x, hx.data(), n_elem * 2, hipMemcpyHostToDevice));
    const size_t n_w = (size_t)C * C * KS;
    std::vector<_Float16> hw(n_w);
    for (size_t i = 0; i < n_w; ++i) {
        s = s * 1664525u + 1013904223u;
        hw[i] = (_Float16)(0.02f * ((float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f));
    }
    _Float16* w = (_Float16*)dalloc_bytes(n_w * 2);
    HIP_CHECK(hipMemcpy(w, hw.data(), n_w * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(C);
    for (int i = 0; i < C; ++i) hb[i] = 0.01f * (i % 7 - 3);
    float* bias = (float*)dalloc_bytes(C * 4);
    HIP_CHECK(hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice));
    std::vector<int> hl(B, base);
    hl[B - 1] = base - 7;   // one ragged utterance
    int* blen = (int*)dalloc_bytes(B * 4);
    HIP_CHECK(hipMemcpy(blen, hl.data(), B * 4, hipMemcpyHostToDevice));
    _Float16* o1 = (_Float16*)dalloc_bytes(n_elem * 2);
    _Float16* o2 = (_Float16*)dalloc_bytes(n_elem * 2);
    HIP_CHECK(hipMemset(o1, 0, n_elem * 2));
    HIP_CHECK(hipMemset(o2, 0, n_elem * 2));

    ConvArgs a{};
    a.x = reinterpret_cast<const float*>(x);
    a.wp16 = w;
    a.bias = bias;
    a.base_len = blen;
    a.len_mul = len_mul;
    a.Cin = a.Mtot = a.Cout = C;
    a.x_stride = a.o_stride = L;
    a.x_bstride = a.o_bstride = (long)C * L;
    a.padl = (KS - 1) * DIL / 2;
    a.slope = 1.0f;
    a.max_len = (int)L;
    a.B = B;
    a.x_f16 = 1;
    a.out_act_f16 = 1;
    a.out_slope = 0.1f;

    auto time_ms = [&](auto&& f) {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        f();
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < 5; ++i) f();
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 5;
    };
    a.out = reinterpret_cast<float*>(o1);
    const float t_ref = time_ms([&] { launch_conv1d_f16(a, KS, DIL, st); });

    constexpr int NCH = 8;
    if (C != 16 * NCH) {
        printf("%s: weights-stationary variant built for %d channels only\n", what, 16 * NCH);
        return;
    }
    a.out = reinterpret_cast<float*>(o2);
    constexpr int XROW = 256 + (KS - 1) * DIL;
    const size_t lds = (size_t)NCH * KS * 2 * 64 * 8 * 2 + (size_t)2 * XROW * 24 * 2;
    constexpr int OCC = (KS <= 3) ? 2 : 1;   // workgroups per CU that fit the LDS
    const int tiles_t = (int)((L + 255) / 256), n_mtiles = C / 64, wgs_per_mtile = 256 * OCC / n_mtiles;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<KS, DIL, NCH, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float t_ws = time_ms([&] {
        hipLaunchKernelGGL((conv_ws_kernel<KS, DIL, NCH, OCC>), dim3(n_mtiles * wgs_per_mtile), dim3(256), lds, st, a, tiles_t, n_mtiles, wgs_per_mtile);
    });
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<_Float16> r1(n_elem), r2(n_elem);
    HIP_CHECK(hipMemcpy(r1.data(), o1, n_elem * 2, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(r2.data(), o2, n_elem * 2, hipMemcpyDeviceToHost));
    const long diff = memcmp(r1.data(), r2.data(), n_elem * 2) ? [&] {
        long d = 0;
        for (size_t i = 0; i < n_elem; ++i) d += memcmp(&r1[i], &r2[i], 2) != 0;
        return d;
    }() : 0;
    const double flop = 2.0 * C * C * KS * (double)L * B, bytes = 4.0 * n_elem;
    printf("%s k=%d d=%d C=%d L=%ld B=%d: current %.3f ms (%.2f TB/s, %.0f TFLOP/s)  weights-stationary %.3f ms (%.2f TB/s, %.0f TFLOP/s, LDS %zu B)  differing outputs %ld of %zu\n",
           what, KS, DIL, C, L, B, t_ref, bytes / t_ref / 1e9, flop / t_ref / 1e9, t_ws, bytes / t_ws / 1e9, flop / t_ws / 1e9, lds, diff, n_elem);
    for (void* p : {(void*)x, (void*)w, (void*)bias, (void*)blen, (void*)o1, (void*)o2}) HIP_CHECK(hipFree(p));
}

int main() {
    HIP_CHECK(hipSetDevice(0));
    run<3, 3>("stage-1 conv1", 64, 128, 64, 1219);
    run<7, 3>("stage-1 conv1", 64, 128, 64, 1219);
    return 0;
}
