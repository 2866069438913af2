// Diagnostic companion of gemm_bench: what bounds a decode GEMM launch.  For each GEMM kind in its default workgroup shape:
// per-launch time with the weight loads / activation loads / MFMAs compiled out (template parameter DBG), with the weights
// cold (24 distinct matrices, HBM) and hot (one matrix, L2 / Infinity Cache), and the back-to-back gap of an empty kernel of
// the same geometry.  Build + run: bash experiments/r03_b.sh
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../auralis_amd/csrc/gpt_kernels.hip"

using namespace aur;

static float* dalloc(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = scale * ((float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f);
    }
    float* d = nullptr;
    HIP_CHECK(hipMalloc(&d, n * 4));
    HIP_CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
__global__ void empty_kernel(float* p) {
    __shared__ float lds[12288];
    if (p == nullptr && threadIdx.x == 9999) p[0] = lds[3];
}

// prefetch-only kernel: workgroup L (XCD L % 8 + rot) touches its share of the lines of the column tiles t = xcd (mod 8) of W
// mode 0: one dword per 128-byte line; 1: one dword per 64 bytes; 2: coalesced float4 (every byte); 3: mode 0 with sc1 loads
__global__ __launch_bounds__(512) void prefetch_kernel(const char* __restrict__ W, int tile_bytes, int tiles, int mode, int rot, float* sink) {
    const int L = blockIdx.x, xcd = ((L & 7) + rot) & 7, slot = L >> 3, n_slot = gridDim.x >> 3;
    const int gran = mode == 0 || mode == 3 ? 128 : (mode == 1 ? 64 : 16);
    const int lpt = tile_bytes / gran;
    const int lines = ((tiles - xcd + 7) >> 3) * lpt;
    const int per = (lines + n_slot - 1) / n_slot;
    float acc = 0.f;
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const int ln = min(slot * per + i, lines - 1);
        const int ti = ln / lpt;
        const char* p = W + ((long)ti * 8 + xcd) * tile_bytes + (long)(ln - ti * lpt) * gran;
        if (mode == 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
            acc += v[0] + v[3];
        } else if (mode == 3) {
            float v;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            acc += v;
        } else {
            acc += *reinterpret_cast<const float*>(p);
        }
    }
    if (acc == 1.2345e30f) sink[0] = acc;
}

template <class F>
static float time_us(hipStream_t st, int iters, F&& f) {
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) f();
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) f();
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}
struct Shape {
    const char* name;
    int N, K;
    bool ln;
    GemmRowsEpi epi;
};
template <int DBG>
static void launch_dbg(const GemmRowsArgs& a, const Shape& s, const GemmRowsShape& c, hipStream_t st) {
    if (s.ln && s.epi == kEpiQkv) launch_gemm_rows_mt<1, true, kEpiQkv, 0, DBG>(a, c.mt, c.nw, st, c.ntl);
    else if (s.ln && s.epi == kEpiBiasGelu) launch_gemm_rows_mt<1, true, kEpiBiasGelu, 0, DBG>(a, c.mt, c.nw, st, c.ntl);
    else if (s.K == 1024 && s.epi == kEpiResidual) launch_gemm_rows_mt<1, false, kEpiResidual, 0, DBG>(a, c.mt, c.nw, st, c.ntl);
    else launch_gemm_rows_mt<4, false, kEpiResidual, 0, DBG>(a, c.mt, c.nw, st, c.ntl);
    HIP_CHECK(hipGetLastError());
}
static void launch_dbg_rt(int dbg, const GemmRowsArgs& a, const Shape& s, const GemmRowsShape& c, hipStream_t st) {
    switch (dbg) {
        case 0: launch_dbg<0>(a, s, c, st); break;
        case 1: launch_dbg<1>(a, s, c, st); break;
        case 2: launch_dbg<2>(a, s, c, st); break;
        case 3: launch_dbg<3>(a, s, c, st); break;
        case 4: launch_dbg<4>(a, s, c, st); break;
        case 5: launch_dbg<5>(a, s, c, st); break;
        case 6: launch_dbg<6>(a, s, c, st); break;
        default: launch_dbg<7>(a, s, c, st); break;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    HIP_CHECK(hipSetDevice(0));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    for (int thr : {256, 1024}) {
        const float e = time_us(st, 500, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(thr), 0, st, (float*)nullptr + 1); });
        printf("empty kernel 256 x %d threads (48 KB LDS), back to back: %.2f us per launch\n", thr, e);
    }
    float* X = dalloc((size_t)256 * 4096, 1.0f, 1);
    float* hres = dalloc((size_t)256 * 4096, 1.0f, 2);
    float* out = dalloc((size_t)256 * 4096, 0.f, 3);
    float* bias = dalloc(4096, 0.1f, 4);
    float* gamma = dalloc(4096, 1.0f, 5);
    float* beta = dalloc(4096, 0.1f, 6);
    float* kv = nullptr;
    HIP_CHECK(hipMalloc(&kv, (size_t)(64 * 66 + 8) * kKvBlockElems * 4));
    std::vector<int> hslot(256), hpos(256, 243), hbt(256 * 66);
    for (int i = 0; i < 256; ++i) hslot[i] = i % 64;
    for (int i = 0; i < 64 * 66; ++i) hbt[i] = i;
    int *dslot, *dpos, *dbt;
    HIP_CHECK(hipMalloc(&dslot, 256 * 4));
    HIP_CHECK(hipMalloc(&dpos, 256 * 4));
    HIP_CHECK(hipMalloc(&dbt, 256 * 66 * 4));
    HIP_CHECK(hipMemcpy(dslot, hslot.data(), 256 * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dpos, hpos.data(), 256 * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dbt, hbt.data(), 64 * 66 * 4, hipMemcpyHostToDevice));
    float2* dstats;
    HIP_CHECK(hipMalloc(&dstats, 256 * 64 * sizeof(float2)));
    {
        std::vector<float2> hs(256 * 64, make_float2(0.f, 16.f / 3.f));
        HIP_CHECK(hipMemcpy(dstats, hs.data(), hs.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    const Shape shapes[] = {{"qkv ", 3072, 1024, true, kEpiQkv},
                            {"proj", 1024, 1024, false, kEpiResidual},
                            {"fc  ", 4096, 1024, true, kEpiBiasGelu},
                            {"prj2", 1024, 4096, false, kEpiResidual}};
    const int NREP = 24;
    const bool skip_sweep = argc > 2 && atoi(argv[2]) == 1;
    if (!skip_sweep)
    for (const Shape& s : shapes) {
        std::vector<float*> wt(NREP);
        float* wsrc = dalloc((size_t)s.K * s.N, 0.05f, 11);
        for (int r = 0; r < NREP; ++r) {
            HIP_CHECK(hipMalloc(&wt[r], (size_t)s.K * s.N * 4));
            launch_pack_wt16(wsrc, s.N, wt[r], s.K, s.N, st);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        const GemmRowsShape c = gemm_rows_shape(M, s.N, s.K, s.ln);
        for (int xmt : {16, 4})
            for (int hot : {0, 1}) {
                printf("%s M=%d %dx%d xmt=%2d weights %s:", s.name, M, 16 * c.mt, 16 * c.ntl, xmt, hot ? "hot " : "cold");
                for (int dbg = 0; dbg < 8; ++dbg) {
                    GemmRowsArgs a{};
                    a.X = X; a.xmt = xmt; a.omt = xmt; a.M = M; a.N = s.N; a.K = s.K; a.bias = bias; a.ln_c1 = s.ln ? gamma : nullptr; a.eps = 1e-5f;
                    a.out = (s.epi == kEpiResidual) ? hres : out; a.ldo = 1024;
                    a.kv_layer = kv; a.row_slot = dslot; a.slot_kvpos = dpos; a.block_tables = dbt; a.max_blocks = 66; a.stats_in = dstats;
                    int it = 0;
                    const float us = time_us(st, 240, [&] {
                        a.Wt = wt[hot ? 0 : (it++ % NREP)];
                        launch_dbg_rt(dbg, a, s, c, st);
                    });
                    printf("  %s%s%s %5.2f", (dbg & 1) ? "-W" : "+W", (dbg & 2) ? "-A" : "+A", (dbg & 4) ? "-M" : "+M", us);
                }
                printf("  us\n");
                fflush(stdout);
            }
        for (int r = 0; r < NREP; ++r) HIP_CHECK(hipFree(wt[r]));
        HIP_CHECK(hipFree(wsrc));
    }
    // ---- does touching the next launch's weights from the previous launch pay?  P = prefetch kernel on the FC weights, G = the FC GEMM
    {
        const Shape s = {"fc  ", 4096, 1024, true, kEpiBiasGelu};
        const int NR = 24;
        std::vector<float*> wt(NR);
        float* wsrc = dalloc((size_t)s.K * s.N, 0.05f, 11);
        for (int r = 0; r < NR; ++r) {
            HIP_CHECK(hipMalloc(&wt[r], (size_t)s.K * s.N * 4));
            launch_pack_wt16(wsrc, s.N, wt[r], s.K, s.N, st);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        const GemmRowsShape c = gemm_rows_shape(M, s.N, s.K, s.ln);
        GemmRowsArgs a{};
        a.X = X; a.xmt = 16; a.omt = 16; a.M = M; a.N = s.N; a.K = s.K; a.bias = bias; a.ln_c1 = s.ln ? gamma : nullptr; a.eps = 1e-5f;
        a.out = out; a.ldo = 1024; a.stats_in = dstats;
        const int tile_bytes = c.ntl * s.K * 64, tiles = s.N / (16 * c.ntl);
        int it = 0;
        const float g_cold = time_us(st, 240, [&] { a.Wt = wt[it++ % NR]; launch_dbg<0>(a, s, c, st); });
        const float g_hot = time_us(st, 240, [&] { a.Wt = wt[0]; launch_dbg<0>(a, s, c, st); });
        printf("fc GEMM alone: cold %.2f us, hot %.2f us\n", g_cold, g_hot);
        for (int mode = 0; mode < 4; ++mode)
            for (int rot : {0, 3}) {
                it = 0;
                const float p_only = time_us(st, 240, [&] {
                    hipLaunchKernelGGL(prefetch_kernel, dim3(256), dim3(512), 0, st, (const char*)wt[it++ % NR], tile_bytes, tiles, mode, rot, out);
                });
                it = 0;
                const float pair = time_us(st, 240, [&] {
                    a.Wt = wt[it++ % NR];
                    hipLaunchKernelGGL(prefetch_kernel, dim3(256), dim3(512), 0, st, (const char*)a.Wt, tile_bytes, tiles, mode, rot, out);
                    launch_dbg<0>(a, s, c, st);
                });
                printf("prefetch mode %d (0 dword/128B, 1 dword/64B, 2 float4, 3 dword/128B sc1) xcd rotation %d: P alone %.2f us, P + G %.2f us -> G after P %.2f us\n",
                       mode, rot, p_only, pair, pair - p_only);
                fflush(stdout);
            }
    }
    return 0;
}
