// EXPERIMENT (r02, measured slower, not part of the library): persistent decode kernel (tools/decode_persistent.h) against the 4-launch-per-layer GEMM chain: bitwise comparison
// of the residual stream after 30 layers and time per layer.  Build + run on the GPU box:  bash tools/persist_bench.sh
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../auralis_amd/csrc/gpt_kernels.hip"
#include "decode_persistent.h"

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

__global__ void stats_init_kernel(const float* h, int mtt, float2* stats, int M) {   // partials of packed rows, as embed_decode emits them
    const int m = blockIdx.x, t = threadIdx.x;   // 64 threads: tile t
    float v[16], mu = 0.f;
    for (int i = 0; i < 16; ++i) {
        v[i] = h[pk_off(m, 16 * t + i, mtt)];
        mu += v[i];
    }
    mu *= 1.0f / 16.0f;
    float m2 = 0.f;
    for (int i = 0; i < 16; ++i) m2 += (v[i] - mu) * (v[i] - mu);
    if (m < M) stats[(long)m * 64 + t] = make_float2(mu, m2);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    const int NL = argc > 2 ? atoi(argv[2]) : 30;
    HIP_CHECK(hipSetDevice(0));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    float* bias = dalloc(4096, 0.1f, 4);
    float* gamma = dalloc(4096, 1.0f, 5);
    float* beta = dalloc(4096, 0.1f, 6);
    float* kv = dalloc((size_t)(64 * 66 + 8) * kKvBlockElems, 0.f, 7);
    std::vector<int> meta(64 * kRowMetaStride, 0);
    for (int m = 0; m < 64; ++m) {
        meta[m * kRowMetaStride] = 100;   // K/V position of the new token
        meta[m * kRowMetaStride + 1] = m;
        for (int b = 0; b < 66; ++b) meta[m * kRowMetaStride + kRowMetaBt + b] = m * 66 + b;
    }
    int* dmeta;
    HIP_CHECK(hipMalloc(&dmeta, meta.size() * 4));
    HIP_CHECK(hipMemcpy(dmeta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
    std::vector<DecodeLayerP> hl(NL);
    float* src = dalloc((size_t)4096 * 1024 + 4096, 0.02f, 21);
    for (int l = 0; l < NL; ++l) {
        float *wq, *wp, *wf, *w2;
        HIP_CHECK(hipMalloc(&wq, (size_t)3072 * 1024 * 4));
        HIP_CHECK(hipMalloc(&wp, (size_t)1024 * 1024 * 4));
        HIP_CHECK(hipMalloc(&wf, (size_t)4096 * 1024 * 4));
        HIP_CHECK(hipMalloc(&w2, (size_t)4096 * 1024 * 4));
        launch_pack_wt16(src + l * 16, 3072, wq, 1024, 3072, st);
        launch_pack_wt16(src + l * 16 + 7, 1024, wp, 1024, 1024, st);
        launch_pack_wt16(src + l * 16 + 3, 4096, wf, 1024, 4096, st);
        launch_pack_wt16(src + l * 16 + 5, 1024, w2, 4096, 1024, st);
        hl[l] = DecodeLayerP{wq, bias, gamma, beta, wp, bias + 64, wf, bias, gamma + 32, beta + 32, w2, bias + 128, kv};
    }
    DecodeLayerP* dl;
    HIP_CHECK(hipMalloc(&dl, NL * sizeof(DecodeLayerP)));
    HIP_CHECK(hipMemcpy(dl, hl.data(), NL * sizeof(DecodeLayerP), hipMemcpyHostToDevice));
    const size_t HN = (size_t)64 * 1024;
    float* h_init = dalloc(HN, 1.0f, 2);
    float* att = dalloc(HN, 0.5f, 23);
    float *h_ref, *qb, *act, *h0, *h_out, *ring;
    HIP_CHECK(hipMalloc(&h_ref, HN * 4));
    HIP_CHECK(hipMalloc(&h0, HN * 4));
    HIP_CHECK(hipMalloc(&h_out, HN * 4));
    HIP_CHECK(hipMalloc(&qb, HN * 4));
    HIP_CHECK(hipMalloc(&act, HN * 4 * 4));
    HIP_CHECK(hipMalloc(&ring, (size_t)NL * kRingLayer * 4));
    float2 *st_ref, *st0;
    HIP_CHECK(hipMalloc(&st_ref, 64 * 64 * sizeof(float2)));
    HIP_CHECK(hipMalloc(&st0, 64 * 64 * sizeof(float2)));
    unsigned* ctl;
    HIP_CHECK(hipMalloc(&ctl, kCtlWords * 4));
    HIP_CHECK(hipStreamSynchronize(st));

    auto reset = [&] {
        HIP_CHECK(hipMemcpyAsync(h_ref, h_init, HN * 4, hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(h0, h_init, HN * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(stats_init_kernel, dim3(64), dim3(64), 0, st, h_init, 4, st_ref, M);
        hipLaunchKernelGGL(stats_init_kernel, dim3(64), dim3(64), 0, st, h_init, 4, st0, M);
    };
    auto chain = [&] {
        for (int l = 0; l < NL; ++l) {
            const DecodeLayerP& L = hl[l];
            GemmRowsArgs a{};
            a.M = M; a.eps = 1e-5f; a.X = h_ref; a.xmt = 4; a.Wt = L.tqkv; a.N = 3072; a.K = 1024; a.bias = L.bqkv; a.gamma = L.ln1w;
            a.beta = L.ln1b; a.stats_in = st_ref; a.out = qb; a.ldo = 1024; a.kv_layer = L.kv_layer; a.row_meta = dmeta; a.max_blocks = 66;
            launch_gemm_rows(a, true, kEpiQkv, st);
            a = GemmRowsArgs{};
            a.M = M; a.X = att; a.xmt = 4; a.Wt = L.tproj; a.N = 1024; a.K = 1024; a.bias = L.bproj; a.out = h_ref; a.omt = 4; a.stats_out = st_ref;
            launch_gemm_rows(a, false, kEpiResidual, st);
            a = GemmRowsArgs{};
            a.M = M; a.eps = 1e-5f; a.X = h_ref; a.xmt = 4; a.Wt = L.tfc; a.N = 4096; a.K = 1024; a.bias = L.bfc; a.gamma = L.ln2w;
            a.beta = L.ln2b; a.stats_in = st_ref; a.out = act; a.omt = 4;
            launch_gemm_rows(a, true, kEpiBiasGelu, st);
            a = GemmRowsArgs{};
            a.M = M; a.X = act; a.xmt = 4; a.Wt = L.tproj2; a.N = 1024; a.K = 4096; a.bias = L.bproj2; a.out = h_ref; a.omt = 4;
            if (l + 1 < NL) a.stats_out = st_ref;
            launch_gemm_rows(a, false, kEpiResidual, st);
        }
    };
    DecodePersistArgs pa{};
    pa.layers = dl; pa.n_layer = NL; pa.M = M; pa.eps = 1e-5f; pa.h0 = h0; pa.mtt0 = 4; pa.stats0 = st0; pa.ring = ring; pa.h_out = h_out;
    pa.row_meta = dmeta; pa.kv_half = 0; pa.ctl = ctl; pa.spin_cap = 1 << 18; pa.att_fixed = att;
    auto persist = [&] { launch_decode_layers(pa, false, st); };

    // ---- correctness: same start, both paths, bitwise comparison of rows < M
    reset();
    chain();
    persist();
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<float> a(HN), b(HN);
    std::vector<unsigned> hc(kCtlWords);
    HIP_CHECK(hipMemcpy(a.data(), h_ref, HN * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(b.data(), h_out, HN * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(hc.data(), ctl, kCtlWords * 4, hipMemcpyDeviceToHost));
    long diff = 0, nan = 0;
    double maxabs = 0;
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < 1024; ++k) {
            const float x = a[pk_off(m, k, 4)], y = b[pk_off(m, k, 4)];
            if (memcmp(&x, &y, 4)) ++diff;
            if (x != x) ++nan;
            maxabs = std::max(maxabs, (double)fabsf(x - y));
        }
    printf("M=%d layers=%d: residual stream after the chain vs persistent kernel: %ld of %d elements differ (max abs %.3e, NaN %ld); err word 0x%x; XCD populations",
           M, NL, diff, M * 1024, maxabs, nan, hc[kCtlErr]);
    for (int i = 0; i < 8; ++i) printf(" %u", hc[kCtlPop + i]);
    printf("\n");
    if (hc[kCtlErr]) return 2;

    auto time_us = [&](int iters, auto&& f) {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) f();
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) f();
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / iters;
    };
    const float us_c = time_us(20, chain);
    const float us_p = time_us(20, persist);
    printf("launch chain     : %.1f us per layer (%.2f ms per %d layers)\n", us_c / NL, us_c / 1000, NL);
    printf("persistent kernel: %.1f us per layer (%.2f ms per %d layers)\n", us_p / NL, us_p / 1000, NL);
    HIP_CHECK(hipMemcpy(hc.data(), ctl, kCtlWords * 4, hipMemcpyDeviceToHost));
    if (hc[kCtlErr]) printf("barrier error word 0x%x after timing\n", hc[kCtlErr]);
    // ---- in-kernel phase stamps of layer 1 (shader clock), averaged over the workgroups
    {
        long long* dprof;
        HIP_CHECK(hipMalloc(&dprof, 256 * 64 * 8));
        HIP_CHECK(hipMemset(dprof, 0, 256 * 64 * 8));
        pa.prof = dprof;
        persist();
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<long long> hp(256 * 64);
        HIP_CHECK(hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost));
        const char* names[4] = {"qkv  ", "proj ", "fc   ", "proj2"};
        const int nlive = std::min(4, (M + 15) / 16);
        for (int ph = 0; ph < 4; ++ph) {
            const int b = ph * 8;
            double s[6] = {0, 0, 0, 0, 0, 0}, mx = 0;
            int n = 0;
            for (int w = 0; w < 256; ++w) {
                if (((w >> 3) & 3) >= nlive) continue;
                const long long* q = &hp[(size_t)w * 64 + b];
                s[0] += q[1] - q[0];   // start -> operands / statistics ready
                s[1] += q[2] - q[1];   // MFMAs + partials to LDS
                s[2] += q[3] - q[2];   // wait for the other waves
                s[3] += q[4] - q[3];   // reduction + epilogue stores issued
                s[4] += q[5] - q[4];   // prefetch issue
                s[5] += q[8] - q[5];   // barrier (store completion + arrival of the last workgroup + release)
                mx = std::max(mx, (double)(q[5] - q[0]));
                ++n;
            }
            printf("%s cycles: operands %5.0f  mfma %5.0f  wave-sync %5.0f  reduce+epilogue %5.0f  prefetch-issue %4.0f | work avg %5.0f max %5.0f | barrier %5.0f\n",
                   names[ph], s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n, (s[0] + s[1] + s[2] + s[3] + s[4]) / n, mx, s[5] / n);
        }
    }
    return 0;
}
