// Standalone micro-benchmark of the decode-regime GEMM kernels (no Python, no engine): per-launch time of every
// workgroup shape back to back on one stream, in-kernel phase stamps (s_memtime), shader-clock calibration, and a
// 5-launch layer chain.  Build + run on the GPU box:  bash tools/gemm_bench.sh
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

__global__ void clock_kernel(long long* out, int spin) {
    const long long c0 = __builtin_amdgcn_s_memtime();
    const long long w0 = wall_clock64();
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fmaf(x, 1.000001f, 0.5f);
    const long long c1 = __builtin_amdgcn_s_memtime();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
        out[2] = (long long)x;
    }
}
__global__ void empty_kernel(float* p) {
    if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f;
}

struct Shape {
    const char* name;
    int N, K;
    bool ln;
    GemmRowsEpi epi;
};

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
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
    return ms * 1000.f / iters;
}

static void launch_variant(const GemmRowsArgs& a, const Shape& s, int mt, int nw, hipStream_t st, int ntl = 1) {
    if (s.ln && s.epi == kEpiQkv) launch_gemm_rows_mt<1, true, kEpiQkv>(a, mt, nw, st, ntl);
    else if (s.ln && s.epi == kEpiBiasGelu) launch_gemm_rows_mt<1, true, kEpiBiasGelu>(a, mt, nw, st, ntl);
    else if (s.ln && s.epi == kEpiBias) launch_gemm_rows_mt<1, true, kEpiBias>(a, mt, nw, st, ntl);
    else if (s.K == 1024 && s.epi == kEpiResidual) launch_gemm_rows_mt<1, false, kEpiResidual>(a, mt, nw, st);
    else if (s.K == 4096 && s.epi == kEpiResidual) launch_gemm_rows_mt<4, false, kEpiResidual>(a, mt, nw, st);
    else if (s.K == 1024 && s.epi == kEpiBias) launch_gemm_rows_mt<1, false, kEpiBias>(a, mt, nw, st);
    else launch_gemm_rows_mt<4, false, kEpiBias>(a, mt, nw, st);
    HIP_CHECK(hipGetLastError());
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    HIP_CHECK(hipSetDevice(0));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    // ---- clock calibration: shader cycles per wall-clock tick (100 MHz) while idle / right after a burst
    long long* dclk;
    HIP_CHECK(hipMalloc(&dclk, 64));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, st, dclk, 200000);
        long long h[3];
        HIP_CHECK(hipMemcpyAsync(h, dclk, 24, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        printf("clock: %lld shader cycles in %lld wall ticks (100 MHz) -> %.0f MHz\n", h[0], h[1], 100.0 * h[0] / h[1]);
    }
    const float e256 = time_us(st, 500, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, st, (float*)dclk); });
    printf("empty kernel, 256 x 1024 threads, back to back: %.2f us per launch\n", e256);

    // ---- buffers
    const int H = 1024;
    float* X = dalloc((size_t)256 * 4096, 1.0f, 1);
    float* hres = dalloc((size_t)256 * 4096, 1.0f, 2);
    float* out = dalloc((size_t)256 * 4096, 0.f, 3);
    float* bias = dalloc(4096, 0.1f, 4);
    float* gamma = dalloc(4096, 1.0f, 5);
    float* beta = dalloc(4096, 0.1f, 6);
    float* kv = dalloc((size_t)(64 * 66 + 8) * kKvBlockElems, 0.f, 7);
    std::vector<int> hslot(256), hpos(256, 100), hbt(256 * 66);
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
    HIP_CHECK(hipMemset(dstats, 0, 256 * 64 * sizeof(float2)));
    long long* dprof;
    HIP_CHECK(hipMalloc(&dprof, 4096 * 8 * 8));

    const Shape shapes[] = {{"qkv  N=3072 K=1024 LN  kv-write", 3072, 1024, true, kEpiQkv},
                            {"proj N=1024 K=1024     residual", 1024, 1024, false, kEpiResidual},
                            {"fc   N=4096 K=1024 LN  gelu    ", 4096, 1024, true, kEpiBiasGelu},
                            {"prj2 N=1024 K=4096     residual", 1024, 4096, false, kEpiResidual},
                            {"head N=1088 K=1024     bias    ", 1088, 1024, false, kEpiBias},
                            {"qkv' N=3072 K=1024 noLN bias   ", 3072, 1024, false, kEpiBias},
                            {"fc'  N=4096 K=1024 noLN bias   ", 4096, 1024, false, kEpiBias}};
    // 30 distinct weight matrices per shape so that the stream really comes from HBM (a single 12-16 MB matrix would sit
    // in the 256 MiB Infinity Cache across iterations)
    const int NREP = 24;
    for (const Shape& s : shapes) {
        std::vector<float*> wt(NREP);
        float* wsrc = dalloc((size_t)s.K * s.N, 0.05f, 11);
        for (int r = 0; r < NREP; ++r) {
            HIP_CHECK(hipMalloc(&wt[r], (size_t)s.K * s.N * 4));
            launch_pack_wt16(wsrc, s.N, wt[r], s.K, s.N, st);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        for (int ntl : {1, 2})
        for (int nt : {0})
        for (int nw : {16, 8, 4})
            for (int mt : {1, 2, 4}) {
                if (nw == 4 && (mt != 1 || s.K != 1024 || ntl != 1)) continue;
                if (ntl == 2 && !(s.ln && nw == 8 && mt <= 2)) continue;
                if (nt && (M + 16 * mt - 1) / (16 * mt) != 1) continue;
                if (s.K == 4096 && mt == 4) continue;
                if (16 * (mt / 2) >= M && mt > 1) continue;
                GemmRowsArgs a{};
                a.X = X; a.xmt = 16; a.omt = 16; a.M = M; a.N = s.N; a.K = s.K; a.bias = bias; a.gamma = gamma; a.beta = beta; a.eps = 1e-5f;
                a.out = (s.epi == kEpiResidual) ? hres : out; a.ldo = (s.epi == kEpiQkv) ? H : s.N;
                a.kv_layer = kv; a.row_slot = dslot; a.slot_kvpos = dpos; a.block_tables = dbt; a.max_blocks = 66; a.nt_w = nt; a.stats_in = dstats; a.stats_out = (s.epi == kEpiResidual && s.N == 1024) ? dstats : nullptr;
                int it = 0;
                const float us = time_us(st, 240, [&] {
                    a.Wt = wt[it++ % NREP];
                    launch_variant(a, s, mt, nw, st, ntl);
                });
                const int n_grp = (M + 16 * mt - 1) / (16 * mt);
                const int nwg = ((s.N / (16 * ntl) + 7) / 8 * 8) * n_grp;
                // phase stamps of one launch
                HIP_CHECK(hipMemsetAsync(dprof, 0, (size_t)nwg * 64, st));
                a.prof = dprof;
                a.Wt = wt[5];
                launch_variant(a, s, mt, nw, st, ntl);
                HIP_CHECK(hipStreamSynchronize(st));
                std::vector<long long> hp((size_t)nwg * 8);
                HIP_CHECK(hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost));
                long long t_min = -1, t_max = 0;
                double ph[5] = {0, 0, 0, 0, 0};
                int live = 0;
                std::vector<double> st_us, en_us;
                for (int g = 0; g < nwg; ++g) {
                    if (hp[g * 8] == 0) continue;   // surplus workgroups of a padded grid exit before the first stamp
                    ++live;
                    t_min = t_min < 0 ? hp[g * 8] : std::min(t_min, hp[g * 8]);
                    t_max = std::max(t_max, hp[g * 8 + 5]);
                }
                for (int g = 0; g < nwg; ++g) {
                    if (hp[g * 8] == 0) continue;
                    for (int k = 0; k < 5; ++k) ph[k] += (double)(hp[g * 8 + k + 1] - hp[g * 8 + k]) / live;
                    st_us.push_back((double)(hp[g * 8] - t_min) / 100.0);
                    en_us.push_back((double)(hp[g * 8 + 5] - t_min) / 100.0);
                }
                std::sort(st_us.begin(), st_us.end());
                std::sort(en_us.begin(), en_us.end());
                printf("    workgroup starts (us after the first): p50 %.2f p90 %.2f max %.2f | ends: p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
                       st_us[st_us.size() / 2], st_us[st_us.size() * 9 / 10], st_us.back(), en_us[en_us.size() / 10], en_us[en_us.size() / 2],
                       en_us[en_us.size() * 9 / 10], en_us.back());
                const double mb = 4.0 * ((double)s.K * s.N + (double)M * s.K + (double)M * s.N) / 1e6;
                printf("%s M=%d cols/wg=%d rows/wg=%2d waves=%2d wgs=%4d : %6.2f us/launch  %5.2f TB/s | 10-ns ticks: span %6lld issue %5.0f ln+wait %6.0f mfma %6.0f bar %5.0f epi %5.0f\n",
                       s.name, M, 16 * ntl, 16 * mt, nw, nwg, us, mb / us, t_max - t_min, ph[0], ph[1], ph[2], ph[3], ph[4]);
            }
        // reference: the round-1 split-K kernel on the same shape (slab sums not included)
        {
            float* P;
            HIP_CHECK(hipMalloc(&P, (size_t)16 * 128 * 4096 * 4));
            const GemmPlan pl = gemm_plan(M, s.K);
            if (s.N % 64 == 0) {
                const float us = time_us(st, 240, [&] { launch_gemm_splitk(X, s.K, wsrc, P, M, s.N, s.K, pl, st, nullptr); });
                printf("%s M=%d round-1 split-K (slabs %d, epilogue launches not included): %6.2f us/launch (weights L3-resident)\n", s.name, M, pl.slabs, us);
            }
            HIP_CHECK(hipFree(P));
        }
        for (int r = 0; r < NREP; ++r) HIP_CHECK(hipFree(wt[r]));
        HIP_CHECK(hipFree(wsrc));
    }
    // ---- the four GEMMs of a block chained over 30 layers with distinct weights (1.5 GB: streams from HBM), default policy
    {
        std::vector<float*> wq(30), wp(30), wf(30), w2(30);
        float* src = dalloc((size_t)4096 * 1024, 0.05f, 21);
        for (int l = 0; l < 30; ++l) {
            HIP_CHECK(hipMalloc(&wq[l], (size_t)3072 * 1024 * 4));
            HIP_CHECK(hipMalloc(&wp[l], (size_t)1024 * 1024 * 4));
            HIP_CHECK(hipMalloc(&wf[l], (size_t)4096 * 1024 * 4));
            HIP_CHECK(hipMalloc(&w2[l], (size_t)4096 * 1024 * 4));
            launch_pack_wt16(src, 3072, wq[l], 1024, 3072, st);
            launch_pack_wt16(src, 1024, wp[l], 1024, 1024, st);
            launch_pack_wt16(src, 4096, wf[l], 1024, 4096, st);
            launch_pack_wt16(src, 1024, w2[l], 4096, 1024, st);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        float* act = dalloc((size_t)256 * 4096, 0.5f, 22);
        float* att = dalloc((size_t)256 * 1024, 0.5f, 23);
        float* qb = dalloc((size_t)256 * 1024, 0.f, 24);
        auto chain = [&] {
            for (int l = 0; l < 30; ++l) {
                GemmRowsArgs a{};
                a.M = M; a.eps = 1e-5f; a.X = hres; a.xmt = 16; a.Wt = wq[l]; a.N = 3072; a.K = 1024; a.bias = bias; a.gamma = gamma;
                a.beta = beta; a.stats_in = dstats; a.out = qb; a.ldo = 1024; a.kv_layer = kv; a.row_slot = dslot; a.slot_kvpos = dpos; a.block_tables = dbt;
                a.max_blocks = 66;
                launch_gemm_rows(a, true, kEpiQkv, st);
                a = GemmRowsArgs{};
                a.M = M; a.X = att; a.xmt = 16; a.Wt = wp[l]; a.N = 1024; a.K = 1024; a.bias = bias; a.out = hres; a.omt = 16; a.stats_out = dstats;
                launch_gemm_rows(a, false, kEpiResidual, st);
                a = GemmRowsArgs{};
                a.M = M; a.eps = 1e-5f; a.X = hres; a.xmt = 16; a.Wt = wf[l]; a.N = 4096; a.K = 1024; a.bias = bias; a.gamma = gamma;
                a.beta = beta; a.stats_in = dstats; a.out = act; a.omt = 16;
                launch_gemm_rows(a, true, kEpiBiasGelu, st);
                a = GemmRowsArgs{};
                a.M = M; a.X = act; a.xmt = 16; a.Wt = w2[l]; a.N = 1024; a.K = 4096; a.bias = bias; a.out = hres; a.omt = 16; a.stats_out = dstats;
                launch_gemm_rows(a, false, kEpiResidual, st);
            }
        };
        const float us = time_us(st, 20, chain);
        printf("chain of 30 x (qkv, proj, fc, proj2), default shapes, M=%d: %.1f us per layer (%.2f ms per step), weights %.2f TB/s\n", M,
               us / 30, us / 1000, 30 * 50.33e6 / us / 1e6);
    }
    return 0;
}
