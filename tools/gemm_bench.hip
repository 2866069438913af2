// Standalone micro-benchmark of the decode-regime kernels (no Python, no engine), on the workgroup shapes the engine's policy
// picks (gemm_rows_shape):
//   * per-launch time of the four block GEMMs and the mel head, back to back on one stream over 24 distinct weight matrices (the
//     stream really comes from HBM), next to the SAME launch with its tile loads and MFMAs compiled out (gemm_rows_kernel.inc,
//     AUR_GR_NO_DATA = 1): the fixed cost of a launch -- boundary, argument fetch, epilogue inputs, LDS reduction, epilogue;
//   * the decode attention launch alone by context length;
//   * the decode layer chained over 30 layers of distinct weights and K/V pools, with and without attention.
// Build + run:  bash tools/gemm_bench.sh <tag> <M> [<M> ...]      (the A/B workgroup shapes of rounds 3-4 are in the git history)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../auralis_amd/csrc/gpt_kernels.hip"

namespace aur {
#define AUR_GR_NAME gemm_rows_nodata_kernel
#define AUR_GR_LAUNCH launch_gemm_rows_nodata
#define AUR_GR_NO_DATA 1
#include "../auralis_amd/csrc/gemm_rows_kernel.inc"
#undef AUR_GR_NAME
#undef AUR_GR_LAUNCH
#undef AUR_GR_NO_DATA
}  // namespace aur

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

__global__ void empty_kernel(int) {}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    const bool quick = argc > 2 && atoi(argv[2]) == 1;   // 1: chains only
    HIP_CHECK(hipSetDevice(0));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    const int H = 1024, MTT = 16;
    float* hres = dalloc((size_t)256 * 4096, 1.0f, 2);
    float* out = dalloc((size_t)256 * 4096, 0.f, 3);
    float* bias = dalloc(4096, 0.1f, 4);
    float* gamma = dalloc(4096, 1.0f, 5);
    const int n_layers = 30;
    const long kv_blocks = 64 * 66 + 8;
    float* kv = nullptr;   // one K/V pool per layer: attention streams 127 MB per launch from HBM, as in the engine
    HIP_CHECK(hipMalloc(&kv, (size_t)n_layers * kv_blocks * kKvBlockElems * 4));
    HIP_CHECK(hipMemset(kv, 0, (size_t)n_layers * kv_blocks * kKvBlockElems * 4));
    std::vector<int> hslot(256), hbt(256 * 66);
    for (int i = 0; i < 256; ++i) hslot[i] = i % 64;
    for (int i = 0; i < 64 * 66; ++i) hbt[i] = i;
    // dense per-row K/V addressing as embed_decode_kernel writes it once per step
    int* drm;
    HIP_CHECK(hipMalloc(&drm, (size_t)256 * kRowMetaStride * 4));
    auto set_positions = [&](int pos) {
        std::vector<int> hrm((size_t)256 * kRowMetaStride, 0);
        for (int i = 0; i < 256; ++i) {
            int* rm = &hrm[(size_t)i * kRowMetaStride];
            const int slot = hslot[i];
            rm[0] = pos;
            rm[1] = slot;
            rm[kRowMetaWblk] = hbt[slot * 66 + pos / kKvBlockTokens];
            for (int j = 0; j < 66; ++j) rm[kRowMetaBt + j] = hbt[slot * 66 + j];
        }
        HIP_CHECK(hipMemcpy(drm, hrm.data(), hrm.size() * 4, hipMemcpyHostToDevice));
    };
    set_positions(243);
    float2* dstats;
    HIP_CHECK(hipMalloc(&dstats, 256 * 64 * sizeof(float2)));
    {   // plausible LayerNorm partials (mean 0, M2 = 16/3 per 16-column tile of uniform(-1, 1) data)
        std::vector<float2> hs(256 * 64, make_float2(0.f, 16.f / 3.f));
        HIP_CHECK(hipMemcpy(dstats, hs.data(), hs.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    float* ksp_buf;
    unsigned* ksp_cnt;
    HIP_CHECK(hipMalloc(&ksp_buf, (size_t)kGemmKspTiles * 16 * 256 * 4));
    HIP_CHECK(hipMalloc(&ksp_cnt, (size_t)kGemmKspTiles * 4));
    HIP_CHECK(hipMemset(ksp_cnt, 0, (size_t)kGemmKspTiles * 4));
    float* act = dalloc((size_t)256 * 4096, 0.5f, 22);
    float* att = dalloc((size_t)256 * 1024, 0.5f, 23);
    float* qb = dalloc((size_t)256 * 1024, 0.f, 24);

    const Shape shapes[] = {{"qkv ", 3072, 1024, true, kEpiQkv},
                            {"proj", 1024, 1024, false, kEpiResidual},
                            {"fc  ", 4096, 1024, true, kEpiBiasGelu},
                            {"prj2", 1024, 4096, false, kEpiResidual},
                            {"head", 1088, 1024, false, kEpiBias}};
    // (a second set of activation buffers, K-split scratch and rows 32.. of the row table: the second chain of the two-stream experiment)
    struct Bufs { float *hres, *act, *att, *qb, *ksp_buf; unsigned* ksp_cnt; int* drm; float2* dstats; };
    const Bufs B0{hres, act, att, qb, ksp_buf, ksp_cnt, drm, dstats};
    Bufs B1 = B0;
    if (getenv("DUAL")) {
        B1.hres = dalloc((size_t)256 * 4096, 1.0f, 32);
        B1.act = dalloc((size_t)256 * 4096, 0.5f, 33);
        B1.att = dalloc((size_t)256 * 1024, 0.5f, 34);
        B1.qb = dalloc((size_t)256 * 1024, 0.f, 35);
        HIP_CHECK(hipMalloc(&B1.ksp_buf, (size_t)kGemmKspTiles * 16 * 256 * 4));
        HIP_CHECK(hipMalloc(&B1.ksp_cnt, (size_t)kGemmKspTiles * 4));
        HIP_CHECK(hipMemset(B1.ksp_cnt, 0, (size_t)kGemmKspTiles * 4));
        B1.drm = drm + (size_t)M * kRowMetaStride;
        HIP_CHECK(hipMalloc(&B1.dstats, 256 * 64 * sizeof(float2)));
        HIP_CHECK(hipMemcpy(B1.dstats, dstats, 256 * 64 * sizeof(float2), hipMemcpyDeviceToDevice));
    }
    auto make_args_b = [&](const Shape& s, const float* wt, int prec, const Bufs& B) {
        GemmRowsArgs a{};
        a.M = M; a.prec = prec; a.eps = 1e-5f; a.xmt = MTT; a.omt = MTT; a.Wt = wt; a.N = s.N; a.K = s.K; a.bias = bias;
        a.ksp_buf = B.ksp_buf; a.ksp_cnt = B.ksp_cnt;
        if (s.ln) { a.ln_c1 = gamma; a.stats_in = B.dstats; }
        if (s.epi == kEpiQkv) { a.X = B.hres; a.out = B.qb; a.ldo = H; a.kv_layer = kv; a.row_meta = B.drm; }
        else if (s.epi == kEpiBiasGelu) { a.X = B.hres; a.out = B.act; }
        else if (s.epi == kEpiResidual) { a.X = s.K == 4096 ? B.act : B.att; a.out = B.hres; a.stats_out = B.dstats + 128 * 64; }
        else { a.X = B.att; a.out = out; a.ldo = s.N; }
        return a;
    };
    auto make_args = [&](const Shape& s, const float* wt, int prec) { return make_args_b(s, wt, prec, B0); };
    {
        const float e_us = time_us(st, 2000, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, 0); });
        printf("empty 256-workgroup kernel, back to back: %.2f us per launch (host-bound floor of this loop)\n", e_us);
    }
    const int NREP = getenv("NREP") ? atoi(getenv("NREP")) : 24;   // distinct weight matrices per shape (NREP=1: the stream comes from L2)
    if (!quick)
        for (const Shape& s : shapes) {
            std::vector<float*> wt(NREP);
            float* wsrc = dalloc((size_t)s.K * s.N, 0.05f, 11);
            for (int r = 0; r < NREP; ++r) {
                HIP_CHECK(hipMalloc(&wt[r], (size_t)s.K * s.N * 4));
                launch_pack_wt16(wsrc, s.N, wt[r], s.K, s.N, st);
            }
            HIP_CHECK(hipStreamSynchronize(st));
            for (int prec : {1, 0}) {
                const GemmRowsShape g = gemm_rows_shape(M, s.N, s.K, s.ln);
                int it = 0;
                const float us = time_us(st, 480, [&] { launch_gemm_rows(make_args(s, wt[it++ % NREP], prec), s.ln, s.epi, st); });
                it = 0;
                const float us0 = time_us(st, 480, [&] { launch_gemm_rows_nodata(make_args(s, wt[it++ % NREP], prec), s.ln, s.epi, st); });
                const double mb = 4.0 * ((double)s.K * s.N + (double)M * s.K + (double)M * s.N * (s.epi == kEpiResidual ? 2 : 1)) / 1e6;
                printf("%s M=%d rows/wg=%2d cols/wg=%2d waves=%2d ksp=%d nt=%d prec=%d : %6.2f us per launch (%.2f TB/s) | fixed (no tile loads, no MFMAs) %5.2f us\n",
                       s.name, M, 16 * g.mt, 16 * g.ntl, g.nw, g.ksp, (int)g.nt, prec, us, mb / us, us0);
                fflush(stdout);
            }
            for (int r = 0; r < NREP; ++r) HIP_CHECK(hipFree(wt[r]));
            HIP_CHECK(hipFree(wsrc));
        }
    // ---- decode attention alone, by context length (all rows at the same position): intercept = the launch's fixed chain,
    // slope = per 64-token iteration
    for (int pos : {0, 63, 127, 243, 383, 639}) {
        set_positions(pos);
        int l = 0;
        const float us = time_us(st, 240, [&] {
            float* kvl = kv + (size_t)(l++ % n_layers) * kv_blocks * kKvBlockElems;
            launch_paged_attention(qb, kvl, drm, 66, att, M, st, MTT, false);
        });
        printf("paged attention alone M=%d context %3d tokens: %.2f us per launch (%.2f TB/s)\n", M, pos + 1, us,
               8.0 * 1024 * (pos + 1) * M / 1e6 / us);
    }
    set_positions(243);
    // ---- the decode layer chained over 30 layers with distinct weights (1.5 GB) and K/V pools (127 MB read per attention launch)
    {
        std::vector<float*> wq(n_layers), wp(n_layers), wf(n_layers), w2(n_layers);
        float* src = dalloc((size_t)4096 * 1024, 0.05f, 21);
        for (int l = 0; l < n_layers; ++l) {
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
        auto chain = [&](bool attn, int prec) {
            for (int l = 0; l < n_layers; ++l) {
                float* kvl = kv + (size_t)l * kv_blocks * kKvBlockElems;
                GemmRowsArgs a = make_args(shapes[0], wq[l], prec);
                a.kv_layer = kvl;
                launch_gemm_rows(a, true, kEpiQkv, st);
                if (attn) launch_paged_attention(qb, kvl, drm, 66, att, M, st, MTT, false);
                launch_gemm_rows(make_args(shapes[1], wp[l], prec), false, kEpiResidual, st);
                launch_gemm_rows(make_args(shapes[2], wf[l], prec), true, kEpiBiasGelu, st);
                launch_gemm_rows(make_args(shapes[3], w2[l], prec), false, kEpiResidual, st);
            }
        };
        if (getenv("DUAL")) {
            // two chains of M rows each on two streams, enqueued layer by layer in turn (rows 0..M-1 and M..2M-1 of the row table: distinct
            // K/V blocks; the same weights, a layer apart at most): do the launch boundaries of one chain hide under the other's data?
            hipStream_t st2;
            HIP_CHECK(hipStreamCreate(&st2));
            auto layer = [&](int l, const Bufs& B, hipStream_t s_, int prec) {
                float* kvl = kv + (size_t)l * kv_blocks * kKvBlockElems;
                GemmRowsArgs a = make_args_b(shapes[0], wq[l], prec, B);
                a.kv_layer = kvl;
                launch_gemm_rows(a, true, kEpiQkv, s_);
                launch_paged_attention(B.qb, kvl, B.drm, 66, B.att, M, s_, MTT, false);
                launch_gemm_rows(make_args_b(shapes[1], wp[l], prec, B), false, kEpiResidual, s_);
                launch_gemm_rows(make_args_b(shapes[2], wf[l], prec, B), true, kEpiBiasGelu, s_);
                launch_gemm_rows(make_args_b(shapes[3], w2[l], prec, B), false, kEpiResidual, s_);
            };
            for (int stagger : {0, 2}) {
                hipEvent_t e0, e1, e2;
                HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventCreate(&e2));
                const int iters = 20;
                for (int rep = 0; rep < 2; ++rep) {   // rep 0 = warm-up
                    HIP_CHECK(hipDeviceSynchronize());
                    HIP_CHECK(hipEventRecord(e0, st));
                    HIP_CHECK(hipStreamWaitEvent(st2, e0, 0));
                    for (int it = 0; it < iters; ++it)
                        for (int l = 0; l < n_layers + stagger; ++l) {   // chain B runs `stagger` layers behind chain A
                            if (l < n_layers) layer(l, B0, st, 1);
                            if (l >= stagger) layer(l - stagger, B1, st2, 1);
                        }
                    HIP_CHECK(hipEventRecord(e1, st));
                    HIP_CHECK(hipEventRecord(e2, st2));
                    HIP_CHECK(hipEventSynchronize(e1));
                    HIP_CHECK(hipEventSynchronize(e2));
                }
                float m1 = 0.f, m2 = 0.f;
                HIP_CHECK(hipEventElapsedTime(&m1, e0, e1));
                HIP_CHECK(hipEventElapsedTime(&m2, e0, e2));
                const float us = (m1 > m2 ? m1 : m2) * 1000.f / iters;
                printf("TWO chains of 30 x (qkv, attention, proj, fc, proj2) M=%d each on two streams, stagger %d layers, prec=1: %.1f us per layer pair (%.3f ms per step of %d rows)\n",
                       M, stagger, us / n_layers, us / 1000, 2 * M);
                fflush(stdout);
            }
        }
        for (int attn : {0, 1})
            for (int prec : {0, 1}) {
                const float us = time_us(st, 20, [&] { chain(attn != 0, prec); });
                printf("chain of 30 x (qkv,%s proj, fc, proj2) M=%d prec=%d: %.1f us per layer (%.3f ms per step)\n", attn ? " attention," : "", M,
                       prec, us / n_layers, us / 1000);
                fflush(stdout);
            }
    }
    return 0;
}
