// Standalone micro-benchmark of the decode-regime GEMM kernels (no Python, no engine): per-launch time of workgroup shapes
// and arithmetic modes back to back on one stream, in-kernel phase stamps, a correctness check of every variant against the
// exact-f32 16 x 16 workgroup result, and the decode layer chain (QKV GEMM, attention, proj, FC, proj2 over 30 layers of
// distinct weights and K/V) in both arithmetic modes.  Build + run:  bash tools/gemm_bench.sh
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define AUR_GEMM_ALL_SHAPES 1   // the A/B workgroup shapes (8 waves, wide tiles) exist only in this tool
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

struct Shape {
    const char* name;
    int N, K;
    bool ln;
    GemmRowsEpi epi;
};
struct Cfg {
    int mt, nw, ntl, prec;
    bool nt = false;
    int ksp = 1;
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

template <int PREC, bool NT>
static void launch_variant_p(const GemmRowsArgs& a, const Shape& s, const Cfg& c, hipStream_t st) {
    if (s.ln && s.epi == kEpiQkv) launch_gemm_rows_mt<1, true, kEpiQkv, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    else if (s.ln && s.epi == kEpiBiasGelu) launch_gemm_rows_mt<1, true, kEpiBiasGelu, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    else if (s.ln && s.epi == kEpiBias) launch_gemm_rows_mt<1, true, kEpiBias, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    else if (s.K == 1024 && s.epi == kEpiResidual) launch_gemm_rows_mt<1, false, kEpiResidual, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    else if (s.K == 4096 && s.epi == kEpiResidual) launch_gemm_rows_mt<4, false, kEpiResidual, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    else if (s.K == 1024 && s.epi == kEpiBias) launch_gemm_rows_mt<1, false, kEpiBias, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    else launch_gemm_rows_mt<4, false, kEpiBias, PREC, NT>(a, c.mt, c.nw, st, c.ntl);
    HIP_CHECK(hipGetLastError());
}
static void launch_variant(const GemmRowsArgs& a, const Shape& s, const Cfg& c, hipStream_t st) {
    if (c.ksp > 1) {
        if (c.prec) launch_gemm_rows_ksp<1>(a, s.epi, st);
        else launch_gemm_rows_ksp<0>(a, s.epi, st);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (c.nt) {
        if (c.prec) launch_variant_p<1, true>(a, s, c, st);
        else launch_variant_p<0, true>(a, s, c, st);
    } else {
        if (c.prec) launch_variant_p<1, false>(a, s, c, st);
        else launch_variant_p<0, false>(a, s, c, st);
    }
}
// the shapes the engine used until round 3 at any M (for the small-M A/B): QKV 16 x 48, FC 16 x 32 at M <= 16, 16 x 16 otherwise
static Cfg r03_policy(int M, const Shape& s, int prec) {
    Cfg c{1, 16, 1, prec, false};
    if (s.ln) {
        if (s.N % 48 == 0 && s.N < 4096) c.ntl = 3;
        else if (s.N % 32 == 0) { c.ntl = 2; c.mt = M > 16 ? 2 : 1; }
    }
    return c;
}
static Cfg r04_policy(int M, const Shape& s, int prec) {
    const GemmRowsShape g = gemm_rows_shape(M, s.N, s.K, s.ln);
    return Cfg{g.mt, g.nw, g.ntl, prec, g.nt, g.ksp};
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    const bool quick = argc > 2 && atoi(argv[2]) == 1;   // 1: chains only
    HIP_CHECK(hipSetDevice(0));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    const int H = 1024, MTT = 16;
    float* X = dalloc((size_t)256 * 4096, 1.0f, 1);
    float* hres = dalloc((size_t)256 * 4096, 1.0f, 2);
    float* hres0 = dalloc((size_t)256 * 4096, 1.0f, 2);
    float* out = dalloc((size_t)256 * 4096, 0.f, 3);
    float* bias = dalloc(4096, 0.1f, 4);
    float* gamma = dalloc(4096, 1.0f, 5);
    float* beta = dalloc(4096, 0.1f, 6);
    const int n_layers = 30;
    const long kv_blocks = 64 * 66 + 8;
    float* kv = nullptr;   // one K/V pool per layer: attention streams 127 MB per launch from HBM, as in the engine
    HIP_CHECK(hipMalloc(&kv, (size_t)n_layers * kv_blocks * kKvBlockElems * 4));
    HIP_CHECK(hipMemset(kv, 0, (size_t)n_layers * kv_blocks * kKvBlockElems * 4));
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
    // dense per-row K/V addressing as embed_decode_kernel writes it once per step
    int* drm;
    {
        std::vector<int> hrm((size_t)256 * kRowMetaStride, 0);
        for (int i = 0; i < 256; ++i) {
            int* rm = &hrm[(size_t)i * kRowMetaStride];
            const int slot = hslot[i], pos = hpos[i];
            rm[0] = pos;
            rm[1] = slot;
            rm[kRowMetaWblk] = hbt[slot * 66 + pos / kKvBlockTokens];
            for (int j = 0; j < 66; ++j) rm[kRowMetaBt + j] = hbt[slot * 66 + j];
        }
        HIP_CHECK(hipMalloc(&drm, hrm.size() * 4));
        HIP_CHECK(hipMemcpy(drm, hrm.data(), hrm.size() * 4, hipMemcpyHostToDevice));
    }
    float2* dstats;
    HIP_CHECK(hipMalloc(&dstats, 256 * 64 * sizeof(float2)));
    {   // plausible LayerNorm partials (mean 0, M2 = 16/3 per 16-column tile of uniform(-1, 1) data)
        std::vector<float2> hs(256 * 64, make_float2(0.f, 16.f / 3.f));
        HIP_CHECK(hipMemcpy(dstats, hs.data(), hs.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    long long* dprof;
    HIP_CHECK(hipMalloc(&dprof, 4096 * 8 * 8));
    float* ksp_buf;
    unsigned* ksp_cnt;
    HIP_CHECK(hipMalloc(&ksp_buf, (size_t)kGemmKspTiles * 16 * 256 * 4));
    HIP_CHECK(hipMalloc(&ksp_cnt, (size_t)kGemmKspTiles * 4));
    HIP_CHECK(hipMemset(ksp_cnt, 0, (size_t)kGemmKspTiles * 4));

    const Shape shapes[] = {{"qkv ", 3072, 1024, true, kEpiQkv},
                            {"proj", 1024, 1024, false, kEpiResidual},
                            {"fc  ", 4096, 1024, true, kEpiBiasGelu},
                            {"prj2", 1024, 4096, false, kEpiResidual},
                            {"head", 1088, 1024, false, kEpiBias}};
    const int NREP = getenv("NREP") ? atoi(getenv("NREP")) : 24;   // distinct weight matrices per shape: the stream really comes from HBM (NREP=1: from L2)
    if (!quick)
    for (const Shape& s : shapes) {
        std::vector<float*> wt(NREP);
        float* wsrc = dalloc((size_t)s.K * s.N, 0.05f, 11);
        for (int r = 0; r < NREP; ++r) {
            HIP_CHECK(hipMalloc(&wt[r], (size_t)s.K * s.N * 4));
            launch_pack_wt16(wsrc, s.N, wt[r], s.K, s.N, st);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<Cfg> cfgs;
        cfgs.push_back({1, 16, 1, 0});   // reference for the correctness check
        if (s.ln) {
            for (int prec : {0, 1}) {
                cfgs.push_back({1, 8, 1, prec});
                if (M > 16) cfgs.push_back({2, 8, 1, prec});
                cfgs.push_back({1, 16, 2, prec});
                if (s.N % 48 == 0) cfgs.push_back({1, 16, 3, prec});
                cfgs.push_back({1, 16, 4, prec});
                if (M > 16) cfgs.push_back({2, 16, 2, prec});
                if (M > 16) cfgs.push_back({2, 16, 1, prec});
                if (M > 48) cfgs.push_back({4, 16, 1, prec});   // 64 rows x 16 columns: every weight line has one reader
                if (prec) cfgs.push_back({1, 16, 1, prec});
            }
        } else {
            cfgs.push_back({1, 16, 1, 1});
            if (M > 16) cfgs.push_back({2, 16, 1, 0});
            if (M > 16) cfgs.push_back({2, 16, 1, 1});
            if (M > 48 && s.K == 1024) cfgs.push_back({4, 16, 1, 1});
            if (s.K == 1024 && s.N % 32 == 0) cfgs.push_back({1, 16, 2, 0});
            if (s.K == 1024 && s.N % 32 == 0) cfgs.push_back({1, 16, 2, 1});
        }
        if (M <= 16) {   // one row group: every weight tile has one reader -> non-temporal weight loads
            cfgs.push_back({1, 16, 1, 1, true});
            cfgs.push_back({1, 16, 1, 0, true});
        }
        if (s.K == 4096) {   // K split over 4 workgroups of 4 waves per tile, last arriver combines (bitwise the unsplit result)
            cfgs.push_back({1, 4, 1, 1, true, kGemmKsp});
            cfgs.push_back({1, 4, 1, 0, true, kGemmKsp});
        }
        std::vector<float> ref;
        for (const Cfg& c : cfgs) {
            GemmRowsArgs a{};
            a.X = X; a.xmt = MTT; a.omt = MTT; a.M = M; a.N = s.N; a.K = s.K; a.bias = bias; a.ln_c1 = s.ln ? gamma : nullptr; a.eps = 1e-5f;
            a.out = (s.epi == kEpiResidual) ? hres : out; a.ldo = (s.epi == kEpiQkv) ? H : s.N;
            a.kv_layer = kv; a.row_slot = dslot; a.slot_kvpos = dpos; a.block_tables = dbt; a.max_blocks = 66; a.stats_in = dstats; a.row_meta = drm;
            a.ksp_buf = ksp_buf; a.ksp_cnt = ksp_cnt;
            // correctness: one launch on a fresh output, compared element-wise with the reference configuration
            HIP_CHECK(hipMemcpyAsync(hres, hres0, (size_t)256 * 4096 * 4, hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipMemsetAsync(out, 0, (size_t)256 * 4096 * 4, st));
            a.Wt = wt[0];
            launch_variant(a, s, c, st);
            HIP_CHECK(hipStreamSynchronize(st));
            std::vector<float> got((size_t)256 * 4096);
            HIP_CHECK(hipMemcpy(got.data(), a.out, got.size() * 4, hipMemcpyDeviceToHost));
            double maxd = 0, maxr = 0;
            if (ref.empty()) ref = got;
            for (size_t i = 0; i < got.size(); ++i) {
                maxd = std::max(maxd, (double)fabsf(got[i] - ref[i]));
                maxr = std::max(maxr, (double)fabsf(ref[i]));
            }
            a.stats_out = (s.epi == kEpiResidual && s.N == 1024) ? dstats + 128 * 64 : nullptr;
            int it = 0;
            const float us = time_us(st, 240, [&] {
                a.Wt = wt[it++ % NREP];
                launch_variant(a, s, c, st);
            });
            const int n_grp = (M + 16 * c.mt - 1) / (16 * c.mt);
            const int nwg = ((s.N / (16 * c.ntl) + 7) / 8 * 8) * n_grp;   // (K split: the stamps of the 4 parts of a tile overwrite each other)
            HIP_CHECK(hipMemsetAsync(dprof, 0, (size_t)nwg * 64, st));
            a.prof = dprof;
            a.Wt = wt[5 % NREP];
            launch_variant(a, s, c, st);
            HIP_CHECK(hipStreamSynchronize(st));
            std::vector<long long> hp((size_t)nwg * 8);
            HIP_CHECK(hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost));
            long long t_min = -1, t_max = 0;
            double ph[5] = {0, 0, 0, 0, 0};
            int live = 0;
            std::vector<double> en_us;
            for (int g = 0; g < nwg; ++g) {
                if (hp[g * 8] == 0) continue;
                ++live;
                t_min = t_min < 0 ? hp[g * 8] : std::min(t_min, hp[g * 8]);
                t_max = std::max(t_max, hp[g * 8 + 5]);
            }
            for (int g = 0; g < nwg; ++g) {
                if (hp[g * 8] == 0) continue;
                for (int k = 0; k < 5; ++k) ph[k] += (double)(hp[g * 8 + k + 1] - hp[g * 8 + k]) / live;
                en_us.push_back((double)(hp[g * 8 + 5] - t_min) / 100.0);
            }
            std::sort(en_us.begin(), en_us.end());
            printf("%s M=%d rows/wg=%2d cols/wg=%2d waves=%2d ksp=%d prec=%d nt=%d wgs=%4d : %6.2f us/launch | max|d| vs ref %.2e (max|ref| %.2e) | 10-ns ticks: span %5lld issue %4.0f ln+wait %5.0f mfma %5.0f bar %4.0f epi %4.0f | ends p10 %.2f p50 %.2f max %.2f\n",
                   s.name, M, 16 * c.mt, 16 * c.ntl, c.nw, c.ksp, c.prec, (int)c.nt, nwg * c.ksp, us, maxd, maxr, t_max - t_min, ph[0], ph[1], ph[2], ph[3], ph[4],
                   en_us[en_us.size() / 10], en_us[en_us.size() / 2], en_us.back());
            fflush(stdout);
        }
        for (int r = 0; r < NREP; ++r) HIP_CHECK(hipFree(wt[r]));
        HIP_CHECK(hipFree(wsrc));
    }
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
        float* act = dalloc((size_t)256 * 4096, 0.5f, 22);
        float* att = dalloc((size_t)256 * 1024, 0.5f, 23);
        float* qb = dalloc((size_t)256 * 1024, 0.f, 24);
        Cfg (*policy)(int, const Shape&, int) = r04_policy;
        auto chain = [&](bool attn, int prec) {
            for (int l = 0; l < n_layers; ++l) {
                float* kvl = kv + (size_t)l * kv_blocks * kKvBlockElems;
                GemmRowsArgs a{};
                a.M = M; a.prec = prec; a.eps = 1e-5f; a.X = hres; a.xmt = MTT; a.Wt = wq[l]; a.N = 3072; a.K = 1024; a.bias = bias; a.ln_c1 = gamma;
                a.stats_in = dstats; a.out = qb; a.ldo = 1024; a.kv_layer = kvl; a.row_slot = dslot; a.slot_kvpos = dpos; a.block_tables = dbt; a.row_meta = drm;
                a.max_blocks = 66;
                launch_variant(a, shapes[0], policy(M, shapes[0], prec), st);
                if (attn) launch_paged_attention(qb, kvl, dslot, nullptr, dpos, dbt, 66, att, M, st, MTT, false, drm);
                a = GemmRowsArgs{};
                a.M = M; a.prec = prec; a.X = att; a.xmt = MTT; a.Wt = wp[l]; a.N = 1024; a.K = 1024; a.bias = bias; a.out = hres; a.omt = MTT; a.stats_out = dstats + 128 * 64;
                launch_variant(a, shapes[1], policy(M, shapes[1], prec), st);
                a = GemmRowsArgs{};
                a.M = M; a.prec = prec; a.eps = 1e-5f; a.X = hres; a.xmt = MTT; a.Wt = wf[l]; a.N = 4096; a.K = 1024; a.bias = bias; a.ln_c1 = gamma;
                a.stats_in = dstats; a.out = act; a.omt = MTT;
                launch_variant(a, shapes[2], policy(M, shapes[2], prec), st);
                a = GemmRowsArgs{};
                a.M = M; a.prec = prec; a.X = act; a.xmt = MTT; a.Wt = w2[l]; a.N = 1024; a.K = 4096; a.bias = bias; a.out = hres; a.omt = MTT; a.stats_out = dstats + 128 * 64;
                a.ksp_buf = ksp_buf; a.ksp_cnt = ksp_cnt;
                launch_variant(a, shapes[3], policy(M, shapes[3], prec), st);
            }
        };
        // decode attention alone, by context length (all rows at the same position): intercept = the launch's fixed chain, slope = per
        // 64-token iteration
        for (int pos : {0, 63, 127, 243, 383, 639}) {
            std::vector<int> hp(256, pos);
            std::vector<int> hrm2((size_t)256 * kRowMetaStride, 0);
            for (int i = 0; i < 256; ++i) {
                int* rm = &hrm2[(size_t)i * kRowMetaStride];
                rm[0] = pos;
                rm[1] = hslot[i];
                rm[kRowMetaWblk] = hbt[hslot[i] * 66 + pos / kKvBlockTokens];
                for (int j = 0; j < 66; ++j) rm[kRowMetaBt + j] = hbt[hslot[i] * 66 + j];
            }
            HIP_CHECK(hipMemcpy(drm, hrm2.data(), hrm2.size() * 4, hipMemcpyHostToDevice));
            int l = 0;
            const float us = time_us(st, 120, [&] {
                float* kvl = kv + (size_t)(l++ % n_layers) * kv_blocks * kKvBlockElems;
                launch_paged_attention(qb, kvl, dslot, nullptr, dpos, dbt, 66, att, M, st, MTT, false, drm);
            });
            printf("paged attention alone M=%d context %3d tokens: %.2f us per launch\n", M, pos + 1, us);
        }
        {   // restore the chain's positions
            std::vector<int> hrm2((size_t)256 * kRowMetaStride, 0);
            for (int i = 0; i < 256; ++i) {
                int* rm = &hrm2[(size_t)i * kRowMetaStride];
                rm[0] = hpos[i];
                rm[1] = hslot[i];
                rm[kRowMetaWblk] = hbt[hslot[i] * 66 + hpos[i] / kKvBlockTokens];
                for (int j = 0; j < 66; ++j) rm[kRowMetaBt + j] = hbt[hslot[i] * 66 + j];
            }
            HIP_CHECK(hipMemcpy(drm, hrm2.data(), hrm2.size() * 4, hipMemcpyHostToDevice));
        }
        for (int pol = 0; pol < 2; ++pol) {
            policy = pol ? r04_policy : r03_policy;
            for (int attn : {0, 1})
                for (int prec : {0, 1}) {
                    const float us = time_us(st, 20, [&] { chain(attn != 0, prec); });
                    printf("chain of 30 x (qkv,%s proj, fc, proj2) M=%d shapes=%s prec=%d: %.1f us per layer (%.3f ms per step)\n",
                           attn ? " attention," : "", M, M > 16 || pol ? "r04" : "r03", prec, us / n_layers, us / 1000);
                    fflush(stdout);
                }
            if (M > 16) break;   // the two policies differ only at M <= 16
        }
    }
    return 0;
}
