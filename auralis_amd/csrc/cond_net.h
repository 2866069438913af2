// Host-side schedule of the speaker-conditioning networks on the GPU (kernels: cond_kernels.h, GEMMs: launch_gemm_tile).
// Follows XTTSv2.get_conditioning_latents (models/xttsv2/XTTSv2.py:409-468): per reference, clip to max_ref_length, speaker
// embedding (16 kHz, pre-emphasis, 64-mel, ResNet-SE, attentive statistics pooling; :312-328), embeddings averaged over the
// references; the references concatenated, clipped to gpt_cond_len, cut into gpt_cond_chunk_len chunks, each chunk -> 80-mel ->
// ConditioningEncoder -> PerceiverResampler, latents averaged over the chunks (:349-407).
// Weight names and layouts: auralis_amd/weights.py: pack_conditioning.
#pragma once
#include <cmath>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "cond_kernels.h"
#include "gpt_kernels.h"
// (included by engine.hip after its DevBuf definition)

namespace aur {

struct CondParams {
    int max_ref_length = 30;       // seconds of each reference that are used at all
    int gpt_cond_len = 6;          // seconds of the concatenated references that feed the latents
    int gpt_cond_chunk_len = 6;    // seconds per chunk
    int sound_norm_refs = 0;
};

class CondNet {
public:
    using WeightFn = std::function<const float*(const std::string&, int64_t)>;
    using HasFn = std::function<bool(const std::string&)>;
    CondNet(WeightFn w, HasFn has, hipStream_t st) : W_(std::move(w)), has_(std::move(has)), st_(st) {}

    // pcm: host pointers, mono float32 at 22 050 Hz; out_cond [32 * 1024], out_spk [512] (host)
    void run(const float* const* pcm, const int* n_samples, int n_refs, const CondParams& p, float* out_cond, float* out_spk) {
        AUR_REQUIRE(n_refs >= 1, "conditioning: at least one reference");
        used_ = 0;
        constexpr int SR = 22050;
        float* spk_acc = alloc(512);
        float* lat_acc = alloc(32 * 1024);
        HIP_CHECK(hipMemsetAsync(spk_acc, 0, 512 * sizeof(float), st_));
        HIP_CHECK(hipMemsetAsync(lat_acc, 0, 32 * 1024 * sizeof(float), st_));
        long total = 0;
        std::vector<int> n_use(n_refs);
        for (int r = 0; r < n_refs; ++r) {
            AUR_REQUIRE(pcm[r] && n_samples[r] > 0, "conditioning: empty reference");
            n_use[r] = std::min<long>(n_samples[r], (long)SR * p.max_ref_length);
            total += n_use[r];
        }
        float* audio = alloc(total);   // the references back to back
        long off = 0;
        for (int r = 0; r < n_refs; ++r) {
            std::vector<float> tmp;
            const float* src = pcm[r];
            if (p.sound_norm_refs) {   // (audio / |audio|.max()) * 0.75, XTTSv2.py:450-451
                float mx = 0.f;
                for (int i = 0; i < n_use[r]; ++i) mx = std::max(mx, std::fabs(src[i]));
                tmp.assign(src, src + n_use[r]);
                for (auto& v : tmp) v = v / mx * 0.75f;
                src = tmp.data();
            }
            HIP_CHECK(hipMemcpyAsync(audio + off, src, (size_t)n_use[r] * sizeof(float), hipMemcpyHostToDevice, st_));
            HIP_CHECK(hipStreamSynchronize(st_));   // (tmp / caller memory is pageable)
            float* emb = speaker_embedding(audio + off, n_use[r]);
            launch_cond_axpy(spk_acc, emb, 1.0f / (float)n_refs, 512, st_);
            off += n_use[r];
        }
        const long n_lat = std::min<long>(total, p.gpt_cond_len > 0 ? (long)SR * p.gpt_cond_len : total);
        const long chunk = (long)SR * p.gpt_cond_chunk_len;
        std::vector<std::pair<long, long>> chunks;
        for (long i = 0; i < n_lat; i += chunk) {
            const long len = std::min(chunk, n_lat - i);
            if ((double)len < SR * 0.33) continue;   // XTTSv2.py:380-381
            chunks.push_back({i, len});
        }
        AUR_REQUIRE(!chunks.empty(), "reference audio shorter than 0.33 s");
        const size_t mark = used_;
        for (auto& c : chunks) {
            used_ = mark;   // chunk workspaces are reused
            float* lat = gpt_latents(audio + c.first, (int)c.second);
            launch_cond_axpy(lat_acc, lat, 1.0f / (float)chunks.size(), 32 * 1024, st_);
        }
        HIP_CHECK(hipMemcpyAsync(out_cond, lat_acc, 32 * 1024 * sizeof(float), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipMemcpyAsync(out_spk, spk_acc, 512 * sizeof(float), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
    }

private:
    WeightFn W_;
    HasFn has_;
    hipStream_t st_;
    std::vector<std::unique_ptr<DevBuf>> pool_;
    size_t used_ = 0;

    // bump allocation out of a list of device buffers that persist between calls (sizes repeat for a given clip length)
    float* alloc(long n) {
        if (used_ == pool_.size()) pool_.emplace_back(new DevBuf());
        DevBuf& b = *pool_[used_++];
        b.ensure((size_t)std::max<long>(n, 1) * sizeof(float));
        return reinterpret_cast<float*>(b.p);
    }
    const float* W(const std::string& n, int64_t numel = -1) { return W_(n, numel); }
    static int pad(int v, int m) { return (v + m - 1) / m * m; }

    // P[M][N] = X[M][K] . W[K][N]   (N, K as padded by the packer)
    float* gemm(const float* X, int ldx, const std::string& wname, int M, int N, int K) {
        float* P = alloc((long)M * N);
        launch_gemm_tile(X, ldx, W(wname, (int64_t)K * N), P, M, N, K, st_);
        return P;
    }

    // mel power spectrogram [T][ld] (ld = 128) of a device signal: frames -> DFT GEMM -> |.|^2 -> filterbank GEMM
    float* mel(const float* x, int n, const char* tag, int n_fft, int hop, float preemph, int& T) {
        const int bins = n_fft / 2 + 1, ncol = pad(2 * bins, 128), kb = pad(bins, 16);
        T = 1 + n / hop;
        float* frames = alloc((long)T * n_fft);
        launch_cond_frames(x, n, W(std::string("cond.win_") + tag, n_fft), n_fft, hop, T, preemph, frames, st_);
        float* spec = gemm(frames, n_fft, std::string("cond.dft_") + tag, T, ncol, n_fft);
        float* pw = alloc((long)T * kb);
        launch_cond_power(spec, ncol, pw, kb, T, bins, st_);
        return gemm(pw, kb, std::string("cond.fb_") + tag, T, 128, kb);
    }

    float* gpt_latents(const float* x, int n) {
        int T = 0;
        float* m = mel(x, n, "gpt", 2048, 256, 0.f, T);
        float* x0 = alloc((long)T * 80);
        launch_cond_logmel_gpt(m, 128, W("cond.mel_stats", 80), x0, 80, T, 80, st_);
        // ConditioningEncoder (latent_encoder.py:209-253): 1x1 conv 80 -> 1024, then attention blocks whose residual is the
        // NORMALISED input (AttentionBlock.forward :196-206)
        float* h = gemm(x0, 80, "cond.enc.init.w", T, 1024, 80);
        launch_cond_bias_act(h, 1024, W("cond.enc.init.b", 1024), T, 1024, 0, st_);
        for (int i = 0; has_("cond.enc." + std::to_string(i) + ".qkv.w"); ++i) {
            const std::string p = "cond.enc." + std::to_string(i) + ".";
            float* xn = alloc((long)T * 1024);
            launch_cond_group_norm(h, xn, W(p + "gn.w", 1024), W(p + "gn.b", 1024), T, 1024, 32, st_);
            float* qkv = gemm(xn, 1024, p + "qkv.w", T, 3072, 1024);
            launch_cond_bias_act(qkv, 3072, W(p + "qkv.b", 3072), T, 3072, 0, st_);
            float* att = alloc((long)T * 1024);
            CondAttn a{};   // channels are head-major: (q | k | v) of head h at columns 192*h (QKVAttention, :95-131)
            a.q = qkv; a.k = qkv + 64; a.v = qkv + 128;
            a.ldq = a.ldk = a.ldv = 3072;
            a.q_head_stride = a.k_head_stride = a.v_head_stride = 192;
            a.out = att; a.ldo = 1024; a.nq = a.nk = T; a.heads = 16;
            a.scale = 0.125f;   // (ch^-1/4 on q) * (ch^-1/4 on k), ch = 64
            launch_cond_attention(a, st_);
            float* pr = gemm(att, 1024, p + "proj.w", T, 1024, 1024);
            launch_cond_bias_act(pr, 1024, W(p + "proj.b", 1024), T, 1024, 0, st_);
            float* hn = alloc((long)T * 1024);
            launch_cond_add(xn, pr, hn, (long)T * 1024, st_);
            h = hn;
        }
        // PerceiverResampler (perceiver_encoder.py:363-442): 32 latents attend to concat(latents, context)
        const int L = 32, inner = 2730, inner_k = pad(inner, 16), ff_n = pad(2 * inner, 128);
        float* lat = alloc((long)L * 1024);
        HIP_CHECK(hipMemcpyAsync(lat, W("cond.per.latents", L * 1024), (size_t)L * 1024 * sizeof(float), hipMemcpyDeviceToDevice, st_));
        for (int li = 0; has_("cond.per." + std::to_string(li) + ".q.w"); ++li) {
            const std::string p = "cond.per." + std::to_string(li) + ".";
            float* ctx = alloc((long)(L + T) * 1024);
            HIP_CHECK(hipMemcpyAsync(ctx, lat, (size_t)L * 1024 * sizeof(float), hipMemcpyDeviceToDevice, st_));
            HIP_CHECK(hipMemcpyAsync(ctx + (long)L * 1024, h, (size_t)T * 1024 * sizeof(float), hipMemcpyDeviceToDevice, st_));
            float* q = gemm(lat, 1024, p + "q.w", L, 512, 1024);
            float* kv = gemm(ctx, 1024, p + "kv.w", L + T, 1024, 1024);
            float* o = alloc((long)L * 512);
            CondAttn a{};
            a.q = q; a.ldq = 512; a.q_head_stride = 64;
            a.k = kv; a.ldk = 1024; a.k_head_stride = 64;
            a.v = kv + 512; a.ldv = 1024; a.v_head_stride = 64;
            a.out = o; a.ldo = 512; a.nq = L; a.nk = L + T; a.heads = 8; a.scale = 0.125f;
            launch_cond_attention(a, st_);
            float* ao = gemm(o, 512, p + "out.w", L, 1024, 512);
            float* lat1 = alloc((long)L * 1024);
            launch_cond_add(ao, lat, lat1, (long)L * 1024, st_);
            float* f1 = gemm(lat1, 1024, p + "ff1.w", L, ff_n, 1024);
            launch_cond_bias_act(f1, ff_n, W(p + "ff1.b", ff_n), L, ff_n, 0, st_);
            float* gg = alloc((long)L * inner_k);
            launch_cond_geglu(f1, ff_n, gg, inner_k, L, inner, st_);
            float* f2 = gemm(gg, inner_k, p + "ff2.w", L, 1024, inner_k);
            launch_cond_bias_act(f2, 1024, W(p + "ff2.b", 1024), L, 1024, 0, st_);
            float* lat2 = alloc((long)L * 1024);
            launch_cond_add(f2, lat1, lat2, (long)L * 1024, st_);
            lat = lat2;
        }
        float* out = alloc((long)L * 1024);
        launch_cond_rms_norm(lat, out, W("cond.per.norm.g", 1024), L, 1024, st_);
        return out;
    }

    // 3x3 convolution (padding 1) of an NHWC image through im2col + GEMM; returns [Ho*Wo][Cout]
    float* conv3(const float* x, int H, int Wd, int C, int stride, const std::string& wname, int Cout, int& Ho, int& Wo) {
        Ho = (H - 1) / stride + 1;
        Wo = (Wd - 1) / stride + 1;
        const int K = pad(9 * C, 16);
        float* cols = alloc((long)Ho * Wo * K);
        if (K != 9 * C) HIP_CHECK(hipMemsetAsync(cols, 0, (size_t)Ho * Wo * K * sizeof(float), st_));
        launch_cond_im2col3(x, H, Wd, C, stride, cols, K, Ho, Wo, st_);
        return gemm(cols, K, wname, Ho * Wo, Cout, K);
    }

    float* speaker_embedding(const float* x22, int n22) {
        const size_t mark = used_;
        // 22 050 -> 16 000 Hz (torchaudio.functional.resample's kernel for 441 -> 320, width 9)
        const int n16 = (int)(((long)n22 * 320 + 440) / 441);
        float* a16 = alloc(n16);
        launch_cond_resample(x22, n22, a16, n16, W("cond.rs_22050_16000", 320 * 459), 441, 320, 9, st_);
        int T = 0;
        float* m = mel(a16, n16, "spk", 512, 160, -0.97f, T);
        const int H0 = 64;
        float* img = alloc((long)H0 * T);
        launch_cond_logmel_spk(m, 128, img, H0, T, st_);
        const std::string s = "cond.spk.";
        int H = H0, Wd = T, Ho, Wo;
        // conv1 (1 -> 32, padded to 64 channels) + bias -> relu -> bn1
        float* cur = conv3(img, H, Wd, 1, 1, s + "conv1.w", 64, Ho, Wo);
        launch_cond_bn(cur, (long)Ho * Wo, 64, W(s + "conv1.b", 64), W(s + "bn1.scale", 64), W(s + "bn1.shift", 64), 1, 0, st_);
        int C = 64;
        const int planes_pad[4] = {64, 64, 128, 256}, n_blocks[4] = {3, 4, 6, 3}, strides[4] = {1, 2, 2, 2};
        float* se_scratch = alloc(130 * 256);   // means, gates, <= 128 partial rows of <= 256 channels
        for (int li = 0; li < 4; ++li)
            for (int bi = 0; bi < n_blocks[li]; ++bi) {
                const std::string p = s + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
                const int stride = bi == 0 ? strides[li] : 1, P = planes_pad[li];
                // conv1 -> relu -> bn1 -> conv2 -> bn2 (SEBasicBlock.forward, hifigan_decoder.py:386-400)
                float* y = conv3(cur, H, Wd, C, stride, p + "conv1.w", P, Ho, Wo);
                launch_cond_bn(y, (long)Ho * Wo, P, nullptr, W(p + "bn1.scale", P), W(p + "bn1.shift", P), 1, 0, st_);
                int H2, W2;
                float* y2 = conv3(y, Ho, Wo, P, 1, p + "conv2.w", P, H2, W2);
                launch_cond_bn(y2, (long)Ho * Wo, P, nullptr, W(p + "bn2.scale", P), W(p + "bn2.shift", P), 0, 0, st_);
                const float* res = cur;
                if (has_(p + "down.w")) {
                    float* rows = alloc((long)Ho * Wo * C);
                    launch_cond_gather_stride(cur, H, Wd, C, stride, rows, Ho, Wo, st_);
                    float* d = gemm(rows, C, p + "down.w", Ho * Wo, P, C);
                    launch_cond_bn(d, (long)Ho * Wo, P, nullptr, W(p + "down.scale", P), W(p + "down.shift", P), 0, 0, st_);
                    res = d;
                }
                const int Cr = P == 64 && li == 0 ? 4 : P / 8;
                launch_cond_se_residual(y2, res, (long)Ho * Wo, P, Cr, W(p + "se1.w", (int64_t)Cr * P), W(p + "se1.b", Cr),
                                        W(p + "se2.w", (int64_t)P * Cr), W(p + "se2.b", P), se_scratch, st_);
                cur = y2;
                H = Ho;
                Wd = Wo;
                C = P;
            }
        // attentive statistics pooling over time on the (C*H)-channel sequence (hifigan_decoder.py:628-646)
        const int F = C * H;   // 256 * 8
        float* feat = alloc((long)Wd * F);
        launch_cond_asp_features(cur, H, Wd, C, feat, st_);
        float* a1 = gemm(feat, F, s + "att0.w", Wd, 128, F);
        launch_cond_bn(a1, Wd, 128, W(s + "att0.b", 128), W(s + "att2.scale", 128), W(s + "att2.shift", 128), 1, 0, st_);
        float* lg = gemm(a1, 128, s + "att3.w", Wd, F, 128);
        launch_cond_bias_act(lg, F, W(s + "att3.b", F), Wd, F, 0, st_);
        float* stats = alloc(2L * F);
        launch_cond_asp_pool(feat, lg, Wd, F, stats, st_);
        float* e = gemm(stats, 2 * F, s + "fc.w", 1, 512, 2 * F);
        launch_cond_bias_act(e, 512, W(s + "fc.b", 512), 1, 512, 0, st_);
        used_ = mark;              // the workspaces above may be reused by the next reference ...
        float* out = alloc(512);   // ... (stream order keeps `e` intact until the normalisation below has run)
        launch_cond_l2_norm(e, out, 512, st_);
        return out;
    }
};

}  // namespace aur
