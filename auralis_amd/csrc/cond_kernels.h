// Speaker conditioning on the GPU: reference audio -> gpt_cond_latent [32][1024] + speaker_embedding [512].
// Replaces the reference's once-per-speaker PyTorch modules (models/xttsv2/XTTSv2.py:312-468: get_speaker_embedding,
// get_gpt_cond_latents, get_conditioning_latents; components/tts/layers/xtts/latent_encoder.py:134-253 ConditioningEncoder,
// perceiver_encoder.py:363-442 PerceiverResampler, hifigan_decoder.py:386-646 ResNetSpeakerEncoder; mel front-ends
// common/utilities.py:9-71 and hifigan_decoder.py:560-600).
// Everything is fp32.  Activations are TIME-MAJOR ([frames][channels]; the ResNet works on NHWC images), so that every
// 1x1 convolution / linear layer / im2col'ed 3x3 convolution is one launch_gemm_tile call (exact-f32 MFMA, gpt_kernels.h) on
// weights stored [K][N] (auralis_amd/weights.py: pack_conditioning pads K to 16 and N to 64 / 128 with zeros); the STFT is a
// GEMM against a [n_fft][2 * bins] cos / -sin matrix.  The remaining kernels here are the small glue around those GEMMs.
#pragma once
#include "common.h"

namespace aur {

// y[j * nnew + i] = sum_k x[j * norig + k - width] * kern[i][k]  (band-limited sinc interpolation, torchaudio's algorithm;
// kern = [nnew][klen], klen = 2 * width + norig, computed by the host in float64)
void launch_cond_resample(const float* x, int n_in, float* y, int n_out, const float* kern, int norig, int nnew, int width, hipStream_t st);
// frames[t][n] = window[n] * xp[t * hop + n], xp = reflect-padded (n_fft / 2 each side) signal; preemph != 0: the signal is
// first replaced by x[i] + preemph * x[i - 1] with one reflected sample in front (PreEmphasis, hifigan_decoder.py:560-580)
void launch_cond_frames(const float* x, int n, const float* window, int n_fft, int hop, int T, float preemph, float* frames, hipStream_t st);
// pw[t][k] = re^2 + im^2 from spec[t][2k], spec[t][2k + 1]; columns k >= bins of pw (ld_pw) are zeroed
void launch_cond_power(const float* spec, int ld_spec, float* pw, int ld_pw, int T, int bins, hipStream_t st);
// gpt mel: out[t][m] = log(max(mel[t][m], 1e-5)) / mel_stats[m]   (XTTSv2.py:386-389)
void launch_cond_logmel_gpt(const float* mel, int ld_mel, const float* mel_stats, float* out, int ld_out, int T, int n_mels, hipStream_t st);
// speaker mel: img[h][w] = instance_norm_over_w(log(mel[w][h] + 1e-6))   (hifigan_decoder.py:620-623), one channel, [H][W]
void launch_cond_logmel_spk(const float* mel, int ld_mel, float* img, int H, int W, hipStream_t st);

// P[m][n] (+)= bias[n], then act (0 none, 1 relu); in place, ld = row pitch
void launch_cond_bias_act(float* P, int ld, const float* bias, int M, int N, int act, hipStream_t st);
// y = a + b (same shape, contiguous)
void launch_cond_add(const float* a, const float* b, float* y, long n, hipStream_t st);
// GroupNorm over [T][C] with `groups` channel groups (statistics over T x C/groups), eps 1e-5
void launch_cond_group_norm(const float* x, float* y, const float* w, const float* b, int T, int C, int groups, hipStream_t st);
// Softmax attention, fp32: out[i][h*dh + d] = sum_j softmax_j(scale * q_i . k_j) v_j[d]; head h of q / k / v starts at
// column h * head_stride of its matrix (row pitches ldq / ldk / ldv), dh = 64
struct CondAttn {
    const float *q, *k, *v;
    int ldq, ldk, ldv, q_head_stride, k_head_stride, v_head_stride;
    float* out;
    int ldo, nq, nk, heads;
    float scale;
};
void launch_cond_attention(const CondAttn& a, hipStream_t st);
// GEGLU (perceiver_encoder.py:334-335): out[m][j] = gelu_erf(h[m][inner + j]) * h[m][j], j < inner; columns inner..ld_out-1 zeroed
void launch_cond_geglu(const float* h, int ld_h, float* out, int ld_out, int M, int inner, hipStream_t st);
// out[m][:] = x[m][:] / max(||x[m]||, 1e-12) * sqrt(C) * gamma   (RMSNorm of the perceiver, :275-276)
void launch_cond_rms_norm(const float* x, float* out, const float* gamma, int M, int C, hipStream_t st);
// acc += s * x   (averages over chunks / references)
void launch_cond_axpy(float* acc, const float* x, float s, long n, hipStream_t st);

// ---- ResNet-SE speaker encoder, NHWC images [H][W][C]
// cols[(ho*Wo + wo)][(ky*3 + kx)*C + c] = x[ho*stride + ky - 1][wo*stride + kx - 1][c] (zero outside); ld_cols >= 9*C
void launch_cond_im2col3(const float* x, int H, int W, int C, int stride, float* cols, int ld_cols, int Ho, int Wo, hipStream_t st);
// rows[(ho*Wo + wo)][c] = x[ho*stride][wo*stride][c]   (1x1 stride-s downsample)
void launch_cond_gather_stride(const float* x, int H, int W, int C, int stride, float* rows, int Ho, int Wo, hipStream_t st);
// y = act_after(scale[c] * act_before(x + bias[c]) + shift[c]); relu_before / relu_after flags; bias may be nullptr; in place
void launch_cond_bn(float* x, long rows, int C, const float* bias, const float* scale, const float* shift, int relu_before, int relu_after, hipStream_t st);
// Squeeze-excite + residual: s = sigmoid(W2 relu(W1 mean_hw(y) + b1) + b2); y = relu(y * s[c] + r)   (hifigan_decoder.py:355-400);
// scratch: 130 * C floats
void launch_cond_se_residual(float* y, const float* r, long rows, int C, int Cr, const float* w1, const float* b1, const float* w2,
                             const float* b2, float* scratch, hipStream_t st);
// feat[w][c*H + h] = x[h][w][c]   (x.reshape(B, C*H, W) of the NCHW tensor, time-major)
void launch_cond_asp_features(const float* x, int H, int W, int C, float* feat, hipStream_t st);
// attentive statistics pooling: wgt = softmax over w of logits[w][f]; out[f] = sum_w x*wgt, out[F + f] = sqrt(max(sum x^2 wgt - mu^2, 1e-5))
void launch_cond_asp_pool(const float* feat, const float* logits, int W, int F, float* out, hipStream_t st);
// e = x / max(||x||, 1e-12)
void launch_cond_l2_norm(const float* x, float* out, int n, hipStream_t st);

}  // namespace aur
