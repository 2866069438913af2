// Glue kernels of the speaker-conditioning path (see cond_kernels.h).  The heavy lifting (STFT, 1x1 convolutions, linear
// layers, im2col'ed 3x3 convolutions) is launch_gemm_tile; what is here runs once per speaker on a few hundred frames, so the
// kernels are written for clarity and exact fp32 arithmetic, one thread per output element or one workgroup per reduction.
#include "cond_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace aur {

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {   // 256 threads
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int n_in, float* __restrict__ y, int n_out,
                                                       const float* __restrict__ kern, int norig, int nnew, int width) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int j = o / nnew, i = o - j * nnew, klen = 2 * width + norig;
    const float* kr = kern + (long)i * klen;
    const long base = (long)j * norig - width;
    float acc = 0.f;
    for (int k = 0; k < klen; ++k) {
        const long s = base + k;
        if (s >= 0 && s < n_in) acc = fmaf(x[s], kr[k], acc);
    }
    y[o] = acc;
}

__global__ __launch_bounds__(256) void frames_kernel(const float* __restrict__ x, int n, const float* __restrict__ window, int n_fft,
                                                     int hop, int T, float preemph, float* __restrict__ frames) {
    const int t = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_fft) return;
    long i = (long)t * hop + c - n_fft / 2;
    if (i < 0) i = -i;
    if (i >= n) i = 2L * (n - 1) - i;
    float v = x[i];
    if (preemph != 0.f) v = fmaf(preemph, x[i == 0 ? 1 : i - 1], v);
    frames[(long)t * n_fft + c] = v * window[c];
}

__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ spec, int ld_spec, float* __restrict__ pw, int ld_pw, int T,
                                                    int bins) {
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ld_pw) return;
    float v = 0.f;
    if (k < bins) {
        const float re = spec[(long)t * ld_spec + 2 * k], im = spec[(long)t * ld_spec + 2 * k + 1];
        v = re * re + im * im;
    }
    pw[(long)t * ld_pw + k] = v;
}

__global__ __launch_bounds__(256) void logmel_gpt_kernel(const float* __restrict__ mel, int ld_mel, const float* __restrict__ stats,
                                                         float* __restrict__ out, int ld_out, int T, int n_mels) {
    const int t = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= ld_out) return;
    out[(long)t * ld_out + m] = m < n_mels ? logf(fmaxf(mel[(long)t * ld_mel + m], 1e-5f)) / stats[m] : 0.f;
}

__global__ __launch_bounds__(256) void logmel_spk_kernel(const float* __restrict__ mel, int ld_mel, float* __restrict__ img, int H, int W) {
    __shared__ float sh[4];
    const int h = blockIdx.x;
    float s = 0.f;
    for (int w = threadIdx.x; w < W; w += 256) s += logf(mel[(long)w * ld_mel + h] + 1e-6f);
    const float mean = block_sum(s, sh) / (float)W;
    float q = 0.f;
    for (int w = threadIdx.x; w < W; w += 256) {
        const float d = logf(mel[(long)w * ld_mel + h] + 1e-6f) - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, sh) / (float)W + 1e-5f);
    for (int w = threadIdx.x; w < W; w += 256) img[(long)h * W + w] = (logf(mel[(long)w * ld_mel + h] + 1e-6f) - mean) * rstd;
}

__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ P, int ld, const float* __restrict__ bias, int M, int N, int act) {
    const int m = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float v = P[(long)m * ld + n] + (bias ? bias[n] : 0.f);
    if (act == 1) v = fmaxf(v, 0.f);
    P[(long)m * ld + n] = v;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ acc, const float* __restrict__ x, float s, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = fmaf(s, x[i], acc[i]);
}

__global__ __launch_bounds__(256) void group_norm_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
                                                         const float* __restrict__ b, int T, int C, int groups) {
    __shared__ float sh[4];
    const int g = blockIdx.x, cg = C / groups, c0 = g * cg;
    const long n = (long)T * cg;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) s += x[(i / cg) * C + c0 + (i % cg)];
    const float mean = block_sum(s, sh) / (float)n;
    float q = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float d = x[(i / cg) * C + c0 + (i % cg)] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, sh) / (float)n + 1e-5f);
    for (long i = threadIdx.x; i < n; i += 256) {
        const long t = i / cg;
        const int c = c0 + (int)(i % cg);
        y[t * C + c] = (x[t * C + c] - mean) * rstd * w[c] + b[c];
    }
}

// one query per thread, 64 keys per LDS tile (all lanes read the same key row: LDS broadcast), online softmax
__global__ __launch_bounds__(64) void attention_kernel(CondAttn a) {
    __shared__ float ks[64][64], vs[64][64];
    const int h = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    const bool live = i < a.nq;
    float q[64], o[64];
    const float* qp = a.q + (long)(live ? i : 0) * a.ldq + h * a.q_head_stride;
#pragma unroll
    for (int d = 0; d < 64; ++d) {
        q[d] = qp[d];
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < a.nk; j0 += 64) {
        __syncthreads();
        {
            const int j = j0 + threadIdx.x;
            const float* kp = a.k + (long)(j < a.nk ? j : 0) * a.ldk + h * a.k_head_stride;
            const float* vp = a.v + (long)(j < a.nk ? j : 0) * a.ldv + h * a.v_head_stride;
#pragma unroll
            for (int d = 0; d < 64; d += 4) {
                *reinterpret_cast<f32x4*>(&ks[threadIdx.x][d]) = *reinterpret_cast<const f32x4*>(kp + d);
                *reinterpret_cast<f32x4*>(&vs[threadIdx.x][d]) = *reinterpret_cast<const f32x4*>(vp + d);
            }
        }
        __syncthreads();
        const int nj = min(64, a.nk - j0);
        for (int j = 0; j < nj; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s = fmaf(q[d], ks[j][d], s);
            s *= a.scale;
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn), p = expf(s - mn);
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(p, vs[j][d], o[d] * alpha);
            m = mn;
        }
    }
    if (live) {
        float* op = a.out + (long)i * a.ldo + h * 64;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < 64; ++d) op[d] = o[d] * inv;
    }
}

__global__ __launch_bounds__(256) void geglu_kernel(const float* __restrict__ h, int ld_h, float* __restrict__ out, int ld_out, int M,
                                                    int inner) {
    const int m = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ld_out) return;
    float v = 0.f;
    if (j < inner) {
        const float a = h[(long)m * ld_h + j], g = h[(long)m * ld_h + inner + j];
        v = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)) * a;
    }
    out[(long)m * ld_out + j] = v;
}

__global__ __launch_bounds__(256) void rms_norm_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ gamma,
                                                       int M, int C) {
    __shared__ float sh[4];
    const int m = blockIdx.x;
    float q = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) q += x[(long)m * C + c] * x[(long)m * C + c];
    const float nrm = fmaxf(sqrtf(block_sum(q, sh)), 1e-12f);
    const float s = sqrtf((float)C) / nrm;
    for (int c = threadIdx.x; c < C; c += 256) out[(long)m * C + c] = x[(long)m * C + c] * s * gamma[c];
}

__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, int H, int W, int C, int stride, float* __restrict__ cols,
                                                      int ld_cols, int Ho, int Wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // (row, tap, c)
    const long total = (long)Ho * Wo * 9 * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int tap = (int)((idx / C) % 9);
    const long row = idx / (9L * C);
    const int wo = (int)(row % Wo), ho = (int)(row / Wo);
    const int hi = ho * stride + tap / 3 - 1, wi = wo * stride + tap % 3 - 1;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = x[((long)hi * W + wi) * C + c];
    cols[row * ld_cols + tap * C + c] = v;
}

__global__ __launch_bounds__(256) void gather_stride_kernel(const float* __restrict__ x, int H, int W, int C, int stride,
                                                            float* __restrict__ rows, int Ho, int Wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)Ho * Wo * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long row = idx / C;
    const int wo = (int)(row % Wo), ho = (int)(row / Wo);
    rows[idx] = x[((long)(ho * stride) * W + wo * stride) * C + c];
}

__global__ __launch_bounds__(256) void bn_kernel(float* __restrict__ x, long rows, int C, const float* __restrict__ bias,
                                                 const float* __restrict__ scale, const float* __restrict__ shift, int relu_before,
                                                 int relu_after) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int c = (int)(idx % C);
    float v = x[idx] + (bias ? bias[c] : 0.f);
    if (relu_before) v = fmaxf(v, 0.f);
    v = fmaf(v, scale[c], shift[c]);
    if (relu_after) v = fmaxf(v, 0.f);
    x[idx] = v;
}

// column means of [rows][C] in two deterministic stages: partial[chunk][c] over row chunks (grid: C/64 x chunks, 4 row phases per
// workgroup), then the chunks in order
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ y, long rows, int C, int chunks,
                                                              float* __restrict__ partial) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6, ch = blockIdx.y;
    const long per = (rows + chunks - 1) / chunks, r0 = ch * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < C)
        for (long r = r0 + ph; r < r1; r += 4) s += y[r * C + c];
    part[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < C) partial[(long)ch * C + c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void col_mean_final_kernel(const float* __restrict__ partial, int chunks, int C, long rows,
                                                             float* __restrict__ mean) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int ch = 0; ch < chunks; ++ch) s += partial[(long)ch * C + c];
    mean[c] = s / (float)rows;
}
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ mean, int C, int Cr, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ gate) {
    __shared__ float hid[64];
    for (int j = threadIdx.x; j < Cr; j += 256) {
        float s = b1[j];
        for (int c = 0; c < C; ++c) s = fmaf(w1[(long)j * C + c], mean[c], s);
        hid[j] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = b2[c];
        for (int j = 0; j < Cr; ++j) s = fmaf(w2[(long)c * Cr + j], hid[j], s);
        gate[c] = 1.0f / (1.0f + expf(-s));
    }
}
__global__ __launch_bounds__(256) void se_apply_kernel(float* __restrict__ y, const float* __restrict__ r, long n, int C,
                                                       const float* __restrict__ gate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    y[idx] = fmaxf(fmaf(y[idx], gate[idx % C], r[idx]), 0.f);
}

__global__ __launch_bounds__(256) void asp_features_kernel(const float* __restrict__ x, int H, int W, int C, float* __restrict__ feat) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over (w, c, h)
    const long total = (long)W * C * H;
    if (idx >= total) return;
    const int h = (int)(idx % H), c = (int)((idx / H) % C), w = (int)(idx / ((long)H * C));
    feat[idx] = x[((long)h * W + w) * C + c];
}

__global__ __launch_bounds__(256) void asp_pool_kernel(const float* __restrict__ feat, const float* __restrict__ logits, int W, int F,
                                                       float* __restrict__ out) {
    __shared__ float sh[4];
    const int f = blockIdx.x;
    float mx = -INFINITY;
    for (int w = threadIdx.x; w < W; w += 256) mx = fmaxf(mx, logits[(long)w * F + f]);
    mx = block_max(mx, sh);
    float z = 0.f, s1 = 0.f, s2 = 0.f;
    for (int w = threadIdx.x; w < W; w += 256) {
        const float p = expf(logits[(long)w * F + f] - mx), v = feat[(long)w * F + f];
        z += p;
        s1 = fmaf(p, v, s1);
        s2 = fmaf(p, v * v, s2);
    }
    z = block_sum(z, sh);
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
        const float mu = s1 / z;
        out[f] = mu;
        out[F + f] = sqrtf(fmaxf(s2 / z - mu * mu, 1e-5f));
    }
}

__global__ __launch_bounds__(256) void l2_norm_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
    __shared__ float sh[4];
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) q += x[i] * x[i];
    const float inv = 1.0f / fmaxf(sqrtf(block_sum(q, sh)), 1e-12f);
    for (int i = threadIdx.x; i < n; i += 256) out[i] = x[i] * inv;
}

inline unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define AUR_LAUNCH(name, grid, block, ...)                              \
    do {                                                                \
        trace_launch(#name);                                            \
        hipLaunchKernelGGL(name, grid, block, 0, st, __VA_ARGS__);      \
        HIP_CHECK(hipGetLastError());                                   \
    } while (0)

void launch_cond_resample(const float* x, int n_in, float* y, int n_out, const float* kern, int norig, int nnew, int width, hipStream_t st) {
    AUR_REQUIRE(n_in > 0 && n_out > 0 && norig > 0 && nnew > 0, "cond_resample: shape");
    AUR_LAUNCH(resample_kernel, dim3(blocks(n_out)), dim3(256), x, n_in, y, n_out, kern, norig, nnew, width);
}
void launch_cond_frames(const float* x, int n, const float* window, int n_fft, int hop, int T, float preemph, float* frames, hipStream_t st) {
    AUR_REQUIRE(n > n_fft / 2 && T >= 1, "cond_frames: the signal must be longer than half a window (reflect padding)");
    AUR_LAUNCH(frames_kernel, dim3(blocks(n_fft), T), dim3(256), x, n, window, n_fft, hop, T, preemph, frames);
}
void launch_cond_power(const float* spec, int ld_spec, float* pw, int ld_pw, int T, int bins, hipStream_t st) {
    AUR_LAUNCH(power_kernel, dim3(blocks(ld_pw), T), dim3(256), spec, ld_spec, pw, ld_pw, T, bins);
}
void launch_cond_logmel_gpt(const float* mel, int ld_mel, const float* mel_stats, float* out, int ld_out, int T, int n_mels, hipStream_t st) {
    AUR_LAUNCH(logmel_gpt_kernel, dim3(blocks(ld_out), T), dim3(256), mel, ld_mel, mel_stats, out, ld_out, T, n_mels);
}
void launch_cond_logmel_spk(const float* mel, int ld_mel, float* img, int H, int W, hipStream_t st) {
    AUR_LAUNCH(logmel_spk_kernel, dim3(H), dim3(256), mel, ld_mel, img, H, W);
}
void launch_cond_bias_act(float* P, int ld, const float* bias, int M, int N, int act, hipStream_t st) {
    AUR_LAUNCH(bias_act_kernel, dim3(blocks(N), M), dim3(256), P, ld, bias, M, N, act);
}
void launch_cond_add(const float* a, const float* b, float* y, long n, hipStream_t st) {
    AUR_LAUNCH(add_kernel, dim3(blocks(n)), dim3(256), a, b, y, n);
}
void launch_cond_axpy(float* acc, const float* x, float s, long n, hipStream_t st) {
    AUR_LAUNCH(axpy_kernel, dim3(blocks(n)), dim3(256), acc, x, s, n);
}
void launch_cond_group_norm(const float* x, float* y, const float* w, const float* b, int T, int C, int groups, hipStream_t st) {
    AUR_REQUIRE(C % groups == 0, "cond_group_norm: channels per group");
    AUR_LAUNCH(group_norm_kernel, dim3(groups), dim3(256), x, y, w, b, T, C, groups);
}
void launch_cond_attention(const CondAttn& a, hipStream_t st) {
    AUR_REQUIRE(a.nq >= 1 && a.nk >= 1 && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.k_head_stride % 4 == 0 &&
                    a.v_head_stride % 4 == 0,
                "cond_attention: shape / alignment");
    AUR_LAUNCH(attention_kernel, dim3((a.nq + 63) / 64, a.heads), dim3(64), a);
}
void launch_cond_geglu(const float* h, int ld_h, float* out, int ld_out, int M, int inner, hipStream_t st) {
    AUR_LAUNCH(geglu_kernel, dim3(blocks(ld_out), M), dim3(256), h, ld_h, out, ld_out, M, inner);
}
void launch_cond_rms_norm(const float* x, float* out, const float* gamma, int M, int C, hipStream_t st) {
    AUR_LAUNCH(rms_norm_kernel, dim3(M), dim3(256), x, out, gamma, M, C);
}
void launch_cond_im2col3(const float* x, int H, int W, int C, int stride, float* cols, int ld_cols, int Ho, int Wo, hipStream_t st) {
    AUR_REQUIRE(ld_cols >= 9 * C, "cond_im2col3: row pitch");
    AUR_LAUNCH(im2col3_kernel, dim3(blocks((long)Ho * Wo * 9 * C)), dim3(256), x, H, W, C, stride, cols, ld_cols, Ho, Wo);
}
void launch_cond_gather_stride(const float* x, int H, int W, int C, int stride, float* rows, int Ho, int Wo, hipStream_t st) {
    AUR_LAUNCH(gather_stride_kernel, dim3(blocks((long)Ho * Wo * C)), dim3(256), x, H, W, C, stride, rows, Ho, Wo);
}
void launch_cond_bn(float* x, long rows, int C, const float* bias, const float* scale, const float* shift, int relu_before, int relu_after,
                    hipStream_t st) {
    AUR_LAUNCH(bn_kernel, dim3(blocks(rows * C)), dim3(256), x, rows, C, bias, scale, shift, relu_before, relu_after);
}
void launch_cond_se_residual(float* y, const float* r, long rows, int C, int Cr, const float* w1, const float* b1, const float* w2,
                             const float* b2, float* scratch, hipStream_t st) {
    AUR_REQUIRE(Cr <= 64, "cond_se: reduced width");
    // scratch: [C] means, [C] gates, [chunks][C] partial sums
    const int chunks = (int)std::min<long>(128, std::max<long>(1, rows / 64));
    AUR_LAUNCH(col_sum_partial_kernel, dim3((C + 63) / 64, chunks), dim3(256), y, rows, C, chunks, scratch + 2 * C);
    AUR_LAUNCH(col_mean_final_kernel, dim3(blocks(C)), dim3(256), scratch + 2 * C, chunks, C, rows, scratch);
    AUR_LAUNCH(se_gate_kernel, dim3(1), dim3(256), scratch, C, Cr, w1, b1, w2, b2, scratch + C);
    AUR_LAUNCH(se_apply_kernel, dim3(blocks(rows * C)), dim3(256), y, r, rows * C, C, scratch + C);
}
void launch_cond_asp_features(const float* x, int H, int W, int C, float* feat, hipStream_t st) {
    AUR_LAUNCH(asp_features_kernel, dim3(blocks((long)W * C * H)), dim3(256), x, H, W, C, feat);
}
void launch_cond_asp_pool(const float* feat, const float* logits, int W, int F, float* out, hipStream_t st) {
    AUR_LAUNCH(asp_pool_kernel, dim3(F), dim3(256), feat, logits, W, F, out);
}
void launch_cond_l2_norm(const float* x, float* out, int n, hipStream_t st) {
    AUR_LAUNCH(l2_norm_kernel, dim3(1), dim3(256), x, out, n);
}

}  // namespace aur
