// Launch wrappers for the HiFi-GAN vocoder kernels (vocoder_kernels.hip).
#pragma once
#include "common.h"

namespace aur {

// One masked 1-D convolution evaluated as an implicit GEMM on fp32 MFMA (32x32x2):
//   y[b][v][q] = sum_{ci<Cin} sum_{j<KS} Wp[v][ci][j] * lrelu(x[b][ci][q + j*DIL - padl], slope)
// with x == 0 outside [0, len_b).  `v` indexes "virtual" output channels: for a plain Conv1d v == co;
// for the polyphase form of ConvTranspose1d (stride s, pad p) v = co*s + r and the value lands at
// t = q*s + r - p.  Epilogue adds bias[co], optional per-(b,co) conditioning, optional residual, and
// optionally folds the 3-way multi-receptive-field mean.
struct ConvArgs {
    const float* x;         // [B][Cin][x_stride]
    const float* wp;        // packed [Mtot/MT][Cin][KS][MT]
    const void* wp16;       // fp16 variant: packed [Mtot/MT][Cin/16][KS][MT][16] halves
    const float* bias;      // [Cout] or nullptr
    const float* cond;      // cond[cond_row[b]*cond_stride + co] or nullptr (1x1 speaker conditioning, precomputed)
    const int* cond_row;    // [B] row of the conditioning table per utterance
    long cond_stride;
    const float* res;       // [B][Cout][o_stride] or nullptr (same layout as out)
    float* mrf;             // [B][Cout][o_stride] accumulator for mrf_mode != 0
    float* out;             // [B][Cout][o_stride]
    const int* base_len;    // [B] vocoder frames T' per utterance
    int len_mul;            // valid input length = base_len[b] * len_mul
    int Cin, Mtot, Cout;
    long x_stride, o_stride;       // elements between channel rows
    long x_bstride, o_bstride;     // elements between batch items
    int padl;
    float slope;            // 1.0f = identity
    int ups_s, ups_p;       // ups_s == 0: plain conv
    int mrf_mode;           // 0 none; 1 mrf = v; 2 mrf += v; 3 out = (mrf + v) / 3
    int max_len;            // max over batch of valid input length (grid sizing)
    int B;
    // fp16 kernel only: the c1 -> c2 intermediate of a ResBlock can live in HBM as fp16.  out_act_f16: the epilogue
    // stores (_Float16)lrelu(value, out_slope) -- exactly what the consumer's staging would have produced from the fp32
    // value -- and the consumer sets x_f16 (input rows are halves, already activated; its `slope` is ignored).
    // Strides stay in elements.
    // The fp16 activated tensors are CHANNEL-CHUNK INTERLEAVED: [B][C/16][stride positions][16 channels] halves, i.e. the 16
    // input channels a staging chunk needs for one position are 32 contiguous bytes (two 16-byte loads per position instead
    // of 16 two-byte ones; the producer stores 4 channels = 8 bytes at a time instead of single halves).
    int x_f16, out_act_f16;
    float out_slope;
    // fp16 residual stream inside a ResBlock (round 3).  The stream is stored ACTIVATED: y = fp16(lrelu(x, 0.1)) in the
    // interleaved layout (out_act_f16 + out_slope on the producer: the polyphase transposed conv that opens a stage, and the
    // residual convs of rounds 0 / 1).  The first conv of a round stages y as is (x_f16); the residual conv of the round reads
    // it back with res_f16 and undoes the activation, x = y >= 0 ? y : y * res_unact (res_unact = 1 / slope; the leaky ReLU is
    // invertible, so one 2-byte tensor serves both uses).  res and out may alias (each element is read and then written by the
    // same lane).  This is the precision of the reference's own GPU path, which runs the ResBlocks under fp16 autocast
    // (hifigan_decoder.py:242).  mrf_f16: the running MRF sum (mrf_mode 1 / 2 write it, 2 / 3 read it) is interleaved halves
    // too.  With out_act_f16 the stage output of mrf_mode 3 is stored as fp16(lrelu(mean, out_slope)) as well (what the next
    // stage's transposed conv, or conv_post, stages anyway), otherwise fp32.
    int res_f16;
    float res_unact;
    int mrf_f16;
    // >= 16 zero bytes (16-byte aligned).  With x_f16 it selects the LDS-DMA staged kernel, whose copies read this page for
    // positions outside the utterance.
    const void* zeros;
};

void launch_conv1d(const ConvArgs& a, int KS, int DIL, hipStream_t st);

// One ResBlock round of the fp16 vocoder in ONE launch (the 64- and 32-channel stages, which are HBM-bound as two launches):
//   h = fp16(lrelu(conv1(y) + b1, 0.1))            conv1: k taps, dilation d, input = the activated stream y = fp16(lrelu(x))
//   v = x + conv2(h) + b2                           conv2: k taps, dilation 1; x = y un-activated (read back from the staged window)
//   mrf_mode 0: out = fp16(lrelu(v, 0.1))  (the next round's stream; out must not alias y: neighbouring tiles read y's halo)
//   mrf_mode 1 / 2 / 3: the ResBlock's last round, as ConvArgs (mrf = halves; mode 3 writes e_out = fp16(lrelu(mean, e_slope)))
// h never leaves LDS: per element the round moves 2 B in + 2 B out instead of 10.  Same chunk, tap and MFMA order as the two
// separate kernels, so the results are equal bit for bit.  All tensors are interleaved halves [C/16][stride][16].
struct RoundArgs {
    const void* y;          // [B][C/16][stride][16] halves
    void* out;              // mode 0
    void* mrf;              // modes 1, 2 (read-modify-write), 3 (read)
    void* e_out;            // mode 3
    const void* w1;         // packed fp16 [C/16][KS][C][16] (ConvLayer::wp16 of c1 / c2)
    const void* w2;
    const float* b1;
    const float* b2;
    const int* base_len;
    int len_mul;
    long stride, bstride;   // positions per channel chunk row / elements per batch item (C * stride)
    int C, mrf_mode;
    float e_slope;
    int B, max_len;
    const void* zeros;      // >= 16 zero bytes
};
void launch_resblock_round_f16(const RoundArgs& a, int KS, int DIL, hipStream_t st);
// same contract on fp16 MFMA inputs (fp32 accumulate, fp32 activations in HBM); uses a.wp16
void launch_conv1d_f16(const ConvArgs& a, int KS, int DIL, hipStream_t st);

// z[b][c][j] = interp(interp(latents[b]^T, x4), x24000/22050)[c][j]   (hifigan_decoder.py:787-800)
// out_f16: z is written as interleaved halves [B][C/16][z_stride][16] = fp16 of the same values (strides in elements), the layout the
// LDS-DMA conv kernel stages from
void launch_interp2(const float* lat, long lat_bstride, const int* lat_row, const int* n_lat, const int* base_len, float* z,
                    long z_stride, long z_bstride, int C, int B, int max_len, hipStream_t st, bool out_f16 = false);

// wav[b][t] = tanh(sum_{ci,j} w[ci][j] * lrelu(x[b][ci][t+j-3], slope))   (conv_post, no bias)
void launch_conv_post(const float* x, const float* w, float* wav, const int* base_len, int len_mul, int Cin,
                      long x_stride, long x_bstride, long wav_bstride, float slope, int B, int max_len,
                      hipStream_t st, bool x_f16_act = false);   // x_f16_act: x = fp16(lrelu(.)) already, interleaved halves

// y[b][r] = bias[r] + sum_k W[r][k] * g[b][k]      (1x1 conditioning convs on the speaker embedding)
void launch_gemv_rows(const float* W, const float* bias, const float* g, float* y, int R, int K, int B,
                      long g_bstride, long y_bstride, hipStream_t st);

}  // namespace aur
