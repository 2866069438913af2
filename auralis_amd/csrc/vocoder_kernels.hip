// HiFi-GAN vocoder kernels for gfx950 (MI355X).  fp32 in, fp32 accumulate, exact-f32 MFMA.
//
// Reference arithmetic: src/auralis/models/xttsv2/components/tts/layers/xtts/hifigan_decoder.py
//   ResBlock1.forward 76-91, HifiganGenerator.forward 228-260, HifiDecoder.forward 776-802.
//
// conv1d_mfma_kernel: one workgroup (4 waves) produces a [MT virtual channels] x [NT positions] output
// tile of one utterance.  Input channels are consumed in chunks of CK: the activated, zero-masked
// input window x[CK][NT + (KS-1)*DIL] and the packed weights Wp[CK][KS][MT] are staged in LDS, then
// every (channel pair, tap) is one K=2 step of v_mfma_f32_32x32x2_f32:
//   A[i = lane&31][k = lane>>5] = Wp[ci0+cc+k][j][m*32 + i]       (consecutive lanes -> consecutive LDS words)
//   B[k = lane>>5][n = lane&31] = xs[cc+k][col + j*DIL + n]      (consecutive lanes -> consecutive LDS words)
// Each wave keeps a 64x64 (MT=64) or 32x128 (MT=32) accumulator tile in registers: 4 MFMAs per
// 4 LDS dword reads.  D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "vocoder_kernels.h"

#include <type_traits>

namespace aur {

// Shared epilogue of the MFMA conv kernels: bias, speaker conditioning, residual, MRF fold, masked store.
// D layout of the 32x32 MFMAs: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Epilogue.  Every global read the epilogue needs (bias / conditioning per output row, residual and MRF accumulator
// per element) is issued as a batch of unconditional loads at clamped addresses, then a scheduling fence, then the
// arithmetic and the predicated stores: a predicated load inside the r/n loops made hipcc emit
// load -> s_waitcnt vmcnt(0) -> store 64 times per wave, i.e. 64 serialized memory round trips per tile.
// add[r] = bias[co] + cond[b][co] for the 16 accumulator rows of 32-row tile m (all loads issued together)
template <int MT, bool UPS>
__device__ __forceinline__ void conv_row_adds(const ConvArgs& a, float (&add)[16], int b, int mtile, int m, int hi) {
    int co[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = mtile * MT + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        co[r] = UPS ? v / a.ups_s : v;
        add[r] = 0.f;
    }
    if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = a.bias[co[r]];
    }
    if (a.cond) {
        const float* cb = a.cond + (long)a.cond_row[b] * a.cond_stride;
        float cv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[r] = cb[co[r]];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] += cv[r];
    }
}

typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// plain conv (ups_s == 0): t == q, len_out == n_q
template <int WM, int WN, int MT, int NTW, int MODE>
__device__ __forceinline__ void conv_epilogue_plain(const ConvArgs& a, f32x16 (&acc)[WM][WN], int b, int mtile, int q0, int wv,
                                                    int l31, int hi, int n_q) {
    const long ob = (long)b * a.o_bstride;
    const bool has_res = a.res != nullptr;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        float add[16];
        conv_row_adds<MT, false>(a, add, b, mtile, m, hi);
        // interleaved fp16 tensors: this lane's registers 4g .. 4g+3 are channels c0 + 8g .. +3 of one position, i.e. 8
        // contiguous bytes at [chunk (c0 + 8g)/16][q][(c0 + 8g) % 16]
        const int c0 = mtile * MT + m * 32 + 4 * hi;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int q = q0 + wv * NTW + n * 32 + l31;
            const bool ok = q < n_q;
            const int qc = min(q, n_q - 1);
            const long base = ob + (long)c0 * a.o_stride + qc;
            auto load_h = [&](const void* src, float (&v)[16]) {
                const _Float16* hb = reinterpret_cast<const _Float16*>(src) + ob;
                h16x4v hv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    hv[g] = *reinterpret_cast<const h16x4v*>(hb + ((long)((c0 + 8 * g) >> 4) * a.o_stride + qc) * 16 + ((c0 + 8 * g) & 15));
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = (float)hv[r >> 2][r & 3];
            };
            float rv[16], mv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            if (has_res && a.res_f16) {
                load_h(a.res, rv);
            } else if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = a.res[base + (long)((r & 3) + 8 * (r >> 2)) * a.o_stride];
            }
            if (MODE >= 2) {
                if (a.mrf_f16) {
                    load_h(a.mrf, mv);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mv[r] = a.mrf[base + (long)((r & 3) + 8 * (r >> 2)) * a.o_stride];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // all loads of this 32x32 tile are in flight before the first use
            if (ok) {
                if (has_res && a.res_f16) {   // the stream is stored activated: undo the (invertible) leaky ReLU
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = rv[r] < 0.f ? rv[r] * a.res_unact : rv[r];
                }
                float val[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) val[r] = acc[m][n][r] + add[r] + rv[r];
                if (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) val[r] += mv[r];
                }
                auto store_h = [&](void* dst, float sl) {
                    _Float16* hb = reinterpret_cast<_Float16*>(dst) + ob;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = c0 + 8 * g;
                        const h16x4v hv = {(_Float16)lrelu(val[4 * g], sl), (_Float16)lrelu(val[4 * g + 1], sl),
                                           (_Float16)lrelu(val[4 * g + 2], sl), (_Float16)lrelu(val[4 * g + 3], sl)};
                        *reinterpret_cast<h16x4v*>(hb + ((long)(c >> 4) * a.o_stride + q) * 16 + (c & 15)) = hv;
                    }
                };
                if (MODE == 3) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) val[r] = (mv[r] + val[r]) / 3.0f;
                }
                if ((MODE == 0 || MODE == 3) && a.out_act_f16) {
                    store_h(a.out, a.out_slope);
                } else if ((MODE == 1 || MODE == 2) && a.mrf_f16) {
                    store_h(a.mrf, 1.0f);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long off = base + (long)((r & 3) + 8 * (r >> 2)) * a.o_stride;
                        if (MODE == 0) a.out[off] = val[r];
                        else if (MODE == 1 || MODE == 2) a.mrf[off] = val[r];
                        else a.out[off] = val[r];
                    }
                }
            }
        }
    }
}

// The same epilogue when EVERY tensor it touches is interleaved halves (the default fp16 vocoder: residual stream, MRF sum and the
// stage outputs; ConvArgs::res_f16 / mrf_f16 / out_act_f16) -- the arithmetic of conv_epilogue_plain element for element, written
// so that a 32 x 32 tile needs its 16 row constants, eight packed loads and four values at a time: the 512-position workgroups of
// eight waves run two per CU, i.e. at 128 registers per wave, where the general form (fp32 and fp16 branches side by side) spills.
template <int WM, int WN, int MT, int NTW, int MODE>
__device__ __forceinline__ void conv_epilogue_plain_h(const ConvArgs& a, f32x16 (&acc)[WM][WN], int b, int mtile, int q0, int wv,
                                                      int l31, int hi, int n_q) {
    const long ob = (long)b * a.o_bstride;
    const bool has_res = a.res != nullptr;
    const _Float16* rb = reinterpret_cast<const _Float16*>(a.res) + ob;
    _Float16* mb = reinterpret_cast<_Float16*>(a.mrf) + ob;
    _Float16* dstb = (MODE == 0 || MODE == 3) ? reinterpret_cast<_Float16*>(a.out) + ob : mb;
    const float dsl = (MODE == 0 || MODE == 3) ? a.out_slope : 1.0f;
    const float unact = a.res_unact;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        float add[16];
        conv_row_adds<MT, false>(a, add, b, mtile, m, hi);
        const int c0 = mtile * MT + m * 32 + 4 * hi;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int q = q0 + wv * NTW + n * 32 + l31;
            const bool ok = q < n_q;
            const int qc = min(q, n_q - 1);
            long off[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) off[g] = ((long)((c0 + 8 * g) >> 4) * a.o_stride + qc) * 16 + ((c0 + 8 * g) & 15);
            h16x4v rh[4], mh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                rh[g] = h16x4v{0, 0, 0, 0};
                mh[g] = h16x4v{0, 0, 0, 0};
            }
            if (has_res) {
#pragma unroll
                for (int g = 0; g < 4; ++g) rh[g] = *reinterpret_cast<const h16x4v*>(rb + off[g]);
            }
            if (MODE >= 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) mh[g] = *reinterpret_cast<const h16x4v*>(mb + off[g]);
            }
            __builtin_amdgcn_sched_barrier(0);   // all loads of this 32x32 tile are in flight before the first use
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    h16x4v o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float rv = has_res ? (float)rh[g][k] : 0.f;
                        if (has_res) rv = rv < 0.f ? rv * unact : rv;   // the stream is stored activated: undo the (invertible) leaky ReLU
                        float val = acc[m][n][4 * g + k] + add[4 * g + k] + rv;
                        if (MODE == 2) val += (float)mh[g][k];
                        if (MODE == 3) val = ((float)mh[g][k] + val) / 3.0f;
                        o[k] = (_Float16)lrelu(val, dsl);
                    }
                    *reinterpret_cast<h16x4v*>(dstb + off[g]) = o;
                }
            }
        }
    }
}

// polyphase transposed conv: virtual row v = co*s + phase lands at t = q*s + phase - p (no residual / MRF on these layers).
// A lane's registers 4g .. 4g+3 are four CONSECUTIVE virtual rows at one position q:
//   s = 8 (k 16, p 4): four consecutive phases of one output channel -> times 8q + 4*hi - 4 .. +3, one aligned 16-byte store;
//       the two half-waves of a position cover 32 contiguous bytes, a wave instruction 1 KiB per channel group
//   s = 2 (k 4, p 1): two channels x two phases -> times 2q - 1, 2q of each, one 8-byte store per channel
// (round 2 scattered single floats at a stride of s elements: 1.1 TB/s on these four layers).  Vectors that straddle the
// ends of the utterance (s = 2: q = 0 and q = len_in) fall back to scalar stores.
template <int WM, int WN, int MT, int NTW>
__device__ __forceinline__ void conv_epilogue_ups(const ConvArgs& a, f32x16 (&acc)[WM][WN], int b, int mtile, int q0, int wv,
                                                  int l31, int hi, int len_in, int n_q) {
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));   // times 2q - 1, 2q: 4-byte aligned only
    const long ob = (long)b * a.o_bstride;
    const int len_out = len_in * a.ups_s;
    if (a.out_act_f16) {
        // fp16(lrelu(.)) into the interleaved layout [C/16][t][16]: the lane's 4 * WM register groups are 4 * WM adjacent
        // channels (s = 8) or, together with the lane 32 places away, 16 adjacent channels (s = 2) of one output time
        _Float16* hb = reinterpret_cast<_Float16*>(a.out) + ob;
        float add[WM][16];
#pragma unroll
        for (int m = 0; m < WM; ++m) conv_row_adds<MT, true>(a, add[m], b, mtile, m, hi);
        const float sl = a.out_slope;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int q = q0 + wv * NTW + n * 32 + l31;
            if (q >= n_q) continue;   // (both lanes of an exchange pair share q)
            if (a.ups_s == 8 && a.ups_p == 4) {
                // rows 8g + 4hi + r of the 32-row tile m: channel mtile*MT/8 + 4m + g, time 8q + 4hi - 4 + r
                const int cb = mtile * (MT / 8);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 8 * q + 4 * hi - 4 + r;
                    if (t < 0 || t >= len_out) continue;
                    _Float16* dst = hb + ((long)(cb >> 4) * a.o_stride + t) * 16 + (cb & 15);
#pragma unroll
                    for (int m = 0; m < WM; ++m) {
                        const h16x4v hv = {(_Float16)lrelu(acc[m][n][r] + add[m][r], sl), (_Float16)lrelu(acc[m][n][4 + r] + add[m][4 + r], sl),
                                           (_Float16)lrelu(acc[m][n][8 + r] + add[m][8 + r], sl), (_Float16)lrelu(acc[m][n][12 + r] + add[m][12 + r], sl)};
                        *reinterpret_cast<h16x4v*>(dst + 4 * m) = hv;
                    }
                }
            } else if (a.ups_s == 2 && a.ups_p == 1) {
                // rows 8g + 4hi + 2c + ph: channel mtile*MT/2 + 16m + 4g + 2hi + c, time 2q - 1 + ph.  The half-wave hi keeps
                // phase hi of all four channels 4g .. 4g+3 and trades its other phase with lane ^ 32.
                const int t = 2 * q - 1 + hi;
                const bool tok = t >= 0 && t < len_out;
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
                    h16x8 o[2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        h16x2v mine, give;   // channels c = 0, 1 at my phase / at the partner's phase
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float v0 = lrelu(acc[m][n][4 * g + 2 * c] + add[m][4 * g + 2 * c], sl);
                            const float v1 = lrelu(acc[m][n][4 * g + 2 * c + 1] + add[m][4 * g + 2 * c + 1], sl);
                            mine[c] = (_Float16)(hi ? v1 : v0);
                            give[c] = (_Float16)(hi ? v0 : v1);
                        }
                        const int got_i = __shfl_xor(__builtin_bit_cast(int, give), 32);
                        const h16x2v got = __builtin_bit_cast(h16x2v, got_i);
                        // channels 4g + {0,1} come from the hi = 0 lane, 4g + {2,3} from the hi = 1 lane
                        const h16x2v lo2 = hi ? got : mine, hi2 = hi ? mine : got;
                        o[g >> 1][4 * (g & 1) + 0] = lo2[0]; o[g >> 1][4 * (g & 1) + 1] = lo2[1];
                        o[g >> 1][4 * (g & 1) + 2] = hi2[0]; o[g >> 1][4 * (g & 1) + 3] = hi2[1];
                    }
                    if (tok) {
                        const int cb = mtile * (MT / 2) + 16 * m;
                        _Float16* dst = hb + ((long)(cb >> 4) * a.o_stride + t) * 16;
                        *reinterpret_cast<h16x8*>(dst) = o[0];
                        *reinterpret_cast<h16x8*>(dst + 8) = o[1];
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int v = mtile * MT + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, co = v / a.ups_s;
                        const int t = q * a.ups_s + (v - co * a.ups_s) - a.ups_p;
                        if (t >= 0 && t < len_out)
                            hb[((long)(co >> 4) * a.o_stride + t) * 16 + (co & 15)] = (_Float16)lrelu(acc[m][n][r] + add[m][r], sl);
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        float add[16];
        conv_row_adds<MT, true>(a, add, b, mtile, m, hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int v0 = mtile * MT + m * 32 + 8 * g + 4 * hi;   // first of the lane's four consecutive virtual rows
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int q = q0 + wv * NTW + n * 32 + l31;
                if (q >= n_q) continue;
                if (a.ups_s == 8 && a.ups_p == 4) {
                    const int co = v0 >> 3, t0 = q * 8 + (v0 & 7) - 4;   // (v0 & 7) is 0 or 4: the vector is all inside or all outside
                    if (t0 >= 0 && t0 + 3 < len_out) {
                        const f32x4 o = {acc[m][n][4 * g] + add[4 * g], acc[m][n][4 * g + 1] + add[4 * g + 1],
                                         acc[m][n][4 * g + 2] + add[4 * g + 2], acc[m][n][4 * g + 3] + add[4 * g + 3]};
                        *reinterpret_cast<f32x4*>(a.out + ob + (long)co * a.o_stride + t0) = o;
                    }
                } else if (a.ups_s == 2 && a.ups_p == 1) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int co = (v0 >> 1) + c, t0 = 2 * q - 1;
                        float* dst = a.out + ob + (long)co * a.o_stride;
                        const float x0 = acc[m][n][4 * g + 2 * c] + add[4 * g + 2 * c], x1 = acc[m][n][4 * g + 2 * c + 1] + add[4 * g + 2 * c + 1];
                        if (t0 >= 0 && t0 + 1 < len_out) {
                            *reinterpret_cast<f32x2u*>(dst + t0) = f32x2u{x0, x1};
                        } else {
                            if (t0 >= 0 && t0 < len_out) dst[t0] = x0;
                            if (t0 + 1 >= 0 && t0 + 1 < len_out) dst[t0 + 1] = x1;
                        }
                    }
                } else {   // any other (stride, padding): element by element
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int v = v0 + r, co = v / a.ups_s;
                        const int t = q * a.ups_s + (v - co * a.ups_s) - a.ups_p;
                        if (t >= 0 && t < len_out) a.out[ob + (long)co * a.o_stride + t] = acc[m][n][4 * g + r] + add[4 * g + r];
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int MT, int NTW>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[WM][WN], int b, int mtile, int q0, int wv,
                                              int l31, int hi, int len_in, int n_q) {
    if (a.ups_s) {
        conv_epilogue_ups<WM, WN, MT, NTW>(a, acc, b, mtile, q0, wv, l31, hi, len_in, n_q);
    } else if (a.mrf_mode == 0) {
        conv_epilogue_plain<WM, WN, MT, NTW, 0>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
    } else if (a.mrf_mode == 1) {
        conv_epilogue_plain<WM, WN, MT, NTW, 1>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
    } else if (a.mrf_mode == 2) {
        conv_epilogue_plain<WM, WN, MT, NTW, 2>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
    } else {
        conv_epilogue_plain<WM, WN, MT, NTW, 3>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
    }
}

// XCD-aware tile order shared by both conv kernels (see conv1d_mfma_kernel).
__device__ __forceinline__ void conv_tile_order(int& mtile, int& ttile) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int L = blockIdx.x + gx * blockIdx.y;
    const int g8 = (gx / 8) * 8;
    if (gy == 1) {
        mtile = 0;
        ttile = L;
    } else if (L < g8 * gy) {
        const int xcd = L & 7, slot = L >> 3;
        mtile = slot % gy;
        ttile = (slot / gy) * 8 + xcd;
    } else {
        const int r = L - g8 * gy;
        mtile = r % gy;
        ttile = g8 + r / gy;
    }
}

template <int KS, int DIL, int MT, int CK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void conv1d_mfma_kernel(ConvArgs a) {
    constexpr int WM = MT / 32;            // 32-row tiles per wave
    constexpr int WN = (MT == 64) ? 2 : 4; // 32-col tiles per wave
    constexpr int NTW = 32 * WN;
    constexpr int NT = 4 * NTW;
    constexpr int HALO = (KS - 1) * DIL;
    constexpr int XROW = NT + HALO;
    __shared__ __attribute__((aligned(16))) float xs[CK][XROW];
    // weights: CK*KS rows of MT floats, rounded up to whole 256-thread float4 passes so the staging stores need no predicate
    constexpr int WI_ = (CK * KS * MT / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float ws_raw[WI_ * 1024];
    float (*ws)[MT] = reinterpret_cast<float (*)[MT]>(ws_raw);

    const int b = blockIdx.z;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest), so the
    // gridDim.y co-tiles that read the SAME input window are mapped to ids that are 8 apart => same XCD/L2, adjacent
    // in time.  Pure speed choice; any mapping is correct.
    int mtile, ttile;
    conv_tile_order(mtile, ttile);
    const int q0 = ttile * NT;
    const int len_in = a.base_len[b] * a.len_mul;
    const int n_q = a.ups_s ? len_in + 1 : len_in;   // polyphase needs q == len_in for the tail phases
    if (q0 >= n_q) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const float* xb = a.x + (long)b * a.x_bstride;
    const float slope = a.slope;
    const float4* wsrc_tile = reinterpret_cast<const float4*>(a.wp + (long)mtile * a.Cin * KS * MT);

    // Software pipeline over channel chunks: the global loads of chunk i+1 are issued before the MFMA loop of chunk i
    // and parked in registers; they are written to LDS after the loop (single LDS buffer, two barriers per chunk).
    constexpr int XI = (XROW + 255) / 256;
    constexpr int N4 = CK * KS * MT / 4;
    constexpr int WI = (N4 + 255) / 256;
    float xv[CK][XI];
    // two half-size register arrays: hipcc left a 6-entry float4 array in scratch (112 B) for the k=11 kernels
    constexpr int WH = (WI + 1) / 2;
    float4 wva[WH], wvb[WH];
    auto load_chunk = [&](int ci0) {
        const float4* src = wsrc_tile + (long)ci0 * KS * MT / 4;
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const float* xr = xb + (long)(ci0 + c) * a.x_stride;
#pragma unroll
            for (int it = 0; it < XI; ++it) {
                // unconditional loads (clamped address): a predicated load makes hipcc drain vmcnt(0) in the middle
                // of the prefetch block; masking happens when the value is written to LDS
                const int t = q0 - a.padl + tid + it * 256;
                xv[c][it] = xr[min(max(t, 0), len_in - 1)];
            }
        }
#pragma unroll
        for (int it = 0; it < WH; ++it) wva[it] = src[min(tid + it * 256, N4 - 1)];
#pragma unroll
        for (int it = WH; it < WI; ++it) wvb[it - WH] = src[min(tid + it * 256, N4 - 1)];
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
            for (int it = 0; it < XI; ++it) {
                const int i = tid + it * 256;
                const int t = q0 - a.padl + i;
                if (i < XROW) xs[c][i] = (t >= 0 && t < len_in) ? lrelu(xv[c][it], slope) : 0.f;
            }
        float4* dst = reinterpret_cast<float4*>(ws_raw);
#pragma unroll
        for (int it = 0; it < WH; ++it) dst[tid + it * 256] = wva[it];
#pragma unroll
        for (int it = WH; it < WI; ++it) dst[tid + it * 256] = wvb[it - WH];
    };

    load_chunk(0);
    for (int ci0 = 0; ci0 < a.Cin; ci0 += CK) {
        __builtin_amdgcn_sched_barrier(0);   // loads stay above, LDS stores below (hipcc sinks loads otherwise)
        __syncthreads();                     // previous chunk fully consumed
        store_chunk();
        __syncthreads();
        if (ci0 + CK < a.Cin) load_chunk(ci0 + CK);
        __builtin_amdgcn_sched_barrier(0);   // prefetch is in flight before the first MFMA
        // ---- MFMA over (channel pair, tap)
#pragma unroll 2
        for (int cc = 0; cc < CK; cc += 2) {
            const float* xrow = &xs[cc + hi][wv * NTW + l31];
            const float* wrow = &ws[(cc + hi) * KS][l31];
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                float av[WM], bv[WN];
#pragma unroll
                for (int m = 0; m < WM; ++m) av[m] = wrow[j * MT + m * 32];
#pragma unroll
                for (int n = 0; n < WN; ++n) bv[n] = xrow[j * DIL + n * 32];
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    conv_epilogue<WM, WN, MT, NTW>(a, acc, b, mtile, q0, wv, l31, hi, len_in, n_q);
}

template <int KS, int DIL, int MT, int CK>
static void launch_conv_t(const ConvArgs& a, hipStream_t st) {
    constexpr int NT = (MT == 64) ? 256 : 512;
    AUR_REQUIRE(a.Cin % CK == 0, "conv: Cin % CK");
    AUR_REQUIRE(!a.x_f16 && !a.out_act_f16, "conv: fp16 tensors need the fp16 kernel");
    AUR_REQUIRE(a.Mtot % MT == 0, "conv: Mtot % MT");
    const int n_q = a.ups_s ? a.max_len + 1 : a.max_len;
    dim3 grid((n_q + NT - 1) / NT, a.Mtot / MT, a.B);
    trace_launch("conv1d_mfma_kernel");
    hipLaunchKernelGGL((conv1d_mfma_kernel<KS, DIL, MT, CK>), grid, dim3(256), 0, st, a);
}

template <int KS, int DIL, int CK>
static void launch_conv_mt(const ConvArgs& a, hipStream_t st) {
    if (a.Mtot % 64 == 0)
        launch_conv_t<KS, DIL, 64, CK>(a, st);
    else
        launch_conv_t<KS, DIL, 32, CK>(a, st);
}

void launch_conv1d(const ConvArgs& a, int KS, int DIL, hipStream_t st) {
    const int key = KS * 16 + DIL;
    switch (key) {
        case 2 * 16 + 1: launch_conv_mt<2, 1, 16>(a, st); break;
        case 3 * 16 + 1: launch_conv_mt<3, 1, 16>(a, st); break;
        case 3 * 16 + 3: launch_conv_mt<3, 3, 16>(a, st); break;
        case 3 * 16 + 5: launch_conv_mt<3, 5, 16>(a, st); break;
        case 7 * 16 + 1: launch_conv_mt<7, 1, 8>(a, st); break;
        case 7 * 16 + 3: launch_conv_mt<7, 3, 8>(a, st); break;
        case 7 * 16 + 5: launch_conv_mt<7, 5, 8>(a, st); break;
        case 11 * 16 + 1: launch_conv_mt<11, 1, 4>(a, st); break;
        case 11 * 16 + 3: launch_conv_mt<11, 3, 4>(a, st); break;
        case 11 * 16 + 5: launch_conv_mt<11, 5, 4>(a, st); break;
        default: throw InvalidArgument("launch_conv1d: unsupported (kernel,dilation)");
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// fp16-input / fp32-accumulate variant (v_mfma_f32_32x32x16_f16): same tiling and epilogue, activations stay fp32 in
// HBM and are rounded to fp16 (RN) when they are staged into LDS.  One MFMA consumes 16 input channels of one tap:
//   A[i = lane&31][k = 8*(lane>>5)+e] = W[co][ci0 + k]   (LDS rows [tap*MT + co][16 ch], 48-B row pitch)
//   B[k][n = lane&31]                 = x[ci0 + k][t]     (LDS rows [t][16 ch], 48-B row pitch => conflict-free b128)

// XH: the input tensor is fp16 in HBM and already activated (ConvArgs::x_f16), staged without conversion.
// Staging is synchronous (load -> barrier -> LDS write -> barrier -> MFMAs) at 3 waves per SIMD; the software-pipelined and
// 512-position forms measured in round 2 (92-111 ms against 94-103 per batch) are gone from the source.
// MT = 64 (a 64 x 64 accumulator tile per wave next to the staging registers) needs more than the 168 registers of three waves per
// SIMD: round 5 compiled it at three and spilled 32 VGPRs per lane to scratch; those instantiations now run two
// waves per SIMD.  None of them is on the default path any more (conv_pre moved to the LDS-DMA kernel): they serve the A/B switches
// (AUR_CONV_DMA=0, AUR_XT_F16=0) and the fp32-input parity entry point.
template <int KS, int DIL, int MT, bool XH>
__global__ __launch_bounds__(256, MT == 64 ? 2 : 3) void conv1d_mfma_f16_kernel(ConvArgs a) {
    constexpr int CK = 16;
    constexpr int WM = MT / 32;
    constexpr int WN = (MT == 64) ? 2 : 4;
    constexpr int NTW = 32 * WN;
    constexpr int NT = 4 * NTW;
    constexpr int HALO = (KS - 1) * DIL;
    constexpr int XROW = NT + HALO;
    constexpr int RS = 24;                                  // halves per LDS row: 16 data + 8 pad (48 B)
    constexpr int XI = (XROW + 255) / 256;
    constexpr int NW = KS * MT * 2;                         // 16-byte weight pieces per chunk
    constexpr int WI = (NW + 255) / 256;
    static_assert(WI <= 6, "weight staging registers");
    __shared__ __attribute__((aligned(16))) _Float16 xs[XROW * RS];
    __shared__ __attribute__((aligned(16))) _Float16 ws[WI * 128 * RS];   // >= KS*MT rows, whole 256-piece passes

    const int b = blockIdx.z;
    int mtile, ttile;
    conv_tile_order(mtile, ttile);
    const int q0 = ttile * NT;
    const int len_in = a.base_len[b] * a.len_mul;
    const int n_q = a.ups_s ? len_in + 1 : len_in;
    if (q0 >= n_q) return;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const float* xb = a.x + (long)b * a.x_bstride;
    const _Float16* xhb = reinterpret_cast<const _Float16*>(a.x) + (long)b * a.x_bstride;
    const float slope = a.slope;
    const uint4* wsrc_tile = reinterpret_cast<const uint4*>(a.wp16) + (long)mtile * (a.Cin / CK) * NW;

    float xv[XH ? 1 : XI][XH ? 1 : CK];
    h16x8 xlo[XH ? XI : 1], xhh[XH ? XI : 1];   // XH: the 16 channels of one position = 32 contiguous bytes (interleaved layout)
    uint4 w0, w1, w2, w3, w4, w5;
    w0 = w1 = w2 = w3 = w4 = w5 = uint4{0, 0, 0, 0};
#define AUR_WLD(i, reg) \
    if constexpr ((i) < WI) reg = src[min(tid + (i) * 256, NW - 1)];
#define AUR_WST(i, reg)                                                                     \
    if constexpr ((i) < WI) {                                                                \
        const int p = tid + (i) * 256;                                                      \
        *reinterpret_cast<uint4*>(&ws[(p >> 1) * RS + (p & 1) * 8]) = reg;                  \
    }
    auto load_chunk = [&](int ci0) {
        const uint4* src = wsrc_tile + (long)(ci0 / CK) * NW;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int t = q0 - a.padl + tid + it * 256;
            const int tc = min(max(t, 0), len_in - 1);
            if constexpr (XH) {
                const _Float16* p = xhb + ((long)(ci0 >> 4) * a.x_stride + tc) * 16;
                xlo[it] = *reinterpret_cast<const h16x8*>(p);
                xhh[it] = *reinterpret_cast<const h16x8*>(p + 8);
            } else {
#pragma unroll
                for (int c = 0; c < CK; ++c) xv[it][c] = xb[(long)(ci0 + c) * a.x_stride + tc];
            }
        }
        AUR_WLD(0, w0) AUR_WLD(1, w1) AUR_WLD(2, w2) AUR_WLD(3, w3) AUR_WLD(4, w4) AUR_WLD(5, w5)
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int i = tid + it * 256;
            const int t = q0 - a.padl + i;
            const bool ok = (t >= 0 && t < len_in);
            h16x8 lo, hh;
            if constexpr (XH) {
                const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                lo = ok ? xlo[it] : zero;
                hh = ok ? xhh[it] : zero;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    lo[c] = (_Float16)(ok ? lrelu(xv[it][c], slope) : 0.f);
                    hh[c] = (_Float16)(ok ? lrelu(xv[it][c + 8], slope) : 0.f);
                }
            }
            if (i < XROW) {
                *reinterpret_cast<h16x8*>(&xs[i * RS]) = lo;
                *reinterpret_cast<h16x8*>(&xs[i * RS + 8]) = hh;
            }
        }
        AUR_WST(0, w0) AUR_WST(1, w1) AUR_WST(2, w2) AUR_WST(3, w3) AUR_WST(4, w4) AUR_WST(5, w5)
    };

    for (int ci0 = 0; ci0 < a.Cin; ci0 += CK) {
        load_chunk(ci0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        store_chunk();
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        const _Float16* xbase = &xs[(wv * NTW + l31) * RS + 8 * hi];
        const _Float16* wbase = &ws[l31 * RS + 8 * hi];
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            h16x8 av[WM], bv[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) av[m] = *reinterpret_cast<const h16x8*>(wbase + (j * MT + m * 32) * RS);
#pragma unroll
            for (int n = 0; n < WN; ++n) bv[n] = *reinterpret_cast<const h16x8*>(xbase + (j * DIL + n * 32) * RS);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }
#undef AUR_WLD
#undef AUR_WST
    conv_epilogue<WM, WN, MT, NTW>(a, acc, b, mtile, q0, wv, l31, hi, len_in, n_q);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA staged variant for the fp16 interleaved inputs (every ResBlock conv of the fp16 vocoder).  Same tiling, fragments
// and epilogue as conv1d_mfma_f16_kernel<.., XH = true, ..>; what changes is how a 16-channel chunk gets into LDS:
//   * `global_load_lds_dwordx4`: each wave instruction copies 64 x 16 B straight from global memory to 1 KiB of LDS (no staging
//     registers, no ds_write pass -- `ds_write_b128` costs 13 cycles per wave instruction against 4 for the matching read);
//   * NBUF chunk buffers: the copies of chunk c + NBUF - 1 are in flight while the MFMAs of chunk c run, one barrier per chunk
//     (the register-staged kernel runs load -> barrier -> LDS write -> barrier -> MFMA per chunk, and tools/conv_diag shows
//     the three phases adding up instead of overlapping);
//   * a DMA lands lane-linear, so the LDS rows are unpadded 32-byte rows ([tap][channel] for the weights, [position] for the
//     input window) and the two 16-byte halves of row i are swapped when bit 3 of i is set -- on the SOURCE address of the copy
//     and on the fragment read -- which keeps `ds_read_b128` conflict-free (rows 8 apart share banks; each b128 lane group
//     holds rows 8 or 24 apart);
//   * positions outside the utterance read a.zeros (a 16-byte zero page) instead of being predicated.
// Ordering (cdna_hip_programming.md, LDS-DMA): every wave waits for its own copies of chunk c with a counted vmcnt, then the
// workgroup barrier, then the fragment reads; the buffer of chunk c - 1 is refilled after that barrier, which every wave
// reaches with its fragment reads retired (lgkmcnt(0)).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;   // M0 carries the LDS destination; hipcc owns M0, so save / restore it inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NWV = waves per workgroup: the tile is MT channels x (NWV * 32 * WN) positions.  The packed weights of a chunk are the same for
// every position tile, so a workgroup of 8 waves (512 positions) pulls them through the CU's L1 once for twice the MFMAs.
// EPIH: the all-halves epilogue (conv_epilogue_plain_h; the launcher checks that the tensors are halves).
// MT = 128 (round 6): the tile spans TWO of the packed 64-channel weight tiles (the packing stays [Mtot/64][Cin/16][KS][64][16]: the
// chunk's two weight images land back to back in LDS), a wave keeps a 128-channel x 64-position accumulator tile (128 registers: one
// workgroup of eight waves per CU), and every staged input window feeds twice the MFMAs: the global -> LDS operand fill, which bounds
// the k = 3 / 7 convs of the 128- and 256-channel stages (~12.5 B/clk/CU measured, profiles/r05_pmc_conv_latency.txt), drops from
// 22.6 to 14.4 KB per 96 MFMAs at k = 3.  Same chunk, tap and MFMA order per output element: bitwise the MT = 64 result.
template <int KS, int DIL, int MT, int NBUF, int NWV = 4, bool EPIH = false>
__global__ __launch_bounds__(64 * NWV, MT == 128 ? 2 : NWV == 4 ? 2 : 4) void conv1d_dma_f16_kernel(ConvArgs a) {   // (two workgroups per CU; one at MT = 128)
    constexpr int CK = 16;
    constexpr int WM = MT / 32;
    constexpr int WN = (MT == 32) ? 4 : 2;
    constexpr int NTW = 32 * WN;
    constexpr int NT = NWV * NTW;
    constexpr int HALO = (KS - 1) * DIL;
    constexpr int XROW = NT + HALO;
    constexpr int MTP = MT == 128 ? 64 : MT;               // channels per packed weight tile
    constexpr int WIP = KS * MTP * 2 / 64;                 // 1-KiB copies of one packed tile's chunk
    constexpr int WI = WIP * (MT / MTP);                   // weight copies per chunk
    constexpr int XI0 = (2 * XROW + 63) / 64;
    constexpr int XI = XI0 + (NWV - (WI + XI0) % NWV) % NWV;   // input-window copies, padded so that every wave issues IPW of them
    constexpr int IPW = (WI + XI) / NWV;
    constexpr int BUF = (WI + XI) * 1024;
    static_assert((KS * MTP * 2) % 64 == 0 && IPW * (NBUF - 2) < 64, "DMA bookkeeping");
    static_assert(NBUF * BUF <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUF];

    const int b = blockIdx.z;
    int mtile, ttile;
    conv_tile_order(mtile, ttile);
    const int q0 = ttile * NT;
    const int len_in = a.base_len[b] * a.len_mul;
    const int n_q = a.ups_s ? len_in + 1 : len_in;   // polyphase transposed conv: one more input position than output frames / s
    if (q0 >= n_q) return;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const long wtile_bytes = (long)(a.Cin / CK) * (WIP * 1024);   // one packed 64- (32-) channel tile, all chunks
    const char* wsrc_tile = reinterpret_cast<const char*>(a.wp16) + (long)mtile * (MT / MTP) * wtile_bytes;
    const _Float16* xhb = reinterpret_cast<const _Float16*>(a.x) + (long)b * a.x_bstride;
    const unsigned sbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const int nch = a.Cin / CK;

    auto issue = [&](int c) {   // the copies of chunk c into buffer c % NBUF; this wave's share: ii = wave, wave + 4, ..
        const unsigned dst0 = sbase + (unsigned)(c % NBUF) * BUF;
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            const int ii = wvs + NWV * k;
            const bool is_w = ii < WI;   // wave-uniform; selects instead of branches keep the copy sequence straight-line
            const int iw = (MT == MTP) ? ii : (ii >= WIP ? ii - WIP : ii);   // copy index inside its packed tile
            const int s = (is_w ? iw : ii - WI) * 64 + lane, row = s >> 1, h = (s & 1) ^ ((row >> 3) & 1);
            const int t = q0 - a.padl + row;
            const char* wsrc = wsrc_tile + ((MT != MTP && ii >= WIP) ? wtile_bytes : 0L) + (long)c * (WIP * 1024) + (row * 2 + h) * 16;
            const char* xsrc = reinterpret_cast<const char*>(xhb + ((long)c * a.x_stride + t) * 16 + 8 * h);
            const char* src = is_w ? wsrc : (t >= 0 && t < len_in) ? xsrc : reinterpret_cast<const char*>(a.zeros);
            glds16(src, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)ii * 1024));
        }
    };

#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < nch) issue(c);
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        // this wave's copies of chunk c have landed once at most IPW * (chunks issued after c) of its copies are outstanding
        const int after = min(NBUF - 2, nch - 1 - c);
        if (NBUF >= 4 && after == 2) wait_vmcnt<2 * IPW>();
        else if (NBUF >= 3 && after == 1) wait_vmcnt<IPW>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + NBUF - 1 < nch) issue(c + NBUF - 1);
        __builtin_amdgcn_sched_barrier(0);
        const char* bufp = smem + (c % NBUF) * BUF;
        const char* wb = bufp + l31 * 32 + 16 * (hi ^ ((l31 >> 3) & 1));
        const char* xb = bufp + WI * 1024 + (wv * NTW) * 32;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            h16x8 av[WM], bv[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m)   // (MT = 128: m = 2, 3 live in the second packed tile's image, WIP KiB further on)
                av[m] = *reinterpret_cast<const h16x8*>(wb + (m * 32 / MTP) * (WIP * 1024) + (j * MTP + (m * 32) % MTP) * 32);
            const int i0 = l31 + j * DIL;
            const char* xp = xb + i0 * 32 + 16 * (hi ^ ((i0 >> 3) & 1));
#pragma unroll
            for (int n = 0; n < WN; ++n) bv[n] = *reinterpret_cast<const h16x8*>(xp + n * 32 * 32);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }
    if constexpr (EPIH) {
        if (a.mrf_mode == 0) conv_epilogue_plain_h<WM, WN, MT, NTW, 0>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
        else if (a.mrf_mode == 1) conv_epilogue_plain_h<WM, WN, MT, NTW, 1>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
        else if (a.mrf_mode == 2) conv_epilogue_plain_h<WM, WN, MT, NTW, 2>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
        else conv_epilogue_plain_h<WM, WN, MT, NTW, 3>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
    } else {
        if (a.ups_s) conv_epilogue_ups<WM, WN, MT, NTW>(a, acc, b, mtile, q0, wv, l31, hi, len_in, n_q);
        else if (a.mrf_mode == 0) conv_epilogue_plain<WM, WN, MT, NTW, 0>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
        else if (a.mrf_mode == 1) conv_epilogue_plain<WM, WN, MT, NTW, 1>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
        else if (a.mrf_mode == 2) conv_epilogue_plain<WM, WN, MT, NTW, 2>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
        else conv_epilogue_plain<WM, WN, MT, NTW, 3>(a, acc, b, mtile, q0, wv, l31, hi, n_q);
    }
}

// Workgroup shape of the 64-channel-tile convs (the 256- and 128-channel stages): 8 = eight waves x 64 positions, two workgroups per
// CU (default since round 5: the chunk's packed weights pass through the CU's L1 once per 512 positions instead of once per 256 --
// PMC showed the L1 stalled on its pending-request limit 45-61 % of the time, profiles/r05_pmc_conv_latency.txt; 61.8 vs 65.4 ms per
// 64 utterances); AUR_CONV_WAVES=4 selects round 4's 256-position tiles (A/B).  Also measured, not kept: one 1024-position workgroup
// of 16 waves per CU 63.2 ms; four waves x 128 positions (256 registers, an A fragment serves four MFMAs) 64.8.
static int conv_dma_waves() {
    static const int w = [] {
        const char* e = getenv("AUR_CONV_WAVES");
        return e ? atoi(e) : 8;
    }();
    return w;
}
// Output-channel tile of the all-halves convs whose Mtot is a multiple of 128 (the ResBlock convs of the 256- and 128-channel
// stages, conv_pre): 64 (default) or 128 (AUR_CONV_MT=128, A/B).  Bit mask AUR_CONV_MT128_KS selects the tap counts that take the
// wide tile (1 = k 3, 2 = k 7, 4 = k 11; default all).  Measured in round 6 on one box, conv time per 64 utterances
// (profiles/r06_conv_mt_ab.log): 64-channel tile 64.2 / 64.0 ms; 128-channel tile for every k 66.0, for k = 3 only 65.2, k = 7 only
// 64.9, k = 11 only 65.2 (256-channel class 0.41 -> 0.38 of the matrix peak, 128-channel class 0.345 -> 0.324).  The wide tile halves
// the window bytes staged per MFMA and cuts the LDS fragment reads per MFMA from 1 to 0.75, and still loses: at 128 accumulator
// registers a CU holds ONE workgroup of eight waves, and what the second workgroup of the 64-channel form hides -- the per-chunk
// barrier and the wait for the chunk's copies -- is exposed.  The operand fill is therefore not what bounds these kernels; the
// vocoder is closed at this tiling.
static int conv_dma_mt() {
    static const int v = [] {
        const char* e = getenv("AUR_CONV_MT");
        return e ? atoi(e) : 64;
    }();
    return v;
}
static int conv_dma_mt128_ks() {
    static const int v = [] {
        const char* e = getenv("AUR_CONV_MT128_KS");
        return e ? atoi(e) : 7;
    }();
    return v;
}
template <int KS, int DIL>
static void launch_conv_dma(const ConvArgs& a, hipStream_t st) {
    constexpr int NBUF = KS >= 11 ? 2 : KS >= 3 ? 3 : 4;   // the fewer taps, the shorter a chunk's MFMA phase and the deeper the prefetch
    AUR_REQUIRE(a.x_f16 && a.zeros && a.Cin % 16 == 0 && a.Cout % 16 == 0 && a.wp16, "conv dma: fp16 interleaved input, zero page");
    AUR_REQUIRE(a.Cin / 16 >= NBUF - 1, "conv dma: fewer input-channel chunks than the prefetch depth");
    trace_launch("conv1d_dma_f16_kernel");
    const int n_q = a.ups_s ? a.max_len + 1 : a.max_len;
    const bool all_halves = a.ups_s == 0 && (!a.res || a.res_f16) && (a.mrf_mode == 0 || a.mrf_f16) &&
                            ((a.mrf_mode == 1 || a.mrf_mode == 2) || a.out_act_f16);
    if constexpr (KS >= 3) {
        const int ksbit = KS == 3 ? 1 : KS == 7 ? 2 : 4;
        if (a.Mtot % 128 == 0 && conv_dma_waves() == 8 && all_halves && conv_dma_mt() == 128 && (conv_dma_mt128_ks() & ksbit)) {
            constexpr int NB128 = KS >= 7 ? 2 : 3;
            dim3 grid((n_q + 511) / 512, a.Mtot / 128, a.B);
            hipLaunchKernelGGL((conv1d_dma_f16_kernel<KS, DIL, 128, NB128, 8, true>), grid, dim3(512), 0, st, a);
            return;
        }
    }
    if (a.Mtot % 64 == 0 && conv_dma_waves() == 8 && KS >= 3 && all_halves) {
        constexpr int NB8 = KS >= 7 ? 2 : 3;
        dim3 grid((n_q + 511) / 512, a.Mtot / 64, a.B);
        hipLaunchKernelGGL((conv1d_dma_f16_kernel<KS, DIL, 64, NB8, 8, true>), grid, dim3(512), 0, st, a);
    } else if (a.Mtot % 64 == 0) {
        dim3 grid((n_q + 255) / 256, a.Mtot / 64, a.B);
        hipLaunchKernelGGL((conv1d_dma_f16_kernel<KS, DIL, 64, NBUF>), grid, dim3(256), 0, st, a);
    } else {
        AUR_REQUIRE(a.Mtot % 32 == 0, "conv dma: Mtot % 32");
        dim3 grid((n_q + 511) / 512, a.Mtot / 32, a.B);
        hipLaunchKernelGGL((conv1d_dma_f16_kernel<KS, DIL, 32, NBUF>), grid, dim3(256), 0, st, a);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused ResBlock round (vocoder_kernels.h, RoundArgs).  Workgroup = C/32 x NT1/(32*WN) waves, each wave a 32-channel x
// 32*WN-position block of a [C] x [NT1] conv1 tile; conv2 then yields NT2 = NT1 - (KS - 1) output positions of it.  LDS: the input window of ALL C
// channels (staged once, by LDS-DMA, also the source of the residual), the conv1 result h as [chunk][position][16] rows, and two
// weight-chunk buffers through which the 2 * C/16 weight chunks of conv1 then conv2 stream (DMA of chunk g + 1 under the MFMAs of
// chunk g).  Row halves are swapped when bit 3 of the row index is set (as conv1d_dma_f16_kernel) in all three images.
__device__ __forceinline__ int swz16(int row, int half) { return 16 * (half ^ ((row >> 3) & 1)); }

template <int KS, int DIL, int C, int WN, int NT1>
__global__ __launch_bounds__(64 * (C / 32) * (NT1 / 32 / WN)) void resblock_round_f16_kernel(RoundArgs a) {
    constexpr int NCH = C / 16, NWN = NT1 / 32 / WN, NW = (C / 32) * NWN, PW = 32 * WN;   // NWN waves along the positions, PW positions each
    constexpr int P1 = (KS - 1) / 2 * DIL, P2 = (KS - 1) / 2;
    constexpr int NT2 = NT1 - (KS - 1);   // NT1 conv1 positions per tile -> NT2 outputs
    constexpr int XROW = (NT1 + (KS - 1) * DIL + 31) / 32 * 32;   // window rows per chunk, whole 1-KiB copies
    constexpr int HROW = (NT1 + (KS - 1) + 31) / 32 * 32;
    constexpr int WCH = KS * C * 32;                               // bytes of one weight chunk
    constexpr int XI = XROW / 32, WI = WCH / 1024;                 // 1-KiB copies per window chunk / weight chunk
    static_assert(WCH % 1024 == 0, "weight chunk in whole copies");
    __shared__ __attribute__((aligned(1024))) char xs[NCH * XROW * 32];
    __shared__ __attribute__((aligned(1024))) char hs[NCH * HROW * 32];
    __shared__ __attribute__((aligned(1024))) char ws[2][WCH];

    const int b = blockIdx.y;
    const int q0 = blockIdx.x * NT2;
    const int len = a.base_len[b] * a.len_mul;
    if (q0 >= len) return;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wvs / NWN, wn = wvs % NWN;
    const _Float16* yb = reinterpret_cast<const _Float16*>(a.y) + (long)b * a.bstride;
    const unsigned xs_l = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)xs;
    const unsigned ws_l = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)&ws[0][0];

    // (a deeper weight ring -- chunk g + 3 requested when chunk g starts, counted waits -- measured no gain: DESIGN section 7)
    auto issue_w = [&](int g) {   // weight chunk g of the sequence conv1[0..NCH), conv2[0..NCH) into buffer g & 1
        const char* src0 = reinterpret_cast<const char*>(g < NCH ? a.w1 : a.w2) + (long)(g < NCH ? g : g - NCH) * WCH;
        for (int ii = wvs; ii < WI; ii += NW) {
            const int s = ii * 64 + lane, row = s >> 1, h = (s & 1) ^ ((row >> 3) & 1);
            glds16(src0 + (row * 2 + h) * 16, __builtin_amdgcn_readfirstlane(ws_l + (unsigned)(g & 1) * WCH + (unsigned)ii * 1024));
        }
    };
    // the input window of every chunk: row i <-> position q0 - P2 - P1 + i
    for (int ii = wvs; ii < NCH * XI; ii += NW) {
        const int c = ii / XI, s = (ii - c * XI) * 64 + lane, row = s >> 1, h = (s & 1) ^ ((row >> 3) & 1);
        const int t = q0 - P2 - P1 + row;
        const char* src = (t >= 0 && t < len) ? reinterpret_cast<const char*>(yb + ((long)c * a.stride + t) * 16 + 8 * h)
                                              : reinterpret_cast<const char*>(a.zeros);
        glds16(src, __builtin_amdgcn_readfirstlane(xs_l + (unsigned)ii * 1024));
    }
    issue_w(0);
    issue_w(1);

    f32x16 acc[WN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    };
    // one weight chunk: B fragments from `img` (rows of 32 B, chunk-major), row = wn * PW + n * 32 + l31 + j * dil
    auto mfma_chunk = [&](const char* img, int rows_per_chunk, int c, int g, auto DILc) {
        constexpr int dil = decltype(DILc)::value;
        const char* wb = &ws[g & 1][0] + (wm * 32 + l31) * 32 + 16 * (hi ^ ((l31 >> 3) & 1));
        const char* xb = img + (long)c * rows_per_chunk * 32;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const h16x8 av = *reinterpret_cast<const h16x8*>(wb + j * C * 32);
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int i = wn * PW + n * 32 + l31 + j * dil;
                const h16x8 bv = *reinterpret_cast<const h16x8*>(xb + i * 32 + swz16(i, hi));
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[n], 0, 0, 0);
            }
        }
    };
    auto sync_chunk = [&](int g) {   // chunk g's copies have landed everywhere; refill the buffer chunk g - 1 used
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g >= 1 && g + 1 < 2 * NCH) issue_w(g + 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- conv1 over the 256 positions q0 - P2 .. of the tile
    zero_acc();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        sync_chunk(c);
        mfma_chunk(xs, XROW, c, c, std::integral_constant<int, DIL>{});
    }
    // h = fp16(lrelu(. + b1)), zero outside the utterance (conv2 pads the conv1 OUTPUT with zeros)
    {
        const int c0 = wm * 32 + 4 * hi;
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = a.b1[c0 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int p = wn * PW + n * 32 + l31, pg = q0 - P2 + p;
            const bool in = pg >= 0 && pg < len;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = c0 + 8 * g4;
                h16x4v hv;
#pragma unroll
                for (int k = 0; k < 4; ++k) hv[k] = (_Float16)(in ? lrelu(acc[n][4 * g4 + k] + bias[4 * g4 + k], 0.1f) : 0.f);
                *reinterpret_cast<h16x4v*>(hs + ((long)(c >> 4) * HROW + p) * 32 + swz16(p, (c & 15) >> 3) + (c & 7) * 2) = hv;
            }
        }
    }
    // ---- conv2 over h (the barrier of its first chunk also publishes h)
    zero_acc();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        sync_chunk(NCH + c);
        mfma_chunk(hs, HROW, c, NCH + c, std::integral_constant<int, 1>{});
    }
    // ---- epilogue: + b2 + residual (the window's own rows, un-activated) -> stream / MRF
    {
        const int c0 = wm * 32 + 4 * hi;
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = a.b2[c0 + (r & 3) + 8 * (r >> 2)];
        const long ob = (long)b * a.bstride;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int p = wn * PW + n * 32 + l31, q = q0 + p;
            const bool ok = p < NT2 && q < len;
            const int qc = min(q, len - 1), i = min(p, NT2 - 1) + P2 + P1;
            h16x4v mold[4];
            if (a.mrf_mode >= 2) {
                const _Float16* mb = reinterpret_cast<const _Float16*>(a.mrf) + ob;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c = c0 + 8 * g4;
                    mold[g4] = *reinterpret_cast<const h16x4v*>(mb + ((long)(c >> 4) * a.stride + qc) * 16 + (c & 15));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!ok) continue;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = c0 + 8 * g4;
                const h16x4v yv = *reinterpret_cast<const h16x4v*>(xs + ((long)(c >> 4) * XROW + i) * 32 + swz16(i, (c & 15) >> 3) + (c & 7) * 2);
                float val[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yk = (float)yv[k];
                    val[k] = acc[n][4 * g4 + k] + bias[4 * g4 + k] + (yk < 0.f ? yk * 10.0f : yk);
                }
                const long off = ((long)(c >> 4) * a.stride + q) * 16 + (c & 15);
                h16x4v o;
                if (a.mrf_mode == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (_Float16)lrelu(val[k], 0.1f);
                    *reinterpret_cast<h16x4v*>(reinterpret_cast<_Float16*>(a.out) + ob + off) = o;
                } else if (a.mrf_mode == 1 || a.mrf_mode == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (_Float16)(a.mrf_mode == 2 ? val[k] + (float)mold[g4][k] : val[k]);
                    *reinterpret_cast<h16x4v*>(reinterpret_cast<_Float16*>(a.mrf) + ob + off) = o;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (_Float16)lrelu(((float)mold[g4][k] + val[k]) / 3.0f, a.e_slope);
                    *reinterpret_cast<h16x4v*>(reinterpret_cast<_Float16*>(a.e_out) + ob + off) = o;
                }
            }
        }
    }
}

template <int KS, int DIL>
static void launch_round_c(const RoundArgs& a, hipStream_t st) {
    trace_launch("resblock_round_f16_kernel");
    // 64 channels, k = 3 / 7: 128-position tiles (8 waves of 32 channels x 32 positions, <= 69 KB of LDS) so that two workgroups
    // share a CU -- 15.6 vs 16.4 ms for the stage; k = 11 needs 120 KB either way and keeps 256 positions (64 per wave).
    // 32 channels: 256 positions, 32 per wave (9.6 vs 10.1 ms with 64 per wave).
    // Round 5, measured on these two stages and not kept (conv time per 64 utterances, same box): one 16-wave workgroup
    // per CU on twice the positions -- half the weight re-streaming per position, one barrier domain -- 63.2 vs 62.3 ms (32-channel class
    // 0.199 vs 0.215 of the matrix peak); two position tiles = two independent accumulators per wave on half the waves per workgroup
    // (VERDICT r04 3 i) 62.7 vs 61.7; a four-deep weight ring with counted waits 62.6 vs 62.2; the weights as A fragments straight from
    // global memory into registers, one chunk ahead, no weight buffers in LDS and two barriers per tile instead of one per chunk
    // (every wave of a channel group re-reads the same lines through the L1) 64.3 vs 61.0 (64-channel class 0.236 vs 0.265, 32-channel
    // 0.190 vs 0.216), bitwise equal.
    if (a.C == 64 && KS < 11) {
        constexpr int nt2 = 128 - (KS - 1);
        hipLaunchKernelGGL((resblock_round_f16_kernel<KS, DIL, 64, 1, 128>), dim3((a.max_len + nt2 - 1) / nt2, a.B), dim3(512), 0, st, a);
        return;
    }
    // (A persistent, weight-stationary form of the 32-channel round -- both convs' weights as A fragments in registers, a loader
    // wave double- / triple-buffering the window by LDS-DMA, seven consumer waves, two barriers per tile -- measured SLOWER than
    // this per-tile kernel in every variant: 10.6 ms per stage with one tile of lead, 11.2 with two, 10.2 with two workgroups
    // per CU for k = 3 / 7, against 9.6; bit-identical results.  DESIGN section 7.)
    constexpr int nt2 = 256 - (KS - 1);
    const dim3 grid((a.max_len + nt2 - 1) / nt2, a.B);
    if (a.C == 64) hipLaunchKernelGGL((resblock_round_f16_kernel<KS, DIL, 64, 2, 256>), grid, dim3(512), 0, st, a);
    else if (a.C == 32) hipLaunchKernelGGL((resblock_round_f16_kernel<KS, DIL, 32, 1, 256>), grid, dim3(512), 0, st, a);
    else throw InvalidArgument("resblock round: 64 or 32 channels");
}

void launch_resblock_round_f16(const RoundArgs& a, int KS, int DIL, hipStream_t st) {
    AUR_REQUIRE(a.y && a.zeros && a.w1 && a.w2 && a.b1 && a.b2, "resblock round: arguments");
    AUR_REQUIRE(a.mrf_mode == 0 ? (a.out && a.out != a.y) : (a.mrf && (a.mrf_mode != 3 || a.e_out)), "resblock round: outputs of the mode");
    switch (KS * 16 + DIL) {
        case 3 * 16 + 1: launch_round_c<3, 1>(a, st); break;
        case 3 * 16 + 3: launch_round_c<3, 3>(a, st); break;
        case 3 * 16 + 5: launch_round_c<3, 5>(a, st); break;
        case 7 * 16 + 1: launch_round_c<7, 1>(a, st); break;
        case 7 * 16 + 3: launch_round_c<7, 3>(a, st); break;
        case 7 * 16 + 5: launch_round_c<7, 5>(a, st); break;
        case 11 * 16 + 1: launch_round_c<11, 1>(a, st); break;
        case 11 * 16 + 3: launch_round_c<11, 3>(a, st); break;
        case 11 * 16 + 5: launch_round_c<11, 5>(a, st); break;
        default: throw InvalidArgument("resblock round: k in {3,7,11}, dilation in {1,3,5}");
    }
    HIP_CHECK(hipGetLastError());
}

template <int KS, int DIL, bool XH>
static void launch_conv_f16_t(const ConvArgs& a, hipStream_t st) {
    AUR_REQUIRE(a.Cin % 16 == 0 && a.wp16, "conv f16: Cin % 16, packed fp16 weights");
    AUR_REQUIRE(!a.out_act_f16 || a.mrf_mode == 0 || a.mrf_mode == 3, "conv f16: fp16 output for plain convs and the MRF mean");
    AUR_REQUIRE(!a.mrf_f16 || a.mrf_mode != 0, "conv f16: mrf_f16 without an MRF mode");
    AUR_REQUIRE((!a.x_f16 && !a.out_act_f16 && !a.res_f16 && !a.mrf_f16) || (a.Cin % 16 == 0 && a.Cout % 16 == 0), "conv f16: interleaved tensors need 16-channel chunks");
    AUR_REQUIRE(!a.res_f16 || (a.res && a.ups_s == 0), "conv f16: fp16 residual only on plain convs");
    const int n_q = a.ups_s ? a.max_len + 1 : a.max_len;
    trace_launch("conv1d_mfma_f16_kernel");
    if (a.Mtot % 64 == 0) {
        constexpr int NT64 = 256;
        dim3 grid((n_q + NT64 - 1) / NT64, a.Mtot / 64, a.B);
        hipLaunchKernelGGL((conv1d_mfma_f16_kernel<KS, DIL, 64, XH>), grid, dim3(256), 0, st, a);
    } else {
        AUR_REQUIRE(a.Mtot % 32 == 0, "conv f16: Mtot % 32");
        constexpr int NT32 = 512;
        dim3 grid((n_q + NT32 - 1) / NT32, a.Mtot / 32, a.B);
        hipLaunchKernelGGL((conv1d_mfma_f16_kernel<KS, DIL, 32, XH>), grid, dim3(256), 0, st, a);
    }
}

void launch_conv1d_f16(const ConvArgs& a, int KS, int DIL, hipStream_t st) {
    // fp16 interleaved inputs (every ResBlock conv): LDS-DMA staged, multi-buffered for 64-channel output tiles; the 32-channel
    // stage (2 chunks per tile, HBM-bound) measured faster register-staged (37.7 vs 40.4 ms per 3 batches)
    if (a.x_f16 && a.zeros && a.Mtot % 64 == 0) {
        switch (KS * 16 + DIL) {
            case 2 * 16 + 1: launch_conv_dma<2, 1>(a, st); break;
            case 3 * 16 + 1: launch_conv_dma<3, 1>(a, st); break;
            case 3 * 16 + 3: launch_conv_dma<3, 3>(a, st); break;
            case 3 * 16 + 5: launch_conv_dma<3, 5>(a, st); break;
            case 7 * 16 + 1: launch_conv_dma<7, 1>(a, st); break;
            case 7 * 16 + 3: launch_conv_dma<7, 3>(a, st); break;
            case 7 * 16 + 5: launch_conv_dma<7, 5>(a, st); break;
            case 11 * 16 + 1: launch_conv_dma<11, 1>(a, st); break;
            case 11 * 16 + 3: launch_conv_dma<11, 3>(a, st); break;
            case 11 * 16 + 5: launch_conv_dma<11, 5>(a, st); break;
            default: throw InvalidArgument("launch_conv1d_f16: fp16 input only for k in {3,7,11}, dilation in {1,3,5}");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (a.x_f16) {   // same inputs, register-staged (32-channel stage, transposed convs; no zero page given: tools/conv_diag's reference point)
        switch (KS * 16 + DIL) {
            case 2 * 16 + 1: launch_conv_f16_t<2, 1, true>(a, st); break;
            case 3 * 16 + 1: launch_conv_f16_t<3, 1, true>(a, st); break;
            case 3 * 16 + 3: launch_conv_f16_t<3, 3, true>(a, st); break;
            case 3 * 16 + 5: launch_conv_f16_t<3, 5, true>(a, st); break;
            case 7 * 16 + 1: launch_conv_f16_t<7, 1, true>(a, st); break;
            case 7 * 16 + 3: launch_conv_f16_t<7, 3, true>(a, st); break;
            case 7 * 16 + 5: launch_conv_f16_t<7, 5, true>(a, st); break;
            case 11 * 16 + 1: launch_conv_f16_t<11, 1, true>(a, st); break;
            case 11 * 16 + 3: launch_conv_f16_t<11, 3, true>(a, st); break;
            case 11 * 16 + 5: launch_conv_f16_t<11, 5, true>(a, st); break;
            default: throw InvalidArgument("launch_conv1d_f16: fp16 input only for k in {3,7,11}, dilation in {1,3,5}");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    switch (KS * 16 + DIL) {
        case 2 * 16 + 1: launch_conv_f16_t<2, 1, false>(a, st); break;
        case 3 * 16 + 1: launch_conv_f16_t<3, 1, false>(a, st); break;
        case 3 * 16 + 3: launch_conv_f16_t<3, 3, false>(a, st); break;
        case 3 * 16 + 5: launch_conv_f16_t<3, 5, false>(a, st); break;
        case 7 * 16 + 1: launch_conv_f16_t<7, 1, false>(a, st); break;
        case 7 * 16 + 3: launch_conv_f16_t<7, 3, false>(a, st); break;
        case 7 * 16 + 5: launch_conv_f16_t<7, 5, false>(a, st); break;
        case 11 * 16 + 1: launch_conv_f16_t<11, 1, false>(a, st); break;
        case 11 * 16 + 3: launch_conv_f16_t<11, 3, false>(a, st); break;
        case 11 * 16 + 5: launch_conv_f16_t<11, 5, false>(a, st); break;
        default: throw InvalidArgument("launch_conv1d_f16: unsupported (kernel,dilation)");
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Two chained linear interpolations (align_corners=False), closed form of SURVEY A7'(i).
//   scale r = float32(1/s); src = max(r*(j+0.5)-0.5, 0); i0 = floor(src); i1 = min(i0+1, L-1)
__device__ __forceinline__ void lin_src(float r, int j, int L, int& i0, int& i1, float& lam) {
    float src = r * ((float)j + 0.5f) - 0.5f;
    src = fmaxf(src, 0.f);
    i0 = (int)floorf(src);
    i1 = min(i0 + 1, L - 1);
    lam = src - (float)i0;
}

// (1 - lam) a + lam b with the contraction spelled out: which of the two products hipcc fuses into an fma depends on the code around
// the expression, and the fp32 and fp16 forms of the kernel must round alike (the DMA-staged and the register-staged vocoder are
// compared bit for bit)
__device__ __forceinline__ float lerp_f(float a, float b, float lam) { return fmaf(lam, b, (1.0f - lam) * a); }

__global__ __launch_bounds__(256) void interp2_kernel(const float* __restrict__ lat, long lat_bstride,
                                                      const int* __restrict__ lat_row,
                                                      const int* __restrict__ n_lat,
                                                      const int* __restrict__ base_len, float* __restrict__ z,
                                                      long z_stride, long z_bstride, int C, float r1, float r2) {
    const int b = blockIdx.z;
    const int c = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int L0 = n_lat[b];
    const int L1 = 4 * L0;
    const int L2 = base_len[b];
    if (j >= L2) return;
    const float* x = lat + (long)(lat_row ? lat_row[b] : b) * lat_bstride + c;   // x[k] = x[k*C]
    int i0, i1;
    float lam2;
    lin_src(r2, j, L1, i0, i1, lam2);
    float y[2];
    const int idx[2] = {i0, i1};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int k0, k1;
        float lam1;
        lin_src(r1, idx[u], L0, k0, k1, lam1);
        y[u] = lerp_f(x[(long)k0 * C], x[(long)k1 * C], lam1);
    }
    z[(long)b * z_bstride + (long)c * z_stride + j] = lerp_f(y[0], y[1], lam2);
}

// The same values as interleaved halves z16[b][C/16][z_stride][16] = fp16(z) -- what conv_pre's staging rounds the fp32 z to anyway
// (its input activation is the identity), so conv_pre can take the LDS-DMA kernel on bit-identical MFMA operands.  A thread = eight
// channels of one position: 32 contiguous bytes of each of the (at most four) latent rows it blends, one 16-byte store.
__global__ __launch_bounds__(256) void interp2_h_kernel(const float* __restrict__ lat, long lat_bstride, const int* __restrict__ lat_row,
                                                        const int* __restrict__ n_lat, const int* __restrict__ base_len,
                                                        _Float16* __restrict__ z, long z_stride, long z_bstride, int C, float r1, float r2) {
    const int b = blockIdx.z;
    const int c8 = blockIdx.y * 8;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int L0 = n_lat[b];
    const int L1 = 4 * L0;
    const int L2 = base_len[b];
    if (j >= L2) return;
    const float* x = lat + (long)(lat_row ? lat_row[b] : b) * lat_bstride + c8;
    int i0, i1;
    float lam2;
    lin_src(r2, j, L1, i0, i1, lam2);
    const int idx[2] = {i0, i1};
    float y[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int k0, k1;
        float lam1;
        lin_src(r1, idx[u], L0, k0, k1, lam1);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + (long)k0 * C), a1 = *reinterpret_cast<const f32x4*>(x + (long)k0 * C + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(x + (long)k1 * C), b1 = *reinterpret_cast<const f32x4*>(x + (long)k1 * C + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[u][e] = lerp_f(a0[e], b0[e], lam1);
            y[u][e + 4] = lerp_f(a1[e], b1[e], lam1);
        }
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)lerp_f(y[0][e], y[1][e], lam2);
    *reinterpret_cast<h16x8*>(z + (long)b * z_bstride + ((long)(c8 >> 4) * z_stride + j) * 16 + (c8 & 15)) = o;
}

void launch_interp2(const float* lat, long lat_bstride, const int* lat_row, const int* n_lat, const int* base_len, float* z,
                    long z_stride, long z_bstride, int C, int B, int max_len, hipStream_t st, bool out_f16) {
    const float r1 = (float)(1.0 / (1024.0 / 256.0));
    const float r2 = (float)(1.0 / (24000.0 / 22050.0));
    if (out_f16) {
        AUR_REQUIRE(C % 16 == 0, "interp2: whole 16-channel chunks for the interleaved fp16 output");
        dim3 grid((max_len + 255) / 256, C / 8, B);
        trace_launch("interp2_h_kernel");
        hipLaunchKernelGGL(interp2_h_kernel, grid, dim3(256), 0, st, lat, lat_bstride, lat_row, n_lat, base_len,
                           reinterpret_cast<_Float16*>(z), z_stride, z_bstride, C, r1, r2);
        HIP_CHECK(hipGetLastError());
        return;
    }
    dim3 grid((max_len + 255) / 256, C, B);
    trace_launch("interp2_kernel");
    hipLaunchKernelGGL(interp2_kernel, grid, dim3(256), 0, st, lat, lat_bstride, lat_row, n_lat, base_len, z, z_stride,
                       z_bstride, C, r1, r2);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// conv_post (Cin -> 1, k7, no bias) + tanh.  Cin <= 32.  HBM-bound: reads Cin floats per sample.
// XH: x is the fp16 vocoder's stage output, already activated, interleaved [C/16][t][16] halves (strides in elements).
template <bool XH>
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ wav, const int* __restrict__ base_len,
                                                        int len_mul, int Cin, long x_stride, long x_bstride,
                                                        long wav_bstride, float slope) {
    constexpr int NT = 256, KS = 7, MAXC = 32;
    __shared__ float xs[MAXC][NT + KS - 1];
    __shared__ float wsm[MAXC * KS];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * NT;
    const int len = base_len[b] * len_mul;
    if (t0 >= len) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < Cin * KS; i += 256) wsm[i] = w[i];
    const float* xb = x + (long)b * x_bstride;
    // stage 16 channels per pass: 16 unconditional loads at clamped addresses in flight per thread (a predicated load
    // per channel serialised 32 memory round trips), masked when written to LDS; lanes 0..5 also fetch the right halo
    constexpr int CU = 16;
    const int tm = t0 - 3 + tid, th = t0 - 3 + NT + tid;
    const int tmc = min(max(tm, 0), len - 1), thc = min(max(th, 0), len - 1);
    const bool okm = tm >= 0 && tm < len, okh = th >= 0 && th < len;
    for (int c0 = 0; c0 < Cin; c0 += CU) {
        float v[CU], h[CU];
        if constexpr (XH) {   // the 16 channels of a position are 32 contiguous bytes
            const _Float16* xh = reinterpret_cast<const _Float16*>(x) + (long)b * x_bstride + (long)(c0 >> 4) * x_stride * 16;
            const h16x8 m0 = *reinterpret_cast<const h16x8*>(xh + (long)tmc * 16), m1 = *reinterpret_cast<const h16x8*>(xh + (long)tmc * 16 + 8);
            h16x8 h0 = m0, h1 = m1;
            if (tid < KS - 1) {
                h0 = *reinterpret_cast<const h16x8*>(xh + (long)thc * 16);
                h1 = *reinterpret_cast<const h16x8*>(xh + (long)thc * 16 + 8);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = (float)m0[u]; v[u + 8] = (float)m1[u];
                h[u] = (float)h0[u]; h[u + 8] = (float)h1[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < CU; ++u) v[u] = xb[(long)min(c0 + u, Cin - 1) * x_stride + tmc];
            if (tid < KS - 1) {
#pragma unroll
                for (int u = 0; u < CU; ++u) h[u] = xb[(long)min(c0 + u, Cin - 1) * x_stride + thc];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CU; ++u)
            if (c0 + u < Cin) xs[c0 + u][tid] = okm ? (XH ? v[u] : lrelu(v[u], slope)) : 0.f;
        if (tid < KS - 1) {
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (c0 + u < Cin) xs[c0 + u][NT + tid] = okh ? (XH ? h[u] : lrelu(h[u], slope)) : 0.f;
        }
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= len) return;
    float s = 0.f;
    for (int c = 0; c < Cin; ++c) {
#pragma unroll
        for (int j = 0; j < KS; ++j) s = fmaf(wsm[c * KS + j], xs[c][tid + j], s);
    }
    wav[(long)b * wav_bstride + t] = tanhf(s);
}

void launch_conv_post(const float* x, const float* w, float* wav, const int* base_len, int len_mul, int Cin,
                      long x_stride, long x_bstride, long wav_bstride, float slope, int B, int max_len,
                      hipStream_t st, bool x_f16_act) {
    AUR_REQUIRE(Cin <= 32 && (!x_f16_act || Cin % 16 == 0), "conv_post: Cin <= 32 (whole 16-channel chunks for the fp16 form)");
    dim3 grid((max_len + 255) / 256, B);
    trace_launch("conv_post_kernel");
    if (x_f16_act)
        hipLaunchKernelGGL(conv_post_kernel<true>, grid, dim3(256), 0, st, x, w, wav, base_len, len_mul, Cin, x_stride,
                           x_bstride, wav_bstride, slope);
    else
        hipLaunchKernelGGL(conv_post_kernel<false>, grid, dim3(256), 0, st, x, w, wav, base_len, len_mul, Cin, x_stride,
                           x_bstride, wav_bstride, slope);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// y[b][r] = bias[r] + W[r][:] . g[b][:]   one wave per output row.
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                        const float* __restrict__ g, float* __restrict__ y, int R,
                                                        int K, long g_bstride, long y_bstride) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* wr = W + (long)r * K;
    const float* gb = g + (long)b * g_bstride;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(wr[k], gb[k], s);
    s = wave_sum(s);
    if (lane == 0) y[(long)b * y_bstride + r] = s + (bias ? bias[r] : 0.f);
}

void launch_gemv_rows(const float* W, const float* bias, const float* g, float* y, int R, int K, int B,
                      long g_bstride, long y_bstride, hipStream_t st) {
    dim3 grid((R + 3) / 4, B);
    trace_launch("gemv_rows_kernel");
    hipLaunchKernelGGL(gemv_rows_kernel, grid, dim3(256), 0, st, W, bias, g, y, R, K, g_bstride, y_bstride);
    HIP_CHECK(hipGetLastError());
}

}  // namespace aur
