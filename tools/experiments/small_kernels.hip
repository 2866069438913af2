// Small-batch decode chain for gfx950 (MI355X): see small_kernels.h.
//
// Reference arithmetic: vLLM's GPT2Block at M <= 4 rows (instantiated at vllm_mm_gpt.py:757-761: c_attn, attention over the paged
// K/V, c_proj, c_fc + activation, mlp.c_proj), pre-LN with LayerNorm folded into c_attn / c_fc (gpt_kernels.h: launch_fold_ln).
//
// One workgroup of four waves per 16-column weight tile (and K part).  The tile's weights are the FIRST thing a launch requests --
// every 1 KiB block of its K range, non-temporal, all in flight at once: 64 KB per workgroup for the wide GEMMs -- and everything
// the prologue needs (residual rows, K slabs, attention partials) is requested behind them, so a launch is one memory round trip
// wide.  x lives in LDS (4 rows x the workgroup's K range), a lane multiplies its float4 of a packed block (four consecutive k of
// one column, gpt_kernels.h: pack_wt16) against the matching x float4 of each row: exact fp32 FMAs, k ascending inside a wave's
// blocks, then lane groups, waves and K slabs summed in a fixed order.
#include <type_traits>

#include "small_kernels.h"

namespace aur {

namespace {

__device__ __forceinline__ long skv_offset(int blk, int kv, int head, int tok) {   // paged K/V layout: [block][K|V][head][token][64]
    return (((long)blk * 2 + kv) * kHeads + head) * (kKvBlockTokens * kHeadDim) + (long)tok * kHeadDim;
}

enum SmallMode { kSmQkv = 0, kSmFc = 1, kSmProj = 2, kSmProj2 = 3 };

struct SmallGemvArgs {
    SmallRowsIn in;        // qkv, fc
    const float* Wt;       // packed (pack_wt16)
    const float* c1;       // qkv, fc: launch_fold_ln's vectors
    const float* c2;
    float eps;
    float* qbuf;           // qkv
    void* kv_layer;
    int kv_half;
    const int* row_meta;   // qkv (K/V addressing), proj (context lengths)
    float* act;            // fc out [kSmallRows][4096]
    int gelu_erf;
    const float* part_o;   // proj in
    const float* part_ml;
    int split;             // proj: tokens per attention split
    const float* act_in;   // proj2 in
    float* slabs;          // proj, proj2 out [kSmallSlabs][kSmallRows][1024]
    int M;
};

// sum of one value per thread over the workgroup's four waves, for kSmallRows values at once; fixed order
__device__ __forceinline__ void block_sum4(float (&v)[kSmallRows], float (*buf)[kSmallRows], int w, int lane) {
#pragma unroll
    for (int m = 0; m < kSmallRows; ++m) {
        const float s = wave_sum(v[m]);
        if (lane == 0) buf[w][m] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kSmallRows; ++m) v[m] = (buf[0][m] + buf[1][m]) + (buf[2][m] + buf[3][m]);
}

template <int MODE>
__global__ __launch_bounds__(256) void small_gemv_kernel(SmallGemvArgs a) {
    constexpr int NB = (MODE == kSmProj) ? 4 : 16;             // 16 x 16 weight blocks per wave
    constexpr int KW = 64 * NB;                                 // k range of the workgroup: 4 waves x NB blocks x 16
    constexpr int KB_TOTAL = (MODE == kSmProj2) ? 256 : 64;     // K / 16
    constexpr bool LN = MODE == kSmQkv || MODE == kSmFc;
    constexpr int NOUT = (MODE == kSmQkv) ? 3 * kHidden : (MODE == kSmFc) ? 4 * kHidden : kHidden;
    __shared__ __attribute__((aligned(16))) float xs[kSmallRows][KW];
    __shared__ float red[4][kSmallRows][16];
    __shared__ float sta[4][kSmallRows], stb[4][kSmallRows];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x, part = blockIdx.y;
    const int M = a.M;

    // ---- the tile's weights: every block of this wave's K range, in flight before anything else
    const f32x4* wt = reinterpret_cast<const f32x4*>(a.Wt) + ((long)nt * KB_TOTAL + part * (4 * NB) + w * NB) * 64 + lane;
    f32x4 bf[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bf[i] = __builtin_nontemporal_load(&wt[i * 64]);
    // epilogue constants of this thread's output element (threads 0..63: row tid >> 4, column tid & 15)
    const int em = (tid >> 4) & 3, en = 16 * nt + (tid & 15);
    float ec1 = 0.f, ec2 = 0.f;
    if (LN) {
        ec1 = a.c1[en];
        ec2 = a.c2[en];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the input rows of this workgroup's K range -> xs
    float mean[kSmallRows], rstd[kSmallRows];
    if constexpr (LN) {
        // h = ((hs + bias) + slab 0) + slab 1 + slab 2 + slab 3, one float4 per thread and row; rows >= M are zeros
        const int n4 = 4 * tid;
        f32x4 v[kSmallRows];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const bool has_slabs = a.in.slabs != nullptr;
        f32x4 bv = zero;
        if (has_slabs) bv = *reinterpret_cast<const f32x4*>(a.in.slab_bias + n4);
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) {
            v[m] = zero;
            if (m < M) {
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(a.in.hs_in + m * kHidden + n4);
                if (has_slabs) {
                    f32x4 s[kSmallSlabs];
#pragma unroll
                    for (int p = 0; p < kSmallSlabs; ++p)
                        s[p] = *reinterpret_cast<const f32x4*>(a.in.slabs + ((long)p * kSmallRows + m) * kHidden + n4);
                    f32x4 t = h0 + bv;
#pragma unroll
                    for (int p = 0; p < kSmallSlabs; ++p) t += s[p];
                    v[m] = t;
                } else {
                    v[m] = h0;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) {
            *reinterpret_cast<f32x4*>(&xs[m][n4]) = v[m];
            if (nt == 0 && a.in.hs_out && m < M) *reinterpret_cast<f32x4*>(a.in.hs_out + m * kHidden + n4) = v[m];
        }
        // LayerNorm statistics of the rows (two passes over the registers, fixed reduction order)
        float s[kSmallRows];
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) s[m] = (v[m][0] + v[m][1]) + (v[m][2] + v[m][3]);
        block_sum4(s, sta, w, lane);
        float q[kSmallRows];
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) {
            mean[m] = s[m] * (1.0f / kHidden);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = v[m][c] - mean[m];
                acc = fmaf(d, d, acc);
            }
            q[m] = acc;
        }
        block_sum4(q, stb, w, lane);   // (its barrier also publishes xs)
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) rstd[m] = 1.0f / sqrtf(q[m] * (1.0f / kHidden) + a.eps);
    } else if constexpr (MODE == kSmProj) {
        // x[m][column] of this K part (four heads): the attention splits of the row's head merged, s ascending:
        //   MX = max_s m_s;  x = (sum_s o_s exp(m_s - MX)) / (sum_s l_s exp(m_s - MX))
        const int head = 4 * part + (tid >> 6), d = tid & 63;
#pragma unroll 1
        for (int m = 0; m < kSmallRows; ++m) {
            float x = 0.f;
            if (m < M) {
                const int pos = a.row_meta[(long)m * kRowMetaStride];
                const int ns = pos / a.split + 1;
                const float* ml = a.part_ml + ((long)(m * kHeads + head) * kSmallMaxSplits) * 2;
                const float* po = a.part_o + ((long)(m * kHeads + head) * kSmallMaxSplits) * kHeadDim + d;
                float mx = ml[0];
                for (int s = 1; s < ns; ++s) mx = fmaxf(mx, ml[2 * s]);
                float num = 0.f, den = 0.f;
                for (int s = 0; s < ns; ++s) {
                    const float wgt = expf(ml[2 * s] - mx);
                    den = fmaf(ml[2 * s + 1], wgt, den);
                    num = fmaf(po[(long)s * kHeadDim], wgt, num);
                }
                x = num / den;
            }
            xs[m][tid] = x;
        }
        __syncthreads();
    } else {
        const int n4 = 4 * tid;
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M) v = *reinterpret_cast<const f32x4*>(a.act_in + (long)m * (4 * kHidden) + part * KW + n4);
            *reinterpret_cast<f32x4*>(&xs[m][n4]) = v;
        }
        __syncthreads();
    }

    // ---- this wave's blocks: k ascending; a lane holds W[16 kb + 4 q + s][column j] for s = 0..3 (q = lane >> 4, j = lane & 15)
    float acc[kSmallRows] = {0.f, 0.f, 0.f, 0.f};
    const int q4 = 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int m = 0; m < kSmallRows; ++m) {
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(&xs[m][(w * NB + i) * 16 + q4]);
            float t = acc[m];
            t = fmaf(bf[i][0], x4[0], t);
            t = fmaf(bf[i][1], x4[1], t);
            t = fmaf(bf[i][2], x4[2], t);
            t = fmaf(bf[i][3], x4[3], t);
            acc[m] = t;
        }
    }
    // the four k quarters of a block (lane groups q), then the four waves
#pragma unroll
    for (int m = 0; m < kSmallRows; ++m) {
        acc[m] += lane_xor<16>(acc[m]);
        acc[m] += lane_xor<32>(acc[m]);
        if (lane < 16) red[w][m][lane] = acc[m];
    }
    __syncthreads();
    if (tid >= 64 || em >= M) return;
    const float sum = (red[0][em][tid & 15] + red[1][em][tid & 15]) + (red[2][em][tid & 15] + red[3][em][tid & 15]);
    static_assert(NOUT % 16 == 0, "whole column tiles");
    if constexpr (MODE == kSmQkv) {
        const float t = rstd[em] * (sum - mean[em] * ec1) + ec2;
        const int u = en >> 10, d = en & (kHidden - 1);
        if (u == 0) {
            a.qbuf[(long)em * kHidden + d] = t;
        } else {
            const int pos = a.row_meta[(long)em * kRowMetaStride], wblk = a.row_meta[(long)em * kRowMetaStride + kRowMetaWblk];
            const long off = skv_offset(wblk, u - 1, d / kHeadDim, pos % kKvBlockTokens) + d % kHeadDim;
            if (a.kv_half) reinterpret_cast<_Float16*>(a.kv_layer)[off] = (_Float16)t;
            else reinterpret_cast<float*>(a.kv_layer)[off] = t;
        }
    } else if constexpr (MODE == kSmFc) {
        const float t = rstd[em] * (sum - mean[em] * ec1) + ec2;
        a.act[(long)em * (4 * kHidden) + en] = a.gelu_erf ? gelu_erf(t) : gelu_new(t);
    } else {
        a.slabs[((long)part * kSmallRows + em) * kHidden + en] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// One attention split: tokens [s * STEP, (s + 1) * STEP) of row m, head `head` -- exactly ONE iteration of paged_attention_kernel's
// token loop (same lane-to-token assignment, same per-group online softmax over the group's UN tokens, same in-workgroup merge of the
// NP groups), leaving the split's UNNORMALISED result: mx = max score, L = sum exp(score - mx), O[d] = sum exp(score - mx) v[d].
typedef _Float16 sh16x8 __attribute__((ext_vector_type(8)));
template <bool KVH>
__global__ __launch_bounds__(256) void small_attention_kernel(const int* __restrict__ row_meta, const float* __restrict__ qbuf,
                                                              const void* __restrict__ kv_layer_v, float* __restrict__ part_o_g,
                                                              float* __restrict__ part_ml_g) {
    constexpr int UN = 4;
    constexpr int LPT = KVH ? 8 : 16;        // lanes per token
    constexpr int EPL = kHeadDim / LPT;      // elements per lane
    constexpr int TPW = 64 / LPT;            // tokens per wave instruction
    constexpr int NP = 4 * TPW;              // partial groups per workgroup
    constexpr int STEP = 4 * TPW * UN;       // tokens per split
    constexpr int NBI = STEP / kKvBlockTokens;
    using KT = typename std::conditional<KVH, _Float16, float>::type;
    using RawT = typename std::conditional<KVH, sh16x8, f32x4>::type;
    __shared__ float part_o[NP][kHeadDim];
    __shared__ float part_m[NP], part_l[NP];
    const int m = blockIdx.x, head = blockIdx.y, sp = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane / LPT, dl = lane % LPT;
    const int* rm = row_meta + (long)m * kRowMetaStride;
    const int pos = rm[0], wblk = rm[kRowMetaWblk];
    const int t0 = sp * STEP;
    if (t0 > pos) return;                     // the row's context ends before this split
    int ids[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) ids[i] = rm[kRowMetaBt + t0 / kKvBlockTokens + i];
    float qv[EPL];
    {
        const float* qp = qbuf + (long)m * kHidden + head * kHeadDim + dl * EPL;
#pragma unroll
        for (int c4 = 0; c4 < EPL / 4; ++c4) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(qp + 4 * c4);
#pragma unroll
            for (int c = 0; c < 4; ++c) qv[4 * c4 + c] = q4[c];
        }
    }
    const int n_keys = pos + 1;
    constexpr int kBlkSh = KVH ? 16 : 17;
    constexpr unsigned kVOff = 1u << (kBlkSh - 1);
    static_assert(sizeof(KT) * 2 * kHeads * kKvBlockTokens * kHeadDim == (1u << kBlkSh) && kKvBlockTokens == 16, "paged K/V layout");
    const char* const kbase = reinterpret_cast<const char*>(kv_layer_v);
    const char* const vbase = kbase + kVOff;
    const unsigned row_bytes = kHeadDim * sizeof(KT);
    const unsigned lane_off = (unsigned)head * (kKvBlockTokens * row_bytes) + (unsigned)dl * (EPL * (unsigned)sizeof(KT));
    const unsigned own_off = lane_off + (unsigned)((wv * TPW + g) & (kKvBlockTokens - 1)) * row_bytes;
    const unsigned last_off = ((unsigned)wblk << kBlkSh) + lane_off + (unsigned)(pos & (kKvBlockTokens - 1)) * row_bytes;
    RawT kraw[UN], vraw[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int traw = t0 + 4 * TPW * u + wv * TPW + g;
        const int blk_u = KVH ? ((wv >> 1) ? ids[(2 * u + 1) % NBI] : ids[(2 * u) % NBI]) : ids[u % NBI];
        const unsigned off = traw > pos ? last_off : (((unsigned)blk_u << kBlkSh) | own_off);
        kraw[u] = *reinterpret_cast<const RawT*>(kbase + off);
        vraw[u] = *reinterpret_cast<const RawT*>(vbase + off);
    }
    __builtin_amdgcn_sched_barrier(0);
    float mi = -INFINITY, li = 0.f;
    float o[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) o[c] = 0.f;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int t = t0 + 4 * TPW * u + wv * TPW + g;
        const bool valid = t < n_keys;
        asm volatile("" ::"v"(vraw[u]));
        float kx[EPL], vx[EPL];
#pragma unroll
        for (int c = 0; c < EPL; ++c) {
            kx[c] = (float)kraw[u][c];
            vx[c] = (float)vraw[u][c];
        }
        float sc = fmaf(qv[0], kx[0], qv[1] * kx[1]) + fmaf(qv[2], kx[2], qv[3] * kx[3]);
        if constexpr (EPL == 8) sc += fmaf(qv[4], kx[4], qv[5] * kx[5]) + fmaf(qv[6], kx[6], qv[7] * kx[7]);
        if constexpr (LPT == 16) {
#define AUR_DPP_ADD(ctrl) sc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sc), (ctrl), 0xF, 0xF, false))
            AUR_DPP_ADD(0x128);   // row_ror:8
            AUR_DPP_ADD(0x124);   // row_ror:4
            AUR_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
            AUR_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
#undef AUR_DPP_ADD
        } else {
#pragma unroll
            for (int sh = LPT / 2; sh > 0; sh >>= 1) sc += __shfl_xor(sc, sh, 64);
        }
        sc *= 0.125f;   // 1/sqrt(64)
        if (valid) {
            const float mn = fmaxf(mi, sc);
            const float alpha = expf(mi - mn);
            const float p = expf(sc - mn);
            li = fmaf(li, alpha, p);
#pragma unroll
            for (int c = 0; c < EPL; ++c) o[c] = fmaf(o[c], alpha, p * vx[c]);
            mi = mn;
        }
    }
    const int pidx = wv * TPW + g;
#pragma unroll
    for (int c = 0; c < EPL; ++c) part_o[pidx][dl * EPL + c] = o[c];
    if (dl == 0) {
        part_m[pidx] = mi;
        part_l[pidx] = li;
    }
    __syncthreads();
    if (threadIdx.x < kHeadDim) {   // wave 0: the NP groups, p ascending (groups without a valid token weigh exp(-inf) = 0)
        float mx = part_m[0];
#pragma unroll
        for (int p = 1; p < NP; ++p) mx = fmaxf(mx, part_m[p]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float wgt = expf(part_m[p] - mx);
            L = fmaf(part_l[p], wgt, L);
            O = fmaf(part_o[p][threadIdx.x], wgt, O);
        }
        const long base = ((long)(m * kHeads + head) * kSmallMaxSplits + sp);
        part_o_g[base * kHeadDim + threadIdx.x] = O;
        if (threadIdx.x == 0) {
            part_ml_g[base * 2] = mx;
            part_ml_g[base * 2 + 1] = L;
        }
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sln_wave(f32x4 (&v)[4], const float* gamma, const float* beta, int lane, float eps) {   // as gpt_kernels.hip: ln_wave
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) sum += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    const float mean = wave_sum(sum) * (1.0f / kHidden);
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d = v[u][c] - mean;
            sq = fmaf(d, d, sq);
        }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / kHidden) + eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = 4 * (lane + 64 * u);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + n);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + n);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = (v[u][c] - mean) * rstd * gm[c] + b[c];
    }
}

// decode tail: the last block's residual rows formed from the chain's slabs, then ln_f and both final_norms as final_rows_kernel
__global__ __launch_bounds__(256) void small_final_rows_kernel(SmallRowsIn in, int mtt, const int* __restrict__ sample_slot,
                                                               const float* __restrict__ lnf_w, const float* __restrict__ lnf_b,
                                                               const float* __restrict__ fn_w, const float* __restrict__ fn_b,
                                                               float* __restrict__ ybuf, float* __restrict__ latents, long lat_slot_stride,
                                                               const int* __restrict__ slot_ngen, int max_lat_rows, int Ms, float eps) {
    const int j = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    if (j >= Ms) return;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = 4 * (lane + 64 * u);
        f32x4 t = *reinterpret_cast<const f32x4*>(in.hs_in + j * kHidden + n);
        if (in.slabs) {
            t += *reinterpret_cast<const f32x4*>(in.slab_bias + n);
#pragma unroll
            for (int p = 0; p < kSmallSlabs; ++p) t += *reinterpret_cast<const f32x4*>(in.slabs + ((long)p * kSmallRows + j) * kHidden + n);
        }
        v[u] = t;
    }
    sln_wave(v, lnf_w, lnf_b, lane, eps);
    sln_wave(v, fn_w, fn_b, lane, eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(ybuf + pk_off(j, 4 * (lane + 64 * u), mtt)) = v[u];
    const int slot = sample_slot[j];
    const int idx = slot_ngen[slot];
    if (latents && idx < max_lat_rows) {
        sln_wave(v, fn_w, fn_b, lane, eps);
        float* dst = latents + (long)slot * lat_slot_stride + (long)idx * kHidden;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dst + 4 * (lane + 64 * u)) = v[u];
    }
}

void check_rows(int M) { AUR_REQUIRE(M >= 1 && M <= kSmallRows, "small decode chain: 1..4 rows"); }

}  // namespace

void launch_small_qkv(const SmallRowsIn& in, const float* Wt, const float* c1, const float* c2, float eps, float* qbuf, void* kv_layer,
                      bool kv_half, const int* row_meta, int M, hipStream_t st) {
    check_rows(M);
    AUR_REQUIRE(in.hs_in && Wt && c1 && c2 && qbuf && kv_layer && row_meta && in.hs_out != in.hs_in && (!in.slabs || in.slab_bias), "small_qkv: arguments");
    SmallGemvArgs a{};
    a.in = in; a.Wt = Wt; a.c1 = c1; a.c2 = c2; a.eps = eps; a.qbuf = qbuf; a.kv_layer = kv_layer; a.kv_half = kv_half ? 1 : 0;
    a.row_meta = row_meta; a.M = M;
    trace_launch("small_gemv_kernel<qkv>");
    hipLaunchKernelGGL(small_gemv_kernel<kSmQkv>, dim3(3 * kHidden / 16, 1), dim3(256), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

void launch_small_attention(const float* qbuf, const void* kv_layer, const int* row_meta, float* part_o, float* part_ml, int M, int n_splits,
                            bool kv_half, hipStream_t st) {
    check_rows(M);
    AUR_REQUIRE(qbuf && kv_layer && row_meta && part_o && part_ml && n_splits >= 1 && n_splits <= kSmallMaxSplits, "small_attention: arguments");
    trace_launch("small_attention_kernel");
    const dim3 grid(M, kHeads, n_splits);
    if (kv_half) hipLaunchKernelGGL(small_attention_kernel<true>, grid, dim3(256), 0, st, row_meta, qbuf, kv_layer, part_o, part_ml);
    else hipLaunchKernelGGL(small_attention_kernel<false>, grid, dim3(256), 0, st, row_meta, qbuf, kv_layer, part_o, part_ml);
    HIP_CHECK(hipGetLastError());
}

void launch_small_proj(const float* part_o, const float* part_ml, const int* row_meta, const float* Wt, float* slabs, int M, int split,
                       hipStream_t st) {
    check_rows(M);
    AUR_REQUIRE(part_o && part_ml && row_meta && Wt && slabs && (split == 64 || split == 128), "small_proj: arguments");
    SmallGemvArgs a{};
    a.Wt = Wt; a.part_o = part_o; a.part_ml = part_ml; a.row_meta = row_meta; a.split = split; a.slabs = slabs; a.M = M;
    trace_launch("small_gemv_kernel<proj>");
    hipLaunchKernelGGL(small_gemv_kernel<kSmProj>, dim3(kHidden / 16, kSmallSlabs), dim3(256), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

void launch_small_fc(const SmallRowsIn& in, const float* Wt, const float* c1, const float* c2, float eps, bool gelu_erf_form, float* act, int M,
                     hipStream_t st) {
    check_rows(M);
    AUR_REQUIRE(in.hs_in && Wt && c1 && c2 && act && in.hs_out != in.hs_in && (!in.slabs || in.slab_bias), "small_fc: arguments");
    SmallGemvArgs a{};
    a.in = in; a.Wt = Wt; a.c1 = c1; a.c2 = c2; a.eps = eps; a.act = act; a.gelu_erf = gelu_erf_form ? 1 : 0; a.M = M;
    trace_launch("small_gemv_kernel<fc>");
    hipLaunchKernelGGL(small_gemv_kernel<kSmFc>, dim3(4 * kHidden / 16, 1), dim3(256), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

void launch_small_proj2(const float* act, const float* Wt, float* slabs, int M, hipStream_t st) {
    check_rows(M);
    AUR_REQUIRE(act && Wt && slabs, "small_proj2: arguments");
    SmallGemvArgs a{};
    a.Wt = Wt; a.act_in = act; a.slabs = slabs; a.M = M;
    trace_launch("small_gemv_kernel<proj2>");
    hipLaunchKernelGGL(small_gemv_kernel<kSmProj2>, dim3(kHidden / 16, kSmallSlabs), dim3(256), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

void launch_small_final_rows(const SmallRowsIn& in, int mtt, const int* sample_slot, const float* lnf_w, const float* lnf_b, const float* fn_w,
                             const float* fn_b, float* ybuf, float* latents, long lat_slot_stride, const int* slot_ngen, int max_lat_rows, int Ms,
                             float eps, hipStream_t st) {
    check_rows(Ms);
    trace_launch("small_final_rows_kernel");
    hipLaunchKernelGGL(small_final_rows_kernel, dim3(1), dim3(256), 0, st, in, mtt, sample_slot, lnf_w, lnf_b, fn_w, fn_b, ybuf, latents,
                       lat_slot_stride, slot_ngen, max_lat_rows, Ms, eps);
    HIP_CHECK(hipGetLastError());
}

}  // namespace aur
