// Launch wrappers for the xtts2-gpt token-loop kernels (gpt_kernels.hip).
#pragma once
#include "common.h"

namespace aur {

constexpr int kHidden = 1024;
constexpr int kHeads = 16;
constexpr int kHeadDim = 64;
constexpr int kKvBlockTokens = 16;                                    // paged-KV block size (vLLM default)
constexpr long kMaxKvBlocksPerLayer = 32767;                            // paged_attention_kernel addresses a layer's pool with 32-bit byte offsets (128 KiB per fp32 block)
constexpr long kKvBlockElems = 2L * kHeads * kKvBlockTokens * kHeadDim;  // one layer, K and V

// Optional bias + GELU epilogue of the prefill FC GEMM: act[m][n] = gelu(total + bias[n]) is written instead of the slab.
// erf: 1 = erf form (config.json "activation_function": "gelu"), 0 = tanh form ("gelu_new").
// QKV mode (qbuf != nullptr, no gelu; split-arithmetic kernel only): the slab is not written either -- column n of row m plus
// bias[n] goes to qbuf[m][n] (n < 1024) or into the K / V page of (row_slot[m], row_pos[m]) (the work of qkv_epilogue_kernel, which
// then is not launched: one launch and one 56 MB write + read per layer less at 4 544 prompt rows).
struct GemmGelu {
    const float* bias;
    float* act;
    int erf;
    float* qbuf = nullptr;
    void* kv_layer = nullptr;
    const int* row_slot = nullptr;
    const int* row_pos = nullptr;
    const int* block_tables = nullptr;
    int max_blocks = 0;
    int kv_half = 0;
};

// ---- decode-regime GEMM (M = live sequences; every weight leaves HBM once per step) ---------------------------------
// One workgroup = 16 waves = one [16*MT rows] x [16 columns] output tile over the FULL K: wave w owns K-slice
// [c*1024 + 64w, +64) of every 1024-deep chunk c, partial tiles are reduced through LDS in fixed wave order, and the
// epilogue (bias | bias+gelu | bias+residual | QKV split + KV page write) runs on the totals, so there are no split-K
// slabs in HBM and no separate LayerNorm / GELU launches.  LN: the A operand is LayerNorm(X) computed in the prologue
// (two-pass statistics over the full row, which the workgroup holds in registers; K == 1024).
// Weights come packed by pack_wt16: Wt[N/16][K/16][64 lanes][4], lane = 16*q + j, component s
//   = W[16*kb + 4*q + s][16*nt + j], i.e. one 16 x 16 K-by-N block is 1 KiB, contiguous, and the float4 a lane loads IS its
// B fragment for four consecutive v_mfma_f32_16x16x4_f32 steps (the A float4 X[row][16*kb + 4*q .. +3] likewise).
// Activations that feed these GEMMs live in HBM in MFMA-fragment order ("packed rows"): element (row m, column k) of an
// [rows][K] matrix sits at pk_off(m, k, mtt), mtt = allocated rows / 16, i.e. one (16 rows x 16 columns) block is 1 KiB and
// lane (q = (k>>2)&3, j = m&15) of a wave owns its float4 X[m][16*kb + 4q .. +3].  Every A load of the kernel is then one
// contiguous 1 KiB request per wave (the row-major form costs 64 address cycles per wave-load: adjacent lanes are adjacent
// ROWS, 4 KB apart; measured 16 B/clk/CU), and every epilogue writes whole 1 KiB blocks.
__host__ __device__ inline long pk_off(int m, int k, int mtt) {
    return ((((long)(k >> 4) * mtt + (m >> 4)) * 64 + ((k >> 2) & 3) * 16 + (m & 15)) << 2) + (k & 3);
}
enum GemmRowsEpi { kEpiBias = 0, kEpiBiasGelu = 1, kEpiResidual = 2, kEpiQkv = 3 };
struct GemmRowsArgs {
    const float* X;      // packed rows, K columns
    int xmt;             // 16-row tiles allocated in X (pk_off's mtt); >= ceil(M / 64) * 4
    const float* Wt;     // packed (pack_wt16)
    int M, N, K;
    const float* bias;   // [N], required (zeros for none); LayerNorm-folded GEMMs: c2 of launch_fold_ln
    const float* ln_c1;  // LayerNorm-folded GEMMs (ln != 0, K == 1024): c1 of launch_fold_ln, Wt = its folded packed weights
    float eps;
    float* out;          // kEpiBias: row-major out[m][n] (ldo);  kEpiBiasGelu: packed rows (omt);  kEpiResidual: packed rows,
                         // out[m][n] += total + bias;  kEpiQkv: row-major q rows [M][1024]
    int ldo;
    int omt;             // 16-row tiles allocated in a packed `out`
    void* kv_layer;      // kEpiQkv: paged K/V of this layer, written at (position row_meta[m][0], block row_meta[m][kRowMetaWblk])
    int kv_half;         // 1: the K/V pool holds fp16 (aur_config.kv_fp16 throughput mode), else fp32
    const int* row_meta; // kEpiQkv (required): the step's dense per-row K/V addressing written by embed_decode_kernel
                         // ([m][kRowMetaStride] ints: position, slot, write block, block table) -- one load per output element
                         // instead of the row_slot -> slot_kvpos -> block_tables chain
    // LayerNorm statistics travel as per-column-tile partials (mean_t, M2_t = sum (x - mean_t)^2 over the tile's 16 columns):
    // the kernels that WRITE the residual stream (embed_decode, the kEpiResidual epilogue) emit stats[row][tile] for the 64
    // tiles of a 1024-wide row, the LN prologue combines them (Chan's parallel variance, fixed order) instead of reducing the
    // row itself, so its MFMAs can start on the first K block that arrives and need no workgroup-wide reduction.
    const float2* stats_in;   // LN prologue: [rows][64]
    float2* stats_out;        // kEpiResidual (N == 1024): [rows][64], may be nullptr
    int gelu_erf;        // kEpiBiasGelu: 1 = erf form ("gelu"), 0 = tanh form ("gelu_new")
    int prec;            // arithmetic: 0 = exact f32 MFMA (v_mfma_f32_16x16x4_f32); 1 = every operand split exactly into three bf16
                         // terms, six bf16 MFMAs per product with fp32 accumulation (same accuracy class as an fp32 dot product,
                         // not bitwise an fma chain; 2.7x less matrix-pipe time).  aur_config.gemm_f32_exact selects 0.
    // K split over kGemmKsp workgroups per output tile (the K = 4096 projection at M <= 16 rows, where one workgroup per tile
    // leaves 3/4 of the CUs idle): every workgroup runs 16 / kGemmKsp of the 16 K-slices (each slice = the MFMA chain of one wave
    // of the unsplit kernel, bit for bit), publishes its per-wave partial tiles in ksp_buf, takes a ticket on ksp_cnt[tile], and
    // the workgroup that draws the last ticket sums the 16 partials in wave order 0..15 -- the unsplit kernel's order -- and runs
    // the epilogue.  Both null = unsplit.
    float* ksp_buf;      // [tiles][16][256] floats
    unsigned* ksp_cnt;   // [tiles], zero before the first launch (the last arriver resets its counter)
};
// The kernel-side remainder of GemmRowsArgs: what the kernel needs only after its first tile loads are in flight (the leading
// scalar kernel arguments -- Wt, X, bias, ln_c1, out, stats_in and M, N, xmt, omt packed into two dwords -- are preloaded into SGPRs,
// gemm_rows_kernel.inc).
struct GemmRowsTail {
    float2* stats_out;
    const int* row_meta;
    void* kv_layer;
    float* ksp_buf;
    unsigned* ksp_cnt;
    float eps;
    int ldo, kv_half, gelu_erf;
};
constexpr int kGemmKsp = 4;
constexpr int kGemmKspTiles = 256;   // output tiles the scratch is sized for: ksp_buf = kGemmKspTiles * 16 * 256 floats, ksp_cnt = kGemmKspTiles
void launch_gemm_rows(const GemmRowsArgs& a, bool ln, GemmRowsEpi epi, hipStream_t st);
// Workgroup shape the launcher picks for a GEMM kind at M rows: 16*mt rows x 16*ntl columns, nw waves (K split nw ways);
// nt: non-temporal weight loads (M <= 16: every weight tile has one reader).
struct GemmRowsShape {
    int mt, nw, ntl;
    bool nt;
    int ksp;   // workgroups per output tile along K (1 = unsplit); > 1 needs GemmRowsArgs::ksp_buf / ksp_cnt
};
GemmRowsShape gemm_rows_shape(int M, int N, int K, bool ln);

// Wt = pack_wt16(W), W row-major [K][ldw], N % 16 == 0, K % 16 == 0
void launch_pack_wt16(const float* W, int ldw, float* Wt, int K, int N, hipStream_t st);
// LayerNorm folded into the GEMM that follows it:  LN(x) W + b = rstd * (x Wf - mean * c1) + c2  with Wf = diag(gamma) W,
// c1[n] = sum_k Wf[k][n], c2[n] = sum_k beta[k] W[k][n] + b[n].  Writes Wt = pack_wt16(Wf), c1 and c2 ([N] each); `scratch` holds
// K * N floats (Wf unpacked).  The decode GEMM then multiplies the RAW residual rows and applies the row statistics in its
// epilogue, so nothing in front of its MFMAs depends on them.
void launch_fold_ln(const float* W, int ldw, const float* gamma, const float* beta, const float* bias, float* scratch, float* Wt,
                    float* c1, float* c2, int K, int N, hipStream_t st);
// decode tail: y[j] = final_norm(ln_f(h[j])) -> ybuf;  latents[slot][ngen[slot]] = final_norm(y[j]);  h and ybuf are
// packed rows with `mtt` 16-row tiles, latents row-major
void launch_final_rows(const float* h, int mtt, const int* sample_slot, const float* lnf_w, const float* lnf_b, const float* fn_w,
                       const float* fn_b, float* ybuf, float* latents, long lat_slot_stride, const int* slot_ngen,
                       int max_lat_rows, int Ms, float eps, hipStream_t st);

// Prefill-regime GEMM (M = sum of prompt rows, hundreds to thousands): P[m][n] = sum_k X[m][k] * W[k][n] as ONE slab, or
// act[m][n] = gelu_new(that + bias[n]) when `gelu` is given.  128 x 128 output tile per workgroup, K in steps of 16 through a
// double-buffered LDS stage, exact-f32 v_mfma_f32_32x32x2_f32 (each wave a 64 x 64 sub-tile); W in the file's [K][N] layout.
// Every output element sums k in ascending order whatever M is, so prefill results do not depend on what else was admitted
// in the same step (all prefill-type calls use this kernel, small M included).  N % 64 == 0, K % 16 == 0.
// prec = 1: the decode GEMMs' three-way bf16 split arithmetic (gemm_tile_split_kernel, v_mfma_f32_32x32x16_bf16, the operands
// split on their way into LDS) — same tiles, same k order per output element; prec = 0: exact-f32 MFMA.
// slabs > 1: split-K, slab s = K / slabs consecutive k, written to P + s * M * N (the consumer sums the slabs in a fixed order,
// rows_ln / qkv_epilogue's S); the slab count is a property of the call site, never of M, so results stay batch-invariant.
// wsplit (prec = 1, N % 128 == 0): W pre-split into its three bf16 planes by launch_pack_wsplit (3 * K * N * 2 bytes); the kernel
// then reads 16-byte pieces of the planes instead of splitting the fp32 weights again per row tile.  Bitwise the same result.
void launch_gemm_tile(const float* X, int ldx, const float* W, float* P, int M, int N, int K, hipStream_t st,
                      const GemmGelu* gelu = nullptr, int prec = 0, int slabs = 1, const void* wsplit = nullptr);
void launch_pack_wsplit(const float* W, int ldw, void* out, int K, int N, hipStream_t st);

// h[m] += sum_s P[s][m] + bias (if S > 0); out[m] = LayerNorm(h[m]; gamma, beta, eps).  Rows of 1024.
void launch_rows_ln(const float* P, int S, const float* bias, float* h, const float* gamma, const float* beta,
                    float* out, int M, float eps, hipStream_t st);

// qkv = sum_s P[s] + bias;  q -> qbuf[m][1024];  k,v -> paged cache of this layer at (slot, pos)
// kv_half (here and below): the pool stores fp16 K/V (same [block][K|V][head][16][64] layout, half the bytes); values are
// rounded to nearest on the way in, scores / softmax / P.V stay fp32
void launch_qkv_epilogue(const float* P, int S, const float* bias, float* qbuf, void* kv_layer,
                         const int* row_slot, const int* row_pos, const int* slot_kvpos,
                         const int* block_tables, int max_blocks, int M, hipStream_t st, bool kv_half = false);

// decode rows: causal attention of every row against its sequence's paged K/V (keys 0..pos), 16 heads x 64.  The row's position
// and block table come from the step's dense row_meta table (embed_decode_kernel), max_blocks = entries per row.
// out_mtt > 0: `out` is written as packed rows with that many 16-row tiles (decode chain, A operand of the proj GEMM)
void launch_paged_attention(const float* qbuf, const void* kv_layer, const int* row_meta, int max_blocks, float* out, int M,
                            hipStream_t st, int out_mtt = 0, bool kv_half = false);

// prompt rows (explicit positions, row-major output): exact-f32 MFMA score / PV tiles over query blocks of up to 32 consecutive
// rows of one sequence, qblk[i] = {first row, rows} (row_pos ascends by one inside a block)
void launch_prompt_attention(const float* qbuf, const void* kv_layer, const int2* qblk, int n_qblk, const int* row_slot, const int* row_pos,
                             const int* block_tables, int max_blocks, float* out, hipStream_t st, bool kv_half = false);

// prompt rows: desc[m] = {kind, a, b, _}: kind 0 -> spk_cond[b][a][:], 1 -> text_emb[a]+text_pos[b], 2 -> wte[a]+wpe[b]
void launch_embed_prompt(const int4* desc, const float* spk_cond, const float* text_emb, const float* text_pos,
                         const float* wte, const float* wpe, float* h, int M, hipStream_t st);
// decode rows: h[m] = wte[tok[slot]] + wpe[pos[slot]]
// h_mtt > 0: h is written as packed rows and stats[row][64] receives the LayerNorm partials of each row (GemmRowsArgs).
// row_meta (optional, [M][kRowMetaStride] ints): the step's per-row K/V addressing, gathered once per step for the 30
// attention launches and QKV epilogues: [0] = K/V position of the new token (slot_kvpos), [1] = slot,
// [kRowMetaWblk] = the block that position falls in (where the QKV epilogue writes K/V), [kRowMetaBt ..] = the slot's block table.
constexpr int kRowMetaStride = 96, kRowMetaBt = 8, kRowMetaWblk = 2;   // (66 table entries + the attention kernel's look-ahead)
void launch_embed_decode(const int* row_slot, const int* slot_tok, const int* slot_pos, const float* wte,
                         const float* wpe, float* h, int M, hipStream_t st, int h_mtt = 0, float2* stats = nullptr,
                         int* row_meta = nullptr, const int* slot_kvpos = nullptr, const int* block_tables = nullptr,
                         int max_blocks = 0);

// y[j] = final_norm(xn[sample_row[j]]);  latents[slot][lat_idx[slot]] = final_norm(y[j])   (double final_norm,
// XTTSv2.py:685-687 on top of vllm_mm_gpt.py:671)
void launch_final_norm(const float* xn, const int* sample_row, const int* sample_slot, const float* gamma,
                       const float* beta, float* ybuf, float* latents, long lat_slot_stride,
                       const int* slot_ngen, int max_lat_rows, int Ms, float eps, hipStream_t st);

void launch_double_norm_rows(const float* src, float* dst, int n, const float* gamma, const float* beta, float eps,
                             hipStream_t st);

// test support: *cnt += number of 32-bit words in which x and y differ
void launch_count_mismatch(const void* x, const void* y, long n_words, unsigned long long* cnt, hipStream_t st);
void launch_lane_xor_selftest(unsigned seed, int blocks, unsigned long long* cnt, hipStream_t st);   // test support: lane_xor / wave_sum / wave_max vs the shuffles they replace

constexpr int kTokFinishedBit = 1 << 30;
struct SamplerArgs {
    const float* P;          // [S][Ms][Npad] logits slabs
    int S, Ms, Npad, V;
    const float* bias;       // [Npad]
    const int* sample_slot;  // [Ms]
    const int* next_kvpos;   // [Ms] or nullptr (decode: kvpos += 1)
    // per-slot state
    unsigned char* seen;     // [slots][V_pad=1040]
    int* slot_tok;
    int* slot_pos;
    int* slot_kvpos;
    int* slot_ngen;
    int* slot_finished;
    const float* temperature;
    const float* top_p;
    const int* top_k;
    const float* rep_penalty;
    const int* max_tokens;
    const int* ignore_stop;
    const unsigned* seed;
    int* out_tok;            // [Ms]: token | kTokFinishedBit when the sequence finished with it; -1 for a ghost row
    float* dbg_logits;       // optional [Ms][V] penalised logits
    int stop_token;
    int force_full_sort;     // 1: always take the 2048-key bitonic path (A/B of the top-k fast path)
};
void launch_sampler(const SamplerArgs& a, hipStream_t st);
constexpr int kSeenStride = 1040;

}  // namespace aur
