// Launch wrappers of the small-batch decode chain (small_kernels.hip): the GPT2Block linears and the decode attention of
// vLLM's GPT2Block (vllm_mm_gpt.py:757-761) at M <= kSmallRows live sequences -- BASELINE configs[1], one utterance.
//
// At one row the MFMA decode chain (gemm_rows_kernel + paged_attention_kernel) is bound by what stands between its launches, not by
// bytes: the K = 4096 projection needs three dependent memory trips (partial tiles -> ticket -> read-back), the attention walks a
// sequence's whole context in ONE workgroup per head (7 us at 244 tokens, 10 at 384), and every launch pays a 16-row MFMA tile for
// one live row.  This chain keeps five launches per layer and gives each of them exactly ONE memory round trip:
//   small_qkv   h = hs + bias + sum of the previous projection's K slabs (formed by every workgroup in its prologue, written back by
//               workgroup 0) -> LayerNorm statistics -> 16-column GEMV tile over the LN-folded packed weights -> q rows, K/V pages
//   small_attn  one workgroup per (row, head, 64-token split): unnormalised partial (max, sum, out[64]) per split, no cross-split merge
//   small_proj  merges the splits of its four heads in its prologue, GEMV over a quarter of K -> one of four K slabs (no epilogue)
//   small_fc    h = hs + bias + proj slabs -> statistics -> GEMV -> GELU -> activation rows
//   small_proj2 GEMV over a quarter of K = 4096 -> K slab
// Exact fp32 arithmetic (v_fma_f32), every sum in a fixed order that does not depend on the number of live rows: a row's bits are the
// same alone and beside up to three others (solo == batch inside this path).  They are NOT the bits of the MFMA chain (bf16 x 3 split
// products, another summation tree): which chain a step takes depends on the live rows alone (M <= kSmallRows), the goldens against the
// fp32 CPU oracle hold on both.  Weights are the decode chain's packed copies (pack_wt16 / launch_fold_ln): nothing is stored twice.
#pragma once
#include "gpt_kernels.h"

namespace aur {

constexpr int kSmallRows = 4;        // live sequences this chain takes
constexpr int kSmallSplit = 64;      // tokens per attention split
constexpr int kSmallMaxSplits = 17;  // ceil(1056 / 64): max_model_len 1047 (kMaxBlocks * 16 = 1056)
constexpr int kSmallSlabs = 4;       // K parts of the two projections

// How a kernel forms its input rows: h[m] = hs_in[m] + slab_bias + slabs[0][m] + .. + slabs[3][m] (slabs == nullptr: hs_in as it is).
struct SmallRowsIn {
    const float* hs_in;      // [kSmallRows][1024] row-major
    const float* slabs;      // [kSmallSlabs][kSmallRows][1024] or nullptr
    const float* slab_bias;  // [1024] (with slabs)
    float* hs_out;           // [kSmallRows][1024]: the rows formed, written by workgroup 0 (must differ from hs_in); may be nullptr
};

// Wt: pack_wt16 copy of the LN-folded [1024][3072] matrix; c1 / c2: launch_fold_ln's epilogue vectors
void launch_small_qkv(const SmallRowsIn& in, const float* Wt, const float* c1, const float* c2, float eps, float* qbuf, void* kv_layer,
                      bool kv_half, const int* row_meta, int M, hipStream_t st);
// part_o [kSmallRows][16][kSmallMaxSplits][64], part_ml [kSmallRows][16][kSmallMaxSplits][2] (max, sum); n_splits = splits of the longest row
void launch_small_attention(const float* qbuf, const void* kv_layer, const int* row_meta, float* part_o, float* part_ml, int M, int n_splits,
                            bool kv_half, hipStream_t st);
// split: tokens per attention split (64 with the fp32 K/V pool, 128 with fp16: one token-loop iteration of the attention kernel)
void launch_small_proj(const float* part_o, const float* part_ml, const int* row_meta, const float* Wt, float* slabs, int M, int split,
                       hipStream_t st);
void launch_small_fc(const SmallRowsIn& in, const float* Wt, const float* c1, const float* c2, float eps, bool gelu_erf_form, float* act, int M,
                     hipStream_t st);
void launch_small_proj2(const float* act, const float* Wt, float* slabs, int M, hipStream_t st);
// decode tail on the chain's rows: as launch_final_rows, the residual rows formed from `in` (ybuf: packed rows with mtt tiles)
void launch_small_final_rows(const SmallRowsIn& in, int mtt, const int* sample_slot, const float* lnf_w, const float* lnf_b, const float* fn_w,
                             const float* fn_b, float* ybuf, float* latents, long lat_slot_stride, const int* slot_ngen, int max_lat_rows, int Ms,
                             float eps, hipStream_t st);

}  // namespace aur
