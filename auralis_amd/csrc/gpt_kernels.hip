// xtts2-gpt token-loop kernels for gfx950 (MI355X): fp32 storage, exact-f32 MFMA GEMMs, paged KV.
//
// Reference arithmetic: src/auralis/models/xttsv2/components/vllm_mm_gpt.py (GPT2Model.forward 787-849,
// compute_logits 664-688, sample 691-712), components/vllm/hijack.py:49-88 (repetition penalty), and the
// vLLM 0.6.4.post1 GPT2Block / Sampler semantics restated in SURVEY.md Appendix A2-A5.
#include <type_traits>

#include "gpt_kernels.h"

namespace aur {

// paged KV layout of one layer: [block][K|V][head][token_in_block(16)][64]
__device__ __forceinline__ long kv_offset(int blk, int kv, int head, int tok) {
    return (((long)blk * 2 + kv) * kHeads + head) * (kKvBlockTokens * kHeadDim) + (long)tok * kHeadDim;
}

// ------------------------------------------------------------------------------------------------
// Sum of S partial slabs of one float4 column group plus the bias, in the fixed order ((p[0] + p[1]) + ... + p[S-1]) + bias
// (the prefill GEMM writes one slab, S == 1; S == 0 callers skip it).  Loads go out four at a time at clamped slab indices.
__device__ __forceinline__ f32x4 slab_sum(const float* __restrict__ p0, long sstride, int S, const float* __restrict__ bias_n) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_n);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 t = zero;
    for (int s = 0; s < S; s += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p0 + (long)s * sstride);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p0 + (long)min(s + 1, S - 1) * sstride);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p0 + (long)min(s + 2, S - 1) * sstride);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p0 + (long)min(s + 3, S - 1) * sstride);
        t += a;
        t += (s + 1 < S) ? b : zero;
        t += (s + 2 < S) ? c : zero;
        t += (s + 3 < S) ? d : zero;
    }
    return t + bv;
}

// ------------------------------------------------------------------------------------------------
// Prefill-regime GEMM: see gpt_kernels.h.  LDS stage per K step of 16: A tile stored k-major ([k][128 rows], so that the MFMA A
// operand — lane = row, two k per instruction — is a conflict-free ds_read_b32), B tile [k][128 columns] as loaded.  The global
// loads of step i+1 are issued before the 32 MFMAs of step i and written to the other buffer after them: one barrier per step.
template <int BM, int BN, bool GELU>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                           float* __restrict__ P, int M, int N, int K, GemmGelu ep) {
    static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64), "tile shapes");
    constexpr int BK = 16, PA = BM + 4, PB = BN + 4;
    constexpr int MI = BM / 64, NI = BN / 64;   // 32 x 32 MFMA tiles per wave (waves form a 2 x 2 grid over the tile)
    constexpr int LA = BM / 64, LB = BN / 64;   // float4 loads per thread and K step for the A / B tile
    __shared__ __attribute__((aligned(16))) float As[2][BK][PA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][PB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 1, wn = wv & 1;
    // XCD-aware order: the row tiles of one column tile get ids 8 apart, so the weight panel of a column tile stays in one
    // XCD's L2 while the activation panels stream past it
    const int n_nt = N / BN, n_mt = (M + BM - 1) / BM;
    int ntile, mtile;
    {
        const int L = blockIdx.x;
        if ((n_nt & 7) == 0) {
            const int xcd = L & 7, slot = L >> 3;
            mtile = slot % n_mt;
            ntile = (slot / n_mt) * 8 + xcd;
        } else {
            mtile = L % n_mt;
            ntile = L / n_mt;
        }
    }
    const int m0 = mtile * BM, n0 = ntile * BN;
    // split-K slabs (grid.y): slab s multiplies columns [s*K, (s+1)*K) of X with the matching rows of W into P[s] (K = slab depth)
    X += (long)blockIdx.y * K;
    W += (long)blockIdx.y * K * N;
    P += (long)blockIdx.y * M * N;
    // staging roles: A = float4 of 4 consecutive k for row tid/4 (+64), B = float4 of 4 columns for k row tid/(BN/4) (+256/(BN/4))
    constexpr int BT = BN / 4;                  // threads per B row
    const int a_row = tid >> 2, a_kq = tid & 3, b_kr = tid / BT, b_n4 = tid % BT;
    const float* ap[LA];
    const float* bp[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = m0 + a_row + 64 * i;
        ap[i] = X + (long)(r < M ? r : M - 1) * ldx + 4 * a_kq;   // rows >= M alias the last row: never stored
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) bp[i] = W + (long)(b_kr + (256 / BT) * i) * N + n0 + 4 * b_n4;
    f32x4 ga[LA], gb[LB];
    auto g_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) ga[i] = *reinterpret_cast<const f32x4*>(ap[i] + k0);
#pragma unroll
        for (int i = 0; i < LB; ++i) gb[i] = *reinterpret_cast<const f32x4*>(bp[i] + (long)k0 * N);
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) As[buf][4 * a_kq + c][a_row + 64 * i] = ga[i][c];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][b_kr + (256 / BT) * i][4 * b_n4]) = gb[i];
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    g_load(0);
    s_store(0);
    __syncthreads();
    const int n_steps = K / BK;
    for (int st = 0; st < n_steps; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < n_steps;
        if (more) g_load((st + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // the next step's loads are in flight before this step's MFMAs
        const float* arow = &As[cur][hi][wm * (BM / 2) + l31];
        const float* brow = &Bs[cur][hi][wn * (BN / 2) + l31];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[MI], bv[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av[mi] = arow[2 * kk * PA + 32 * mi];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bv[ni] = brow[2 * kk * PB + 32 * ni];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) s_store(cur ^ 1);
        __syncthreads();
    }
    // D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * (BN / 2) + ni * 32 + l31;
            float bv = 0.f;
            if (GELU) bv = ep.bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < M) {
                    if (GELU) ep.act[(long)row * N + col] = ep.erf ? gelu_erf(acc[mi][ni][r] + bv) : gelu_new(acc[mi][ni][r] + bv);
                    else P[(long)row * N + col] = acc[mi][ni][r];
                }
            }
        }
}

// The same GEMM in the decode GEMMs' split arithmetic (gemm_rows_kernel, PREC = 1): every fp32 operand is h + m + l exactly
// (three bf16), a product keeps the six terms down to 2^-24 of it, on v_mfma_f32_32x32x16_bf16 with an accumulator for the h*h
// term and one for the cross terms — 6 MFMAs of 32 cycles per 32 x 32 x 16 block against 8 x 64 for v_mfma_f32_32x32x2_f32.
// The operands are split once per tile, on the way into LDS: both tiles are staged k-contiguous ([row][16 k] bf16 rows, 48-B
// pitch: conflict-free ds_read_b128 fragments), the activations from float4 loads along k, the weights ([K][N] in memory) from
// four coalesced dword loads k .. k+3 of one column.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3_bf16x4(const f32x4& x, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 hh = (__bf16)x[i];
        const float r = x[i] - (float)hh;
        const __bf16 mm = (__bf16)r;
        h[i] = hh;
        m[i] = mm;
        l[i] = (__bf16)(r - (float)mm);
    }
}

// BD: the weight operand comes PRE-SPLIT (launch_pack_wsplit: the three bf16 planes of W, per 128-column tile and k step one
// contiguous 12 KB block [plane][128 n][16 k]).  Weights are static, and half of this kernel's VALU work and LDS-write
// instructions was splitting the same weights again for every row tile of every prefill: with BD a thread fetches three 16-byte
// pieces (8 consecutive k of one column, one per plane; consecutive threads, consecutive addresses) and writes them to LDS as
// they are -- 3 x global_load_dwordx4 + 3 x ds_write_b128 instead of 8 x global_load_dword + the split + 6 x ds_write_b64; the
// two-steps-ahead register pipeline, the padded rows and the fragment reads are unchanged.  Same split, same MFMA order: bitwise the
// result of splitting on the fly.  (Staging the planes by LDS-DMA instead measured SLOWER than no pre-split at all: 30.8 ms per
// prefill with one k step of lead, 34.0 with two and three buffers, against 29.4 -- three 1-KiB copies per wave and step cost
// more issue time inside the MFMA stream than they save)
template <int BM, int BN, bool GELU, bool BD = false>
__global__ __launch_bounds__(256, 2) void gemm_tile_split_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                                 float* __restrict__ P, int M, int N, int K, GemmGelu ep,
                                                                 const __bf16* __restrict__ Wsp = nullptr) {
    static_assert((BM == 128 || BM == 64) && (BN == 128 || BN == 64), "tile shapes");
    constexpr int BK = 16, RS = 24;              // bf16 per LDS row: 16 k + 8 pad
    constexpr int MI = BM / 64, NI = BN / 64;    // 32 x 32 MFMA tiles per wave (waves form a 2 x 2 grid over the tile)
    constexpr int LA = BM / 64;                  // float4 loads (4 k of one row) per thread and K step
    constexpr int LB = BN / 64;                  // k-quads of one column per thread and K step (4 dword loads each)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][3][BM * RS];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][3][BN * RS];
    __shared__ int2 rowkv[BM];   // QKV mode: (K/V block, token in block) of the tile's rows
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5, wm = wv >> 1, wn = wv & 1;
    const int n_nt = N / BN, n_mt = (M + BM - 1) / BM;
    int ntile, mtile;
    {   // XCD-aware order, as gemm_tile_kernel
        const int L = blockIdx.x;
        if ((n_nt & 7) == 0) {
            const int xcd = L & 7, slot = L >> 3;
            mtile = slot % n_mt;
            ntile = (slot / n_mt) * 8 + xcd;
        } else {
            mtile = L % n_mt;
            ntile = L / n_mt;
        }
    }
    const int m0 = mtile * BM, n0 = ntile * BN;
    if (!GELU && ep.qbuf && tid < BM) {   // (read in the epilogue, behind the K loop's barriers)
        const int r = min(m0 + tid, M - 1), slot = ep.row_slot[r], pos = ep.row_pos[r];
        rowkv[tid] = make_int2(ep.block_tables[(long)slot * ep.max_blocks + pos / kKvBlockTokens], pos % kKvBlockTokens);
    }
    // split-K slabs (grid.y): slab s multiplies columns [s*K, (s+1)*K) of X with the matching rows of W into P[s] (K = slab depth)
    X += (long)blockIdx.y * K;
    if constexpr (!BD) W += (long)blockIdx.y * K * N;
    P += (long)blockIdx.y * M * N;
    // pre-split weights: block (128-column tile t, k step s) of 3 x 128 x 16 bf16 at ((t * ksteps_total) + s) * 6144 elements
    const int ks_total = (int)gridDim.y * (K / BK), ks0 = (int)blockIdx.y * (K / BK);
    const __bf16* wsp_tile = BD ? Wsp + ((long)(n0 / 128) * ks_total + ks0) * (3 * 128 * 16) + (n0 % 128) * 16 : nullptr;
    // pre-split B: piece j of this thread = plane (tid + 256 j) / (2 BN), column ((tid + 256 j) / 2) % BN, k half (tid & 1)
    constexpr int LBD = 3 * BN * 2 / 256;        // 16-byte pieces per thread and k step (3 at BN = 128; 1.5 -> 2 with a guard at BN = 64)
    constexpr int LBDc = (3 * BN * 2 + 255) / 256;
    const int a_row = tid >> 2, a_kq = tid & 3;          // A: rows a_row (+64), k-quad a_kq
    const int b_n = tid % BN, b_kq = tid / BN;           // B: column b_n, k-quads b_kq (+ 256 / BN)
    const float* ap[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = m0 + a_row + 64 * i;
        ap[i] = X + (long)(r < M ? r : M - 1) * ldx + 4 * a_kq;   // rows >= M alias the last row: never stored
    }
    const float* bp = W + n0 + b_n;
    // Two register sets: the global loads of step s + 2 are issued before the MFMAs of step s and parked until step s + 1 has
    // been computed -- two steps of cover for the operand latency (the activation panel streams from HBM / Infinity Cache and a
    // step's 24 MFMAs last only ~0.3 us; with one step of cover the kernel ran at a quarter of the matrix pipe).
    // (with BD the register set of B holds the 16-byte pieces of the planes, bit-cast into the same f32x4 slots)
    constexpr int LBR = BD ? LBDc : LB;
    f32x4 ga[2][LA], gb[2][LBR];
    auto g_load = [&](f32x4 (&a4)[LA], f32x4 (&b4)[LBR], int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) a4[i] = *reinterpret_cast<const f32x4*>(ap[i] + k0);
        if constexpr (BD) {
            const __bf16* blk = wsp_tile + (long)(k0 / BK) * (3 * 128 * 16);
#pragma unroll
            for (int j = 0; j < LBDc; ++j) {
                // piece index in [0, 3 * BN * 2); at BN = 64 the upper half of the threads has no second piece and loads the first
                // one again (not stored): no branch around a load, see the note on the step lambda
                const int u = (LBD * 256 == 3 * BN * 2 || tid + 256 * j < 3 * BN * 2) ? tid + 256 * j : tid;
                const int p = u / (2 * BN), r = u - p * (2 * BN);
                b4[j] = *reinterpret_cast<const f32x4*>(blk + p * (128 * 16) + (r >> 1) * 16 + (r & 1) * 8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < LB; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) b4[i][c] = bp[(long)(k0 + 4 * (b_kq + (256 / BN) * i) + c) * N];
        }
    };
    auto s_store = [&](const f32x4 (&a4)[LA], const f32x4 (&b4)[LBR], int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            bf16x4 h, m, l;
            split3_bf16x4(a4[i], h, m, l);
            const int o = (a_row + 64 * i) * RS + 4 * a_kq;
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = h;
            *reinterpret_cast<bf16x4*>(&As[buf][1][o]) = m;
            *reinterpret_cast<bf16x4*>(&As[buf][2][o]) = l;
        }
        if constexpr (BD) {
#pragma unroll
            for (int j = 0; j < LBDc; ++j) {
                const int u = tid + 256 * j;
                if (LBD * 256 == 3 * BN * 2 || u < 3 * BN * 2) {
                    const int p = u / (2 * BN), r = u - p * (2 * BN);
                    *reinterpret_cast<f32x4*>(&Bs[buf][p][(r >> 1) * RS + (r & 1) * 8]) = b4[j];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                bf16x4 h, m, l;
                split3_bf16x4(b4[i], h, m, l);
                const int o = b_n * RS + 4 * (b_kq + (256 / BN) * i);
                *reinterpret_cast<bf16x4*>(&Bs[buf][0][o]) = h;
                *reinterpret_cast<bf16x4*>(&Bs[buf][1][o]) = m;
                *reinterpret_cast<bf16x4*>(&Bs[buf][2][o]) = l;
            }
        }
    };
    f32x16 acc[MI][NI], lo[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = lo[mi][ni][r] = 0.f;
    const int n_steps = K / BK;
    g_load(ga[0], gb[0], 0);
    g_load(ga[1], gb[1], BK);
    s_store(ga[0], gb[0], 0);
    __syncthreads();
    const int ao = (wm * (BM / 2) + l31) * RS + 8 * hi, bo = (wn * (BN / 2) + l31) * RS + 8 * hi;
    // step st: LDS buffer st & 1 holds it, register set (st + 1) & 1 holds step st + 1, set st & 1 is free for step st + 2
    auto step = [&](auto PAR, int st) {
        constexpr int par = decltype(PAR)::value;
        // No branch around the loads or the LDS stores (the last two steps load the last k block again and the last step stores
        // it where nobody reads): hipcc's waitcnt insertion takes the stricter of the two paths at every join, and with
        // `if (st + 2 < n_steps)` around g_load it drained vmcnt to 0 ahead of the LDS stores of EVERY step (r04 ISA:
        // `s_waitcnt vmcnt(4..0)` before the ds_writes, `vmcnt(2..0)` before the next load group) -- the set loaded "two steps
        // ahead" had one step of cover and each step stalled on its own loads.  Straight-line, the count is exact (vmcnt(LA+LB)).
        g_load(ga[par], gb[par], min(st + 2, n_steps - 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // those loads are in flight before this step's MFMAs
        bf16x8 a[3][MI], b[3][NI];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[p][mi] = *reinterpret_cast<const bf16x8*>(&As[par][p][ao + 32 * mi * RS]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[p][ni] = *reinterpret_cast<const bf16x8*>(&Bs[par][p][bo + 32 * ni * RS]);
        }
        // term order per accumulator as in gemm_rows_kernel (l*h.. first, h*h into its own accumulator); the tiles are the
        // inner loop so that back-to-back MFMAs never share an accumulator
#define AUR_TERM(DST, PA, PB)                                                                                           \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                  \
        DST[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][mi], b[PB][ni], DST[mi][ni], 0, 0, 0);
        AUR_TERM(lo, 0, 2) AUR_TERM(lo, 2, 0) AUR_TERM(lo, 1, 1) AUR_TERM(lo, 0, 1) AUR_TERM(lo, 1, 0) AUR_TERM(acc, 0, 0)
#undef AUR_TERM
        s_store(ga[par ^ 1], gb[par ^ 1], par ^ 1);
        __syncthreads();
    };
    for (int st = 0; st < n_steps; st += 2) {   // K % 32 == 0 (launcher)
        step(std::integral_constant<int, 0>{}, st);
        step(std::integral_constant<int, 1>{}, st + 1);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = n0 + wn * (BN / 2) + ni * 32 + l31;
            const bool qkv = !GELU && ep.qbuf != nullptr;
            float bv = 0.f;
            if (GELU || qkv) bv = ep.bias[col];
            const int qu = col / kHidden, qd = col - qu * kHidden;   // QKV mode: q / k / v and the column inside it
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, row = m0 + rl;
                const float v = acc[mi][ni][r] + lo[mi][ni][r];
                if (row < M) {
                    if (GELU) {
                        ep.act[(long)row * N + col] = ep.erf ? gelu_erf(v + bv) : gelu_new(v + bv);
                    } else if (qkv) {
                        const float t = v + bv;
                        if (qu == 0) {
                            ep.qbuf[(long)row * kHidden + qd] = t;
                        } else {
                            const int2 kb = rowkv[rl];
                            const long off = kv_offset(kb.x, qu - 1, qd / kHeadDim, kb.y) + qd % kHeadDim;
                            if (ep.kv_half) reinterpret_cast<_Float16*>(ep.kv_layer)[off] = (_Float16)t;
                            else reinterpret_cast<float*>(ep.kv_layer)[off] = t;
                        }
                    } else {
                        P[(long)row * N + col] = v;
                    }
                }
            }
        }
}

// W[K][ldw] fp32 -> the three bf16 planes of its exact split, blocked for gemm_tile_split_kernel<.., BD = true>:
// out[((n / 128) * (K / 16) + k / 16) * 3 + plane][n % 128][k % 16]
__global__ __launch_bounds__(256) void pack_wsplit_kernel(const float* __restrict__ W, int ldw, __bf16* __restrict__ out, int K, int N) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // one (n, k-quad)
    if (idx >= (long)N * (K / 4)) return;
    const int n = (int)(idx % N), kq = (int)(idx / N);
    const int k = 4 * kq;
    const float* src = W + (long)k * ldw + n;
    const f32x4 v = {src[0], src[ldw], src[2L * ldw], src[3L * ldw]};
    bf16x4 h, m, l;
    split3_bf16x4(v, h, m, l);
    __bf16* o = out + (((long)(n / 128) * (K / 16) + k / 16) * 3) * (128 * 16) + (n % 128) * 16 + (k % 16);
    *reinterpret_cast<bf16x4*>(o) = h;
    *reinterpret_cast<bf16x4*>(o + 128 * 16) = m;
    *reinterpret_cast<bf16x4*>(o + 2 * 128 * 16) = l;
}
void launch_pack_wsplit(const float* W, int ldw, void* out, int K, int N, hipStream_t st) {
    AUR_REQUIRE(N % 128 == 0 && K % 16 == 0 && ldw >= N, "pack_wsplit: N % 128, K % 16");
    trace_launch("pack_wsplit_kernel");
    const long total = (long)N * (K / 4);
    hipLaunchKernelGGL(pack_wsplit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, ldw, reinterpret_cast<__bf16*>(out), K, N);
    HIP_CHECK(hipGetLastError());
}

void launch_gemm_tile(const float* X, int ldx, const float* W, float* P, int M, int N, int Kfull, hipStream_t st,
                      const GemmGelu* gelu, int prec, int slabs, const void* wsplit) {
    AUR_REQUIRE(slabs >= 1 && Kfull % slabs == 0 && (slabs == 1 || !gelu), "gemm_tile: slabs");
    AUR_REQUIRE(!gelu || !gelu->qbuf || (prec == 1 && !gelu->act && N == 3 * kHidden && gelu->bias && gelu->kv_layer && gelu->row_slot && gelu->row_pos && gelu->block_tables),
                "gemm_tile: the QKV epilogue needs the split arithmetic, N = 3072, bias, K/V pool and row addressing");
    const int K = Kfull / slabs;
    AUR_REQUIRE(N % 64 == 0 && K % 16 == 0 && (!prec || K % 32 == 0) && ldx % 4 == 0 && M >= 1, "gemm_tile: shape");
    trace_launch("gemm_tile_kernel");
    const GemmGelu none{nullptr, nullptr, 0};
    const GemmGelu& g = gelu ? *gelu : none;
    const bool ge = gelu && gelu->act;   // (a descriptor without `act` is the QKV mode of the plain kernel)
    // N = 1024 GEMMs (attention and MLP projections): 128 x 128 tiles give 8 x ceil(M/128) workgroups — 288 for a 64-prompt
    // prefill, 1.1 per CU, half the chip idle in the second round — so they run on 64 x 64 tiles (1136 workgroups).  The k order
    // of every output element is the same for both shapes.
    const bool small = N <= 1024 || N % 128 != 0;
#define AUR_GT(KERN, BM_, BN_, GE) \
    hipLaunchKernelGGL((KERN<BM_, BN_, GE>), dim3((unsigned)((N / BN_) * ((M + BM_ - 1) / BM_)), (unsigned)slabs), dim3(256), 0, st, X, ldx, W, P, M, N, K, g)
#define AUR_GTD(BM_, BN_, GE) \
    hipLaunchKernelGGL((gemm_tile_split_kernel<BM_, BN_, GE, true>), dim3((unsigned)((N / BN_) * ((M + BM_ - 1) / BM_)), (unsigned)slabs), dim3(256), 0, st, X, ldx, W, P, M, N, K, g, \
                       reinterpret_cast<const __bf16*>(wsplit))
    if (prec && wsplit && N % 128 == 0) {   // pre-split weight planes: the largest tile shape that still gives ~every CU a workgroup
        const long n128 = (long)slabs * (N / 128) * ((M + 127) / 128), n64 = (long)slabs * (N / 64) * ((M + 127) / 128);
        // (wide GEMMs too: a 500-row pass of a few prompts is 96 tiles of 128 x 128 on 256 CUs)
        if (n128 >= 200) { if (ge) AUR_GTD(128, 128, true); else AUR_GTD(128, 128, false); }
        else if (n64 >= 200) { if (ge) AUR_GTD(128, 64, true); else AUR_GTD(128, 64, false); }
        else { if (ge) AUR_GTD(64, 64, true); else AUR_GTD(64, 64, false); }
    } else if (prec) {
        // split arithmetic, narrow GEMMs (N <= 1024): the largest tile that still gives every CU a workgroup -- 128 x 128 for a
        // 64-prompt prefill (288 workgroups, one round: 34.8 ms per prefill against 36.4 on 128 x 64 = 576 workgroups on 512
        // slots), 128 x 64 and 64 x 64 for smaller batches.  The k order of an output element is the same for every shape.
        const long n128 = (long)slabs * (N / 128) * ((M + 127) / 128), n64 = (long)slabs * (N / 64) * ((M + 127) / 128);
        if (!small || (N % 128 == 0 && n128 >= 200)) { if (ge) AUR_GT(gemm_tile_split_kernel, 128, 128, true); else AUR_GT(gemm_tile_split_kernel, 128, 128, false); }
        else if (n64 >= 200) { if (ge) AUR_GT(gemm_tile_split_kernel, 128, 64, true); else AUR_GT(gemm_tile_split_kernel, 128, 64, false); }
        else { if (ge) AUR_GT(gemm_tile_split_kernel, 64, 64, true); else AUR_GT(gemm_tile_split_kernel, 64, 64, false); }
    } else {
        if (small) { if (ge) AUR_GT(gemm_tile_kernel, 64, 64, true); else AUR_GT(gemm_tile_kernel, 64, 64, false); }
        else { if (ge) AUR_GT(gemm_tile_kernel, 128, 128, true); else AUR_GT(gemm_tile_kernel, 128, 128, false); }
    }
#undef AUR_GT
#undef AUR_GTD
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Decode-regime GEMM: see gpt_kernels.h.  Reduction order of one output element: the MFMA k-chain of wave w over its
// 64-wide slice of every chunk (chunks in order), then waves 0..15 in order — independent of M and of the other rows, so
// continuous batching stays bitwise batch-invariant.
__global__ __launch_bounds__(256) void pack_wt16_kernel(const float* __restrict__ W, int ldw, float* __restrict__ Wt, int K,
                                                        int N) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // one float4 of Wt
    const long total = (long)(N >> 4) * (K >> 4) * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long blk = idx >> 6;
    const int kb = (int)(blk % (K >> 4)), nt = (int)(blk / (K >> 4));
    const int j = lane & 15, q = lane >> 4;
    const float* src = W + (long)(16 * kb + 4 * q) * ldw + 16 * nt + j;
    f32x4 v = {src[0], src[ldw], src[2L * ldw], src[3L * ldw]};
    *reinterpret_cast<f32x4*>(Wt + idx * 4) = v;
}

// LayerNorm folded into a [K][N] weight matrix: Wf[k][n] = gamma[k] * W[k][n] (then packed like pack_wt16), c1[n] = sum_k Wf[k][n],
// c2[n] = sum_k beta[k] * W[k][n] + bias[n] (sums in double).  A workgroup takes 16 columns; its 16 k-slices are combined in a
// fixed order.
__global__ __launch_bounds__(256) void fold_ln_vectors_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias,
                                                              float* __restrict__ c1, float* __restrict__ c2, int K, int N) {
    __shared__ double p1[16][16], p2[16][16];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + col;
    const int k0 = (int)((long)K * sl / 16), k1 = (int)((long)K * (sl + 1) / 16);
    double s1 = 0.0, s2 = 0.0;
    if (n < N)
        for (int k = k0; k < k1; ++k) {
            const float wv = W[(long)k * ldw + n];
            s1 += (double)(gamma[k] * wv);   // the rounded product the GEMM multiplies with
            s2 += (double)beta[k] * (double)wv;
        }
    p1[sl][col] = s1;
    p2[sl][col] = s2;
    __syncthreads();
    if (sl == 0 && n < N) {
        double t1 = 0.0, t2 = 0.0;
        for (int i = 0; i < 16; ++i) {
            t1 += p1[i][col];
            t2 += p2[i][col];
        }
        c1[n] = (float)t1;
        c2[n] = (float)(t2 + (bias ? (double)bias[n] : 0.0));
    }
}
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ gamma,
                                                         float* __restrict__ out, int K, int N) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)K * N) return;
    const int k = (int)(idx / N), n = (int)(idx - (long)k * N);
    out[idx] = gamma[k] * W[(long)k * ldw + n];
}
void launch_fold_ln(const float* W, int ldw, const float* gamma, const float* beta, const float* bias, float* scratch, float* Wt,
                    float* c1, float* c2, int K, int N, hipStream_t st) {
    AUR_REQUIRE(N % 16 == 0 && K % 16 == 0 && ldw >= N, "fold_ln: shape");
    trace_launch("fold_ln");
    hipLaunchKernelGGL(fold_ln_vectors_kernel, dim3((N + 15) / 16), dim3(256), 0, st, W, ldw, gamma, beta, bias, c1, c2, K, N);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)(((long)K * N + 255) / 256)), dim3(256), 0, st, W, ldw, gamma, scratch, K, N);
    HIP_CHECK(hipGetLastError());
    launch_pack_wt16(scratch, N, Wt, K, N, st);
}

void launch_pack_wt16(const float* W, int ldw, float* Wt, int K, int N, hipStream_t st) {
    AUR_REQUIRE(N % 16 == 0 && K % 16 == 0 && ldw >= N, "pack_wt16: shape");
    const long total = (long)(N >> 4) * (K >> 4) * 64;
    trace_launch("pack_wt16_kernel");
    hipLaunchKernelGGL(pack_wt16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, ldw, Wt, K, N);
    HIP_CHECK(hipGetLastError());
}

// PREC = 1: every fp32 operand is split exactly into three bf16 terms (x = h + m + l, round-to-nearest at each step) and a
// product runs as six bf16 MFMAs with fp32 accumulation (h*h into `acc`; h*m, m*h, m*m, h*l, l*h into `lo`; the three dropped
// terms are below 2^-25 of the product): 6 x 17 cycles per 16 x 16 x 32 block instead of 8 x 32 for v_mfma_f32_16x16x4_f32.
// The two float4 a lane holds for K blocks (2p, 2p + 1) form its 8-element fragment; A and B use the same assignment, so the
// packed layouts stay as they are.
__device__ __forceinline__ void split3_bf16(const f32x4& x0, const f32x4& x1, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? x0[i] : x1[i - 4];
        const __bf16 hh = (__bf16)x;
        const float r = x - (float)hh;
        const __bf16 mm = (__bf16)r;
        h[i] = hh;
        m[i] = mm;
        l[i] = (__bf16)(r - (float)mm);
    }
}

// NT: the packed weight tiles are requested with the non-temporal policy.  For M <= 16 rows a tile is read by exactly one
// workgroup (one row group): streamed weights that nobody re-reads should not displace the activations in L2.  (At M = 64 the
// four row groups of a column tile share its weights through L2 and NT measured neutral, profiles/r03_gemm_nt_weights.log.)
// KSP: workgroups per output tile along K (GemmRowsArgs::ksp_buf).  The workgroup has NW waves and runs K-slices KSP * ... of the
// NW * KSP slices: wave w of part p is slice p * NW + w, with exactly the loads and MFMAs wave p * NW + w of the unsplit kernel runs.
// sc1 (write-through) stores publish the partial tiles, a drained vmcnt and a relaxed agent-scope ticket order them, the last
// arriver reads them with sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility: "sc0 sc1 stores and loads both sides").
__device__ __forceinline__ void store_sc1(float* p, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float load_sc1(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// Shape policy.  The waves per workgroup fix the K grouping of the reduction, so they depend on the GEMM kind only (16
// everywhere; the K-split form runs the same 16 slices in 4 workgroups), never on M: a row's result must not change with the
// number of live rows.  Rows x columns per workgroup do not enter the arithmetic (the K order of an output element is the same for
// every tile shape: bitwise equal).  The shapes aim at 192-256 workgroups (256 CUs) for every number g = ceil(M / 16) of 16-row
// groups (tools/gemm_bench, profiles/r04_gemm_bench_m{1,16,32,48,64}.log):
//   LN GEMM, N = 3072 (QKV): 16 rows x 16 g columns for g <= 3 (192 workgroups), 16 x 48 from g = 4 (64 g workgroups)
//   LN GEMM, N = 4096 (FC) : 16 x 16 (g = 1), 16 x 32 (g = 2), 32 x 32 from g = 3 -- 256 workgroups up to 64 rows
//   N = 1024, K = 1024 (proj), head: 16 x 16
//   N = 1024, K = 4096 (proj2): 16 x 16; for g <= 2 split over 4 workgroups of 4 waves per tile (GemmRowsArgs::ksp_buf): 7.9 vs 12.1 us
//       at 1 row, 9.8 vs 11.9 at 32, 12.4 vs 12.0 at 48
// g = 1 (a single utterance is M = 1): every weight tile has exactly one reader, the 16.8 MB matrices are streamed with
// non-temporal loads (-0.3 us each; proj and the head measured 0.7 us SLOWER with them).  Chain of 30 layers without attention at
// M = 1: 26.7 us per layer against 34.0 with the round-3 shapes, at M = 32: 30.4 against 35.4.
GemmRowsShape gemm_rows_shape(int M, int N, int K, bool ln) {
    GemmRowsShape s{1, 16, 1, false, 1};
    const int g = (M + 15) / 16;
    if (K == 4096 && !ln && g <= 2) {
        s.nw = 4;
        s.ksp = kGemmKsp;
        s.nt = true;
    } else if (ln && N % 48 == 0 && N < 4096) {
        s.ntl = g >= 3 ? 3 : g;
    } else if (ln && N % 32 == 0) {
        s.ntl = g >= 2 ? 2 : 1;
        s.mt = g >= 3 ? 2 : 1;
        s.nt = g == 1 && (long)N * K >= 4L * 1024 * 1024;
    }
    return s;
}

#define AUR_GR_NAME gemm_rows_kernel
#define AUR_GR_LAUNCH launch_gemm_rows
#define AUR_GR_NO_DATA 0
#include "gemm_rows_kernel.inc"
#undef AUR_GR_NAME
#undef AUR_GR_LAUNCH
#undef AUR_GR_NO_DATA

// ------------------------------------------------------------------------------------------------
// residual + LayerNorm rows (one wave per 1024-wide row; statistics via wavefront shuffles)
__global__ __launch_bounds__(256) void rows_ln_kernel(const float* __restrict__ P, int S,
                                                      const float* __restrict__ bias, float* __restrict__ h,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ out,
                                                      int M, float eps) {
    // one workgroup per row: 256 lanes x float4 = 1024; slab loads are independent (unrolled), statistics go
    // wave shuffle -> 4-entry LDS
    __shared__ float part[2][4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = 4 * tid;
    // gamma/beta are fetched first: behind the two barriers below they would cost one more memory round trip
    const f32x4 gam = *reinterpret_cast<const f32x4*>(gamma + n);
    const f32x4 bet = *reinterpret_cast<const f32x4*>(beta + n);
    f32x4 v = *reinterpret_cast<const f32x4*>(h + (long)row * kHidden + n);
    if (S > 0) {
        v += slab_sum(P + (long)row * kHidden + n, (long)M * kHidden, S, bias + n);
        *reinterpret_cast<f32x4*>(h + (long)row * kHidden + n) = v;
    }
    float sum = wave_sum((v[0] + v[1]) + (v[2] + v[3]));
    if (lane == 0) part[0][wv] = sum;
    __syncthreads();
    const float mean = ((part[0][0] + part[0][1]) + (part[0][2] + part[0][3])) * (1.0f / kHidden);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float d = v[c] - mean;
        sq = fmaf(d, d, sq);
    }
    sq = wave_sum(sq);
    if (lane == 0) part[1][wv] = sq;
    __syncthreads();
    const float var = ((part[1][0] + part[1][1]) + (part[1][2] + part[1][3])) * (1.0f / kHidden);
    const float rstd = 1.0f / sqrtf(var + eps);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (v[c] - mean) * rstd * gam[c] + bet[c];
    *reinterpret_cast<f32x4*>(out + (long)row * kHidden + n) = o;
}

void launch_rows_ln(const float* P, int S, const float* bias, float* h, const float* gamma, const float* beta,
                    float* out, int M, float eps, hipStream_t st) {
    trace_launch("rows_ln_kernel");
    hipLaunchKernelGGL(rows_ln_kernel, dim3(M), dim3(256), 0, st, P, S, bias, h, gamma, beta, out, M, eps);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8g __attribute__((ext_vector_type(8)));

template <bool KVH>
__global__ __launch_bounds__(256) void qkv_epilogue_kernel(const float* __restrict__ P, int S,
                                                           const float* __restrict__ bias, float* __restrict__ qbuf,
                                                           void* __restrict__ kv_layer,
                                                           const int* __restrict__ row_slot,
                                                           const int* __restrict__ row_pos,
                                                           const int* __restrict__ slot_kvpos,
                                                           const int* __restrict__ block_tables, int max_blocks,
                                                           int M) {
    const int m = blockIdx.x;
    const int slot = row_slot[m];
    const int pos = row_pos ? row_pos[m] : slot_kvpos[slot];
    const int blk = block_tables[(long)slot * max_blocks + pos / kKvBlockTokens];
    const int tok = pos % kKvBlockTokens;
    constexpr int N = 3 * kHidden;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int n = 4 * (threadIdx.x + 256 * u);
        const f32x4 t = slab_sum(P + (long)m * N + n, (long)M * N, S, bias + n);
        if (u == 0) {
            *reinterpret_cast<f32x4*>(qbuf + (long)m * kHidden + n) = t;
        } else {
            const int d = n - u * kHidden;
            const int head = d / kHeadDim, dd = d % kHeadDim;
            const long off = kv_offset(blk, u - 1, head, tok) + dd;
            if (KVH) {
                const h16x4 hv = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
                *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(kv_layer) + off) = hv;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(kv_layer) + off) = t;
            }
        }
    }
}

void launch_qkv_epilogue(const float* P, int S, const float* bias, float* qbuf, void* kv_layer,
                         const int* row_slot, const int* row_pos, const int* slot_kvpos,
                         const int* block_tables, int max_blocks, int M, hipStream_t st, bool kv_half) {
    trace_launch("qkv_epilogue_kernel");
    if (kv_half)
        hipLaunchKernelGGL(qkv_epilogue_kernel<true>, dim3(M), dim3(256), 0, st, P, S, bias, qbuf, kv_layer, row_slot, row_pos,
                           slot_kvpos, block_tables, max_blocks, M);
    else
        hipLaunchKernelGGL(qkv_epilogue_kernel<false>, dim3(M), dim3(256), 0, st, P, S, bias, qbuf, kv_layer, row_slot, row_pos,
                           slot_kvpos, block_tables, max_blocks, M);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Paged causal attention: one workgroup per (row, head).  16 lanes x float4 (fp32 pool) or 8 lanes x 8 halves (fp16 pool) span
// the 64-wide head, so one wave instruction covers 4 / 8 consecutive cached tokens (1 KiB contiguous); 4 waves stride the
// context.  Scores are reduced with wavefront shuffles; online softmax per lane group; groups merged through LDS.
//
// What stands between the launch and the first K/V request (round 5).  Until round 4: two dependent kernel-argument fetches, the
// row's position (scalar load), its block table (vector load -> vmcnt(0) -> ds_write -> s_barrier), one more argument fetch, then
// the first K/V addresses -- six scalar waits, a vector round trip and a barrier.  Now: the arguments are preloaded SGPRs (10
// dwords, -amdgpu-kernarg-preload-count), and ONE scalar round trip fetches the row's position, its write block and the block ids
// of the first 64 (128) tokens together, next to the q load.  The block ids of a token step are WAVE-UNIFORM -- a step of the
// workgroup covers 16 (32) consecutive tokens starting at a multiple of 16 -- so they are scalar loads out of the row's dense table
// (row_meta, written once per decode step by embed_decode_kernel) into SGPRs, requested one iteration ahead: no LDS copy of the
// table, no barrier in front of the loop.  A lane past the end of the context reads the row's last token (position `pos`, block
// row_meta[kRowMetaWblk]) as before.  Same K/V rows in the same order into the same arithmetic: bitwise equal to round 4.
template <bool KVH>
__global__ __launch_bounds__(256) void paged_attention_kernel(const int* __restrict__ row_meta, const float* __restrict__ qbuf,
                                                              const void* __restrict__ kv_layer_v, float* __restrict__ out, int out_mtt) {
    // (16 token steps in flight for M <= 16 rows -- one loop iteration per 256 tokens: at one row 8.2 vs 9.0 us at 244 tokens, but
    // 6.3 vs 4.9 at 64 and 12.5 vs 12.2 at 384, profiles/r04_gemm_bench_attention.log.  The launch is 3.5 us + 1.5 us per 64 tokens: one
    // CU per (row, head) pulls its K/V at 21-25 KB/us whatever the unroll.  Not kept.)
    constexpr int UN = 4;                    // token steps in flight per workgroup iteration (8: slower in round 3 -- registers -- and, re-measured on the round-5 loop at 91 VGPRs, 23.4-24.0 vs 23.4-23.6 us at 244 tokens, 4.45 vs 3.4 at one)
    constexpr int LPT = KVH ? 8 : 16;        // lanes per token
    constexpr int EPL = kHeadDim / LPT;      // elements per lane
    constexpr int TPW = 64 / LPT;            // tokens per wave instruction
    constexpr int NP = 4 * TPW;              // partial (m, l, o) groups per workgroup
    constexpr int STEP = 4 * TPW * UN;       // tokens per workgroup iteration
    constexpr int NBI = STEP / kKvBlockTokens;   // K/V blocks per workgroup iteration (4 / 8)
    using KT = typename std::conditional<KVH, _Float16, float>::type;
    __shared__ float part_o[NP][kHeadDim];
    __shared__ float part_m[NP], part_l[NP];
    const int m = blockIdx.x, head = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane / LPT, dl = lane % LPT;
    const int* rm = row_meta + (long)m * kRowMetaStride;   // uniform: everything read through it is a scalar load
    int ids[NBI];
    // (contiguous, unclamped: the row's table is padded -- kRowMetaStride -- so that the ids of the iteration behind the last one are
    // still inside the row; entries past the sequence's blocks are never used, the lanes they would serve are past `pos`)
    auto load_ids = [&](int t0) {
        const int* p = rm + kRowMetaBt + t0 / kKvBlockTokens;
#pragma unroll
        for (int i = 0; i < NBI; ++i) ids[i] = p[i];
    };
    const int pos = rm[0], wblk = rm[kRowMetaWblk];
    load_ids(0);
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

    // UN steps unrolled: all 2*UN K/V loads of a lane are issued before the first softmax update (addresses are clamped instead
    // of predicated so that the loads can be hoisted).
    // (the fp16 pool's values stay packed, 4 registers per 8 halves, until they are used: as floats they cost 104 VGPRs)
    using RawT = typename std::conditional<KVH, h16x8g, f32x4>::type;
    RawT kraw[UN], vraw[UN];
    // Addresses as 32-bit BYTE offsets from the layer's K base and V base (two SGPR pairs): the fields of
    // [block][K|V][head][token][64] do not overlap, so an offset is (block << kBlkSh) | lane part, and the lane part -- head, the
    // token slot inside its block, the lane's 16 bytes of the row -- is the same for every step and every iteration, because a step
    // starts at a multiple of 16 tokens (4 * TPW * u + t0) and a wave covers TPW consecutive ones.  Per step that leaves one scalar
    // shift, an OR, a compare and a select, against the 22 VALU instructions (64-bit multiplies-by-shift, a signed modulo) hipcc
    // made of kv_offset(): 115 -> 40 instructions in front of the eighth request of a launch whose instruction cache is cold, and a
    // shorter loop.  A layer's pool therefore has to stay below 4 GiB (kMaxKvBlocksPerLayer; the engine checks it when it sizes the pool).
    constexpr int kBlkSh = KVH ? 16 : 17;                                   // log2(bytes of one block: K and V, 16 heads x 16 tokens x 64)
    constexpr unsigned kVOff = 1u << (kBlkSh - 1);                          // bytes from a block's K half to its V half
    static_assert(sizeof(KT) * 2 * kHeads * kKvBlockTokens * kHeadDim == (1u << kBlkSh) && kKvBlockTokens == 16, "paged K/V layout");
    const char* const kbase = reinterpret_cast<const char*>(kv_layer_v);
    const char* const vbase = kbase + kVOff;
    const unsigned row_bytes = kHeadDim * sizeof(KT);                       // one token of one head
    const unsigned lane_off = (unsigned)head * (kKvBlockTokens * row_bytes) + (unsigned)dl * (EPL * (unsigned)sizeof(KT));
    // token slot of this lane inside its block, for every step: (wv * TPW + g) mod 16 (fp16 pool: TPW = 8, waves 2 and 3 start at slot 0 / 8 of the next block)
    const unsigned own_off = lane_off + (unsigned)((wv * TPW + g) & (kKvBlockTokens - 1)) * row_bytes;
    auto load_kv = [&](int t0) {
        // (a lane past the end of the context reads the row's last token: clamped to the context, not to the table -- making the
        // address independent of the row's position made the masked lanes fetch real, distinct rows: 24.2 vs 22.6 us per launch)
        const unsigned last_off = ((unsigned)wblk << kBlkSh) + lane_off + (unsigned)(pos & (kKvBlockTokens - 1)) * row_bytes;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int traw = t0 + 4 * TPW * u + wv * TPW + g;
            // block of this step inside the iteration: (4 TPW u + TPW wv) / 16 -- u for the fp32 pool, 2 u + (wv >> 1) for fp16
            const int blk_u = KVH ? ((wv >> 1) ? ids[(2 * u + 1) % NBI] : ids[(2 * u) % NBI]) : ids[u % NBI];
            const unsigned off = traw > pos ? last_off : (((unsigned)blk_u << kBlkSh) | own_off);
            kraw[u] = *reinterpret_cast<const RawT*>(kbase + off);
            vraw[u] = *reinterpret_cast<const RawT*>(vbase + off);
        }
        __builtin_amdgcn_sched_barrier(0);   // all 2 * UN requests leave before the first score (hipcc otherwise starts step 0's arithmetic, and its wait, in front of the last loads)
    };
    float mi = -INFINITY, li = 0.f;
    float o[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) o[c] = 0.f;
    int t0 = 0;
    // (n_keys >= 1, hence do-while: there is no guard for hipcc to sink the first loads under.  ONE copy of the body: with the first
    // iteration's loads issued in front of the loop -- `if (t0 > 0) load` inside -- hipcc peeled the whole iteration, and a launch
    // fetched both copies through its cold instruction cache: 6.7 KB of code, now 3.5)
    do {
        load_kv(t0);
        load_ids(t0 + STEP);   // one iteration ahead: the scalar round trip of iteration i + 1 runs under the K/V loads of iteration i
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + 4 * TPW * u + wv * TPW + g;
            const bool valid = t < n_keys;
            // (V is used under `valid` only, and hipcc SINKS a load into the one branch that uses it: until round 5 the V load of the
            // first step of every iteration sat inside the branch, followed by s_waitcnt vmcnt(0) -- one dependent memory round trip
            // and a drain of the seven other loads per 64 tokens.  Naming the register here keeps the load where it is issued.)
            asm volatile("" ::"v"(vraw[u]));
            float kx[EPL], vx[EPL];
#pragma unroll
            for (int c = 0; c < EPL; ++c) {
                kx[c] = (float)kraw[u][c];
                vx[c] = (float)vraw[u][c];
            }
            // (explicit fused multiply-adds, here and in the update below: which products of `a * b + c * d` hipcc contracts depends
            // on the code around them -- a branch-free form of this loop compiled to packed multiplies and separate adds -- and the
            // rounding of a score must not depend on the build.  This is the form every build so far compiled to.)
            float sc = fmaf(qv[0], kx[0], qv[1] * kx[1]) + fmaf(qv[2], kx[2], qv[3] * kx[3]);
            if constexpr (EPL == 8) sc += fmaf(qv[4], kx[4], qv[5] * kx[5]) + fmaf(qv[6], kx[6], qv[7] * kx[7]);
            if constexpr (LPT == 16) {
                // the butterfly sc += sc[lane ^ 8], ^ 4, ^ 2, ^ 1 over the token's 16 lanes (= one DPP row) as four v_add_f32_dpp
                // instead of four ds_bpermute round trips through the LDS crossbar: lane ^ 8 is a row rotation by 8; after it the
                // values are invariant under ^ 8, so the rotation by 4 delivers sc[lane ^ 4] (or its equal sc[lane ^ 4 ^ 8]);
                // ^ 2 and ^ 1 are quad permutations.  Same operand pairs in the same order: bitwise the shuffle version.
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
                const float alpha = expf(mi - mn);   // mi = -inf on first use -> 0
                const float p = expf(sc - mn);
                li = fmaf(li, alpha, p);
#pragma unroll
                for (int c = 0; c < EPL; ++c) o[c] = fmaf(o[c], alpha, p * vx[c]);
                mi = mn;
            }
        }
        t0 += STEP;
    } while (t0 < n_keys);
    const int pidx = wv * TPW + g;
#pragma unroll
    for (int c = 0; c < EPL; ++c) part_o[pidx][dl * EPL + c] = o[c];
    if (dl == 0) {
        part_m[pidx] = mi;
        part_l[pidx] = li;
    }
    __syncthreads();
    if (threadIdx.x < kHeadDim) {   // wave 0
        // The NP group weights exp(m_p - max) ONCE, in parallel: lane p computes weight p (one expf in the code instead of NP copies
        // -- 1.6 KB of straight-line code one wave would pull through the cold instruction cache at the very end of the launch -- or
        // NP serial trips through a rolled loop), the wave exchanges them through LDS (same wave: LDS operations execute in order,
        // no barrier), and every lane then sums its output column in the order p = 0 .. NP - 1 as before: same values, same order.
        static_assert(NP == 16 || NP == 32, "merge: one lane per partial group");
        const float pm = part_m[lane & (NP - 1)];
        float mx = pm;
#define AUR_DPP_MAX(ctrl) mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), (ctrl), 0xF, 0xF, false)))
        AUR_DPP_MAX(0x128);   // row_ror:8, 4, 2, 1: the maximum of every 16-lane row in all of its lanes
        AUR_DPP_MAX(0x124);
        AUR_DPP_MAX(0x122);
        AUR_DPP_MAX(0x121);
#undef AUR_DPP_MAX
        if constexpr (NP == 32) mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        if (lane < NP) part_m[lane] = expf(pm - mx);   // (every lane of the wave has read its part_m element above)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float w = part_m[p];
            L += part_l[p] * w;
            O += part_o[p][threadIdx.x] * w;
        }
        const int col = head * kHeadDim + threadIdx.x;
        out[out_mtt > 0 ? pk_off(m, col, out_mtt) : (long)m * kHidden + col] = O / L;
    }
}

// Prompt rows (prefill): attention on exact-f32 MFMA tiles.  A QUERY BLOCK is up to 32 consecutive prompt rows of one sequence
// (qblk[i] = {first row, rows}: built by the host, a block never crosses a sequence, positions inside it ascend by one); one wave
// takes one (query block, head), a workgroup four heads.  Both products keep the queries on the MFMA column axis, so everything
// per-query (running max, running sum, the rescaling of the output) is lane-local and P never has to be transposed:
//   S^T[key][query]  = K[key][:] . Q[query][:]     v_mfma_f32_32x32x2_f32 x 32, A = K rows (8 float4 loads per lane straight from the
//                                                   pages), B = Q rows (kept in registers for the whole kernel); dims split 0..31 /
//                                                   32..63 over the two lane halves
//   O^T[dim][query] += V^T[dim][key] . P^T[key][query]   x 16 per 32-dim tile, A from the V tile staged in the wave's 8 KB of LDS, B = the
//                                                   lane's own 16 probabilities (D layout of S^T: register r of half h is key
//                                                   (r & 3) + 8 (r >> 2) + 4 h, which is then also the k index of the second product)
// Keys go in blocks of 32 from key 0 (causal mask per element), online softmax per query.  A row's result depends on its own
// sequence only: every output element has its own accumulator chain, and the key partition starts at key 0 whatever the row block.
// (Until round 3 this was a VALU kernel, 8 lanes per row walking the context key by key: 125 us per launch at 4 544 rows.)
template <bool KVH>
__global__ __launch_bounds__(256) void prompt_attention_kernel(const float* __restrict__ qbuf, const void* __restrict__ kv_layer_v,
                                                               const int2* __restrict__ qblk, const int* __restrict__ row_slot,
                                                               const int* __restrict__ row_pos, const int* __restrict__ block_tables,
                                                               int max_blocks, float* __restrict__ out) {
    using KT = typename std::conditional<KVH, _Float16, float>::type;
    constexpr int VP = kHeadDim + 4;   // LDS row pitch of the V tile (floats): b128 writes and b32 reads conflict-free
    __shared__ __attribute__((aligned(16))) float vs[4][32 * VP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int head = blockIdx.y * 4 + wv;
    const int2 qb = qblk[blockIdx.x];
    const int row0 = qb.x, nrows = qb.y;
    const int slot = row_slot[row0], pos0 = row_pos[row0];
    const int* bt = block_tables + (long)slot * max_blocks;
    const KT* kv_layer = reinterpret_cast<const KT*>(kv_layer_v);
    const int my_pos = pos0 + l31;                 // position of this lane's query (queries >= nrows compute and are dropped)
    const int max_pos = pos0 + nrows - 1;
    const int n_kb = (max_pos >> 5) + 1;

    auto load32 = [&](const KT* p, float (&dst)[32]) {   // 32 consecutive elements
        if constexpr (KVH) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h16x8g v = *reinterpret_cast<const h16x8g*>(p + 8 * i);
#pragma unroll
                for (int c = 0; c < 8; ++c) dst[8 * i + c] = (float)v[c];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
                for (int c = 0; c < 4; ++c) dst[4 * i + c] = v[c];
            }
        }
    };
    float qreg[32];
    {
        const float* qp = qbuf + (long)(row0 + min(l31, nrows - 1)) * kHidden + head * kHeadDim + 32 * hi;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * i);
#pragma unroll
            for (int c = 0; c < 4; ++c) qreg[4 * i + c] = v[c];
        }
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float* vw = &vs[wv][0];

    for (int kb = 0; kb < n_kb; ++kb) {
        // K rows of this block as the A operand (keys beyond the block's last needed key are clamped: their scores are masked)
        float kreg[32];
        {
            const int key = min(kb * 32 + l31, max_pos);
            load32(kv_layer + kv_offset(bt[key / kKvBlockTokens], 0, head, key % kKvBlockTokens) + 32 * hi, kreg);
        }
        // V rows of this block: lane -> (key lane >> 1, half lane & 1), 32 elements each
        float vreg[32];
        {
            const int key = min(kb * 32 + (lane >> 1), max_pos);
            load32(kv_layer + kv_offset(bt[key / kKvBlockTokens], 1, head, key % kKvBlockTokens) + 32 * (lane & 1), vreg);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s], qreg[s], acc, 0, 0, 0);
        // scores of this lane's query against keys kb*32 + keyl(r), keyl(r) = (r & 3) + 8 (r >> 2) + 4 hi
        float mb = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            acc[r] = key <= my_pos ? acc[r] * 0.125f : -INFINITY;   // 1/sqrt(64)
            mb = fmaxf(mb, acc[r]);
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_run, mb);          // finite from the first block on: key 0 is visible to every query
        const float alpha = expf(m_run - m_new);        // m_run = -inf on the first block -> 0
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = expf(acc[r] - m_new);              // masked: exp(-inf) = 0
            ps += acc[r];
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // stage the V tile (the previous block's reads are complete: same wave, LDS executes in order; the wave barriers keep the
        // compiler from moving accesses of different lanes across the phases)
        __builtin_amdgcn_wave_barrier();
        {
            float* dst = vw + (lane >> 1) * VP + 32 * (lane & 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dst + 4 * i) = f32x4{vreg[4 * i], vreg[4 * i + 1], vreg[4 * i + 2], vreg[4 * i + 3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* vrow = vw + ((s & 3) + 8 * (s >> 2) + 4 * hi) * VP + l31;
#pragma unroll
            for (int t = 0; t < 2; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * t], acc[s], o[t], 0, 0, 0);
        }
    }
    if (l31 < nrows) {
        const float inv = 1.0f / l_run;
        float* op = out + (long)(row0 + l31) * kHidden + head * kHeadDim + 4 * hi;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<f32x4*>(op + 32 * t + 8 * g4) =
                    f32x4{o[t][4 * g4] * inv, o[t][4 * g4 + 1] * inv, o[t][4 * g4 + 2] * inv, o[t][4 * g4 + 3] * inv};
    }
}

void launch_prompt_attention(const float* qbuf, const void* kv_layer, const int2* qblk, int n_qblk, const int* row_slot, const int* row_pos,
                             const int* block_tables, int max_blocks, float* out, hipStream_t st, bool kv_half) {
    trace_launch("prompt_attention_kernel");
    static_assert(kHeads % 4 == 0 && kHeadDim == 64, "prompt attention: four heads per workgroup, 64-wide heads");
    const dim3 grid(n_qblk, kHeads / 4);
    if (kv_half)
        hipLaunchKernelGGL(prompt_attention_kernel<true>, grid, dim3(256), 0, st, qbuf, kv_layer, qblk, row_slot, row_pos, block_tables, max_blocks, out);
    else
        hipLaunchKernelGGL(prompt_attention_kernel<false>, grid, dim3(256), 0, st, qbuf, kv_layer, qblk, row_slot, row_pos, block_tables, max_blocks, out);
    HIP_CHECK(hipGetLastError());
}

void launch_paged_attention(const float* qbuf, const void* kv_layer, const int* row_meta, int max_blocks, float* out, int M,
                            hipStream_t st, int out_mtt, bool kv_half) {
    trace_launch("paged_attention_kernel");
    // the kernel reads block ids one iteration (8 blocks with the fp16 pool) past the last block a sequence can own
    AUR_REQUIRE(row_meta && max_blocks >= 1 && kRowMetaBt + (max_blocks + 7) / 8 * 8 + 8 <= kRowMetaStride,
                "paged attention: the step's row_meta table (embed_decode_kernel), padded for the kernel's look-ahead");
    if (kv_half) hipLaunchKernelGGL(paged_attention_kernel<true>, dim3(M, kHeads), dim3(256), 0, st, row_meta, qbuf, kv_layer, out, out_mtt);
    else hipLaunchKernelGGL(paged_attention_kernel<false>, dim3(M, kHeads), dim3(256), 0, st, row_meta, qbuf, kv_layer, out, out_mtt);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_prompt_kernel(const int4* __restrict__ desc,
                                                           const float* __restrict__ spk_cond,
                                                           const float* __restrict__ text_emb,
                                                           const float* __restrict__ text_pos,
                                                           const float* __restrict__ wte,
                                                           const float* __restrict__ wpe, float* __restrict__ h) {
    const int m = blockIdx.x;
    const int4 d = desc[m];
    const int n = 4 * threadIdx.x;
    // NOTE: written with pointer selects instead of a 3-way if/else: hipcc (ROCm 7.2) left the store address
    // register undefined on the third arm of the branchy form (seen in the .s; aperture violation on gfx950).
    const float* pa = (d.x == 0) ? spk_cond + ((long)d.z * 32 + d.y) * kHidden
                                 : ((d.x == 1) ? text_emb + (long)d.y * kHidden : wte + (long)d.y * kHidden);
    const float* pb = (d.x == 1) ? text_pos + (long)d.z * kHidden : wpe + (long)d.z * kHidden;
    f32x4 v = *reinterpret_cast<const f32x4*>(pa + n);
    const f32x4 w = *reinterpret_cast<const f32x4*>(pb + n);
    if (d.x != 0) v += w;
    *reinterpret_cast<f32x4*>(h + (long)m * kHidden + n) = v;
}

void launch_embed_prompt(const int4* desc, const float* spk_cond, const float* text_emb, const float* text_pos,
                         const float* wte, const float* wpe, float* h, int M, hipStream_t st) {
    trace_launch("embed_prompt_kernel");
    hipLaunchKernelGGL(embed_prompt_kernel, dim3(M), dim3(256), 0, st, desc, spk_cond, text_emb, text_pos, wte, wpe, h);
    HIP_CHECK(hipGetLastError());
}

// What the launch waits for (round 5): the row's slot, then the slot's token / position / K/V position TOGETHER, then the two embedding
// rows and the row's block table together -- three dependent memory trips, the minimum for row -> slot -> token -> embedding.  hipcc had
// made seven of them: every kernel argument behind the preloaded ones and every per-slot word was fetched where it was first used.
// The leading arguments (everything the first two trips need) arrive in SGPRs; the rest is one scalar burst next to the first trip.
struct EmbedDecodeTail {
    const float* wpe;
    float* h;
    float2* stats;
    int* row_meta;
};
__global__ __launch_bounds__(256) void embed_decode_kernel(const int* __restrict__ row_slot, const int* __restrict__ slot_tok,
                                                           const int* __restrict__ slot_pos, const int* __restrict__ slot_kvpos,
                                                           const int* __restrict__ block_tables, int max_blocks, int h_mtt,
                                                           const float* __restrict__ wte, EmbedDecodeTail tl) {
    const int m = blockIdx.x;
    const int slot = row_slot[m];
    const float* const wpe = tl.wpe;
    float* const h = tl.h;
    float2* const stats = tl.stats;
    int* const row_meta = tl.row_meta;
    asm volatile("; EmbedDecodeTail resident" ::"s"(wpe), "s"(h), "s"(stats), "s"(row_meta));
    const int tok = slot_tok[slot], pe = slot_pos[slot];
    const int kvpos = row_meta ? slot_kvpos[slot] : 0;
    asm volatile("; per-slot words resident" ::"s"(tok), "s"(pe), "s"(kvpos));
    const int n = 4 * threadIdx.x;
    const int* const bt = block_tables + (long)slot * max_blocks;
    const int wblk = row_meta ? bt[kvpos / kKvBlockTokens] : 0;   // (uniform: a scalar load, requested with the rows below)
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(wte + (long)tok * kHidden + n);
    const f32x4 e1 = *reinterpret_cast<const f32x4*>(wpe + (long)pe * kHidden + n);
    const int btv = row_meta ? bt[min((int)threadIdx.x, max_blocks - 1)] : 0;
    asm volatile("; write block resident" ::"s"(wblk), "v"(btv));
    if (row_meta) {   // the step's K/V addressing of this row, dense (kRowMetaStride ints)
        int* rm = row_meta + (long)m * kRowMetaStride;
        if ((int)threadIdx.x < max_blocks) rm[kRowMetaBt + threadIdx.x] = btv;
        if (threadIdx.x == 0) {
            rm[0] = kvpos;
            rm[1] = slot;
            rm[kRowMetaWblk] = wblk;
        }
    }
    const f32x4 v = e0 + e1;
    *reinterpret_cast<f32x4*>(h + (h_mtt > 0 ? pk_off(m, n, h_mtt) : (long)m * kHidden + n)) = v;
    if (stats) {   // LayerNorm partials per 16-column tile (4 adjacent lanes), same definition as the kEpiResidual epilogue
        // (lane ^ 2, lane ^ 1 as DPP quad permutations: the shuffles' operand pairs without their ds_bpermute round trips)
#define AUR_QUAD_ADD(x, ctrl) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (ctrl), 0xF, 0xF, false))
        float sm = (v[0] + v[1]) + (v[2] + v[3]);
        AUR_QUAD_ADD(sm, 0x4E);   // quad_perm [2,3,0,1]
        AUR_QUAD_ADD(sm, 0xB1);   // quad_perm [1,0,3,2]
        const float mu = sm * (1.0f / 16.0f);
        float m2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d = v[c] - mu;
            m2 = fmaf(d, d, m2);
        }
        AUR_QUAD_ADD(m2, 0x4E);
        AUR_QUAD_ADD(m2, 0xB1);
#undef AUR_QUAD_ADD
        if ((threadIdx.x & 3) == 0) stats[(long)m * 64 + (threadIdx.x >> 2)] = make_float2(mu, m2);
    }
}

void launch_embed_decode(const int* row_slot, const int* slot_tok, const int* slot_pos, const float* wte,
                         const float* wpe, float* h, int M, hipStream_t st, int h_mtt, float2* stats, int* row_meta,
                         const int* slot_kvpos, const int* block_tables, int max_blocks) {
    AUR_REQUIRE(!row_meta || (slot_kvpos && block_tables && max_blocks + kRowMetaBt <= kRowMetaStride), "embed_decode: row meta");
    trace_launch("embed_decode_kernel");
    hipLaunchKernelGGL(embed_decode_kernel, dim3(M), dim3(256), 0, st, row_slot, slot_tok, slot_pos, slot_kvpos, block_tables, max_blocks, h_mtt,
                       wte, EmbedDecodeTail{wpe, h, stats, row_meta});
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ln_wave(f32x4 (&v)[4], const float* gamma, const float* beta, int lane, float eps) {
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
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + n);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + n);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = (v[u][c] - mean) * rstd * g[c] + b[c];
    }
}

__global__ __launch_bounds__(256) void final_norm_kernel(const float* __restrict__ xn,
                                                         const int* __restrict__ sample_row,
                                                         const int* __restrict__ sample_slot,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ ybuf,
                                                         float* __restrict__ latents, long lat_slot_stride,
                                                         const int* __restrict__ slot_ngen, int max_lat_rows,
                                                         int Ms, float eps) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= Ms) return;
    const int row = sample_row[j];
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(xn + (long)row * kHidden + 4 * (lane + 64 * u));
    ln_wave(v, gamma, beta, lane, eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(ybuf + (long)j * kHidden + 4 * (lane + 64 * u)) = v[u];
    const int slot = sample_slot[j];
    const int idx = slot_ngen[slot];
    if (latents && idx < max_lat_rows) {
        ln_wave(v, gamma, beta, lane, eps);
        float* dst = latents + (long)slot * lat_slot_stride + (long)idx * kHidden;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dst + 4 * (lane + 64 * u)) = v[u];
    }
}

// dst[j] = final_norm(final_norm(src[j]))  for n contiguous rows (literal second pass, XTTSv2.py:685-687 on top of
// vllm_mm_gpt.py:671)
__global__ __launch_bounds__(256) void double_norm_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)j * kHidden + 4 * (lane + 64 * u));
    ln_wave(v, gamma, beta, lane, eps);
    ln_wave(v, gamma, beta, lane, eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dst + (long)j * kHidden + 4 * (lane + 64 * u)) = v[u];
}

void launch_double_norm_rows(const float* src, float* dst, int n, const float* gamma, const float* beta, float eps,
                             hipStream_t st) {
    trace_launch("double_norm_rows_kernel");
    hipLaunchKernelGGL(double_norm_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, src, dst, n, gamma, beta, eps);
    HIP_CHECK(hipGetLastError());
}

void launch_final_norm(const float* xn, const int* sample_row, const int* sample_slot, const float* gamma,
                       const float* beta, float* ybuf, float* latents, long lat_slot_stride,
                       const int* slot_ngen, int max_lat_rows, int Ms, float eps, hipStream_t st) {
    trace_launch("final_norm_kernel");
    hipLaunchKernelGGL(final_norm_kernel, dim3((Ms + 3) / 4), dim3(256), 0, st, xn, sample_row, sample_slot, gamma, beta,
                       ybuf, latents, lat_slot_stride, slot_ngen, max_lat_rows, Ms, eps);
    HIP_CHECK(hipGetLastError());
}

// decode tail of the gemm_rows chain: the last block's residual add leaves h; ln_f and both final_norms run here
__global__ __launch_bounds__(256) void final_rows_kernel(const float* __restrict__ h, int mtt, const int* __restrict__ sample_slot,
                                                         const float* __restrict__ lnf_w, const float* __restrict__ lnf_b,
                                                         const float* __restrict__ fn_w, const float* __restrict__ fn_b,
                                                         float* __restrict__ ybuf, float* __restrict__ latents,
                                                         long lat_slot_stride, const int* __restrict__ slot_ngen,
                                                         int max_lat_rows, int Ms, float eps) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= Ms) return;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(h + pk_off(j, 4 * (lane + 64 * u), mtt));
    ln_wave(v, lnf_w, lnf_b, lane, eps);
    ln_wave(v, fn_w, fn_b, lane, eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(ybuf + pk_off(j, 4 * (lane + 64 * u), mtt)) = v[u];
    const int slot = sample_slot[j];
    const int idx = slot_ngen[slot];
    if (latents && idx < max_lat_rows) {
        ln_wave(v, fn_w, fn_b, lane, eps);
        float* dst = latents + (long)slot * lat_slot_stride + (long)idx * kHidden;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dst + 4 * (lane + 64 * u)) = v[u];
    }
}

void launch_final_rows(const float* h, int mtt, const int* sample_slot, const float* lnf_w, const float* lnf_b, const float* fn_w,
                       const float* fn_b, float* ybuf, float* latents, long lat_slot_stride, const int* slot_ngen,
                       int max_lat_rows, int Ms, float eps, hipStream_t st) {
    trace_launch("final_rows_kernel");
    hipLaunchKernelGGL(final_rows_kernel, dim3((Ms + 3) / 4), dim3(256), 0, st, h, mtt, sample_slot, lnf_w, lnf_b, fn_w, fn_b, ybuf,
                       latents, lat_slot_stride, slot_ngen, max_lat_rows, Ms, eps);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Fused sampler: slab-sum + bias -> repetition penalty -> greedy argmax | (/T -> top-k -> top-p -> softmax ->
// exponential-race argmax), then per-slot state update.  V = 1026 fits one workgroup.
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float exp_noise(unsigned seed, unsigned step, unsigned v) {
    const unsigned a = lowbias32(v * 0x9E3779B1u + seed);
    const unsigned b = lowbias32(a ^ (step * 0x85EBCA77u + 0x165667B1u));
    const float u = ((float)(b >> 8) + 1.0f) * (1.0f / 16777216.0f);
    return fmaxf(-logf(u), 2.98023224e-08f);   // 2^-25: u == 1 would give -0.0, and p / -0.0 is NaN for a masked-out id (p = 0)
}
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

// block-wide argmax with "first index wins" tie rule (a total order on (value, index): any reduction tree gives the same
// winner, so the waves reduce with shuffles and meet once in LDS instead of eight barrier-separated LDS levels)
__device__ __forceinline__ int block_argmax(float val, int idx, float* sv, int* si) {
    const int tid = threadIdx.x;
    // (the butterfly over lane ^ 32, .., ^ 1 on lane_xor: DPP / lane swaps instead of twelve ds_bpermute round trips)
#define AUR_ARGMAX_STEP(J)                                   \
    {                                                        \
        const float ov = lane_xor<J>(val);                   \
        const int oi = lane_xor<J>(idx);                     \
        if (ov > val || (ov == val && oi < idx)) {           \
            val = ov;                                        \
            idx = oi;                                        \
        }                                                    \
    }
    AUR_ARGMAX_STEP(32) AUR_ARGMAX_STEP(16) AUR_ARGMAX_STEP(8) AUR_ARGMAX_STEP(4) AUR_ARGMAX_STEP(2) AUR_ARGMAX_STEP(1)
#undef AUR_ARGMAX_STEP
    if ((tid & 63) == 0) {
        sv[tid >> 6] = val;
        si[tid >> 6] = idx;
    }
    __syncthreads();
    float bv = sv[0];
    int bi = si[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) {
            bv = sv[w];
            bi = si[w];
        }
    __syncthreads();
    return bi;
}

// sum of one value per thread in the order of the binary tree sv[t] += sv[t + s], s = 128, 64, .., 1 (the order this kernel has
// always used: the softmax denominator's bits feed the sampling race): the two cross-wave levels go through LDS, the six levels
// inside wave 0 are shuffles over the same pairs (t, t + s)
__device__ __forceinline__ float block_sum_tree(float v, float* sv) {
    const int tid = threadIdx.x;
    sv[tid] = v;
    __syncthreads();
    if (tid < 128) sv[tid] += sv[tid + 128];
    __syncthreads();
    float x = 0.f;
    if (tid < 64) {
        x = sv[tid] + sv[tid + 64];
        // lane t: x[t] + x[t + sh], sh = 32 .. 1; lane 0 ends with the tree's root.  On the lanes that feed lane 0 (t < sh at every level)
        // t + sh == t ^ sh, so wave_sum's butterfly (lane swaps + DPP, common.h) forms the same sums in the same order there.
        x = wave_sum(x);
        if (tid == 0) sv[0] = x;
    }
    __syncthreads();
    const float r = sv[0];
    __syncthreads();
    return r;
}

// (the arguments the kernel needs first -- the row's slot, its logits -- are leading scalars: preloaded SGPRs, item 7 of DESIGN section 3)
__global__ __launch_bounds__(256) void sampler_kernel(const int* __restrict__ sample_slot_, const float* __restrict__ P_, const float* __restrict__ bias_,
                                                      int Npad_, int V_, int S_, int Ms_, SamplerArgs a) {
    __shared__ float z[1040];
    __shared__ unsigned long long keys[2048];
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ float sh_f[4];
    __shared__ int sh_i[2];
    const int j = blockIdx.x;
    const int tid = threadIdx.x;
    const int slot = sample_slot_[j];
    const int V = V_;
    // every per-slot parameter in ONE burst of scalar loads (hipcc sinks each to its first use otherwise: a dependent round trip in
    // front of the penalty loop, another in front of the temperature division, the top-k search, the top-p scan, the noise ...)
    const int finished = a.slot_finished[slot];
    const float pen = a.rep_penalty[slot];
    const float T = a.temperature[slot];
    const int topk = a.top_k[slot];
    const float topp = a.top_p[slot];
    const unsigned seed = a.seed[slot];
    const int ngen0 = a.slot_ngen[slot];
    const int max_tok = a.max_tokens[slot];
    const int ign_stop = a.ignore_stop[slot];
    asm volatile("; sampler: slot parameters resident" ::"s"(finished), "s"(pen), "s"(T), "s"(topk), "s"(topp), "s"(seed), "s"(ngen0), "s"(max_tok),
                 "s"(ign_stop));
    if (finished) {   // ghost row (engine.hip, pipelined decode): leave every piece of slot state untouched
        if (tid == 0) a.out_tok[j] = -1;
        return;
    }
    const unsigned char* seen = a.seen + (long)slot * kSeenStride;

    // the row's logits, bias and penalty flags: every load of the thread (V <= 1040 = 5 x 256: up to five ids per thread) issued
    // before the first use.  As a loop with the loads next to their uses hipcc emitted load -> wait -> load -> wait ...: fifteen
    // dependent memory round trips per row, roughly half of the kernel's 27 us (round 4).
    {
        float pv[5], bv[5];
        unsigned char sn[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int v = min(tid + 256 * u, V - 1);
            pv[u] = P_[(long)j * Npad_ + v];
            bv[u] = bias_[v];
            sn[u] = seen[v];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int v = tid + 256 * u;
            if (v < V) {
                float s = pv[u];
                for (int sl = 1; sl < S_; ++sl) s += P_[((long)sl * Ms_ + j) * Npad_ + v];   // (prefill-time callers; S == 1 on the decode path)
                s += bv[u];
                if (pen != 1.0f && sn[u]) s = (s > 0.f) ? s / pen : s * pen;
                z[v] = s;
                if (a.dbg_logits) a.dbg_logits[(long)j * V + v] = s;
            }
        }
    }
    __syncthreads();

    int tok;
    if (T < 1e-5f) {
        float bv = -INFINITY;
        int bi = 0;   // NaN/-inf rows fall back to id 0 instead of indexing out of range
        for (int v = tid; v < V; v += 256)
            if (z[v] > bv) {
                bv = z[v];
                bi = v;
            }
        tok = block_argmax(bv, bi, sv, si);
    } else {
        for (int v = tid; v < V; v += 256) z[v] = z[v] / T;
        __syncthreads();
        // ---- top-k fast path (0 < k <= 64, the XTTS default is 50): the k-th largest value is found by a 32-step
        // bisection on the order-preserving integer image of the logits (one ballot/popcount per element, one barrier
        // per step; a 16-way search with 8 barriers measured slower, 29.7 vs 25.0 us: 15 x 5 ballots per step), the >= threshold survivors (k plus ties) are compacted and sorted by ONE wave, no workgroup
        // barriers.  Same threshold, same survivor set and same (value, id) order as the full sort below, so both paths
        // give identical tokens; it replaces 66 barrier-separated passes over 2048 keys.
        bool sorted = false;
        if (topk > 0 && topk <= 64 && topk < V && !a.force_full_sort) {
            // every wave runs the whole search on its own copy of the row (17 values per lane): a step is 17 compares + ballots and
            // scalar adds, no LDS and no barrier (with the row split over the four waves it was 32 barriers, ~11 of the kernel's 25 us)
            unsigned wvv[17];
#pragma unroll
            for (int u = 0; u < 17; ++u) {
                const int v = (tid & 63) + 64 * u;
                wvv[u] = (v < V) ? f2ord(z[v] + 0.0f) : 0u;   // (-0 -> +0: equal floats, equal images); 0 sorts below every float
            }
            // (the search may stop at the first x with EXACTLY k values >= x: the survivor set {v : image(v) >= x} is then the top k, and
            // everything below uses the threshold only as that mask -- the same survivors, keys and order as the full search, which
            // goes on to the k-th value's own image; with ties at the k-th value no x has exactly k and all 32 steps run)
            unsigned lo = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned x = lo | (1u << bit);
                int c = 0;
#pragma unroll
                for (int u = 0; u < 17; ++u) c += __popcll(__ballot(wvv[u] >= x));
                if (c >= topk) lo = x;
                // (not below the image of the smallest finite float: with fewer than k finite logits the search must run on to the k-th
                // value's own image, -inf, or the survivor set would differ from the full sort's "k plus ties")
                if (c == topk && x >= 0x00800000u) break;
            }
            if (lo < 0x007FFFFFu) lo = 0x007FFFFFu;   // image of -inf: below it lie only NaN patterns, and `z < NaN` would mask nothing
            unsigned ov[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int v = tid + 256 * u;
                ov[u] = (v < V) ? f2ord(z[v] + 0.0f) : 0u;
            }
            // lo = image of the k-th largest logit; survivors: everything >= lo
            if (tid == 0) sh_i[1] = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int v = tid + 256 * u;
                if (v < V && ov[u] >= lo) {
                    const int pos = atomicAdd(&sh_i[1], 1);
                    if (pos < 128) keys[pos] = ((unsigned long long)ov[u] << 32) | (unsigned)v;
                }
            }
            __syncthreads();
            const int n_surv = sh_i[1];
            if (n_surv <= 64) {
                // k (+ ties) <= 64 survivors: one key per lane, bitonic network on shuffles -- 21 compare-exchange steps in registers
                // instead of 28 LDS passes over 128 keys (~2.7 us of the kernel).  Keys are distinct (the id is part of the key), so
                // every correct descending sort gives the same order.
                if (tid < 64) {
                    unsigned long long key = tid < n_surv ? keys[tid] : 0ull;
                    // (partner keys through lane_xor -- DPP / lane swaps -- instead of 42 ds_bpermute round trips)
#define AUR_SORT_STEP(K, JJ)                                                                                                   \
    {                                                                                                                          \
        const unsigned lo32 = (unsigned)lane_xor<JJ>((int)(unsigned)key);                                                      \
        const unsigned hi32 = (unsigned)lane_xor<JJ>((int)(unsigned)(key >> 32));                                              \
        const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;                                              \
        const bool lower = (tid & JJ) == 0; /* this lane is element e < x = e ^ jj of the pair */                             \
        const bool up = (tid & K) == 0;     /* descending block (as the LDS network below: larger key first) */                \
        const bool take_max = lower == up;                                                                                     \
        key = take_max ? (key > other ? key : other) : (key < other ? key : other);                                            \
    }
                    AUR_SORT_STEP(2, 1)
                    AUR_SORT_STEP(4, 2) AUR_SORT_STEP(4, 1)
                    AUR_SORT_STEP(8, 4) AUR_SORT_STEP(8, 2) AUR_SORT_STEP(8, 1)
                    AUR_SORT_STEP(16, 8) AUR_SORT_STEP(16, 4) AUR_SORT_STEP(16, 2) AUR_SORT_STEP(16, 1)
                    AUR_SORT_STEP(32, 16) AUR_SORT_STEP(32, 8) AUR_SORT_STEP(32, 4) AUR_SORT_STEP(32, 2) AUR_SORT_STEP(32, 1)
                    AUR_SORT_STEP(64, 32) AUR_SORT_STEP(64, 16) AUR_SORT_STEP(64, 8) AUR_SORT_STEP(64, 4) AUR_SORT_STEP(64, 2) AUR_SORT_STEP(64, 1)
#undef AUR_SORT_STEP
                    keys[tid] = key;
                    keys[tid + 64] = 0ull;
                    if (tid == 0) {
                        sh_f[0] = ord2f(lo);
                        sh_i[0] = n_surv;
                    }
                }
                sorted = true;
            } else if (n_surv <= 128) {
                if (tid < 64) {   // one wave: LDS accesses of a single wave execute in order, no barrier needed
                    for (int e = tid; e < 128; e += 64)
                        if (e >= n_surv) keys[e] = 0ull;
                    for (int k = 2; k <= 128; k <<= 1)
                        for (int jj = k >> 1; jj > 0; jj >>= 1) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int e = tid + 64 * h;
                                const int x = e ^ jj;
                                if (x > e) {
                                    const bool up = (e & k) == 0;
                                    const unsigned long long ka = keys[e], kb = keys[x];
                                    if ((ka < kb) == up) {
                                        keys[e] = kb;
                                        keys[x] = ka;
                                    }
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    if (tid == 0) {
                        sh_f[0] = ord2f(lo);
                        sh_i[0] = n_surv;
                    }
                }
                sorted = true;
            }
            __syncthreads();
        }
        if (!sorted) {
            for (int e = tid; e < 2048; e += 256)
                keys[e] = (e < V) ? (((unsigned long long)f2ord(z[e]) << 32) | (unsigned)e) : 0ull;
            __syncthreads();
            for (int k = 2; k <= 2048; k <<= 1)
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    for (int e = tid; e < 2048; e += 256) {
                        const int x = e ^ jj;
                        if (x > e) {
                            const bool up = (e & k) == 0;
                            const unsigned long long ka = keys[e], kb = keys[x];
                            if ((ka < kb) == up) {
                                keys[e] = kb;
                                keys[x] = ka;
                            }
                        }
                    }
                    __syncthreads();
                }
            // top-k threshold (ties at the k-th value are kept)
            if (tid == 0) {
                int n1 = V;
                float thr = -INFINITY;
                if (topk > 0 && topk < V) {
                    thr = ord2f((unsigned)(keys[topk - 1] >> 32));
                    n1 = topk;
                    while (n1 < V && ord2f((unsigned)(keys[n1] >> 32)) == thr) ++n1;
                }
                sh_f[0] = thr;
                sh_i[0] = n1;
            }
            __syncthreads();
        }
        const float thr = sh_f[0];
        const int n1 = sh_i[0];
        for (int v = tid; v < V; v += 256)
            if (z[v] < thr) z[v] = -INFINITY;
        __syncthreads();
        const float maxv = ord2f((unsigned)(keys[0] >> 32));
        // top-p on the ascending-sorted softmax: mask cumsum <= 1-p, never the largest.  The two sums are serial BY DEFINITION (their
        // bits decide the cut: sum in rank order, cumulative sum in double from the smallest probability up), but their terms are
        // not: for n1 <= 256 candidates (always on the top-k path) every thread computes its rank's exp and probability, and only the
        // additions are serial -- until round 5 one thread evaluated 2 n1 expf and n1 divisions one after the other (~7 us of the
        // kernel's 27).  Same operands into the same additions in the same order: the same bits.
        if (topp < 1.0f) {
            if (n1 <= 64) {
                // the usual case (k = 50 plus ties): one wave does all of it in registers, no LDS and no barrier -- the serial float
                // sum runs over v_readlane values in rank order (ranks >= n1 hold +0: adding them changes no bit)
                if (tid < 64) {
                    const float ev = tid < n1 ? expf(ord2f((unsigned)(keys[min(tid, n1 - 1)] >> 32)) - maxv) : 0.f;
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 64; ++r) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), r));
                    const float pr = ev / sum;
                    // c_r = pr[n1-1] + ... + pr[r], accumulated in double in exactly that order -- the reference's cumsum over the
                    // ascending-sorted probabilities (torch accumulates a float cumsum in double on the CPU) -- so the cut is the serial
                    // form's bit for bit, by construction: 63 dependent adds on readlane values (~0.5 us), every lane keeps its rank's sum
                    double c = 0.0, mine = 0.0;
#pragma unroll 7
                    for (int r = 63; r >= 1; --r) {
                        c += (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr), r));
                        if (tid == r) mine = c;
                    }
                    if (tid >= 1 && tid < n1 && (float)mine <= 1.0f - topp) z[(unsigned)(keys[tid] & 0xffffffffull)] = -INFINITY;
                }
            } else if (n1 <= 256) {
                const float ev = tid < n1 ? expf(ord2f((unsigned)(keys[min(tid, n1 - 1)] >> 32)) - maxv) : 0.f;
                sv[tid] = ev;
                __syncthreads();
                if (tid == 0) {
                    float sum = 0.f;
                    for (int r = 0; r < n1; ++r) sum += sv[r];
                    sh_f[1] = sum;
                }
                __syncthreads();
                const float pr = ev / sh_f[1];
                __syncthreads();
                sv[tid] = pr;
                __syncthreads();
                // (65..256 candidates only when dozens of logits tie at the k-th value: the serial form itself, on the probabilities the
                // threads computed)
                if (tid == 0) {
                    const float lim = 1.0f - topp;
                    double c = 0.0;
                    for (int r = n1 - 1; r >= 1; --r) {
                        c += (double)sv[r];
                        if ((float)c <= lim)
                            z[(unsigned)(keys[r] & 0xffffffffull)] = -INFINITY;
                        else
                            break;
                    }
                }
            } else if (tid == 0) {
                float sum = 0.f;
                for (int r = 0; r < n1; ++r) sum += expf(ord2f((unsigned)(keys[r] >> 32)) - maxv);
                const float lim = 1.0f - topp;
                double c = 0.0;
                for (int r = n1 - 1; r >= 1; --r) {
                    const float pr = expf(ord2f((unsigned)(keys[r] >> 32)) - maxv) / sum;
                    c += (double)pr;
                    if ((float)c <= lim)
                        z[(unsigned)(keys[r] & 0xffffffffull)] = -INFINITY;
                    else
                        break;
                }
            }
        }
        __syncthreads();
        // softmax over the survivors, then argmax(probs / Exp(1))
        float part = 0.f;
        for (int v = tid; v < V; v += 256) part += expf(z[v] - maxv);
        const float sum2 = block_sum_tree(part, sv);
        const unsigned step = (unsigned)ngen0;
        float bv = -INFINITY;
        int bi = 0;   // NaN/-inf rows fall back to id 0 instead of indexing out of range
        for (int v = tid; v < V; v += 256) {
            const float p = expf(z[v] - maxv) / sum2;
            const float qv = p / exp_noise(seed, step, (unsigned)v);
            if (qv > bv) {
                bv = qv;
                bi = v;
            }
        }
        tok = block_argmax(bv, bi, sv, si);
    }

    if (tid == 0) {
        a.seen[(long)slot * kSeenStride + tok] = 1;
        const int ng = ngen0 + 1;
        a.slot_ngen[slot] = ng;
        a.slot_tok[slot] = tok;
        a.slot_pos[slot] = ng;   // k-th generated token enters at mel position k (vllm_mm_gpt.py:480)
        a.slot_kvpos[slot] = a.next_kvpos ? a.next_kvpos[j] : a.slot_kvpos[slot] + 1;
        const bool stop = (tok == a.stop_token) && !ign_stop;
        const bool fin = stop || ng >= max_tok;
        if (fin) a.slot_finished[slot] = 1;
        a.out_tok[j] = tok | (fin ? kTokFinishedBit : 0);   // one read-back word per row: the token and "this was the last one"
    }
}

// test support: number of 32-bit words in which two buffers differ (bitwise), accumulated into *cnt
__global__ __launch_bounds__(256) void count_mismatch_kernel(const unsigned* __restrict__ x, const unsigned* __restrict__ y, long n,
                                                             unsigned long long* __restrict__ cnt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool bad = i < n && x[i] != y[i];
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(cnt, (unsigned long long)__popcll(m));
}
void launch_count_mismatch(const void* x, const void* y, long n_words, unsigned long long* cnt, hipStream_t st) {
    hipLaunchKernelGGL(count_mismatch_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const unsigned*>(x),
                       reinterpret_cast<const unsigned*>(y), n_words, cnt);
    HIP_CHECK(hipGetLastError());
}

// test support: lane_xor<J> / wave_sum / wave_max (common.h: DPP + gfx950 lane swaps) against the shuffles they replace, on four
// waves of pseudo-random words; *cnt += number of (lane, J) results that differ bitwise
__global__ __launch_bounds__(256) void lane_xor_selftest_kernel(unsigned seed, unsigned long long* __restrict__ cnt) {
    unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    const int xi = (int)x;
    int bad = 0;
    bad += lane_xor<1>(xi) != __shfl_xor(xi, 1, 64);
    bad += lane_xor<2>(xi) != __shfl_xor(xi, 2, 64);
    bad += lane_xor<4>(xi) != __shfl_xor(xi, 4, 64);
    bad += lane_xor<8>(xi) != __shfl_xor(xi, 8, 64);
    bad += lane_xor<16>(xi) != __shfl_xor(xi, 16, 64);
    bad += lane_xor<32>(xi) != __shfl_xor(xi, 32, 64);
    const float f = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f;   // sums whose bits depend on the order of the additions
    float rs = f, rm = f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rs += __shfl_xor(rs, o, 64);
        rm = fmaxf(rm, __shfl_xor(rm, o, 64));
    }
    bad += __float_as_uint(wave_sum(f)) != __float_as_uint(rs);
    bad += __float_as_uint(wave_max(f)) != __float_as_uint(rm);
    const unsigned long long any = __ballot(bad != 0);
    if (bad) atomicAdd(cnt, (unsigned long long)bad);
    (void)any;
}
void launch_lane_xor_selftest(unsigned seed, int blocks, unsigned long long* cnt, hipStream_t st) {
    hipLaunchKernelGGL(lane_xor_selftest_kernel, dim3(blocks), dim3(256), 0, st, seed, cnt);
    HIP_CHECK(hipGetLastError());
}

void launch_sampler(const SamplerArgs& a, hipStream_t st) {
    AUR_REQUIRE(a.V <= 1040, "sampler: V <= 1040");
    trace_launch("sampler_kernel");
    hipLaunchKernelGGL(sampler_kernel, dim3(a.Ms), dim3(256), 0, st, a.sample_slot, a.P, a.bias, a.Npad, a.V, a.S, a.Ms, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace aur
