// EXPERIMENT (round 2), NOT part of the library: the GEMM phases of the 30 GPT-2 blocks of one decode step as ONE resident grid
// (one workgroup of 8 waves per CU) instead of one launch per GEMM.  Built by tools/persist_bench.hip only.
//
// Result on MI355X (profiles/r02_persist_bench.log, M = 64, 30 layers, 4 GEMM phases per layer): bitwise equal to the launch
// chain, 47.6 us per layer against 42.4 us for the four launches.  In-kernel stamps: per phase ~2.8 us until the activations
// written by the other XCDs arrive from memory, ~3.2 us of MFMA (two waves per SIMD, at the issue limit), 1.5 us waiting for
// the slower waves, ~2 us reduction + epilogue, 2.5-3.3 us device-wide barrier (tools/gridbar_bench: 2.15 us for the barrier
// alone).  One unit per workgroup per phase leaves nothing to overlap those latencies with, whereas the launch chain keeps 2-3
// workgroups per CU in different stages; a kernel boundary (2.5 us) is not more expensive than a cross-XCD barrier on this part.
//
// Why: at M = 64 rows every GEMM of the step lives for ~10 us, of which ~2.5 us is launch ramp and ~2 us is the first
// weight fetch from HBM.  Inside one resident grid the next phase's weights (they do not depend on the activations) are
// requested BEFORE the device-wide barrier and land while the grid waits; the residual tile of a workgroup stays in
// registers for the whole step.
//
// Arithmetic: every output element is produced by exactly the instruction sequence of gemm_rows_kernel /
// paged_attention_kernel (same K slices per wave, same LDS reduction order, same epilogue expressions), so the result is
// bitwise the launch chain's (checked by tools/persist_bench.hip).
//
// Exchange between workgroups without cache maintenance (measured, tools/gridbar_bench: an agent-scope release/acquire
// fence pair, i.e. buffer_wbl2 + buffer_inv, costs 8-70 us per barrier on this part):
//  * every tensor a phase hands to the next one is written ONCE per kernel to an address that no cache can hold yet (a
//    per-layer ring, caches are invalidated at kernel boundaries) with write-through stores (sc1); whole 128 B lines have a
//    single writing workgroup wherever the reader uses plain loads (packed 1 KiB blocks);
//  * the few values whose lines are shared by several writers (q rows, LayerNorm partials, the new K/V token) are read with
//    sc1 loads (they bypass the non-coherent caches);
//  * the barrier is hierarchical: arrivals are counted per XCD (the XCD id comes from the hardware register, the populations
//    are counted once per launch), the last arriver of an XCD publishes one flag, everybody polls the line of 8 flags.
#pragma once

namespace aur {

struct DecodeLayerP {
    const float *tqkv, *bqkv, *ln1w, *ln1b;
    const float *tproj, *bproj;
    const float *tfc, *bfc, *ln2w, *ln2b;
    const float *tproj2, *bproj2;
    void* kv_layer;
};

// ring (floats) of one layer: q rows [64][1024] row-major, the packed (mtt = 4) att / hB / act / hA-next tiles, statistics
constexpr long kRingQ = 0, kRingAtt = 65536, kRingHb = 131072, kRingAct = 196608, kRingHa = 458752, kRingStB = 524288,
               kRingStA = 532480, kRingLayer = 540672;
// control block (unsigned words, one 128 B line per hot word)
constexpr int kCtlCount = 0, kCtlPop = 32, kCtlLocal = 64, kCtlXflag = 576, kCtlErr = 608, kCtlWords = 640;

struct DecodePersistArgs {
    const DecodeLayerP* layers;
    int n_layer, M;
    float eps;
    const float* h0;          // packed rows (mtt0 tiles): the embedded tokens
    int mtt0;
    const float2* stats0;     // their LayerNorm partials [rows][64]
    float* ring;              // n_layer * kRingLayer floats
    float* h_out;             // packed rows (mtt = 4): residual stream after the last block
    const int* row_meta;
    int kv_half;
    unsigned* ctl;            // kCtlWords, zeroed before the launch
    int spin_cap;
    const float* att_fixed;   // ATTN == false (tools/persist_bench): constant packed input of the proj GEMM
    long long* prof;          // optional: 64 s_memtime stamps per workgroup (layer 1), written by thread 0
};

__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt2(float2* p, float2 v) {
    union {
        float2 f;
        unsigned long long u;
    } c;
    c.f = v;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), c.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_mem2(const float2* p) {
    union {
        float2 f;
        unsigned long long u;
    } c;
    c.u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return c.f;
}
// loads through pointers that were themselves read from memory (DecodeLayerP): the compiler cannot prove the address space and
// would emit FLAT loads, which also count on lgkmcnt, so that the s_waitcnt lgkmcnt(0) in front of every s_barrier would wait
// for the prefetch
typedef const f32x4 __attribute__((address_space(1))) * gptr4_t;
typedef const float __attribute__((address_space(1))) * gptr1_t;
__device__ __forceinline__ f32x4 ldg4(const f32x4* p) { return *(gptr4_t)p; }
__device__ __forceinline__ float ldg1(const float* p) { return *(gptr1_t)p; }
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

struct PersistCtx {
    unsigned* ctl;
    unsigned* s_ctl;   // LDS: [0] abort, [1] population of this XCD, [2] XCD id, [3] mask of populated XCDs
    unsigned round;
    int spin_cap;
};

// PF = vector-memory loads this wave issued AFTER its last exchange store (they stay in flight across the barrier)
template <int PF>
__device__ __forceinline__ bool grid_sync(PersistCtx& c) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");
    __syncthreads();
    const unsigned r = ++c.round;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned x = c.s_ctl[2], pop = c.s_ctl[1], mask = c.s_ctl[3];
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&c.ctl[kCtlLocal + 32 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        int spins = 0;
        bool dead = false;
        // last arriver of this XCD publishes the XCD's flag; every workgroup polls the 8 flags (one 128 B line, L2-bypassing loads).
        // Measured (tools/gridbar_bench, 256 workgroups): 2.15 us per barrier; a single counter or 256 flags cost 3.8-4.0 us (the
        // pollers of one line serialise at the memory side), a second-level "go" word per XCD 2.24 us.
        if (old + 1 == pop * r && lane == 0) __hip_atomic_store(&c.ctl[kCtlXflag + x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
            bool ok = true;
            if (lane < 8 && ((mask >> lane) & 1))
                ok = __hip_atomic_load(&c.ctl[kCtlXflag + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= r;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > c.spin_cap) {
                dead = true;
                break;
            }
        }
        if (dead && lane == 0) {
            atomicOr(&c.ctl[kCtlErr], 1u);
            c.s_ctl[0] = 1;
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    return c.s_ctl[0] == 0;
}

// Workgroup = 8 waves (512 threads, one per CU, up to 256 VGPRs per lane): a wave plays TWO of the K-slice roles of the launch
// kernels (two independent accumulator chains), so the 8-slice / 16-slice reductions keep their order.
//
// LayerNorm-prologue GEMM phase (K = 1024): each 4-wave half of the workgroup takes two 16-column tiles (`tile0`, `tile1` or -1;
// their weights are already in pw) for the workgroup's 16 rows; wave w4 of the half owns K-slices 2*w4 and 2*w4 + 1 of the 8:
// the arithmetic of gemm_rows_kernel<1, 1, true, EPI, 8, NTL> (NTL = 1 and 2 give the same bits).
template <class EpiF>
__device__ __forceinline__ void ln_phase(const float* __restrict__ X, int xmt, const float2* __restrict__ stats, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float eps, int tile0, int tile1, const float* __restrict__ bias,
                                         int m_tile, f32x4 (&pw)[32], float (*red)[512], float (*rs)[2], float (*gb)[1024], long long* pp, EpiF&& epi) {
    constexpr int NB = 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), half = wv >> 2, w4 = wv & 3;
    const int j = lane & 15, q = lane >> 4;
    const int m0 = 16 * m_tile;
    // statistics of rows wv and wv + 8, gamma / beta, activations: all requested before the first wait
    const float2 pt0 = ld_mem2(stats + (long)(m0 + wv) * 64 + lane);
    const float2 pt1 = ld_mem2(stats + (long)(m0 + wv + 8) * 64 + lane);
    const f32x4 gbv = ldg4(reinterpret_cast<const f32x4*>(((tid < 256) ? gamma : beta) + 4 * (tid & 255)));
    const f32x4* xp = reinterpret_cast<const f32x4*>(X) + (long)m_tile * 64 + lane;
    f32x4 af[2][NB];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int b = 0; b < NB; ++b) af[u][b] = xp[(long)(NB * (2 * w4 + u) + b) * xmt * 64];
    const float bias0 = ldg1(bias + tile0 * 16 + (tid & 15));   // (the epilogue's column is 16*tile + (tid & 15))
    const float bias1 = tile1 >= 0 ? ldg1(bias + tile1 * 16 + (tid & 15)) : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<f32x4*>(&gb[tid >> 8][4 * (tid & 255)]) = gbv;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float2 pt = u ? pt1 : pt0;
        const float mu = wave_sum_dpp(pt.x) * (1.0f / 64.0f);
        const float d = pt.x - mu;
        const float m2 = wave_sum_dpp(fmaf(16.0f * d, d, pt.y));
        if (lane == 0) {
            rs[wv + 8 * u][0] = mu;
            rs[wv + 8 * u][1] = 1.0f / sqrtf(m2 * (1.0f / 1024.0f) + eps);
        }
    }
    __syncthreads();
    if (pp) pp[0] = (long long)__builtin_amdgcn_s_memtime();
    const float mean = rs[j][0], rstd = rs[j][1];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int k0 = 16 * (NB * (2 * w4 + u) + b) + 4 * q;
            const f32x4 gam = *reinterpret_cast<const f32x4*>(&gb[0][k0]);
            const f32x4 bet = *reinterpret_cast<const f32x4*>(&gb[1][k0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) af[u][b][s] = (af[u][b][s] - mean) * rstd * gam[s] + bet[s];
        }
    f32x4 acc[2][2];   // [tile][slice]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[r][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[r][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][b][s], pw[(r * 2 + u) * NB + b][s], acc[r][u], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[8 * half + 2 * w4 + u][r * 256 + c * 64 + lane] = acc[r][u][c];
    if (pp) pp[1] = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();
    if (pp) pp[2] = (long long)__builtin_amdgcn_s_memtime();
    const int e = tid & 255;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int tile = r == 0 ? tile0 : tile1;
        if (tile < 0) continue;
        float t = red[8 * half][r * 256 + e];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) t += red[8 * half + ww][r * 256 + e];
        const int c = (e >> 6) & 3, l = e & 63;
        const int m = m0 + 4 * (l >> 4) + c, n = tile * 16 + (l & 15);
        t += r == 0 ? bias0 : bias1;
        epi(m, n, t);
    }
}

// Plain GEMM phase (K = 1024 * KCH, N = 1024): the workgroup's own (16 rows x 16 columns) unit; wave wv owns the K-slices
// 2*wv and 2*wv + 1 of the 16, i.e. [c*1024 + 64*slice, +64) of every chunk c (the arithmetic of
// gemm_rows_kernel<1, KCH, false, kEpiResidual, 16, 1>).  pw[(c*2 + u)*4 + b]: weight chunk c (c < 2 prefetched when KCH == 4).
template <int KCH>
__device__ __forceinline__ float plain_phase(const float* __restrict__ X, int xmt, const float* __restrict__ Wt, int tile, int m_tile,
                                             f32x4 (&pw)[32], float (*red)[512], long long* pp) {
    constexpr int NB = 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long wt_tile = 64L * 64 * KCH;
    const f32x4* wt = reinterpret_cast<const f32x4*>(Wt) + (long)tile * wt_tile + lane;
    const f32x4* xp = reinterpret_cast<const f32x4*>(X) + (long)m_tile * 64 + lane;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (pp) pp[1] = (long long)__builtin_amdgcn_s_memtime();
    auto ldx = [&](int c, f32x4* d) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) d[u * NB + b] = xp[(long)(c * 64 + NB * (2 * wv + u) + b) * xmt * 64];
    };
    auto ldw = [&](int c, f32x4* d) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) d[u * NB + b] = ldg4(&wt[(long)(c * 64 + NB * (2 * wv + u) + b) * 64]);
    };
    auto mma = [&](const f32x4* x, const f32x4* w) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u * NB + b][s], w[u * NB + b][s], acc[u], 0, 0, 0);
    };
    if constexpr (KCH == 1) {
        f32x4 xa[8];
        ldx(0, xa);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, &pw[0]);
    } else {
        f32x4 xa[8], xb[8], wc[8];
        ldx(0, xa);
        ldx(1, xb);
        ldw(2, wc);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, &pw[0]);
        __builtin_amdgcn_sched_barrier(0);
        ldx(2, xa);
        ldw(3, &pw[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(xb, &pw[8]);
        __builtin_amdgcn_sched_barrier(0);
        ldx(3, xb);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, wc);
        mma(xb, &pw[0]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[2 * wv + u][c * 64 + lane] = acc[u][c];
    if (pp) pp[2] = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();
    if (pp) pp[3] = (long long)__builtin_amdgcn_s_memtime();
    float t = 0.f;
    if (tid < 256) {
        t = red[0][tid];
#pragma unroll
        for (int ww = 1; ww < 16; ++ww) t += red[ww][tid];
    }
    return t;
}

template <bool ATTN, bool KVH>
__global__ __launch_bounds__(512) void decode_layers_kernel(DecodePersistArgs a) {
    __shared__ __attribute__((aligned(16))) float red[16][512];
    __shared__ float rs[16][2];
    __shared__ __attribute__((aligned(16))) float gb[2][1024];
    __shared__ unsigned s_ctl[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), half = wv >> 2, w4 = wv & 3;
    const int W = blockIdx.x, xcd = W & 7, li = W >> 3, m_tile = li & 3, g = li >> 2;
    const int m0 = 16 * m_tile;
    const bool live = m0 < a.M;   // (workgroups of row tiles beyond M only take part in the barriers)
    PersistCtx cx{a.ctl, s_ctl, 0u, a.spin_cap};
    // ---- once per launch: which XCD am I on, how many workgroups does it hold
    if (tid == 0) {
        const unsigned x = xcc_id() & 7;
        s_ctl[0] = 0;
        s_ctl[2] = x;
        __hip_atomic_fetch_add(&a.ctl[kCtlPop + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.ctl[kCtlCount], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&a.ctl[kCtlCount], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > a.spin_cap) {
                atomicOr(&a.ctl[kCtlErr], 2u);
                s_ctl[0] = 1;
                break;
            }
        }
        unsigned mask = 0;
        for (int i = 0; i < 8; ++i)
            if (__hip_atomic_load(&a.ctl[kCtlPop + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) mask |= 1u << i;
        s_ctl[1] = __hip_atomic_load(&a.ctl[kCtlPop + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[3] = mask;
    }
    __syncthreads();
    if (s_ctl[0]) return;

    // the workgroup's residual tile (16 rows x 16 columns, column tile c_res) lives in the registers of threads 0..255
    const int c_res = g * 8 + xcd;
    const int er = (tid >> 6) & 3, el = tid & 63;
    const int em = m0 + 4 * (el >> 4) + er, en = 16 * c_res + (el & 15);
    float res = 0.f;
    if (tid < 256 && live) res = a.h0[pk_off(em, en, a.mtt0)];
    const bool eok = em < a.M;

    // K/V page address of this thread's epilogue row (LN-phase epilogue: element tid & 255 -> row m0 + 4*((tid&63)>>4) + ((tid>>6)&3))
    int kv_pos = 0, kv_blk = 0;
    if (eok) {
        kv_pos = a.row_meta[(long)em * kRowMetaStride];
        kv_blk = a.row_meta[(long)em * kRowMetaStride + kRowMetaBt + kv_pos / kKvBlockTokens];
    }
    f32x4 pw[32];
    // weights of an LN phase: tiles t0 / t1 of this half, K-slices 2*w4 and 2*w4 + 1: pw[(r*2 + u)*8 + b]
    auto prefetch_ln = [&](const float* Wt, int t0, int t1) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int tile = r == 0 ? t0 : t1;
            if (tile < 0) continue;
            const f32x4* wt = reinterpret_cast<const f32x4*>(Wt) + (long)tile * 4096 + lane;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int b = 0; b < 8; ++b) pw[(r * 2 + u) * 8 + b] = ldg4(&wt[(long)(8 * (2 * w4 + u) + b) * 64]);
        }
    };
    // QKV units of this half: 128 tiles in the first round (two halves x 64 workgroups-with-this-row-tile), 64 in the second (half 0)
    const int q_t0 = (g * 2 + half) * 8 + xcd, q_t1 = half == 0 ? 128 + g * 8 + xcd : -1;
    const int f_t0 = 2 * ((g * 2 + half) * 8 + xcd), f_t1 = f_t0 + 1;
    auto prefetch_proj = [&](const DecodeLayerP& L) {
        const f32x4* wt = reinterpret_cast<const f32x4*>(L.tproj) + (long)c_res * 4096 + lane;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int b = 0; b < 4; ++b) pw[u * 4 + b] = ldg4(&wt[(long)(4 * (2 * wv + u) + b) * 64]);
    };
    auto prefetch_proj2 = [&](const DecodeLayerP& L) {   // chunks 0 and 1 of the K = 4096 tile
        const f32x4* wt = reinterpret_cast<const f32x4*>(L.tproj2) + (long)c_res * 16384 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int b = 0; b < 4; ++b) pw[c * 8 + u * 4 + b] = ldg4(&wt[(long)(c * 64 + 4 * (2 * wv + u) + b) * 64]);
    };
    // residual epilogue shared by proj / proj2: new residual, its LayerNorm partials, write-through copy for the next GEMM
    auto residual_epi = [&](float t, float bias_n, float* hdst, float2* stdst) {
        if (tid < 256) {
            t += bias_n;
            const float v = res + t;
            res = v;
            if (eok) st_wt(hdst + pk_off(em, en, 4), v);
            if (stdst) {
                float sm = v;
                sm += __shfl_xor(sm, 8, 64);
                sm += __shfl_xor(sm, 4, 64);
                sm += __shfl_xor(sm, 2, 64);
                sm += __shfl_xor(sm, 1, 64);
                const float mu = sm * (1.0f / 16.0f);
                const float d = v - mu;
                float m2 = d * d;
                m2 += __shfl_xor(m2, 8, 64);
                m2 += __shfl_xor(m2, 4, 64);
                m2 += __shfl_xor(m2, 2, 64);
                m2 += __shfl_xor(m2, 1, 64);
                if (eok && (el & 15) == 0) st_wt2(stdst + (long)em * 64 + c_res, make_float2(mu, m2));
            }
        }
    };

    if (live) prefetch_ln(a.layers[0].tqkv, q_t0, q_t1);
    for (int l = 0; l < a.n_layer; ++l) {
        long long* pp = (a.prof && l == 1 && tid == 0) ? a.prof + (long)W * 64 : nullptr;
#define AUR_STAMP(i) if (pp) pp[i] = (long long)__builtin_amdgcn_s_memtime()
        const DecodeLayerP L = a.layers[l];
        float* ring = a.ring + (long)l * kRingLayer;
        const float* xin = l == 0 ? a.h0 : ring - kRingLayer + kRingHa;
        const int xin_mtt = l == 0 ? a.mtt0 : 4;
        const float2* stin = l == 0 ? a.stats0 : reinterpret_cast<const float2*>(ring - kRingLayer + kRingStA);
        // ---- QKV (LN1 prologue): q rows + the new token's K / V
        AUR_STAMP(0);
        if (live) {
            float* qdst = ring + kRingQ;
            ln_phase(xin, xin_mtt, stin, L.ln1w, L.ln1b, a.eps, q_t0, q_t1, L.bqkv, m_tile, pw, red, rs, gb, pp ? pp + 1 : nullptr, [&](int m, int n, float t) {
                if (m >= a.M) return;
                const int u = n / kHidden, d = n - u * kHidden;
                if (u == 0) {
                    st_wt(qdst + (long)m * kHidden + d, t);
                } else {
                    const long off = kv_offset(kv_blk, u - 1, d / kHeadDim, kv_pos % kKvBlockTokens) + d % kHeadDim;
                    if (KVH) {
                        union {
                            _Float16 h;
                            unsigned short s;
                        } cv;
                        cv.h = (_Float16)t;
                        __hip_atomic_store(reinterpret_cast<unsigned short*>(L.kv_layer) + off, cv.s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        st_wt(reinterpret_cast<float*>(L.kv_layer) + off, t);
                    }
                }
            });
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            AUR_STAMP(4);
            prefetch_proj(L);
        }
        AUR_STAMP(5);
        if (!grid_sync<8>(cx)) return;
        AUR_STAMP(8);
        // ---- attention (ATTN) ... the prototype feeds the proj GEMM from a constant tile instead
        const float* attx = ATTN ? ring + kRingAtt : a.att_fixed;
        // ---- proj + residual
        if (live) {
            const float bn = ldg1(L.bproj + en);
            const float t = plain_phase<1>(attx, 4, L.tproj, c_res, m_tile, pw, red, pp ? pp + 8 : nullptr);
            residual_epi(t, bn, ring + kRingHb, reinterpret_cast<float2*>(ring + kRingStB));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            AUR_STAMP(12);
            prefetch_ln(L.tfc, f_t0, f_t1);
        }
        AUR_STAMP(13);
        if (!grid_sync<32>(cx)) return;
        AUR_STAMP(16);
        // ---- FC (LN2 prologue) + gelu
        if (live) {
            float* adst = ring + kRingAct;
            ln_phase(ring + kRingHb, 4, reinterpret_cast<const float2*>(ring + kRingStB), L.ln2w, L.ln2b, a.eps, f_t0, f_t1, L.bfc, m_tile, pw,
                     red, rs, gb, pp ? pp + 17 : nullptr, [&](int m, int n, float t) {
                         if (m < a.M) st_wt(adst + pk_off(m, n, 4), gelu_new(t));
                     });
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            AUR_STAMP(20);
            prefetch_proj2(L);
        }
        AUR_STAMP(21);
        if (!grid_sync<16>(cx)) return;
        AUR_STAMP(24);
        // ---- proj2 + residual
        if (live) {
            const float bn = ldg1(L.bproj2 + en);
            const float t = plain_phase<4>(ring + kRingAct, 4, L.tproj2, c_res, m_tile, pw, red, pp ? pp + 24 : nullptr);
            const bool last = l + 1 == a.n_layer;
            residual_epi(t, bn, last ? a.h_out : ring + kRingHa, last ? nullptr : reinterpret_cast<float2*>(ring + kRingStA));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            AUR_STAMP(28);
            if (!last) prefetch_ln(a.layers[l + 1].tqkv, q_t0, q_t1);
        }
        AUR_STAMP(29);
        if (l + 1 < a.n_layer) {
            if (!grid_sync<32>(cx)) return;
        }
        AUR_STAMP(32);
#undef AUR_STAMP
    }
}

inline void launch_decode_layers(const DecodePersistArgs& a, bool attn, hipStream_t st) {
    AUR_REQUIRE(a.M >= 1 && a.M <= 64 && a.n_layer >= 1, "decode_layers: 1..64 rows");
    HIP_CHECK(hipMemsetAsync(a.ctl, 0, kCtlWords * sizeof(unsigned), st));
    trace_launch("decode_layers_kernel");
    if (attn) throw InvalidArgument("decode_layers: attention phase not built yet");
    if (a.kv_half) hipLaunchKernelGGL((decode_layers_kernel<false, true>), dim3(256), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((decode_layers_kernel<false, false>), dim3(256), dim3(512), 0, st, a);
    HIP_CHECK(hipGetLastError());
}

}  // namespace aur
