// Shared host/device helpers for the gfx950 kernels of the XTTSv2 hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace aur {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Typed failures: the C ABI maps the exception TYPE (not the message) to its AUR_E_* code.
struct HipError : std::runtime_error {          // HIP runtime / kernel failure            -> AUR_E_HIP
    using std::runtime_error::runtime_error;
};
struct InvalidArgument : std::runtime_error {   // bad argument, unknown id, shape mismatch -> AUR_E_INVALID
    using std::runtime_error::runtime_error;
};
struct StateError : std::runtime_error {        // call not valid in the current state      -> AUR_E_STATE
    using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
        throw HipError(buf);
    }
}
#define HIP_CHECK(x) ::aur::hip_check((x), #x, __FILE__, __LINE__)
#define AUR_REQUIRE(cond, msg)                                                          \
    do {                                                                                \
        if (!(cond)) throw ::aur::InvalidArgument(std::string("requirement failed: ") + (msg)); \
    } while (0)

// AUR_DEBUG_SYNC=1: print each launch before it is issued and synchronise after it (fault localisation).
inline bool debug_sync() {
    static const bool on = [] {
        const char* e = getenv("AUR_DEBUG_SYNC");
        return e && e[0] == '1';
    }();
    return on;
}
inline void trace_launch(const char* name) {
    if (debug_sync()) {
        (void)hipDeviceSynchronize();   // a fault of the previous launch surfaces before this line prints
        fprintf(stderr, "[aur] launch %s\n", name);
        fflush(stderr);
    }
}
inline void post_launch(const char* name, hipStream_t st) {
    hip_check(hipGetLastError(), name, __FILE__, __LINE__);
    if (debug_sync()) hip_check(hipStreamSynchronize(st), name, __FILE__, __LINE__);
}

// The butterfly v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 (every lane ends with the sum, the operand pairs and their order are
// those of six __shfl_xor steps: the same bits) without the six dependent ds_bpermute round trips through the LDS crossbar:
//   ^ 32, ^ 16  gfx950's lane swaps: v_permlane32_swap / v_permlane16_swap of a register with itself leave {lower, lower} /
//               {upper, upper} halves (resp. even / odd 16-lane rows) in the two results; their sum is v[l] + v[l ^ 32] in every lane;
//   ^ 8         DPP row rotation by 8;
//   ^ 4         DPP row rotation by 4: the value is invariant under ^ 8 by then, so lane l - 4 holds what lane l ^ 4 holds;
//   ^ 2, ^ 1    DPP quad permutations.
// Callers run it with all 64 lanes active (a wave per row).
#define AUR_WAVE_BUTTERFLY(OP)                                                                                                    \
    {                                                                                                                             \
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);                    \
        v = OP(__uint_as_float(r[0]), __uint_as_float(r[1]));                                                                      \
    }                                                                                                                             \
    {                                                                                                                             \
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);                    \
        v = OP(__uint_as_float(r[0]), __uint_as_float(r[1]));                                                                      \
    }                                                                                                                             \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false)));   /* row_ror:8 */       \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false)));   /* row_ror:4 */       \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));    /* quad_perm [2,3,0,1] */ \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));    /* quad_perm [1,0,3,2] */
__device__ __forceinline__ float aur_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) {
    AUR_WAVE_BUTTERFLY(aur_addf)
    return v;
}
// 64-lane sum on the DPP network (6 VALU adds + one readlane) instead of 6 dependent ds_bpermute round trips (~100 cycles
// each): quad swaps, half-row and row mirrors leave every lane of a 16-lane row with the row sum; row_bcast15 / row_bcast31
// carry it across the four rows into lane 63.  Fixed order => deterministic.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define AUR_DPP_ADD(ctrl, rmask)                                                                                         \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xF, false))
    AUR_DPP_ADD(0xB1, 0xF);    // quad_perm [1,0,3,2]
    AUR_DPP_ADD(0x4E, 0xF);    // quad_perm [2,3,0,1]
    AUR_DPP_ADD(0x141, 0xF);   // row_half_mirror
    AUR_DPP_ADD(0x140, 0xF);   // row_mirror
    AUR_DPP_ADD(0x142, 0xA);   // row_bcast15 into rows 1 and 3
    AUR_DPP_ADD(0x143, 0xC);   // row_bcast31 into rows 2 and 3
#undef AUR_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    AUR_WAVE_BUTTERFLY(fmaxf)
    return v;
}
#undef AUR_WAVE_BUTTERFLY

// x of lane (lane ^ J), J a power of two below 64: what __shfl_xor(x, J, 64) returns, without its ds_bpermute round trip through the
// LDS crossbar (~100 cycles, and a counted wait behind it).  J = 1, 2: DPP quad permutations; 4: two row shifts, each written to the
// banks (groups of four lanes) it serves; 8: row rotation; 16, 32: gfx950's v_permlane16_swap / v_permlane32_swap of the register with
// itself (results {R0,R0,R2,R2} / {R1,R1,R3,R3} by 16-lane rows, resp. {lower,lower} / {upper,upper} by halves) and a select on the
// lane's own bit.  All 64 lanes active.  aur_dbg_lane_xor_selftest (engine.hip) checks every J against __shfl_xor on the GPU.
template <int J>
__device__ __forceinline__ int lane_xor(int x) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "lane_xor: a power of two below 64");
    if constexpr (J == 1) {
        return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);   // row_shl:4 -> banks 0, 2: lane l reads l + 4
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);          // row_shr:4 -> banks 1, 3: lane l reads l - 4
    } else if constexpr (J == 8) {
        return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);  // row_ror:8
    } else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((__lane_id() & 16) ? r[0] : r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((__lane_id() & 32) ? r[0] : r[1]);
    }
}
template <int J>
__device__ __forceinline__ float lane_xor(float x) {
    return __int_as_float(lane_xor<J>(__float_as_int(x)));
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// tanh-form GELU ("gelu_new"), same expression order as the oracle.
__device__ __forceinline__ float gelu_new(float x) {
    const float k = 0.7978845608028654f;  // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}

// erf-form GELU (config.json "activation_function": "gelu", the XTTSGPTConfig class default, xttsv2_gpt_config.py:184)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

}  // namespace aur
