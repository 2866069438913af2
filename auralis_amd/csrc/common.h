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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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
