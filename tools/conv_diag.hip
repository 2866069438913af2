// What bounds a vocoder conv launch: the 128-channel ResBlock convs (stage 1: 64 utterances x 78 016 positions) with the global
// loads of the staging / the LDS fragment reads + MFMAs / the epilogue compiled out (template parameter DBG of
// conv1d_mfma_f16_kernel).  Build + run: bash experiments/r03_i.sh
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../auralis_amd/csrc/vocoder_kernels.hip"

using namespace aur;

template <class F>
static float time_ms(hipStream_t st, int iters, F&& f) {
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    f();
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) f();
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

template <int KS, int DIL, int DBG>
static float run(const ConvArgs& a, hipStream_t st) {
    const int n_q = a.max_len;
    dim3 grid((n_q + 255) / 256, a.Mtot / 64, a.B);
    return time_ms(st, 5, [&] { hipLaunchKernelGGL((conv1d_mfma_f16_kernel<KS, DIL, 64, true, DBG>), grid, dim3(256), 0, st, a); });
}

template <int KS, int DIL, int NBUF>
static float run_dma(const ConvArgs& a, hipStream_t st) {
    dim3 grid((a.max_len + 255) / 256, a.Mtot / 64, a.B);
    return time_ms(st, 5, [&] { hipLaunchKernelGGL((conv1d_dma_f16_kernel<KS, DIL, 64, NBUF>), grid, dim3(256), 0, st, a); });
}

template <int KS, int DIL>
static void sweep(const char* what, const ConvArgs& a, hipStream_t st) {
    printf("%s k=%2d d=%d :  full %.3f | no loads %.3f | no mfma %.3f | no epilogue %.3f | no loads+mfma %.3f | no mfma+epilogue (staging only) %.3f | "
           "barriers + LDS writes only %.3f  ms\n", what, KS, DIL, run<KS, DIL, 0>(a, st), run<KS, DIL, 1>(a, st), run<KS, DIL, 4>(a, st),
           run<KS, DIL, 8>(a, st), run<KS, DIL, 5>(a, st), run<KS, DIL, 12>(a, st), run<KS, DIL, 13>(a, st));
    printf("%s k=%2d d=%d :  LDS-DMA staged, 2 buffers %.3f", what, KS, DIL, run_dma<KS, DIL, 2>(a, st));
    if constexpr (KS < 11) printf(" | 3 buffers %.3f", run_dma<KS, DIL, 3>(a, st));
    if constexpr (KS < 7) printf(" | 4 buffers %.3f", run_dma<KS, DIL, 4>(a, st));
    printf("  ms\n");
    fflush(stdout);
}

int main() {
    HIP_CHECK(hipSetDevice(0));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    const int B = 64, C = 128, L = 78016;
    const size_t n = (size_t)B * C * L;
    _Float16 *xh, *act2;
    float *res, *out, *bias;
    void* wp16;
    int* len;
    HIP_CHECK(hipMalloc(&xh, n * 2));
    HIP_CHECK(hipMalloc(&act2, n * 2));
    HIP_CHECK(hipMalloc(&res, n * 4));
    HIP_CHECK(hipMalloc(&out, n * 4));
    HIP_CHECK(hipMalloc(&bias, C * 4));
    HIP_CHECK(hipMalloc(&wp16, (size_t)C * C * 11 * 2));
    HIP_CHECK(hipMalloc(&len, B * 4));
    HIP_CHECK(hipMemset(xh, 0x2e, n * 2));      // halves 0x2e2e ~ 0.096
    HIP_CHECK(hipMemset(act2, 0x2e, n * 2));
    HIP_CHECK(hipMemset(res, 0, n * 4));
    HIP_CHECK(hipMemset(bias, 0, C * 4));
    HIP_CHECK(hipMemset(wp16, 0x22, (size_t)C * C * 11 * 2));   // ~ 0.012
    std::vector<int> hl(B, L);
    HIP_CHECK(hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice));
    void* zeros;
    HIP_CHECK(hipMalloc(&zeros, 256));
    HIP_CHECK(hipMemset(zeros, 0, 256));
    ConvArgs a{};
    a.zeros = zeros;
    a.x = reinterpret_cast<const float*>(xh); a.wp16 = wp16; a.bias = bias; a.base_len = len; a.len_mul = 1; a.Cin = C; a.Mtot = C; a.Cout = C;
    a.x_stride = L; a.o_stride = L; a.x_bstride = (long)C * L; a.o_bstride = (long)C * L; a.slope = 0.1f; a.max_len = L; a.B = B; a.x_f16 = 1;
    // residual conv, rounds 0 / 1: fp16 activated input, activated fp16 residual stream updated in place (6 B per element)
    ConvArgs s2 = a;
    s2.res = reinterpret_cast<const float*>(act2); s2.res_f16 = 1; s2.res_unact = 10.f; s2.out = reinterpret_cast<float*>(act2); s2.out_act_f16 = 1; s2.out_slope = 0.1f;
    s2.padl = 1; sweep<3, 1>("residual conv            ", s2, st);
    s2.padl = 3; sweep<7, 1>("residual conv            ", s2, st);
    s2.padl = 5; sweep<11, 1>("residual conv            ", s2, st);
    // residual conv, round 2, MRF accumulate (mode 2): fp16 residual + fp16 accumulator read-modify-write (8 B per element)
    ConvArgs s3 = a;
    s3.res = reinterpret_cast<const float*>(act2); s3.res_f16 = 1; s3.res_unact = 10.f; s3.mrf = out; s3.mrf_f16 = 1; s3.out = res; s3.mrf_mode = 2;
    s3.padl = 3; sweep<7, 1>("residual conv (MRF +=)   ", s3, st);
    // first conv: activated fp16 stream in -> fp16 activated output (4 B per element)
    ConvArgs s1 = a;
    s1.x = reinterpret_cast<const float*>(act2);
    s1.out = reinterpret_cast<float*>(xh); s1.out_act_f16 = 1; s1.out_slope = 0.1f;
    s1.padl = 3; sweep<3, 3>("first conv               ", s1, st);
    s1.padl = 9; sweep<7, 3>("first conv               ", s1, st);
    s1.padl = 15; sweep<11, 3>("first conv               ", s1, st);
    return 0;
}
