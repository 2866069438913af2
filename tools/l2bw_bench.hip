// Micro-benchmark: bytes per clock a CU can pull from its XCD's L2 with the decode GEMM's access pattern (one contiguous 1 KiB
// request per wave instruction, 16 B per lane), as a function of waves per CU and loads in flight per wave.  Tells whether the
// decode GEMMs (~20 B/clk/CU) sit at a bandwidth limit of the L2 -> L1 path or are latency-bound.
// Build + run on the GPU box:  bash tools/l2bw_bench.sh
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every workgroup streams `region_kb` KiB starting at (blockIdx.x % n_regions) * region: with n_regions small the data stays in
// L2 (hits), with SHARED == false every workgroup has its own region (HBM stream).  U independent loads per wave per step.
template <int U>
__global__ __launch_bounds__(1024) void stream_kernel(const f32x4* __restrict__ buf, long region_f4, int n_regions, int iters,
                                                      float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const f32x4* base = buf + (long)(blockIdx.x % n_regions) * region_f4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long per_step = (long)nw * U * 64;   // float4s per workgroup step
    const long steps = region_f4 / per_step;
    for (int it = 0; it < iters; ++it) {
        for (long s = 0; s < steps; ++s) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = base[s * per_step + ((long)u * nw + wv) * 64 + lane];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int U>
static void run(const f32x4* buf, long region_kb, int n_regions, int wgs, int threads, float* sink, const char* what) {
    const long region_f4 = region_kb * 1024 / 16;
    const int iters = (int)std::max<long>(1, (4L << 20) / (region_kb * 1024));   // ~4 MiB per workgroup in total
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(stream_kernel<U>, dim3(wgs), dim3(threads), 0, 0, buf, region_f4, n_regions, iters, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)wgs * iters * region_kb * 1024.0;
    const double cus = std::min(256, wgs);
    printf("%-10s region %5ld KiB x %3d  wgs %4d x %4d thr  U=%2d : %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (2.4 GHz, %d CUs busy)\n", what,
           region_kb, n_regions, wgs, threads, U, best, bytes / best / 1e9, bytes / (best * 1e-3) / 2.4e9 / cus, (int)cus);
}

int main() {
    const size_t total = (size_t)2 << 30;   // 2 GiB pool
    f32x4* buf;
    float* sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, total));
    // L2-resident: 8 regions of 256 KiB (one per XCD when workgroup i reads region i % 8)
    for (int threads : {512, 1024})
        for (int wgs : {256, 512, 768}) {
            run<2>(buf, 256, 8, wgs, threads, sink, "L2 hits");
            run<4>(buf, 256, 8, wgs, threads, sink, "L2 hits");
            run<8>(buf, 256, 8, wgs, threads, sink, "L2 hits");
            run<16>(buf, 256, 8, wgs, threads, sink, "L2 hits");
        }
    // HBM stream: every workgroup its own 2 MiB
    for (int threads : {512, 1024})
        for (int wgs : {256, 512, 768}) {
            run<4>(buf, 2048, wgs, wgs, threads, sink, "HBM");
            run<8>(buf, 2048, wgs, wgs, threads, sink, "HBM");
            run<16>(buf, 2048, wgs, wgs, threads, sink, "HBM");
        }
    return 0;
}
