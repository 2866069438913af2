// Back-to-back launch cost on one stream by grid size, block size, static LDS and bytes left dirty by the predecessor.
// Build + run: bash experiments/r03_c.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LDS_FLOATS>
__global__ void k_empty(float* p, int n_dirty) {
    __shared__ float lds[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
    if (p == nullptr && threadIdx.x == 9999) p[0] = lds[3];
    // leave n_dirty floats per thread dirty in L2
    for (int i = 0; i < n_dirty; ++i) p[((long)blockIdx.x * blockDim.x + threadIdx.x) * n_dirty + i] = (float)i;
}

template <class F>
static float time_us(hipStream_t st, int iters, F&& f) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) f();
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    float* buf;
    CHECK(hipMalloc(&buf, (size_t)64 << 20));
    for (int grid : {1, 64, 256, 768, 1024})
        for (int block : {64, 256, 1024}) {
            const float a = time_us(st, 1000, [&] { hipLaunchKernelGGL(k_empty<0>, dim3(grid), dim3(block), 0, st, buf, 0); });
            const float b = time_us(st, 1000, [&] { hipLaunchKernelGGL(k_empty<4096>, dim3(grid), dim3(block), 0, st, buf, 0); });
            const float c = time_us(st, 1000, [&] { hipLaunchKernelGGL(k_empty<16384>, dim3(grid), dim3(block), 0, st, buf, 0); });
            printf("grid %4d x %4d threads: no LDS %.2f us | 16 KB LDS %.2f us | 64 KB LDS %.2f us per launch\n", grid, block, a, b, c);
        }
    for (int nd : {1, 4, 16}) {
        const float a = time_us(st, 1000, [&] { hipLaunchKernelGGL(k_empty<0>, dim3(256), dim3(1024), 0, st, buf, nd); });
        printf("grid 256 x 1024, each launch leaves %d KB dirty: %.2f us per launch\n", nd * 1024, a);
    }
    // the same through a graph of 100 launches
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty<16384>, dim3(256), dim3(1024), 0, st, buf, 0);
        CHECK(hipStreamEndCapture(st, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const float a = time_us(st, 50, [&] { CHECK(hipGraphLaunch(ge, st)); });
        printf("graph of 100 x (256 x 1024, 64 KB LDS): %.2f us per launch\n", a / 100);
    }
    return 0;
}
