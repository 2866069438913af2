// What does a DEPENDENT kernel boundary cost on this GPU, and can a flag protocol under out-of-order launches undercut it?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pdl_bench.hip -o /tmp/pdl_bench && /tmp/pdl_bench
//
// A decode step is ~150 dependent launches of 4-20 us each: every one pays the boundary.  Three chains of N
// launches of the same small kernel (256 workgroups x 1024 threads, one per CU -- 96 KB of LDS -- like the decode GEMMs):
//   (a) stream order: the ordinary launch; the command processor holds kernel i + 1 until kernel i has retired (barrier bit),
//       with the cache maintenance of a boundary;
//   (b) stream order, captured once in a hipGraph and replayed (no host launch cost in the period);
//   (c) hipExtAnyOrderLaunch (no barrier bit: the workgroups of kernel i + 1 are placed as soon as CUs free up) + a counter per
//       kernel: every workgroup of kernel i adds one to flag[i] on exit (release, agent scope), every workgroup of kernel i + 1
//       spins until flag[i] == workgroups (acquire) before its "dependent" phase.  What a programmatic dependent launch would
//       cost per boundary; `work_us` of busy-waiting in front of the wait stands for the weight prefetch that could overlap it.
// The spin gives up after ~20 ms (and reports it) so that a protocol error cannot hang the GPU.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

extern __shared__ float lds[];

// mode 0: no flags.  mode 1: wait for flag[i - 1] (i > 0), signal flag[i], acquire / release at agent scope (the compiler's
// buffer_inv sc1 / buffer_wbl2 sc1).  mode 2: the same counters RELAXED, the data itself moved with write-through stores and
// L2-bypassing loads (sc1), a vmcnt(0) drain in front of the counter -- the protocol of the decode GEMM's K split.
__global__ __launch_bounds__(1024) void step_kernel(unsigned* flags, float* data, int i, int nwg, int mode, int work_cycles, unsigned* err) {
    const int tid = threadIdx.x;
    lds[tid] = (float)tid;   // (touch the LDS so that the allocation is real)
    if (work_cycles > 0) {   // independent prologue work (stands for the weight-tile prefetch)
        const long t0 = wall_clock64();
        while (wall_clock64() - t0 < work_cycles) __builtin_amdgcn_s_sleep(1);
    }
    if (mode >= 1 && i > 0) {
        if (tid == 0) {
            const long t0 = wall_clock64();
            while ((mode == 1 ? __hip_atomic_load(&flags[i - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                              : __hip_atomic_load(&flags[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned)nwg) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 2000000) {   // 100 MHz counter: 20 ms
                    atomicAdd(err, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
    // dependent phase: read what the previous kernel wrote (one element per thread), write this kernel's
    const long off = (long)blockIdx.x * 1024 + tid;
    float v = 1.0f;
    const float* src = data + (long)(i - 1) * nwg * 1024 + ((off * 7) % ((long)nwg * 1024));
    if (i > 0) v = mode == 2 ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __builtin_nontemporal_load(src);
    if (mode >= 1) __hip_atomic_store(data + (long)i * nwg * 1024 + off, v + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else data[(long)i * nwg * 1024 + off] = v + 1.0f;
    if (mode == 1) {
        __syncthreads();   // every wave's stores are issued; the release below orders them before the counter
        if (tid == 0) __hip_atomic_fetch_add(&flags[i], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 2) {
        __builtin_amdgcn_s_waitcnt(0);   // the write-through stores of this wave have been acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&flags[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    const int nwg = 256, lds_bytes = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *flags, *err;
    float* data;
    CK(hipMalloc(&flags, N * sizeof(unsigned)));
    CK(hipMalloc(&err, sizeof(unsigned)));
    CK(hipMalloc(&data, (size_t)N * nwg * 1024 * sizeof(float)));
    CK(hipMemset(err, 0, sizeof(unsigned)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto check = [&](const char* what) {   // element (i, x) = i + 2 for every chain that kept its dependencies
        std::vector<float> h((size_t)nwg * 1024);
        CK(hipMemcpy(h.data(), data + (size_t)(N - 1) * nwg * 1024, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        long bad = 0;
        for (float x : h) bad += x != (float)(N + 1);
        unsigned he = 0;
        CK(hipMemcpy(&he, err, sizeof(he), hipMemcpyDeviceToHost));
        printf("    %s: %ld wrong elements of %zu in the last kernel's output, %u spin time-outs\n", what, bad, h.size(), he);
    };
    auto run = [&](const char* name, int mode, bool any_order, int work_cycles, int reps) {
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(flags, 0, N * sizeof(unsigned), st));
            CK(hipMemsetAsync(data, 0, (size_t)N * nwg * 1024 * sizeof(float), st));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) {
                if (any_order)
                    hipExtLaunchKernelGGL(step_kernel, dim3(nwg), dim3(1024), lds_bytes, st, nullptr, nullptr, hipExtAnyOrderLaunch, flags, data, i,
                                          nwg, mode, work_cycles, err);
                else
                    hipLaunchKernelGGL(step_kernel, dim3(nwg), dim3(1024), lds_bytes, st, flags, data, i, nwg, mode, work_cycles, err);
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-78s %6.2f us per launch\n", name, best * 1e3f / N);
        check(name);
    };
    run("(a) stream order, no flags", 0, false, 0, 5);
    run("(a') stream order, flags kept (cost of the counter alone)", 1, false, 0, 5);
    run("(a) stream order, no flags, 2 us of prologue work", 0, false, 200, 5);
    // (b) graph replay of (a)
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(step_kernel, dim3(nwg), dim3(1024), lds_bytes, st, flags, data, i, nwg, 0, 0, err);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-78s %6.2f us per launch\n", "(b) stream order, hipGraph replay", best * 1e3f / N);
        check("(b)");
    }
    run("(c) any-order launches + flags", 1, true, 0, 5);
    run("(c) any-order launches + flags, 2 us of prologue work in front of the wait", 1, true, 200, 5);
    run("(c) any-order launches + flags, 4 us of prologue work in front of the wait", 1, true, 400, 5);
    run("(a) stream order, no flags, 4 us of prologue work", 0, false, 400, 5);
    run("(a'') stream order, relaxed counters + sc1 data kept", 2, false, 0, 5);
    run("(d) any-order launches + relaxed counters + sc1 data", 2, true, 0, 5);
    run("(d) any-order + relaxed counters + sc1 data, 2 us of prologue work", 2, true, 200, 5);
    run("(d) any-order + relaxed counters + sc1 data, 4 us of prologue work", 2, true, 400, 5);
    return 0;
}
