// Micro-benchmark: cost of a device-wide barrier inside one resident grid (the building block of a persistent decode kernel),
// with and without the release / acquire traffic needed to exchange activations between workgroups on different XCDs.
// Build + run on the GPU box:  bash tools/gridbar_bench.sh
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

// MODE 0: barrier only.  1: one float per thread written, release fence (L2 write-back) + acquire fence (invalidate), plain
// loads of another workgroup's values.  2: the same exchange with write-through stores and L2-bypassing loads, no fences.
// 3: as 1, with the fences executed by every wave instead of wave 0 only.
template <int MODE>
__global__ __launch_bounds__(1024) void bar_kernel(unsigned* ctr, float* buf, int rounds, int* err, long long* cyc, int spin_cap) {
    const int nwg = gridDim.x, w = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    int bad = 0;
    __shared__ int s_abort;
    if (t == 0) s_abort = 0;
    const long long c0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        float* b = buf + (long)(r & 1) * nwg * nt;
        if (MODE == 1 || MODE == 3) b[(long)w * nt + t] = (float)(r * 7 + w);
        if (MODE == 2) __hip_atomic_store(&b[(long)w * nt + t], (float)(r * 7 + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (t == 0) {
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nwg;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_cap) {   // never hang the box: report and leave
                    atomicAdd(err, 1 << 20);
                    s_abort = 1;
                    break;
                }
            }
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (s_abort) break;
        if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (MODE != 0) {
            const int src = (w + 1 + r) % nwg;
            float v;
            if (MODE == 2) v = __hip_atomic_load(&b[(long)src * nt + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = b[(long)src * nt + t];
            if (v != (float)(r * 7 + src)) ++bad;
        }
    }
    const long long c1 = wall_clock64();
    if (bad) atomicAdd(err, bad);
    if (t == 0 && w == 0) cyc[0] = c1 - c0;
}

// All-to-all flag barrier: every workgroup publishes its round number with one write-through store and wave 0 polls all the
// flags (1 KB for 256 workgroups) with L2-bypassing loads: no atomics, nothing serialises on one address.
// XMODE 0: barrier only.  1: exchange through per-round UNIQUE addresses: write-through stores, PLAIN (cacheable) loads - no
// line of the round's buffer can be cached anywhere before it is written, so no fence / invalidate is needed.
template <int XMODE>
__global__ __launch_bounds__(1024) void flagbar_kernel(unsigned* flags, float* buf, int rounds, int* err, long long* cyc, int spin_cap) {
    const int nwg = gridDim.x, w = blockIdx.x, t = threadIdx.x;
    int bad = 0;
    __shared__ int s_abort;
    if (t == 0) s_abort = 0;
    const long long c0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        float* b = buf + (long)r * nwg * 64;
        if (XMODE == 1 && t < 64) __hip_atomic_store(&b[(long)w * 64 + t], (float)(r * 7 + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t < 64) {
            if (t == 0) __hip_atomic_store(&flags[w], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (true) {
                bool ok = true;
                for (int i = t; i < nwg; i += 64)
                    ok = ok && (__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(r + 1));
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_cap) {
                    if (t == 0) {
                        atomicAdd(err, 1 << 20);
                        s_abort = 1;
                    }
                    break;
                }
            }
        }
        __syncthreads();
        if (s_abort) break;
        if (XMODE == 1 && t < 64) {
            const int src = (w + 1 + r) % nwg;
            if (b[(long)src * 64 + t] != (float)(r * 7 + src)) ++bad;
        }
    }
    const long long c1 = wall_clock64();
    if (bad) atomicAdd(err, bad);
    if (t == 0 && w == 0) cyc[0] = c1 - c0;
}

template <int XMODE>
static void run_flag(const char* name, int nwg, int nthr, int rounds) {
    unsigned* flags;
    float* buf;
    int* err;
    long long* cyc;
    CK(hipMalloc(&flags, 4096));
    CK(hipMalloc(&buf, (size_t)rounds * nwg * 64 * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&cyc, 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    int herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(flags, 0, 4096));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(buf, 0, (size_t)rounds * nwg * 64 * 4));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(flagbar_kernel<XMODE>, dim3(nwg), dim3(nthr), 0, 0, flags, buf, rounds, err, cyc, 1 << 20);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        int e;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        herr |= e;
    }
    printf("%-44s grid %4d x %4d: %7.3f us per round  errors 0x%x\n", name, nwg, nthr, best * 1e3f / rounds, herr);
    CK(hipFree(flags));
    CK(hipFree(buf));
    CK(hipFree(err));
    CK(hipFree(cyc));
}

// Hierarchical barrier: arrivals counted with L2-local (workgroup-scope) atomics per XCD (the XCD comes from the hardware id
// register, the populations are counted once at kernel start), the last arriver of an XCD publishes one flag across the fabric,
// polls the 8 flags and releases its XCD through an L2-local word.
// XMODE 1: + exchange through unique addresses where every 128 B line is written by TWO workgroups of different XCDs (half a
// line each, write-through) and read whole with plain loads by a third: checks that a partially written line is never served
// stale from the writer's / reader's L2.
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
constexpr int kCnt = 0, kPop = 32, kLoc = 64, kGo = 320, kXf = 576;
template <int XMODE, int HV>
__global__ __launch_bounds__(1024) void hierbar_kernel(unsigned* ctl, float* buf, int rounds, int* err, long long* cyc, int spin_cap) {
    const int nwg = gridDim.x, w = blockIdx.x, t = threadIdx.x;
    __shared__ unsigned s_ctl[4];
    if (t == 0) {
        const unsigned x = xcc_id() & 7;
        s_ctl[0] = 0;
        s_ctl[2] = x;
        __hip_atomic_fetch_add(&ctl[kPop + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl[kCnt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&ctl[kCnt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_cap) {
                s_ctl[0] = 1;
                break;
            }
        }
        unsigned mask = 0;
        for (int i = 0; i < 8; ++i)
            if (__hip_atomic_load(&ctl[kPop + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) mask |= 1u << i;
        s_ctl[1] = __hip_atomic_load(&ctl[kPop + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[3] = mask;
        if (x != (unsigned)(w & 7)) atomicAdd(err, 1 << 12);   // workgroup id -> XCD is not the round-robin we assume elsewhere
    }
    __syncthreads();
    int bad = 0;
    const long long c0 = wall_clock64();
    for (int r = 1; r <= rounds && !s_ctl[0]; ++r) {
        float* b = buf + (long)(r - 1) * nwg * 32;
        // line L (32 floats) of the round: first half by workgroup L, second half by workgroup (L + 3) % nwg (another XCD)
        if (XMODE == 1) {
            if (t < 16) __hip_atomic_store(&b[(long)w * 32 + t], (float)(r * 7 + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (t < 32) {
                const int L = (w + nwg - 3) % nwg;
                __hip_atomic_store(&b[(long)L * 32 + t], (float)(r * 7 + L + 1000), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t < 64) {
            const unsigned x = s_ctl[2], pop = s_ctl[1], mask = s_ctl[3];
            unsigned old = 0;
            if (t == 0) old = __hip_atomic_fetch_add(&ctl[kLoc + 32 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old);
            int spins = 0;
            bool dead = false;
            const bool last = old + 1 == pop * (unsigned)r;
            if (last && t == 0) __hip_atomic_store(&ctl[kXf + x], (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last || HV == 2) {
                while (true) {
                    bool ok = true;
                    if (t < 8 && ((mask >> t) & 1)) ok = __hip_atomic_load(&ctl[kXf + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_cap) {
                        dead = true;
                        break;
                    }
                }
                if (HV == 1 && t == 0) __hip_atomic_store(&ctl[kGo + 32 * x], dead ? 0xFFFFFFFFu : (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (true) {
                    const unsigned v = __hip_atomic_load(&ctl[kGo + 32 * x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == 0xFFFFFFFFu) {
                        dead = true;
                        break;
                    }
                    if (v >= (unsigned)r) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_cap) {
                        dead = true;
                        break;
                    }
                }
            }
            if (dead && t == 0) {
                atomicAdd(err, 1 << 20);
                s_ctl[0] = 1;
            }
        }
        asm volatile("" ::: "memory");
        __syncthreads();
        if (XMODE == 1 && t < 32 && !s_ctl[0]) {
            const int L = (w + 5 + r) % nwg;
            const float want = t < 16 ? (float)(r * 7 + L) : (float)(r * 7 + L + 1000);
            if (b[(long)L * 32 + t] != want) ++bad;
        }
    }
    const long long c1 = wall_clock64();
    if (bad) atomicAdd(err, bad);
    if (t == 0 && w == 0) cyc[0] = c1 - c0;
}

template <int XMODE, int HV>
static void run_hier(const char* name, int nwg, int nthr, int rounds) {
    unsigned* ctl;
    float* buf;
    int* err;
    long long* cyc;
    CK(hipMalloc(&ctl, 4096));
    CK(hipMalloc(&buf, (size_t)rounds * nwg * 32 * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&cyc, 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    int herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, 4096));
        CK(hipMemset(err, 0, 4));
        CK(hipMemset(buf, 0, (size_t)rounds * nwg * 32 * 4));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((hierbar_kernel<XMODE, HV>), dim3(nwg), dim3(nthr), 0, 0, ctl, buf, rounds, err, cyc, 1 << 20);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        int e;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        herr |= e;
    }
    unsigned hc[640];
    CK(hipMemcpy(hc, ctl, sizeof hc, hipMemcpyDeviceToHost));
    printf("%-44s grid %4d x %4d: %7.3f us per round  errors 0x%x  XCD populations", name, nwg, nthr, best * 1e3f / rounds, herr);
    for (int i = 0; i < 8; ++i) printf(" %u", hc[kPop + i]);
    printf("\n");
    CK(hipFree(ctl));
    CK(hipFree(buf));
    CK(hipFree(err));
    CK(hipFree(cyc));
}

template <int MODE>
static void run(const char* name, int nwg, int nthr, int rounds) {
    unsigned* ctr;
    float* buf;
    int* err;
    long long* cyc;
    CK(hipMalloc(&ctr, 4));
    CK(hipMalloc(&buf, (size_t)2 * nwg * nthr * 4));
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&cyc, 8));
    CK(hipMemset(buf, 0, (size_t)2 * nwg * nthr * 4));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bar_kernel<MODE>, nthr, 0));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    int herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4));
        CK(hipMemset(err, 0, 4));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(bar_kernel<MODE>, dim3(nwg), dim3(nthr), 0, 0, ctr, buf, rounds, err, cyc, 1 << 20);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        int e;
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        herr |= e;
    }
    long long hc;
    CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s grid %4d x %4d (occupancy %d/CU): %7.3f us per round  (in-kernel %.3f us at 100 MHz ticks)  errors 0x%x\n", name, nwg,
           nthr, occ, best * 1e3f / rounds, (double)hc / 100.0 / rounds, herr);
    CK(hipFree(ctr));
    CK(hipFree(buf));
    CK(hipFree(err));
    CK(hipFree(cyc));
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int R = 2000;
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int nwg = cfg == 0 ? 256 : cfg == 1 ? 512 : 128, nthr = cfg == 1 ? 512 : 1024;
        run<0>("barrier only", nwg, nthr, R);
        run<1>("exchange, wave-0 release/acquire fences", nwg, nthr, R);
        run<3>("exchange, fences in every wave", nwg, nthr, R);
        run<2>("exchange, write-through stores + sc1 loads", nwg, nthr, R);
        run_flag<0>("flag barrier only", nwg, nthr, R);
        run_flag<1>("flag barrier + unique-address exchange", nwg, nthr, R);
        run_hier<0, 1>("hierarchical (go word via memory)", nwg, nthr, R);
        run_hier<1, 1>("  + split-line exchange", nwg, nthr, R);
        run_hier<0, 2>("per-XCD count, all poll the 8 XCD flags", nwg, nthr, R);
        run_hier<1, 2>("  + split-line exchange", nwg, nthr, R);
    }
    return 0;
}
