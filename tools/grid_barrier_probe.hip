// grid_barrier_probe.hip -- what does a device-wide barrier INSIDE a kernel cost on the MI355X (256 workgroups, one per CU, eight
// XCDs with their own L2s), against the 1.55 us of a kernel boundary in a graph (tools/boundary_probe.hip)?  A persistent kernel of
// 256 workgroups x 256 threads runs R rounds of nothing but the barrier; us per barrier = kernel time / R.  Variants:
//   counter   every workgroup's thread 0: release fence, atomicAdd on ONE counter (agent scope), then polls it (acquire loads)
//   xcd       per-XCD counters first (the workgroups of an XCD meet in their own L2 line), the last arrival of each XCD adds to the
//             global counter, everybody polls the global counter
//   xcd-relaxed  the same, polling with relaxed loads and no sleep, one acquire fence after the poll
//   flags     every workgroup stores the round number to ITS flag; one wave per workgroup polls all 256 flags with one 16-byte
//             load per lane (no read-modify-write anywhere)
// All polls are BOUNDED (a workgroup that gives up sets an error word): a logic error cannot hang the GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr unsigned kSpinLimit = 200000;  // ~ tens of milliseconds of polling: far beyond any healthy barrier

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE> __global__ void __launch_bounds__(256) barrier_kernel(unsigned *ctr, unsigned *xctr, unsigned *flags, unsigned *err, unsigned rounds,
                                                                         unsigned per_xcd) {
    const unsigned nwg = gridDim.x;
    for (unsigned r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (MODE == 0) {
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r * nwg) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) { *err = 1; break; }
                }
            }
        } else if (MODE == 1) {
            if (threadIdx.x == 0) {
                const unsigned x = xcc_id();
                const unsigned old = __hip_atomic_fetch_add(xctr + 32 * x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == r * per_xcd) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r * 8u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) { *err = 2; break; }
                }
            }
        } else if (MODE == 3) {  // as `xcd`, polling with RELAXED loads (no cache invalidate per poll), no sleep, one acquire fence at the end
            if (threadIdx.x == 0) {
                const unsigned x = xcc_id();
                const unsigned old = __hip_atomic_fetch_add(xctr + 32 * x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == r * per_xcd) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r * 8u)
                    if (++spins > 100u * kSpinLimit) { *err = 4; break; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) {  // 64 lanes x 4 flags: all 256 in one load instruction per poll
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    for (unsigned i = 0; i < 4; ++i) {
                        const unsigned idx = threadIdx.x * 4 + i;
                        if (idx < nwg && __hip_atomic_load(flags + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r) ok = false;
                    }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kSpinLimit) { *err = 3; break; }
                }
            }
        }
        __syncthreads();
    }
}

template <int MODE> static int run(const char *name, unsigned *d, hipStream_t s, unsigned per_xcd) {
    unsigned *ctr = d, *xctr = d + 64, *flags = d + 1024, *err = d + 2048;
    const unsigned rounds = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    unsigned herr = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(d, 0, 4096 * sizeof(unsigned), s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(barrier_kernel<MODE>, dim3(256), dim3(256), 0, s, ctr, xctr, flags, err, rounds, per_xcd);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(&herr, err, sizeof herr, hipMemcpyDeviceToHost));
        if (herr) break;
    }
    printf("%-10s %8.3f us per barrier%s\n", name, 1e3f * best / rounds, herr ? "   (GAVE UP: a poll ran into its bound)" : "");
    return 0;
}

__global__ void xcd_census(unsigned *out) {
    if (threadIdx.x == 0) atomicAdd(out + xcc_id(), 1u);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *d;
    CK(hipMalloc(&d, 4096 * sizeof(unsigned)));
    CK(hipMemset(d, 0, 4096 * sizeof(unsigned)));
    hipLaunchKernelGGL(xcd_census, dim3(256), dim3(256), 0, s, d);
    CK(hipStreamSynchronize(s));
    unsigned census[16];
    CK(hipMemcpy(census, d, sizeof census, hipMemcpyDeviceToHost));
    printf("workgroups per XCD of a 256-workgroup launch:");
    bool even = true;
    for (int i = 0; i < 8; ++i) { printf(" %u", census[i]); even = even && census[i] == 32; }
    printf("\n");
    if (run<0>("counter", d, s, 32)) return 1;
    if (even) { if (run<1>("xcd", d, s, 32)) return 1; }
    else printf("xcd        skipped (the launch is not spread 32 per XCD)\n");
    if (even) { if (run<3>("xcd-relaxed", d, s, 32)) return 1; }
    if (run<2>("flags", d, s, 32)) return 1;
    return 0;
}
