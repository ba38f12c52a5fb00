// boundary_probe.hip -- what does a KERNEL BOUNDARY cost inside a HIP graph on this box, and what does it depend on?
// The single 2^20 transform is three dependent kernels; bench.py reads its step as the sum of the three kernels' own event times
// plus 0.5 us on some boxes and plus 1.7 us on others (profiles/README.md, round 6).  This program captures chains of K dependent
// launches of one kernel into a graph, replays it and reports us per launch for:
//   empty      256 workgroups x 256 threads that return at once              -> the boundary itself (launch + drain + fences)
//   args       the same with a 232-byte by-value argument (TileArgs' size)
//   lds        the same with 64 KiB of dynamic LDS
//   dirty N    each workgroup stores N KiB (plain / non-temporal stores)      -> is the gap the write-back of the previous kernel?
//   stream     a 32 MiB copy (16 in, 16 out: one pass of the transform without arithmetic), in place / between two buffers
//   hipcc --offload-arch=gfx950 -O3 tools/boundary_probe.hip -o tools/boundary_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { unsigned long long v[29]; };  // 232 bytes

__global__ void __launch_bounds__(256) k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void __launch_bounds__(256) k_args(Big b, int *p) { if (p && b.v[3] == 12345 && threadIdx.x == 9999) *p = 1; }
__global__ void __launch_bounds__(256) k_lds(int *p) {
    extern __shared__ int sm[];
    if (p && threadIdx.x == 9999) { sm[0] = 1; *p = sm[0]; }
}
template <bool NT> __global__ void __launch_bounds__(256) k_dirty(double *buf, unsigned doubles_per_thread) {
    double *q = buf + ((size_t)blockIdx.x * 256 + threadIdx.x);
    for (unsigned i = 0; i < doubles_per_thread; ++i) {
        if (NT) __builtin_nontemporal_store(1.0, q + (size_t)i * 65536);
        else q[(size_t)i * 65536] = 1.0;
    }
}
typedef double d2 __attribute__((ext_vector_type(2)));
template <int PER> __global__ void __launch_bounds__(256) k_stream(const d2 *in, d2 *out) {  // (PER a constant: v[] stays in registers)
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    d2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = __builtin_nontemporal_load(in + t + (size_t)i * 65536);
#pragma unroll
    for (int i = 0; i < PER; ++i) __builtin_nontemporal_store(v[i], out + t + (size_t)i * 65536);
}

template <typename F> static int timed(const char *name, int K, hipStream_t s, F launch) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s K=%4d: %7.3f us per launch\n", name, K, 1e3f * best / K);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *a, *b;
    const size_t bytes = (size_t)64 << 20;
    CK(hipMalloc(&a, bytes * 8));
    CK(hipMalloc(&b, bytes * 8));
    CK(hipMemset(a, 0, bytes * 8));
    CK(hipMemset(b, 0, bytes * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
    Big big{};
    for (int K : {60, 600}) {
        if (timed("empty 256 x 256", K, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, (int *)nullptr); })) return 1;
        if (timed("empty 1024 x 64", K, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, s, (int *)nullptr); })) return 1;
        if (timed("232-byte argument", K, s, [&](int) { hipLaunchKernelGGL(k_args, dim3(256), dim3(256), 0, s, big, (int *)nullptr); })) return 1;
        if (timed("64 KiB dynamic LDS", K, s, [&](int) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 64 << 10, s, (int *)nullptr); })) return 1;
        for (unsigned kib : {1u, 16u, 64u}) {   // per workgroup: 256 threads x 8 B x n
            char nm[64];
            snprintf(nm, sizeof nm, "dirty %3u KiB/wg plain stores", 2 * kib);
            if (timed(nm, K, s, [&](int i) { hipLaunchKernelGGL(k_dirty<false>, dim3(256), dim3(256), 0, s, a + (size_t)(i % 16) * (4 << 20), kib); })) return 1;
            snprintf(nm, sizeof nm, "dirty %3u KiB/wg nt stores", 2 * kib);
            if (timed(nm, K, s, [&](int i) { hipLaunchKernelGGL(k_dirty<true>, dim3(256), dim3(256), 0, s, a + (size_t)(i % 16) * (4 << 20), kib); })) return 1;
        }
        // one "pass": 16 MiB in, 16 MiB out, every step on another 32 MiB of a 512 MiB ring (cold)
        if (timed("stream 16+16 MiB, a -> b", K, s, [&](int i) {
                hipLaunchKernelGGL(k_stream<16>, dim3(256), dim3(256), 0, s, (const d2 *)(a + (size_t)(i % 16) * (2 << 20) * 2), (d2 *)(b + (size_t)(i % 16) * (2 << 20) * 2));
            })) return 1;
        if (timed("stream 16+16 MiB, in place", K, s, [&](int i) {
                hipLaunchKernelGGL(k_stream<16>, dim3(256), dim3(256), 0, s, (const d2 *)(a + (size_t)(i % 16) * (2 << 20) * 2), (d2 *)(a + (size_t)(i % 16) * (2 << 20) * 2));
            })) return 1;
    }
    return 0;
}
