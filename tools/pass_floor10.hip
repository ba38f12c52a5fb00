// pass_floor10.hip -- what does a kernel boundary cost in a graph-replayed chain of three dependent wave-tile copy
// kernels, and does it depend on the launch's dynamic LDS request or on the size of the kernel-argument block?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned LOG_N = 20;
struct Big { const double *ir, *ii; double *orr, *oi; double c, d; unsigned long long pad[20]; };
__device__ inline double ld(const double *p) { double v; asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ inline void st(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory"); }
__device__ inline size_t off(unsigned tile, unsigned j, unsigned tau, unsigned col) { return ((size_t)(j * 4 + tau) << 14) + tile * 16 + col; }
__global__ void __launch_bounds__(256) k(const Big a) {
    extern __shared__ double dyn[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x, blocks = gridDim.x;
    const unsigned bb = (b & 7u) * (blocks >> 3) + (b >> 3), tile = bb * 4 + wave, col = lane & 15, tau = lane >> 4;
    double r[16], m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { r[j] = ld(a.ir + off(tile, j, tau, col)); m[j] = ld(a.ii + off(tile, j, tau, col)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.pad[19] == 12345) dyn[threadIdx.x] = r[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) { st(a.orr + off(tile, j, tau, col), r[j] + a.c); st(a.oi + off(tile, j, tau, col), m[j] + a.d); }
}
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out, *tmp; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8)); CK(hipMalloc(&tmp, 4 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8)); CK(hipMemset(tmp, 0, 4 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    for (size_t lds : {(size_t)0, (size_t)2048, (size_t)69 * 1024}) {
        float res[2];
        for (int chain = 0; chain < 2; ++chain) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < RING; ++i) {
                const double *x = in + (size_t)i * 2 * n; double *y = out + (size_t)i * 2 * n;
                Big a{}; a.c = 1.0; a.d = 1.0;
                if (!chain) { a.ir = x; a.ii = x + n; a.orr = y; a.oi = y + n; hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, s, a); }
                else {
                    a.ir = x; a.ii = x + n; a.orr = tmp; a.oi = tmp + n; hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, s, a);
                    a.ir = tmp; a.ii = tmp + n; a.orr = tmp + 2 * n; a.oi = tmp + 3 * n; hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, s, a);
                    a.ir = tmp + 2 * n; a.ii = tmp + 3 * n; a.orr = y; a.oi = y + n; hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, s, a);
                }
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best; }
            res[chain] = 1e3f * best / RING;
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
        printf("208-byte argument block, dynamic LDS %6zu B: one pass %6.2f us   3-pass chain %6.2f us\n", lds, res[0], res[1]); fflush(stdout);
    }
    return 0;
}
