// pass_floor9.hip -- EXECUTION time (start/stop events bound to the dispatch, as bench.py's pass_ms and rocprofv3 report it)
// of the wave-tile copy kernel (tools/pass_floor7.hip, R rounds of FMAs, optional stagger) against its per-launch time in
// a HIP graph of independent launches, on a cold ring: what part of a graph's per-launch time is the kernel itself?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned LOG_N = 20;
__device__ inline double ld(const double *p) { double v; asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ inline void st(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory"); }
__device__ inline size_t off(unsigned tile, unsigned j, unsigned tau, unsigned col) { return ((size_t)(j * 4 + tau) << 14) + tile * 16 + col; }
template <int R>
__global__ void __launch_bounds__(256) one(const double *ir, const double *ii, double *orr, double *oi, double c, double d, int units, int mask) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x, blocks = gridDim.x;
    const unsigned bb = (b & 7u) * (blocks >> 3) + (b >> 3), tile = bb * 4 + wave, col = lane & 15, tau = lane >> 4;
    for (unsigned k = (wave & (unsigned)mask) * (unsigned)units; k > 0; --k) __builtin_amdgcn_s_sleep(1);
    double r[16], m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { r[j] = ld(ir + off(tile, j, tau, col)); m[j] = ld(ii + off(tile, j, tau, col)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int k = 0; k < R; ++k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { r[j] = __builtin_fma(r[j], c, d); m[j] = __builtin_fma(m[j], c, d); }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { st(orr + off(tile, j, tau, col), r[j]); st(oi + off(tile, j, tau, col), m[j]); }
}
template <int R> int row(const double *in, double *out, size_t n, int RING, hipStream_t s, int units, int mask) {
    std::vector<hipEvent_t> ev(2 * RING);
    for (auto &e : ev) CK(hipEventCreate(&e));
    float best_exec = 1e9f, best_wall = 1e9f;
    hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(w0, s));
        for (int i = 0; i < RING; ++i) {
            const double *x = in + (size_t)i * 2 * n; double *y = out + (size_t)i * 2 * n;
            hipExtLaunchKernelGGL(one<R>, dim3(256), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, x, x + n, y, y + n, 1.0000001, 1e-9, units, mask);
        }
        CK(hipEventRecord(w1, s)); CK(hipEventSynchronize(w1));
        std::vector<float> ts;
        for (int i = 0; i < RING; ++i) { float t; CK(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1])); ts.push_back(t); }
        std::sort(ts.begin(), ts.end());
        best_exec = std::min(best_exec, 1e3f * ts[RING / 2]);
        float t; CK(hipEventElapsedTime(&t, w0, w1)); best_wall = std::min(best_wall, 1e3f * t / RING);
    }
    // the same launches from a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < RING; ++i) { const double *x = in + (size_t)i * 2 * n; double *y = out + (size_t)i * 2 * n; hipLaunchKernelGGL(one<R>, dim3(256), dim3(256), 0, s, x, x + n, y, y + n, 1.0000001, 1e-9, units, mask); }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best_graph = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(w0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(w1, s)); CK(hipEventSynchronize(w1)); float t; CK(hipEventElapsedTime(&t, w0, w1)); best_graph = std::min(best_graph, 1e3f * t / RING); }
    printf("R = %2d stagger %2d,%d: execution (median of %d dispatches) %6.2f us | eager back-to-back per launch %6.2f us | graph per launch %6.2f us\n", R, units, mask, RING, best_exec, best_wall, best_graph);
    fflush(stdout);
    return 0;
}
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    row<0>(in, out, n, RING, s, 0, 0); row<0>(in, out, n, RING, s, 6, 3);
    row<10>(in, out, n, RING, s, 0, 0); row<10>(in, out, n, RING, s, 6, 3);
    row<15>(in, out, n, RING, s, 0, 0); row<15>(in, out, n, RING, s, 6, 3);
    return 0;
}
