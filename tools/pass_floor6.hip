// pass_floor6.hip -- would software pipelining inside a wave hide the arithmetic of a wave-tile pass?  One-pass "copy"
// of a 2^20-point f64 transform (64 rows x 16 columns per tile, nt loads and stores) with R rounds of 32 independent f64
// FMAs per tile between the loads and the stores (R = 10 is about the arithmetic of a real pass), two ways:
//   one tile per wave, 1024 waves (what wave_fft_kernel does): load, wait, compute, store
//   two tiles per wave, 512 waves: load both, wait for the first, compute it and store it while the second is in flight
// us per pass from a HIP graph on a cold ring.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned LOG_N = 20;
__device__ inline double ld(const double *p) { double v; asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ inline void st(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory"); }
template <int R> __device__ inline void work(double (&r)[16], double (&m)[16], double c, double d) {
#pragma unroll 1
    for (int k = 0; k < R; ++k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { r[j] = __builtin_fma(r[j], c, d); m[j] = __builtin_fma(m[j], c, d); }
    }
}
__device__ inline size_t off(unsigned tile, unsigned j, unsigned tau, unsigned col) { return ((size_t)(j * 4 + tau) << 14) + tile * 16 + col; }

template <int R, int WPB>
__global__ void __launch_bounds__(64 * WPB) one(const double *ir, const double *ii, double *orr, double *oi, double c, double d) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x, blocks = gridDim.x;
    const unsigned bb = (b & 7u) * (blocks >> 3) + (b >> 3), tile = bb * WPB + wave, col = lane & 15, tau = lane >> 4;
    double r[16], m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { r[j] = ld(ir + off(tile, j, tau, col)); m[j] = ld(ii + off(tile, j, tau, col)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    work<R>(r, m, c, d);
#pragma unroll
    for (int j = 0; j < 16; ++j) { st(orr + off(tile, j, tau, col), r[j]); st(oi + off(tile, j, tau, col), m[j]); }
}

// two tiles per wave; ADJ = 1: tiles 2t, 2t+1 (256-byte rows together); ADJ = 0: tiles t and t + 512
template <int R, int WPB, int ADJ>
__global__ void __launch_bounds__(64 * WPB) two(const double *ir, const double *ii, double *orr, double *oi, double c, double d) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x, blocks = gridDim.x;
    const unsigned bb = (b & 7u) * (blocks >> 3) + (b >> 3), w = bb * WPB + wave, col = lane & 15, tau = lane >> 4;
    const unsigned t0 = ADJ ? 2 * w : w, t1 = ADJ ? 2 * w + 1 : w + 512;
    double r0[16], m0[16], r1[16], m1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { r0[j] = ld(ir + off(t0, j, tau, col)); m0[j] = ld(ii + off(t0, j, tau, col)); }
#pragma unroll
    for (int j = 0; j < 16; ++j) { r1[j] = ld(ir + off(t1, j, tau, col)); m1[j] = ld(ii + off(t1, j, tau, col)); }
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    work<R>(r0, m0, c, d);
#pragma unroll
    for (int j = 0; j < 16; ++j) { st(orr + off(t0, j, tau, col), r0[j]); st(oi + off(t0, j, tau, col), m0[j]); }
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // the 32 stores just issued may still be out; the second tile's loads are older
    work<R>(r1, m1, c, d);
#pragma unroll
    for (int j = 0; j < 16; ++j) { st(orr + off(t1, j, tau, col), r1[j]); st(oi + off(t1, j, tau, col), m1[j]); }
}

typedef void (*L)(const double *, double *, size_t, hipStream_t);
template <int R, int WPB> void l_one(const double *x, double *y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL((one<R, WPB>), dim3(1024 / WPB), dim3(64 * WPB), 0, s, x, x + n, y, y + n, 1.0000001, 1e-9);
}
template <int R, int WPB, int ADJ> void l_two(const double *x, double *y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL((two<R, WPB, ADJ>), dim3(512 / WPB), dim3(64 * WPB), 0, s, x, x + n, y, y + n, 1.0000001, 1e-9);
}
static float run(L f, const double *in, double *out, size_t n, int RING, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < RING; ++i) f(in + (size_t)i * 2 * n, out + (size_t)i * 2 * n, n, s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); best = t < best ? t : best; }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 1e3f * best / RING;
}
template <int R> int row(const double *in, double *out, size_t n, int RING, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    printf("R = %2d FMA rounds: 1 tile/wave x4 %6.2f  x2 %6.2f  x1 %6.2f | 2 tiles/wave (t, t+512) x4 %6.2f  x2 %6.2f  x1 %6.2f | (2t, 2t+1) x4 %6.2f  x2 %6.2f  x1 %6.2f us\n", R,
           run(l_one<R, 4>, in, out, n, RING, s, e0, e1), run(l_one<R, 2>, in, out, n, RING, s, e0, e1), run(l_one<R, 1>, in, out, n, RING, s, e0, e1),
           run(l_two<R, 4, 0>, in, out, n, RING, s, e0, e1), run(l_two<R, 2, 0>, in, out, n, RING, s, e0, e1), run(l_two<R, 1, 0>, in, out, n, RING, s, e0, e1),
           run(l_two<R, 4, 1>, in, out, n, RING, s, e0, e1), run(l_two<R, 2, 1>, in, out, n, RING, s, e0, e1), run(l_two<R, 1, 1>, in, out, n, RING, s, e0, e1));
    fflush(stdout);
    return 0;
}
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    row<0>(in, out, n, RING, s, e0, e1); row<5>(in, out, n, RING, s, e0, e1); row<10>(in, out, n, RING, s, e0, e1);
    row<15>(in, out, n, RING, s, e0, e1); row<20>(in, out, n, RING, s, e0, e1);
    return 0;
}
