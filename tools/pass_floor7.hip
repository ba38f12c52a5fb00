// pass_floor7.hip -- staggered waves: the copy-with-arithmetic model of tools/pass_floor6.hip (one 64 x 16 tile per wave,
// four waves per workgroup, R rounds of 32 f64 FMAs between loads and stores), with a fraction of the waves put to
// sleep before they issue their loads, so that the early group's arithmetic and stores overlap the late group's loads
// instead of every wave waiting for HBM, computing and storing in step.  us per pass, HIP graph on a cold ring.
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

// GROUPS = 1: no stagger; 2: odd waves sleep `units` x 64 cycles; 4: wave w sleeps w x units x 64 cycles.
// BYBLOCK: the group is chosen per workgroup (block index) instead of per wave.
template <int R, int GROUPS, int BYBLOCK>
__global__ void __launch_bounds__(256) one(const double *ir, const double *ii, double *orr, double *oi, double c, double d, int units) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x, blocks = gridDim.x;
    const unsigned bb = (b & 7u) * (blocks >> 3) + (b >> 3), tile = bb * 4 + wave, col = lane & 15, tau = lane >> 4;
    const unsigned g = (BYBLOCK ? (b >> 3) : wave) & (GROUPS - 1);
    for (unsigned k = 0; k < g * (unsigned)units; ++k) __builtin_amdgcn_s_sleep(1);
    double r[16], m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { r[j] = ld(ir + off(tile, j, tau, col)); m[j] = ld(ii + off(tile, j, tau, col)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    work<R>(r, m, c, d);
#pragma unroll
    for (int j = 0; j < 16; ++j) { st(orr + off(tile, j, tau, col), r[j]); st(oi + off(tile, j, tau, col), m[j]); }
}
typedef void (*K)(const double *, const double *, double *, double *, double, double, int);
static float run(K k, int units, const double *in, double *out, size_t n, int RING, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < RING; ++i) { const double *x = in + (size_t)i * 2 * n; double *y = out + (size_t)i * 2 * n; hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, x, x + n, y, y + n, 1.0000001, 1e-9, units); }
    (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1); float t; (void)hipEventElapsedTime(&t, e0, e1); best = t < best ? t : best; }
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 1e3f * best / RING;
}
template <int R> void row(const double *in, double *out, size_t n, int RING, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    printf("R = %2d: none %6.2f |", R, run(one<R, 1, 0>, 0, in, out, n, RING, s, e0, e1));
    const int us[5] = {8, 16, 24, 32, 48};   // x 64 cycles: 0.2 .. 1.3 us at 2.4 GHz
    printf(" 2 groups by wave:");
    for (int u : us) printf(" %6.2f", run(one<R, 2, 0>, u, in, out, n, RING, s, e0, e1));
    printf(" | 4 groups by wave (step):");
    for (int u : us) printf(" %6.2f", run(one<R, 4, 0>, u / 2, in, out, n, RING, s, e0, e1));
    printf(" | 2 groups by block:");
    for (int u : us) printf(" %6.2f", run(one<R, 2, 1>, u, in, out, n, RING, s, e0, e1));
    printf("   us (sleep units 8 16 24 32 48 x 64 cycles)\n"); fflush(stdout);
}
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    row<0>(in, out, n, RING, s, e0, e1); row<10>(in, out, n, RING, s, e0, e1); row<15>(in, out, n, RING, s, e0, e1); row<20>(in, out, n, RING, s, e0, e1);
    return 0;
}
