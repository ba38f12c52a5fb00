// pass_floor8.hip -- copy model of a 128-row wave tile: one wave per 128 x 16 tile of a 2^20-point f64 transform (512 waves,
// 32 points per lane, 64 loads in flight per lane), R rounds of 64 f64 FMAs per tile between loads and stores, waves
// staggered as tools/pass_floor7.hip; WPB waves per workgroup.  Compare with pass_floor7's 64-row tiles (1024 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned LOG_N = 20;
__device__ inline double ld(const double *p) { double v; asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ inline void st(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory"); }
// rows 4 j + tau, j < 32, of a 128-row tile: row stride 2^13 elements (N / 128)
__device__ inline size_t off(unsigned tile, unsigned j, unsigned tau, unsigned col) { return ((size_t)(j * 4 + tau) << 13) + tile * 16 + col; }
template <int R, int WPB>
__global__ void __launch_bounds__(64 * WPB) k(const double *ir, const double *ii, double *orr, double *oi, double c, double d, int units, int mask) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x, blocks = gridDim.x;
    const unsigned bb = (b & 7u) * (blocks >> 3) + (b >> 3), tile = bb * WPB + wave, col = lane & 15, tau = lane >> 4;
    const unsigned g = (((b >> 3) & 1u) * WPB + wave) & (unsigned)mask;
    for (unsigned i = 0; i < g * (unsigned)units; ++i) __builtin_amdgcn_s_sleep(1);
    double r[32], m[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { r[j] = ld(ir + off(tile, j, tau, col)); m[j] = ld(ii + off(tile, j, tau, col)); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int q = 0; q < R; ++q) {
#pragma unroll
        for (int j = 0; j < 32; ++j) { r[j] = __builtin_fma(r[j], c, d); m[j] = __builtin_fma(m[j], c, d); }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) { st(orr + off(tile, j, tau, col), r[j]); st(oi + off(tile, j, tau, col), m[j]); }
}
typedef void (*K)(const double *, const double *, double *, double *, double, double, int, int);
static float run(K kk, int wpb, int units, int mask, const double *in, double *out, size_t n, int RING, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < RING; ++i) { const double *x = in + (size_t)i * 2 * n; double *y = out + (size_t)i * 2 * n; hipLaunchKernelGGL(kk, dim3(512 / wpb), dim3(64 * wpb), 0, s, x, x + n, y, y + n, 1.0000001, 1e-9, units, mask); }
    (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1); float t; (void)hipEventElapsedTime(&t, e0, e1); best = t < best ? t : best; }
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 1e3f * best / RING;
}
template <int R> void row(const double *in, double *out, size_t n, int RING, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    printf("R = %2d:", R);
    const int us[5] = {0, 6, 12, 18, 24};
    printf(" 1 wave/wg, mask 1:");
    for (int u : us) printf(" %6.2f", run(k<R, 1>, 1, u, 1, in, out, n, RING, s, e0, e1));
    printf(" | 2 waves/wg, mask 1:");
    for (int u : us) printf(" %6.2f", run(k<R, 2>, 2, u, 1, in, out, n, RING, s, e0, e1));
    printf(" | 2 waves/wg, mask 3:");
    for (int u : us) printf(" %6.2f", run(k<R, 2>, 2, u, 3, in, out, n, RING, s, e0, e1));
    printf(" | 4 waves/wg, mask 3:");
    for (int u : us) printf(" %6.2f", run(k<R, 4>, 4, u, 3, in, out, n, RING, s, e0, e1));
    printf("   us (sleep units 0 6 12 18 24 x 64 cycles per group step)\n"); fflush(stdout);
}
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    row<0>(in, out, n, RING, s, e0, e1); row<10>(in, out, n, RING, s, e0, e1); row<15>(in, out, n, RING, s, e0, e1); row<20>(in, out, n, RING, s, e0, e1);
    return 0;
}
