// pass_floor5.hip -- cache-policy bits on the loads and stores of a one-pass copy of a 2^20-point f64 transform
// (64 rows x 16 columns per wave, the pass-C pattern), gfx950: every combination of {plain, nt, sc1, sc0 sc1, sc1 nt,
// sc0 sc1 nt} on the loads and on the stores; one pass and a chain of three (HIP graph, cold ring), us.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned LOG_N = 20;
#define LOADV(POL) asm volatile("global_load_dwordx2 %0, %1, off " POL : "=v"(v) : "v"(p) : "memory")
#define STOREV(POL) asm volatile("global_store_dwordx2 %0, %1, off " POL : : "v"(p), "v"(v) : "memory")
template <int LP> __device__ inline double ld(const double *p) {
    double v;
    if (LP == 0) LOADV("");
    else if (LP == 1) LOADV("nt");
    else if (LP == 2) LOADV("sc1");
    else if (LP == 3) LOADV("sc0 sc1");
    else if (LP == 4) LOADV("sc1 nt");
    else LOADV("sc0 sc1 nt");
    return v;
}
template <int SP> __device__ inline void st(double *p, double v) {
    if (SP == 0) STOREV("");
    else if (SP == 1) STOREV("nt");
    else if (SP == 2) STOREV("sc1");
    else if (SP == 3) STOREV("sc0 sc1");
    else if (SP == 4) STOREV("sc1 nt");
    else STOREV("sc0 sc1 nt");
}
template <int LP, int SP>
__global__ void __launch_bounds__(256) k(const double *ir, const double *ii, double *orr, double *oi, unsigned tiles) {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x;
    const unsigned blocks = tiles >> 2, bb = (b & 7u) * (blocks >> 3) + (b >> 3), tile = bb * 4 + wave;
    const unsigned col = lane & 15, tau = lane >> 4;
    double r[16], m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { const size_t o = ((size_t)(j * 4 + tau) << 14) + tile * 16 + col; r[j] = ld<LP>(ir + o); m[j] = ld<LP>(ii + o); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) { const size_t o = ((size_t)(j * 4 + tau) << 14) + tile * 16 + col; st<SP>(orr + o, r[j] + 1.0); st<SP>(oi + o, m[j] + 1.0); }
}
typedef void (*L)(const double *, double *, size_t, hipStream_t);
template <int LP, int SP> void launch(const double *x, double *y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL((k<LP, SP>), dim3(256), dim3(256), 0, s, x, x + n, y, y + n, 1024u);
}
template <int LP> void fill(L *t) { t[0] = launch<LP, 0>; t[1] = launch<LP, 1>; t[2] = launch<LP, 2>; t[3] = launch<LP, 3>; t[4] = launch<LP, 4>; t[5] = launch<LP, 5>; }
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out, *tmp; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8)); CK(hipMalloc(&tmp, 4 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8)); CK(hipMemset(tmp, 0, 4 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    L tab[6][6]; fill<0>(tab[0]); fill<1>(tab[1]); fill<2>(tab[2]); fill<3>(tab[3]); fill<4>(tab[4]); fill<5>(tab[5]);
    const char *names[6] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
    for (int lp = 0; lp < 6; ++lp) for (int sp = 0; sp < 6; ++sp) {
        float res[2];
        for (int chain = 0; chain < 2; ++chain) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < RING; ++i) {
                const double *x = in + (size_t)i * 2 * n; double *y = out + (size_t)i * 2 * n;
                if (!chain) tab[lp][sp](x, y, n, s);
                else { tab[lp][sp](x, tmp, n, s); tab[lp][sp](tmp, tmp + 2 * n, n, s); tab[lp][sp](tmp + 2 * n, y, n, s); }
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best; }
            res[chain] = 1e3f * best / RING;
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
        printf("loads %-10s stores %-10s: one pass %6.2f us   3-pass chain %6.2f us\n", names[lp], names[sp], res[0], res[1]); fflush(stdout);
    }
    return 0;
}
