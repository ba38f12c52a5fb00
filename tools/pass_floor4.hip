// pass_floor4.hip -- copy floor of the THREE access patterns of the (6,8,6) plan for one 2^20-point f64 transform:
//   A: rows 2^14 apart (64 rows x 16 cols), stores = 16 contiguous runs of 64 elements
//   B: rows 2^6 apart inside 2^14-element blocks (256 rows x 16 cols: four tiles interleave in every 512-byte row)
//   C: rows 2^14 apart (64 rows x 16 cols), same pattern out
// each as its own kernel (HIP graph of 48 launches on a cold ring), 256 threads x 16 values or 512 x 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr unsigned LOG_N = 20;
// pattern: 0 = A-in/C (64 rows, stride 2^14), 1 = B (256 rows, stride 2^6 within blocks of 2^14), 2 = A-out (runs of 64)
template <int NT, int P, int PAT> __device__ inline size_t addr(unsigned tile, unsigned e) {
    if (PAT == 0) { const unsigned row = e >> 4, col = e & 15; return ((size_t)row << 14) + tile * 16 + col; }
    if (PAT == 1) { const unsigned row = e >> 4, col = e & 15, g = tile * 16 + col; return ((size_t)(g >> 6) << 14) | ((size_t)row << 6) | (g & 63); }
    const unsigned c = e >> 6, k = e & 63; return ((size_t)(tile * 16 + c) << 6) + k;
}
template <int NT, int P, int RD, int WR>
__global__ void __launch_bounds__(NT) k(const double *__restrict__ ir, const double *__restrict__ ii, double *__restrict__ orr, double *__restrict__ oi, unsigned tiles) {
    const unsigned t = blockIdx.x, tile = (t & 7u) * (tiles >> 3) + (t >> 3);
    double r[P], m[P];
#pragma unroll
    for (int j = 0; j < P; ++j) { const size_t o = addr<NT, P, RD>(tile, j * NT + threadIdx.x); r[j] = __builtin_nontemporal_load(ir + o); m[j] = __builtin_nontemporal_load(ii + o); }
#pragma unroll
    for (int j = 0; j < P; ++j) { const size_t o = addr<NT, P, WR>(tile, j * NT + threadIdx.x); __builtin_nontemporal_store(r[j] + 1.0, orr + o); __builtin_nontemporal_store(m[j] + 1.0, oi + o); }
}
typedef void (*L)(const double *, double *, size_t, hipStream_t);
template <int NT, int P, int RD, int WR> void launch(const double *x, double *y, size_t n, hipStream_t s) {
    constexpr unsigned tiles = (1u << LOG_N) / (NT * P);
    hipLaunchKernelGGL((k<NT, P, RD, WR>), dim3(tiles), dim3(NT), 0, s, x, x + n, y, y + n, tiles);
}
int main() {
    const size_t n = (size_t)1 << LOG_N; const int RING = 48;
    double *in, *out; CK(hipMalloc(&in, RING * 2 * n * 8)); CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8)); CK(hipMemset(out, 0, RING * 2 * n * 8));
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char *name; L f; };
    const V vs[] = {{"A: 64x16 in (stride 2^14), runs out   64thr x p16", launch<64, 16, 0, 2>}, {"A:                                   128thr x p8 ", launch<128, 8, 0, 2>},
                    {"C: 64x16 in/out (stride 2^14)          64thr x p16", launch<64, 16, 0, 0>}, {"C:                                   128thr x p8 ", launch<128, 8, 0, 0>},
                    {"B: 256x16 in/out (stride 2^6)         256thr x p16", launch<256, 16, 1, 1>}, {"B:                                   512thr x p8 ", launch<512, 8, 1, 1>}};
    for (const V &v : vs) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < RING; ++i) v.f(in + (size_t)i * 2 * n, out + (size_t)i * 2 * n, n, s);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best; }
        printf("%s: %6.2f us\n", v.name, 1e3 * best / RING); fflush(stdout);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
