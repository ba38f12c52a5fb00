// strided_copy_nt.hip -- does a non-temporal hint change the copy rate of the tile access pattern?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LR, int LC, int MODE> __global__ void __launch_bounds__((1 << (LR + LC)) / 16)
copy_kernel(const double* __restrict__ in_re, const double* __restrict__ in_im, double* __restrict__ out_re,
            double* __restrict__ out_im, unsigned log_s, unsigned tiles) {
    constexpr int COLS = 1 << LC, M = (1 << LR) / 16;
    const int tid = threadIdx.x, col = tid & (COLS - 1), tau = tid >> LC;
    const unsigned chunk = tiles >> 3;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned tile = (t & 7u) * chunk + (t >> 3);
        const unsigned g = (tile << LC) + col;
        const size_t base = ((size_t)(g >> log_s) << (log_s + LR)) | (g & ((1u << log_s) - 1u));
        double r[16], m[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t off = base + ((size_t)(j * M + tau) << log_s);
            if (MODE & 1) { r[j] = __builtin_nontemporal_load(in_re + off); m[j] = __builtin_nontemporal_load(in_im + off); }
            else { r[j] = in_re[off]; m[j] = in_im[off]; }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t off = base + ((size_t)(j * M + tau) << log_s);
            if (MODE & 2) { __builtin_nontemporal_store(r[j] * 1.0000001, out_re + off); __builtin_nontemporal_store(m[j] * 1.0000001, out_im + off); }
            else { out_re[off] = r[j] * 1.0000001; out_im[off] = m[j] * 1.0000001; }
        }
    }
}

template <int LR, int LC, int MODE> void run(double* a, double* b, double* c, double* d, size_t n, unsigned log_s, int wg, bool inplace) {
    constexpr int NT = (1 << (LR + LC)) / 16;
    const unsigned tiles = (unsigned)(n >> (LR + LC));
    unsigned grid = 256u * wg; if (grid > tiles) grid = tiles;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double* o1 = inplace ? a : c; double* o2 = inplace ? b : d;
    hipLaunchKernelGGL((copy_kernel<LR, LC, MODE>), dim3(grid), dim3(NT), 0, 0, a, b, o1, o2, log_s, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_kernel<LR, LC, MODE>), dim3(grid), dim3(NT), 0, 0, a, b, o1, o2, log_s, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("rows=2^%d cols=%d (%3d B) stride=2^%u %s nt_load=%d nt_store=%d wg/cu=%d: %.3f ms %.0f GB/s\n", LR, 1 << LC, 8 << LC,
           log_s, inplace ? "in-place " : "out-place", MODE & 1, (MODE >> 1) & 1, wg, ms, 32.0 * n / ms / 1e6);
}

int main() {
    const size_t n = (size_t)1 << 27;
    double *a, *b, *c, *d;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&d, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
#define ALL(LR, LC, LS, WG, IP) run<LR, LC, 0>(a, b, c, d, n, LS, WG, IP); run<LR, LC, 1>(a, b, c, d, n, LS, WG, IP); \
    run<LR, LC, 2>(a, b, c, d, n, LS, WG, IP); run<LR, LC, 3>(a, b, c, d, n, LS, WG, IP);
    ALL(10, 3, 10, 2, false) ALL(10, 3, 10, 2, true) ALL(9, 4, 9, 2, true) ALL(7, 5, 7, 4, true) ALL(8, 5, 18, 2, true) ALL(9, 3, 17, 4, false)
    return 0;
}
