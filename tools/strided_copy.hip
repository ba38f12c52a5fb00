// strided_copy.hip -- microbenchmark: what does the MI355X memory system give a pure copy with the tile
// kernels' access pattern (ROWS rows at a power-of-two stride, COLS contiguous f64 per row)?
// Separates "the pattern is slow" from "the FFT kernel is slow".   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LR, int LC, int PER> __global__ void __launch_bounds__((1 << (LR + LC)) / PER)
copy_kernel(const double* __restrict__ in_re, const double* __restrict__ in_im, double* __restrict__ out_re,
            double* __restrict__ out_im, unsigned log_s, unsigned tiles, int mode) {
    constexpr int ROWS = 1 << LR, COLS = 1 << LC, NT = ROWS * COLS / PER, M = ROWS / PER;
    const int tid = threadIdx.x, col = tid & (COLS - 1), tau = tid >> LC;
    const unsigned chunk = tiles >> 3;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned tile = (mode & 1) ? t : (t & 7u) * chunk + (t >> 3);
        const unsigned g = (tile << LC) + col;
        const size_t base = ((size_t)(g >> log_s) << (log_s + LR)) | (g & ((1u << log_s) - 1u));
        double r[PER], m[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const size_t off = base + ((size_t)(j * M + tau) << log_s);
            r[j] = in_re[off];
            m[j] = in_im[off];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const size_t off = base + ((size_t)(j * M + tau) << log_s);
            out_re[off] = r[j] * 1.0000001;
            out_im[off] = m[j] * 1.0000001;
        }
    }
}

template <int LR, int LC, int PER> float run(double* a, double* b, double* c, double* d, size_t n, unsigned log_s,
                                             int wg_per_cu, int mode, int reps) {
    constexpr int NT = (1 << (LR + LC)) / PER;
    const unsigned tiles = (unsigned)(n >> (LR + LC));
    unsigned grid = 256u * wg_per_cu;
    if (grid > tiles) grid = tiles;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((copy_kernel<LR, LC, PER>), dim3(grid), dim3(NT), 0, 0, a, b, c, d, log_s, tiles, mode);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((copy_kernel<LR, LC, PER>), dim3(grid), dim3(NT), 0, 0, a, b, c, d, log_s, tiles, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t n = (size_t)1 << 27;  // 1 GiB per plane: 4 GiB of traffic per pass
    double *a, *b, *c, *d;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&d, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    const double bytes = 32.0 * n;
#define RUN(LR, LC, PER, LS, WG, MODE)                                                                     \
    {                                                                                                      \
        float ms = run<LR, LC, PER>(a, b, c, d, n, LS, WG, MODE, 3);                                       \
        printf("rows=2^%d cols=%d (%d B seg) per_thread=%d stride=2^%u el wg/cu=%d %s: %.3f ms  %.0f GB/s\n", LR, \
               1 << LC, 8 << LC, PER, (unsigned)LS, WG, MODE ? "linear-order" : "xcd-order", ms, bytes / ms / 1e6); \
    }
    for (int wg = 2; wg <= 8; wg *= 2) {
        RUN(10, 3, 16, 10, wg, 0)   // 2^20-style pass B: stride 8 KiB, 64 B segments
        RUN(10, 4, 16, 10, wg, 0)   // 128 B
        RUN(10, 5, 16, 10, wg, 0)   // 256 B
        RUN(7, 5, 16, 7, wg, 0)     // (7,7,6) pass B: stride 1 KiB, 256 B
        RUN(9, 3, 16, 17, wg, 0)    // 2^26 pass A: stride 1 MiB, 64 B
        RUN(9, 4, 16, 17, wg, 0)
        RUN(9, 5, 16, 17, wg, 0)
        RUN(10, 3, 16, 10, wg, 1)   // same as first, linear tile order
    }
    RUN(10, 3, 8, 10, 4, 0)
    RUN(10, 3, 4, 10, 8, 0)
    RUN(10, 2, 16, 10, 4, 0)
    return 0;
}
