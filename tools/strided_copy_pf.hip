// strided_copy_pf.hip -- can a wide "warming" read make the narrow-row tile pattern run at the wide-row rate?
// A workgroup owns GROUP adjacent 1024 x 8 tiles (together 1024 x 8*GROUP columns).  Variant PF first reads the
// whole wide region with full-width rows (data discarded: it only pulls the lines into L2 / Infinity Cache), then
// copies the GROUP narrow tiles one after the other exactly like the FFT pass would.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LR, int LC, int GROUP, bool PF> __global__ void __launch_bounds__((1 << (LR + LC)) / 16)
copy_kernel(const double* __restrict__ in_re, const double* __restrict__ in_im, double* __restrict__ out_re,
            double* __restrict__ out_im, unsigned log_s, unsigned groups, double* sink) {
    constexpr int ROWS = 1 << LR, COLS = 1 << LC, NT = ROWS * COLS / 16, M = ROWS / 16, WIDE = COLS * GROUP;
    const int tid = threadIdx.x, col = tid & (COLS - 1), tau = tid >> LC;
    const unsigned chunk = groups >> 3;
    for (unsigned t = blockIdx.x; t < groups; t += gridDim.x) {
        const unsigned grp = (t & 7u) * chunk + (t >> 3);
        const unsigned g0 = grp * WIDE;
        const size_t base0 = ((size_t)(g0 >> log_s) << (log_s + LR)) | (g0 & ((1u << log_s) - 1u));
        if (PF) {
            double acc = 0;
            const int wc = tid % WIDE, wr0 = tid / WIDE;  // full-width rows: WIDE*8 contiguous bytes per row
            for (int row = wr0; row < ROWS; row += NT / WIDE) {
                acc += in_re[base0 + ((size_t)row << log_s) + wc];
                acc += in_im[base0 + ((size_t)row << log_s) + wc];
            }
            if (acc == 1.2345e300) sink[0] = acc;
        }
        for (int sub = 0; sub < GROUP; ++sub) {
            const size_t base = base0 + sub * COLS + col;
            double r[16], m[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const size_t off = base + ((size_t)(j * M + tau) << log_s);
                r[j] = in_re[off];
                m[j] = in_im[off];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const size_t off = base + ((size_t)(j * M + tau) << log_s);
                out_re[off] = r[j] * 1.0000001;
                out_im[off] = m[j] * 1.0000001;
            }
        }
    }
}

template <int LR, int LC, int GROUP, bool PF> void run(double* a, double* b, double* c, double* d, size_t n, unsigned log_s, int wg) {
    constexpr int NT = (1 << (LR + LC)) / 16;
    const unsigned groups = (unsigned)(n >> (LR + LC)) / GROUP;
    unsigned grid = 256u * wg; if (grid > groups) grid = groups;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((copy_kernel<LR, LC, GROUP, PF>), dim3(grid), dim3(NT), 0, 0, a, b, c, d, log_s, groups, c);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_kernel<LR, LC, GROUP, PF>), dim3(grid), dim3(NT), 0, 0, a, b, c, d, log_s, groups, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("rows=2^%d cols=%d group=%d (%d B wide) stride=2^%u prefetch=%d wg/cu=%d: %.3f ms %.0f GB/s\n", LR, 1 << LC, GROUP,
           8 * GROUP << LC, log_s, (int)PF, wg, ms, 32.0 * n / ms / 1e6);
}

int main() {
    const size_t n = (size_t)1 << 27;
    double *a, *b, *c, *d;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&d, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    for (int wg = 1; wg <= 2; ++wg) {
        run<10, 3, 1, false>(a, b, c, d, n, 10, wg);
        run<10, 3, 4, false>(a, b, c, d, n, 10, wg);
        run<10, 3, 4, true>(a, b, c, d, n, 10, wg);
        run<10, 3, 8, true>(a, b, c, d, n, 10, wg);
        run<10, 3, 2, true>(a, b, c, d, n, 10, wg);
        run<9, 3, 4, false>(a, b, c, d, n, 17, wg);
        run<9, 3, 4, true>(a, b, c, d, n, 17, wg);
        run<10, 2, 8, false>(a, b, c, d, n, 10, wg);
        run<10, 2, 8, true>(a, b, c, d, n, 10, wg);
    }
    return 0;
}
