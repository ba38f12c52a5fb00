// pad_stride_probe.hip -- does a NON-power-of-two row stride in the planner's scratch help the far-strided passes?
// Copy model of the last pass of N = 2^26 f64 (256 rows x 64 columns per tile, rows 2^18 elements = 2 MiB apart on both
// sides): the read side's row stride is 2^18 + pad elements (the scratch is ours to lay out), the write side stays at 2^18
// (the caller's array).  Also: both sides padded (what a fully padded layout could reach) and both contiguous.
//   hipcc --offload-arch=gfx950 -O3 tools/pad_stride_probe.hip -o tools/pad_stride_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <unsigned ROWS, unsigned COLS, unsigned NT>
__global__ void __launch_bounds__(NT) copy_tiles(const double *in_re, const double *in_im, double *out_re, double *out_im,
                                                 size_t in_stride, size_t out_stride, size_t in_xf, size_t out_xf, unsigned tiles_per_xf,
                                                 unsigned tiles) {
    constexpr unsigned P = ROWS * COLS / NT, TAUS = NT / COLS;
    const unsigned col = threadIdx.x & (COLS - 1), tau = threadIdx.x / COLS;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned tile = (t & 7u) * (tiles >> 3) + (t >> 3);  // XCD-aware order as the product
        const unsigned xf = tile / tiles_per_xf, ti = tile % tiles_per_xf;
        double r[P], m[P];
#pragma unroll
        for (unsigned j = 0; j < P; ++j) {
            const size_t off = (size_t)xf * in_xf + (size_t)(tau + TAUS * j) * in_stride + (size_t)ti * COLS + col;
            r[j] = __builtin_nontemporal_load(in_re + off);
            m[j] = __builtin_nontemporal_load(in_im + off);
        }
#pragma unroll
        for (unsigned j = 0; j < P; ++j) {
            const size_t off = (size_t)xf * out_xf + (size_t)(tau + TAUS * j) * out_stride + (size_t)ti * COLS + col;
            __builtin_nontemporal_store(r[j] + 1.0, out_re + off);
            __builtin_nontemporal_store(m[j] + 1.0, out_im + off);
        }
    }
}

template <unsigned ROWS, unsigned COLS, unsigned NT>
int run(const char *what, unsigned log_row, size_t xforms, bool in_place) {
    // `xforms` independent blocks of ROWS rows; a row is 2^log_row elements (+ pad), every tile takes COLS of them
    const size_t row = (size_t)1 << log_row, maxpad = 4096;
    const size_t xf_elems = ROWS * (row + maxpad), plane = xforms * xf_elems;
    double *in, *out;
    CK(hipMalloc(&in, 2 * plane * 8));
    CK(hipMalloc(&out, 2 * plane * 8));
    CK(hipMemset(in, 0, 2 * plane * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned tiles_per_xf = (unsigned)(row / COLS), tiles = (unsigned)(tiles_per_xf * xforms);
    const size_t pads[] = {0, 16, 32, 48, 80, 272, 1040};
    printf("%s: %u rows x %u cols, rows 2^%u (+pad) elements apart, %zu block(s)%s\n", what, ROWS, COLS, log_row, xforms,
           in_place ? ", in place" : "");
    for (int mode = 0; mode < (in_place ? 1 : 2); ++mode)
        for (size_t pad : pads) {
            const size_t is = row + pad, os = (mode || in_place) ? row + pad : row;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, 0));
                copy_tiles<ROWS, COLS, NT><<<256, NT>>>(in, in + plane, in_place ? in : out, in_place ? in + plane : out + plane, is, os,
                                                         xf_elems, xf_elems, tiles_per_xf, tiles);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("  read stride +%-5zu write stride +%-5zu: %8.1f us  %.2f TB/s\n", pad, os - row, 1e3 * best,
                   4.0 * ROWS * row * xforms * 8 / (best * 1e-3) / 1e12);
        }
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}

int main() {
    if (run<256, 64, 512>("2^26 pass C", 18, 1, false)) return 1;     // rows 2 MiB apart
    if (run<512, 32, 512>("2^26 pass B", 9, 512, true)) return 1;     // rows 4 KiB apart, 512 u-blocks, in place
    if (run<1024, 16, 512>("2^20 x 256 pass B", 10, 256, false)) return 1;  // rows 8 KiB apart, 256 transforms
    return 0;
}
