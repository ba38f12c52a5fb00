// strided_rw.hip -- microbenchmark: copy rate on MI355X when the READ and the WRITE side of a pass use
// different segment lengths.  A tile is 8192 f64 per plane (re + im), 512 threads x 16 elements, exactly the
// (10,3) tile of the 2^20 plan.  Pattern = (segment length in elements, distance between segments).
//   hipcc --offload-arch=gfx950 -O3 tools/strided_rw.hip -o tools/strided_rw.bin
#include <hip/hip_runtime.h>
#include <cstdio>

struct Pat { unsigned lseg, lstride; };  // log2 segment elements, log2 distance between segments (elements)

__device__ inline size_t tile_off(unsigned tile, unsigned e, Pat p) {
    // a transform is 2^20 elements; tiles of one transform interleave at segment granularity
    const unsigned per_xf = 128;  // tiles per transform
    const unsigned xf = tile / per_xf, t = tile % per_xf;
    const unsigned seg = e >> p.lseg, in = e & ((1u << p.lseg) - 1u);
    size_t base;
    if (p.lseg == 13) base = (size_t)t << 13;                       // whole tile contiguous
    else base = ((size_t)t << p.lseg);                              // tiles adjacent inside a segment row
    return ((size_t)xf << 20) + base + ((size_t)seg << p.lstride) + in;
}

__global__ void __launch_bounds__(512) copy_kernel(const double* __restrict__ in_re, const double* __restrict__ in_im,
                                                   double* __restrict__ out_re, double* __restrict__ out_im,
                                                   unsigned tiles, Pat rd, Pat wr) {
    const unsigned chunk = tiles >> 3;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned tile = (t & 7u) * chunk + (t >> 3);
        double r[16], m[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t off = tile_off(tile, j * 512 + threadIdx.x, rd);
            r[j] = in_re[off];
            m[j] = in_im[off];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t off = tile_off(tile, j * 512 + threadIdx.x, wr);
            out_re[off] = r[j] * 1.0000001;
            out_im[off] = m[j] * 1.0000001;
        }
    }
}

int main() {
    const size_t n = (size_t)1 << 27;
    double *a, *b, *c, *d;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&d, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    const unsigned tiles = (unsigned)(n >> 13);
    // segment patterns: 8 el (64 B) rows 2^10 apart; 16-el; 64 el (512 B) blocks 2^13 apart; 1024 el runs 2^17 apart; contiguous
    const Pat pats[] = {{3, 10}, {6, 13}, {10, 17}, {13, 13}};
    const char* names[] = {"64B-rows", "512B-blocks", "8KiB-runs", "contiguous"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg = 2; wg <= 4; wg *= 2)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                hipLaunchKernelGGL(copy_kernel, dim3(256 * wg), dim3(512), 0, 0, a, b, c, d, tiles, pats[i], pats[j]);
                hipEventRecord(e0);
                for (int k = 0; k < 3; ++k)
                    hipLaunchKernelGGL(copy_kernel, dim3(256 * wg), dim3(512), 0, 0, a, b, c, d, tiles, pats[i], pats[j]);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ms /= 3;
                printf("wg/cu=%d read %-12s write %-12s: %.3f ms %.0f GB/s\n", wg, names[i], names[j], ms, 32.0 * n / ms / 1e6);
            }
    return 0;
}
