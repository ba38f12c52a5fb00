// mirror_load_probe.hip -- what does the MIRRORED stream of the fused C2R first pass (c2r_fused.hpp) cost?  A wave reads rows of
// 16 f32 (64-byte segments, rows `stride` elements apart: the 256 x 16 first-pass tile of c2r_fft_f32 at 2^24) in four ways:
//   asc      lane col reads element base + col                       (the tile's own stream)
//   asc+1    the same one element off the 64-byte alignment
//   desc+1   lane col reads element base + 16 - col                  (the mirrored partner stream as the kernel issues it)
//   asc+1 dpp   ascending one element off, then reversed across the 16 lanes with a DPP row_mirror move (the candidate)
// Every variant reads the same bytes in total (64 MiB per plane pair) and adds them up; GB/s of the load stream alone.
//   hipcc --offload-arch=gfx950 -O3 tools/mirror_load_probe.hip -o tools/mirror_load_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int XCD> __global__ void __launch_bounds__(256) probe(const float *in, float *out, unsigned log_cols, unsigned tiles) {
    const unsigned col = threadIdx.x & 15u, tau = threadIdx.x >> 4;  // 16 columns x 16 row groups, 16 rows per thread
    const size_t mcols = (size_t)1 << log_cols;
    float acc = 0.f;
    for (unsigned b = blockIdx.x; b < tiles; b += gridDim.x) {
        // XCD-aware order (workgroup b runs on XCD b % 8: each XCD takes one contiguous run of tiles, so that the two tiles
        // sharing a 128-byte line meet in one L2 -- as TileBody::locate; XCD = 0: plain order, every line fetched by two XCDs)
        const unsigned t = XCD ? (b & 7u) * (tiles >> 3) + (b >> 3) : b;
        const unsigned g0 = t << 4;
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *row = in + (size_t)(j * 16 + tau) * mcols;
            if (MODE == 0) v[j] = row[g0 + col];
            if (MODE == 1) v[j] = row[g0 + 1 + col];
            if (MODE == 2) v[j] = row[g0 + 16 - col];
            if (MODE == 3) v[j] = row[g0 + 1 + col];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 3) v[j] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[j]), 0x140, 0xf, 0xf, true));  // row_mirror
            acc += v[j];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const unsigned log_cols = 15, rows = 256;                 // 256 x 2^15 f32 = 32 MiB: one plane of the 2^23-point half-spectrum
    const size_t n = ((size_t)rows << log_cols) + 64;
    float *in, *out;
    CK(hipMalloc(&in, n * 4 * 8));                            // 8 planes: rotate so that every launch reads HBM-cold data
    CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMemset(in, 0, n * 4 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned tiles = 1u << (log_cols - 4);
    const char *names[4] = {"asc", "asc+1", "desc+1 (the kernel today)", "asc+1 then DPP row_mirror"};
    for (int xcd = 1; xcd >= 0; --xcd)
    for (int wg = 1024; wg <= 2048; wg *= 2)
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 9; ++rep) {
                const float *p = in + (size_t)(rep % 8) * n;
                CK(hipEventRecord(e0));
                if (xcd) {
                    if (mode == 0) probe<0, 1><<<wg, 256>>>(p, out, log_cols, tiles);
                    if (mode == 1) probe<1, 1><<<wg, 256>>>(p, out, log_cols, tiles);
                    if (mode == 2) probe<2, 1><<<wg, 256>>>(p, out, log_cols, tiles);
                    if (mode == 3) probe<3, 1><<<wg, 256>>>(p, out, log_cols, tiles);
                } else {
                    if (mode == 0) probe<0, 0><<<wg, 256>>>(p, out, log_cols, tiles);
                    if (mode == 1) probe<1, 0><<<wg, 256>>>(p, out, log_cols, tiles);
                    if (mode == 2) probe<2, 0><<<wg, 256>>>(p, out, log_cols, tiles);
                    if (mode == 3) probe<3, 0><<<wg, 256>>>(p, out, log_cols, tiles);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%s grid %4d  %-28s %7.2f us  %7.1f GB/s\n", xcd ? "XCD-aware" : "plain    ", wg, names[mode], 1e3 * best, (double)rows * (1u << log_cols) * 4 / (best * 1e-3) / 1e9);
        }
    return 0;
}
