// twiddle.hip -- the inter-factor twiddle of a four-step split N = N1 * N2: element (r, c) of a row-major block
// is multiplied by W_N^(r*c).  Used by the single-transform-over-several-GPUs path (phastft_amd/distributed.py,
// SURVEY.md 8 f-3): between the two local FFT stages every rank scales its slab [n2 (its block)][k1 (all)].
// Inside one GPU the same factor is the PRE_TW load of tile_fft.hpp; the reference has no counterpart (its
// recursion never leaves one address space, algorithms/dit.rs:33-164).
//
// W_N^e comes from the same three-level tables as everywhere else (plan.hpp: host_tw3), staged in LDS: a
// streamed N-entry table would double the traffic of this otherwise pure read-modify-write sweep.
#include "kernels.hpp"

namespace phast {

template <typename T> __global__ void __launch_bounds__(256) twiddle_grid_kernel(const TwiddleGridArgs a) {
    using cx = cx_t<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx *tab = reinterpret_cast<cx *>(smem);
    for (unsigned i = threadIdx.x; i < (3u << a.tw_bits); i += blockDim.x) tab[i] = reinterpret_cast<const cx *>(a.tw3)[i];
    __syncthreads();
    T *re = reinterpret_cast<T *>(a.re), *im = reinterpret_cast<T *>(a.im);
    const unsigned long long mask = (1ull << a.log_n) - 1ull;
    for (unsigned long long r = blockIdx.y; r < a.rows; r += gridDim.y) {
        const unsigned long long gr = a.row0 + r;
        for (unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; c < a.cols;
             c += (unsigned long long)gridDim.x * blockDim.x) {
            const unsigned e = (unsigned)((gr * (a.col0 + c)) & mask);
            T wr, wi;
            tw3_lookup<T>(tab, a.tw_bits, e, wr, wi);
            const size_t at = (size_t)r * a.row_pitch + c;
            const T x = re[at], y = im[at];
            re[at] = x * wr - y * wi;
            im[at] = x * wi + y * wr;
        }
    }
}

template <typename T> hipError_t launch_twiddle_grid(const TwiddleGridArgs &a, hipStream_t stream) {
    if (a.rows == 0 || a.cols == 0) return hipSuccess;
    const size_t lds = ((size_t)3 << a.tw_bits) * sizeof(cx_t<T>);
    if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;  // TwiddleGrid::init rejects such sizes up front
    static PerDeviceLimit lds_limit;  // raised only when the request grows (as launch_tile_inst): steady state = launch only
    if (hipError_t e = raise_lds_limit(lds_limit, reinterpret_cast<const void *>(twiddle_grid_kernel<T>), lds); e != hipSuccess)
        return e;
    unsigned gx = (unsigned)((a.cols + 255) / 256);
    if (gx > 64) gx = 64;
    unsigned gy = (unsigned)(a.rows < 65535 ? a.rows : 65535);
    while ((unsigned long long)gx * gy > 16384ull && gy > 1) gy = (gy + 1) / 2;  // persistent rows: the table is staged once per block
    hipLaunchKernelGGL(twiddle_grid_kernel<T>, dim3(gx, gy), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template hipError_t launch_twiddle_grid<float>(const TwiddleGridArgs &, hipStream_t);
template hipError_t launch_twiddle_grid<double>(const TwiddleGridArgs &, hipStream_t);

}  // namespace phast
