// small_fft.hip -- whole-transform-in-LDS radix-2 DIT FFT for N <= 2048 (gfx950).
//
// GPU counterpart of the reference's L1-resident leaf: bit-reverse (algorithms/bravo.rs:225-251,
// scalar/BRAVO regimes) then stages 0..log_n-1 (algorithms/dit.rs:44-65, kernels/dit.rs).  One
// workgroup per transform; the bit reversal is the LDS store index of the load, each stage is one
// sweep over LDS with W_{2^(s+1)}^j read from a W_N table.  Small N is launch-bound, not a
// bandwidth problem, so this kernel favours being obviously correct.
#include <hip/hip_ext.h>

#include "kernels.hpp"

namespace phast {

template <typename T> __global__ void __launch_bounds__(256) small_fft_kernel(const SmallArgs a) {
    using cx = cx_t<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned n = 1u << a.log_n;
    T *s_re = reinterpret_cast<T *>(smem);
    T *s_im = s_re + n;
    const cx *tw = reinterpret_cast<const cx *>(a.tw);

    for (unsigned xf = blockIdx.x; xf < a.batch; xf += gridDim.x) {
        __syncthreads();
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned j = a.log_n ? (__brev(i) >> (32u - a.log_n)) : 0u;
            if (a.in_interleaved) {
                cx v = reinterpret_cast<const cx *>(a.in_re)[(size_t)xf * a.in_dist + i];
                s_re[j] = a.in_interleaved == 2 ? v.y : v.x;
                s_im[j] = a.in_interleaved == 2 ? v.x : v.y;
            } else {
                s_re[j] = reinterpret_cast<const T *>(a.in_re)[(size_t)xf * a.in_dist + i];
                s_im[j] = reinterpret_cast<const T *>(a.in_im)[(size_t)xf * a.in_dist + i];
            }
        }
        for (unsigned s = 0; s < a.log_n; ++s) {
            __syncthreads();
            const unsigned dist = 1u << s;
            for (unsigned b = threadIdx.x; b < (n >> 1); b += blockDim.x) {
                const unsigned j = b & (dist - 1u);
                const unsigned i0 = ((b >> s) << (s + 1)) | j;
                const unsigned i1 = i0 + dist;
                const cx w = tw[j << (a.log_n - 1u - s)];
                const T br = s_re[i1], bi = s_im[i1];
                const T tr = br * w.x - bi * w.y;
                const T ti = br * w.y + bi * w.x;
                const T ar = s_re[i0], ai = s_im[i0];
                s_re[i0] = ar + tr;
                s_im[i0] = ai + ti;
                s_re[i1] = ar - tr;
                s_im[i1] = ai - ti;
            }
        }
        __syncthreads();
        const T scale = (T)a.scale;
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
            const T r = s_re[i] * scale, m = s_im[i] * scale;
            if (a.out_interleaved) {
                cx v;
                v.x = a.out_interleaved == 2 ? m : r;
                v.y = a.out_interleaved == 2 ? r : m;
                reinterpret_cast<cx *>(a.out_re)[(size_t)xf * a.out_dist + i] = v;
            } else {
                reinterpret_cast<T *>(a.out_re)[(size_t)xf * a.out_dist + i] = r;
                reinterpret_cast<T *>(a.out_im)[(size_t)xf * a.out_dist + i] = m;
            }
        }
    }
}

template <typename T>
hipError_t launch_small_fft(const SmallArgs &a, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const size_t lds = (size_t)2 * sizeof(T) << a.log_n;
    const unsigned grid = a.batch < 65536u ? a.batch : 65536u;
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(small_fft_kernel<T>, dim3(grid), dim3(256), (uint32_t)lds, stream, ev_start, ev_stop, 0, a);
    else
        hipLaunchKernelGGL(small_fft_kernel<T>, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template hipError_t launch_small_fft<float>(const SmallArgs &, hipStream_t, hipEvent_t, hipEvent_t);
template hipError_t launch_small_fft<double>(const SmallArgs &, hipStream_t, hipEvent_t, hipEvent_t);

}  // namespace phast
