// r2c.hip -- real-FFT post/pre-processing passes on gfx950.
//
//   untangle_kernel       <- simd_untangle_inplace_f32/f64   (algorithms/r2c.rs:150-242)
//   c2r_preprocess_kernel <- simd_c2r_preprocess_f32/f64     (algorithms/r2c.rs:263-433)
// The deinterleave (r2c.rs:73-128) and interleave (r2c.rs:446-489) sweeps of the reference do not
// exist here: they are the load of the first FFT pass / the store of the last one (tile_fft.hpp,
// in_interleaved / out_interleaved).
//
// One thread per mirror pair (k, half-k): 4 loads + 4 stores, both sides of a wave's accesses are
// contiguous 64-lane runs (the mirror side descending).  The twiddle 0.5*W_N^k is a product of three
// small-table entries (read through L1/L2: consecutive lanes hit consecutive level-0 entries and one
// shared entry of the upper levels); unlike planner.rs:120-162 there is no rotation recurrence, so
// there is no drift to reproduce -- the values are correctly rounded to ~1 ulp.
#include <hip/hip_ext.h>

#include "kernels.hpp"

namespace phast {

template <typename T> __global__ void __launch_bounds__(256) untangle_kernel(const UntangleArgs a) {
    using cx = cx_t<T>;
    const unsigned half = a.half, q = half >> 1;
    const cx *tab = reinterpret_cast<const cx *>(a.tw3);
    for (unsigned xf = blockIdx.y; xf < a.batch; xf += gridDim.y) {
        T *re = reinterpret_cast<T *>(a.re) + (size_t)xf * a.dist;
        T *im = reinterpret_cast<T *>(a.im) + (size_t)xf * a.dist;
        for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k <= q; k += gridDim.x * blockDim.x) {
            if (k == 0) {  // r2c.rs:161-166
                const T a0 = re[0], b0 = im[0];
                re[0] = a0 + b0;
                im[0] = (T)0;
                re[half] = a0 - b0;
                im[half] = (T)0;
                if (q != 0) continue;
            }
            T wr, wi;
            tw3_lookup<T>(tab, a.tw_bits, k, wr, wi);
            wr *= (T)0.5;
            wi *= (T)0.5;
            if (k == q) {  // r2c.rs:233-236 (for half == 1 this runs after the k == 0 branch, as in the reference)
                const T x = re[q], y = im[q];
                re[q] = x + (T)2 * wr * y;
                im[q] = (T)2 * wi * y;
                continue;
            }
            const unsigned mirror = half - k;
            const T x = re[k], y = im[k], c = re[mirror], d = im[mirror];
            const T s_re = (T)0.5 * (x + c);
            const T s_im = (T)0.5 * (y - d);
            const T t_re = y + d;
            const T t_im = c - x;
            const T wzr = wr * t_re - wi * t_im;
            const T wzi = wr * t_im + wi * t_re;
            re[k] = s_re + wzr;
            im[k] = s_im + wzi;
            re[mirror] = s_re - wzr;
            im[mirror] = wzi - s_im;
        }
    }
}

template <typename T> __global__ void __launch_bounds__(256) c2r_preprocess_kernel(const C2rPreArgs a) {
    using cx = cx_t<T>;
    const unsigned half = a.half;
    const cx *tab = reinterpret_cast<const cx *>(a.tw3);
    for (unsigned xf = blockIdx.y; xf < a.batch; xf += gridDim.y) {
        const T *in_re = reinterpret_cast<const T *>(a.in_re) + (size_t)xf * a.in_dist;
        const T *in_im = reinterpret_cast<const T *>(a.in_im) + (size_t)xf * a.in_dist;
        T *z_re = reinterpret_cast<T *>(a.z_re) + (size_t)xf * a.z_dist;
        T *z_im = reinterpret_cast<T *>(a.z_im) + (size_t)xf * a.z_dist;
        for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < half; k += gridDim.x * blockDim.x) {
            const unsigned mirror = half - k;
            T c_h, s_h;
            tw3_lookup<T>(tab, a.tw_bits, k, c_h, s_h);
            c_h *= (T)0.5;
            s_h *= (T)0.5;
            const T re_first = in_re[k], im_first = in_im[k];
            const T re_second = in_re[mirror], im_second = -in_im[mirror];
            const T zx_re = (T)0.5 * (re_first + re_second);
            const T zx_im = (T)0.5 * (im_first + im_second);
            const T dr = re_first - re_second;
            const T di = im_first - im_second;
            const T zy_re = c_h * dr + s_h * di;
            const T zy_im = c_h * di - s_h * dr;
            z_re[k] = zx_re - zy_im;
            z_im[k] = zx_im + zy_re;
        }
    }
}

static inline dim3 grid_for(unsigned work, unsigned batch) {
    unsigned gx = (work + 255u) / 256u;
    if (gx > 4096u) gx = 4096u;
    if (gx == 0) gx = 1;
    return dim3(gx, batch < 65535u ? batch : 65535u);
}

template <typename T> hipError_t launch_untangle(const UntangleArgs &a, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    if (e0 && e1)  // measurement hook: events bound to the dispatch (kernel execution time)
        hipExtLaunchKernelGGL(untangle_kernel<T>, grid_for(a.half / 2 + 1, a.batch), dim3(256), 0, stream, e0, e1, 0, a);
    else
        hipLaunchKernelGGL(untangle_kernel<T>, grid_for(a.half / 2 + 1, a.batch), dim3(256), 0, stream, a);
    return hipGetLastError();
}
template <typename T> hipError_t launch_c2r_preprocess(const C2rPreArgs &a, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
    if (e0 && e1)
        hipExtLaunchKernelGGL(c2r_preprocess_kernel<T>, grid_for(a.half, a.batch), dim3(256), 0, stream, e0, e1, 0, a);
    else
        hipLaunchKernelGGL(c2r_preprocess_kernel<T>, grid_for(a.half, a.batch), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template hipError_t launch_untangle<float>(const UntangleArgs &, hipStream_t, hipEvent_t, hipEvent_t);
template hipError_t launch_untangle<double>(const UntangleArgs &, hipStream_t, hipEvent_t, hipEvent_t);
template hipError_t launch_c2r_preprocess<float>(const C2rPreArgs &, hipStream_t, hipEvent_t, hipEvent_t);
template hipError_t launch_c2r_preprocess<double>(const C2rPreArgs &, hipStream_t, hipEvent_t, hipEvent_t);

}  // namespace phast
