// small_fft.hip -- N <= 2048: launcher of the one-pass small-transform kernels (row_fft.hpp).
//
// GPU counterpart of the reference's leaf (algorithms/dit.rs:44-65) for transforms that fit on chip whole:
// a workgroup runs 2..256 transforms at once through the register/LDS digit chain of tile_fft.hpp; N = 1 is a
// scaled copy (the reference's recursion does nothing for a single point, algorithms/dit.rs:33-43).
#include "row_fft.hpp"

#include "kernels.hpp"

namespace phast {

template <typename T> __global__ void __launch_bounds__(256) one_point_kernel(const SmallArgs a) {
    using cx = cx_t<T>;
    const T scale = (T)a.scale;
    for (size_t xf = (size_t)blockIdx.x * blockDim.x + threadIdx.x; xf < a.batch; xf += (size_t)gridDim.x * blockDim.x) {
        T re, im;
        if (a.in_interleaved) {
            const cx v = reinterpret_cast<const cx *>(a.in_re)[xf * a.in_dist];
            re = a.in_interleaved == 2 ? v.y : v.x;
            im = a.in_interleaved == 2 ? v.x : v.y;
        } else {
            re = reinterpret_cast<const T *>(a.in_re)[xf * a.in_dist];
            im = reinterpret_cast<const T *>(a.in_im)[xf * a.in_dist];
        }
        re *= scale;
        im *= scale;
        if (a.out_interleaved) {
            cx v;
            v.x = a.out_interleaved == 2 ? im : re;
            v.y = a.out_interleaved == 2 ? re : im;
            reinterpret_cast<cx *>(a.out_re)[xf * a.out_dist] = v;
        } else {
            reinterpret_cast<T *>(a.out_re)[xf * a.out_dist] = re;
            reinterpret_cast<T *>(a.out_im)[xf * a.out_dist] = im;
        }
    }
}

template <typename T>
hipError_t launch_small_fft(const SmallArgs &a, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a.batch == 0) return hipSuccess;
    if (a.log_n == 0) {  // (never with real_mode: a real transform has at least 4 points, its core 2)
        const unsigned grid = (unsigned)((a.batch + 255) / 256 < 4096 ? (a.batch + 255) / 256 : 4096);
        if (ev_start && ev_stop)
            hipExtLaunchKernelGGL(one_point_kernel<T>, dim3(grid), dim3(256), 0, stream, ev_start, ev_stop, 0, a);
        else
            hipLaunchKernelGGL(one_point_kernel<T>, dim3(grid), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    RowArgs r{};
    r.in_re = a.in_re;
    r.in_im = a.in_im;
    r.out_re = a.out_re;
    r.out_im = a.out_im;
    r.twr = a.tw;
    r.in_dist = a.in_dist;
    r.out_dist = a.out_dist;
    r.batch = a.batch;
    r.in_interleaved = a.in_interleaved;
    r.out_interleaved = a.out_interleaved;
    r.scale = a.scale;
    r.real_mode = a.real_mode;
    r.rtw_bits = a.rtw_bits;
    r.rtw3 = a.rtw3;
    const unsigned lc = row_tile_cols_log(a.log_n);
    r.tiles_total = (unsigned)((a.batch + ((1ull << lc) - 1)) >> lc);
    if (a.log_n == 5 && a.real_mode)  // 64-point real transforms: 16 points per thread (row_fft.hpp: kRealRow5LP)
        return launch_row_inst<T, 5, 7, kRealRow5LP, true>(r, stream, ev_start, ev_stop);
#define PHAST_ROW_CASE(LR_, LC_, LP_)                                                                       \
    if (a.log_n == LR_)                                                                                     \
        return a.real_mode ? launch_row_inst<T, LR_, LC_, LP_, true>(r, stream, ev_start, ev_stop)          \
                           : launch_row_inst<T, LR_, LC_, LP_, false>(r, stream, ev_start, ev_stop);
    PHAST_ROW_SHAPES(PHAST_ROW_CASE)
#undef PHAST_ROW_CASE
    return hipErrorInvalidValue;
}

template hipError_t launch_small_fft<float>(const SmallArgs &, hipStream_t, hipEvent_t, hipEvent_t);
template hipError_t launch_small_fft<double>(const SmallArgs &, hipStream_t, hipEvent_t, hipEvent_t);

}  // namespace phast
