// complex_nums.hip -- the data-format helpers either side of the transform: complex_nums.rs (public with the feature
// `bench-internals`, like the bit reversal): `deinterleave` ([1, 2, 3, 4] -> ([1, 3], [2, 4]), complex_nums.rs:11-17, which is
// what `deinterleave_complex64 / 32` are on a cast slice, :25-39) and `combine_re_im` (:47-56).  The interleaved transforms
// (lib.rs:41-140) do NOT call these on the device -- their first pass reads pairs and their last pass writes them -- but a caller
// that holds Complex<T> data and wants planes (or the reverse) needs the sweep itself.
//
// Pure data movement, bit-exact: every element read once, written once (2 * sizeof(T) bytes per scalar), bounded by the box's
// copy rate.  A thread moves 32 bytes of pairs with 16-byte accesses where the three pointers allow it (16-byte aligned:
// any allocation, any even offset into one) and falls back to element accesses otherwise (`&v[1..]`: the reference takes plain
// slices); rows of a wave are contiguous in all three streams, non-temporal on both sides (nothing is read twice).
#include "kernels.hpp"

#ifndef PHAST_CN_UNROLL  // 16-byte groups per thread
#define PHAST_CN_UNROLL 1
#endif

namespace phast {

template <typename T> struct Vec16;  // 16 bytes of T
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };

// pairs [i0, i0 + N) of `in` -> a[i0 ..], b[i0 ..]: two 16-byte loads, two 16-byte stores
template <typename T>
__global__ void __launch_bounds__(256) deinterleave_vec_kernel(const T *__restrict__ in, T *__restrict__ a, T *__restrict__ b, size_t groups) {
    using V = typename Vec16<T>::type;
    constexpr int N = Vec16<T>::N;
    const V *vin = reinterpret_cast<const V *>(in);
    V *va = reinterpret_cast<V *>(a), *vb = reinterpret_cast<V *>(b);
    // ONE step per thread, one workgroup per 256 * U groups, workgroups dispatched in address order: measured against the
    // persistent grid-stride form of the same loop (tools/complex_nums_rate.py, profiles/r06_complex_nums.log): 6.3 against
    // 5.1-5.7 TB/s for 4 GiB of f64 pairs -- more than the grid-stride copy probe reaches on the same box
    const size_t g0 = (size_t)blockIdx.x * (256 * PHAST_CN_UNROLL) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < PHAST_CN_UNROLL; ++u) {
        const size_t g = g0 + (size_t)u * 256;
        if (g >= groups) break;
        const V lo = __builtin_nontemporal_load(vin + 2 * g), hi = __builtin_nontemporal_load(vin + 2 * g + 1);
        V x, y;
        if constexpr (N == 2) {
            x = V{lo[0], hi[0]};
            y = V{lo[1], hi[1]};
        } else {
            x = V{lo[0], lo[2], hi[0], hi[2]};
            y = V{lo[1], lo[3], hi[1], hi[3]};
        }
        __builtin_nontemporal_store(x, va + g);
        __builtin_nontemporal_store(y, vb + g);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) deinterleave_scalar_kernel(const T *__restrict__ in, T *__restrict__ a, T *__restrict__ b, size_t first, size_t pairs) {
    for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        a[i] = in[2 * i];
        b[i] = in[2 * i + 1];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) combine_vec_kernel(const T *__restrict__ re, const T *__restrict__ im, T *__restrict__ out, size_t groups) {
    using V = typename Vec16<T>::type;
    constexpr int N = Vec16<T>::N;
    const V *vr = reinterpret_cast<const V *>(re), *vi = reinterpret_cast<const V *>(im);
    V *vo = reinterpret_cast<V *>(out);
    const size_t g0 = (size_t)blockIdx.x * (256 * PHAST_CN_UNROLL) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < PHAST_CN_UNROLL; ++u) {
        const size_t g = g0 + (size_t)u * 256;
        if (g >= groups) break;
        const V x = __builtin_nontemporal_load(vr + g), y = __builtin_nontemporal_load(vi + g);
        V lo, hi;
        if constexpr (N == 2) {
            lo = V{x[0], y[0]};
            hi = V{x[1], y[1]};
        } else {
            lo = V{x[0], y[0], x[1], y[1]};
            hi = V{x[2], y[2], x[3], y[3]};
        }
        __builtin_nontemporal_store(lo, vo + 2 * g);
        __builtin_nontemporal_store(hi, vo + 2 * g + 1);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) combine_scalar_kernel(const T *__restrict__ re, const T *__restrict__ im, T *__restrict__ out, size_t first, size_t n) {
    for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        out[2 * i] = re[i];
        out[2 * i + 1] = im[i];
    }
}

static inline unsigned sweep_grid(size_t items) {  // the element-wise tail kernels: grid-stride, a handful of workgroups per CU
    const size_t want = (items + 255) / 256;
    return (unsigned)(want < 1 ? 1 : want > 8192 ? 8192 : want);
}
// the 16-byte kernels: one workgroup per 256 * PHAST_CN_UNROLL groups; beyond 2^31 - 1 workgroups (2^43 scalars) the call is split
static constexpr size_t kVecBlockGroups = (size_t)256 * PHAST_CN_UNROLL;
static constexpr size_t kVecMaxGroups = kVecBlockGroups * 0x7fffffffull;
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// in[0 .. 2 pairs) -> a[0 .. pairs), b[0 .. pairs)
template <typename T> hipError_t launch_deinterleave(const T *in, T *a, T *b, size_t pairs, hipStream_t stream) {
    if (pairs == 0) return hipSuccess;
    constexpr int N = Vec16<T>::N;
    size_t done = 0;
    if (aligned16(in) && aligned16(a) && aligned16(b) && pairs >= (size_t)N) {
        const size_t groups = pairs / N;
        for (size_t g0 = 0; g0 < groups; g0 += kVecMaxGroups) {
            const size_t cnt = groups - g0 < kVecMaxGroups ? groups - g0 : kVecMaxGroups;
            hipLaunchKernelGGL(deinterleave_vec_kernel<T>, dim3((unsigned)((cnt + kVecBlockGroups - 1) / kVecBlockGroups)), dim3(256), 0, stream,
                               in + 2 * N * g0, a + N * g0, b + N * g0, cnt);
        }
        done = groups * N;
    }
    if (done < pairs)  // the tail of an aligned call (< N pairs), or all of an element-aligned one
        hipLaunchKernelGGL(deinterleave_scalar_kernel<T>, dim3(sweep_grid(pairs - done)), dim3(256), 0, stream, in, a, b, done, pairs);
    return hipGetLastError();
}

// re[0 .. n), im[0 .. n) -> out[0 .. 2 n)
template <typename T> hipError_t launch_combine(const T *re, const T *im, T *out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    constexpr int N = Vec16<T>::N;
    size_t done = 0;
    if (aligned16(re) && aligned16(im) && aligned16(out) && n >= (size_t)N) {
        const size_t groups = n / N;
        for (size_t g0 = 0; g0 < groups; g0 += kVecMaxGroups) {
            const size_t cnt = groups - g0 < kVecMaxGroups ? groups - g0 : kVecMaxGroups;
            hipLaunchKernelGGL(combine_vec_kernel<T>, dim3((unsigned)((cnt + kVecBlockGroups - 1) / kVecBlockGroups)), dim3(256), 0, stream,
                               re + N * g0, im + N * g0, out + 2 * N * g0, cnt);
        }
        done = groups * N;
    }
    if (done < n) hipLaunchKernelGGL(combine_scalar_kernel<T>, dim3(sweep_grid(n - done)), dim3(256), 0, stream, re, im, out, done, n);
    return hipGetLastError();
}

template hipError_t launch_deinterleave<double>(const double *, double *, double *, size_t, hipStream_t);
template hipError_t launch_deinterleave<float>(const float *, float *, float *, size_t, hipStream_t);
template hipError_t launch_combine<double>(const double *, const double *, double *, size_t, hipStream_t);
template hipError_t launch_combine<float>(const float *, const float *, float *, size_t, hipStream_t);

}  // namespace phast
