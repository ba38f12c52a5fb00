// fill.hip -- harness kernels: deterministic on-device synthetic inputs and per-transform digests.
//
// Inputs mirror the reference's bench/test generators (uniform [-1, 1): utilities/src/lib.rs:26-75,
// benches/bit_reversal.rs:14 seed 0xCAFE) but are counter-based so that the host oracle
// (oracle/pho_fill_*) and the GPU produce bit-identical arrays without a PCIe copy.
#include "kernels.hpp"

namespace phast {

template <typename T>
__global__ void __launch_bounds__(256)
    fill_kernel(T *re, T *im, size_t n, size_t dist, unsigned long long seed, unsigned long long first_id) {
    const size_t xf = blockIdx.y;
    T *r = re + xf * dist;
    T *m = im ? im + xf * dist : nullptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        r[i] = (T)uniform_pm1(seed, first_id + xf, 2ull * i);
        if (m) m[i] = (T)uniform_pm1(seed, first_id + xf, 2ull * i + 1ull);
    }
}

// digest[xf] = {sum re, sum im, sum (re^2 + im^2), re[probe]} accumulated in f64
template <typename T>
__global__ void __launch_bounds__(256)
    digest_kernel(const T *re, const T *im, size_t n, size_t dist, size_t probe, double *digest) {
    __shared__ double red[3][256];
    const size_t xf = blockIdx.x;
    const T *r = re + xf * dist;
    const T *m = im + xf * dist;
    double s0 = 0, s1 = 0, s2 = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)r[i], y = (double)m[i];
        s0 += x;
        s1 += y;
        s2 += x * x + y * y;
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        digest[4 * xf + 0] = red[0][0];
        digest[4 * xf + 1] = red[1][0];
        digest[4 * xf + 2] = red[2][0];
        digest[4 * xf + 3] = (double)r[probe < n ? probe : 0];
    }
}

template <typename T>
hipError_t launch_fill(T *re, T *im, size_t n, size_t batch, size_t dist, unsigned long long seed,
                       unsigned long long first_id, hipStream_t stream) {
    if (n == 0 || batch == 0) return hipSuccess;
    unsigned gx = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const unsigned gy = (unsigned)((batch - b0) < 65535 ? (batch - b0) : 65535);
        hipLaunchKernelGGL(fill_kernel<T>, dim3(gx, gy), dim3(256), 0, stream, re + b0 * dist,
                           im ? im + b0 * dist : nullptr, n, dist, seed, first_id + b0);
    }
    return hipGetLastError();
}

template <typename T>
hipError_t launch_digest(const T *re, const T *im, size_t n, size_t batch, size_t dist, size_t probe, double *digest,
                         hipStream_t stream) {
    if (n == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(digest_kernel<T>, dim3((unsigned)batch), dim3(256), 0, stream, re, im, n, dist, probe, digest);
    return hipGetLastError();
}

template hipError_t launch_fill<float>(float *, float *, size_t, size_t, size_t, unsigned long long,
                                       unsigned long long, hipStream_t);
template hipError_t launch_fill<double>(double *, double *, size_t, size_t, size_t, unsigned long long,
                                        unsigned long long, hipStream_t);
template hipError_t launch_digest<float>(const float *, const float *, size_t, size_t, size_t, size_t, double *,
                                         hipStream_t);
template hipError_t launch_digest<double>(const double *, const double *, size_t, size_t, size_t, size_t, double *,
                                          hipStream_t);

}  // namespace phast
