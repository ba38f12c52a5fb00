// probe.hip -- harness kernels: what THIS box's HBM gives plain streaming kernels (read-only, write-only, 1:1 copy).
//
// SURVEY.md 8(d) "bounding roofline": HBM bandwidth, vendor peak confirmed with a device probe in the same run.  A pass
// of the transform reads every byte once and writes it once, so the number its rate should be read against is the COPY
// figure of the box it ran on, measured by hand-written kernels of the same kind (grid-stride, 8 / 16 bytes per lane,
// non-temporal hints) -- not a library memcpy.  bench.py puts {read, write, copy} GB/s on its JSON line
// (roofline.stream_probe) and quotes every dominant pass against `copy`.  Not part of the transform path.
#include "kernels.hpp"

namespace phast {

typedef double double2_t __attribute__((ext_vector_type(2)));

template <typename V, bool NT>
__global__ void __launch_bounds__(256) probe_copy_kernel(const V *__restrict__ in, V *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
        else out[i] = in[i];
    }
}
template <typename V, bool NT>
__global__ void __launch_bounds__(256) probe_read_kernel(const V *__restrict__ in, V *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    V acc = in[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        acc = acc + v;
    }
    // never true for the finite fill the probe uses; keeps the loads alive without a store per thread
    if (reinterpret_cast<const double *>(&acc)[0] == 12345.678) out[0] = acc;
}
template <typename V, bool NT>
__global__ void __launch_bounds__(256) probe_write_kernel(const V *__restrict__, V *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    V one;
    if constexpr (sizeof(V) == 16) one = V{1.0, 1.0};
    else one = (V)1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (NT) __builtin_nontemporal_store(one, out + i);
        else out[i] = one;
    }
}

template <typename V> using ProbeKernel = void (*)(const V *, V *, size_t);

// best of `reps` back-to-back launches (HIP events on `stream` around the K launches), in GB/s of `moved` bytes
template <typename V>
static hipError_t run_variant(ProbeKernel<V> k, int wg_per_cu, int cus, const void *a, void *b, size_t bytes, double moved,
                              int reps, hipStream_t stream, hipEvent_t e0, hipEvent_t e1, double *best) {
    const size_t n = bytes / sizeof(V);
    // wg_per_cu == 0: no persistent workgroups -- one element per thread, workgroups dispatched in address order (round 6: the form
    // csrc/complex_nums.hip's sweeps run fastest in; a yardstick must not be slower than what is measured against it)
    const dim3 grid(wg_per_cu ? (unsigned)(cus * wg_per_cu) : (unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k, grid, block, 0, stream, (const V *)a, (V *)b, n);  // warm
    hipError_t e = hipEventRecord(e0, stream);
    if (e != hipSuccess) return e;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, grid, block, 0, stream, (const V *)a, (V *)b, n);
    e = hipEventRecord(e1, stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e != hipSuccess) return e;
    float ms = 0;
    e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) return e;
    const double gbs = moved * reps / (ms * 1e-3) / 1e9;
    if (gbs > *best) *best = gbs;
    return hipGetLastError();
}

// d_a, d_b: two device buffers of `bytes` each (a is read, b written).  out_gbps = {read, write, copy (read + write
// counted)}: the best variant of each over grid sizes and access widths.  Blocks until done.
hipError_t stream_probe(const void *d_a, void *d_b, size_t bytes, int reps, int cus, double *out_gbps, hipStream_t stream) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double rd = 0, wr = 0, cp = 0;
    const double B = (double)bytes;
#define PHAST_PROBE(K, V, NT, WG, MOVED, BEST) \
    if (e == hipSuccess) e = run_variant<V>(K<V, NT>, WG, cus, d_a, d_b, bytes, MOVED, reps, stream, e0, e1, &BEST)
    for (int wg : {0, 8, 16}) {
        PHAST_PROBE(probe_read_kernel, double, true, wg, B, rd);
        PHAST_PROBE(probe_read_kernel, double2_t, true, wg, B, rd);
        PHAST_PROBE(probe_read_kernel, double, false, wg, B, rd);
    }
    for (int wg : {0, 16, 32}) {
        PHAST_PROBE(probe_write_kernel, double, false, wg, B, wr);
        PHAST_PROBE(probe_write_kernel, double2_t, false, wg, B, wr);
        PHAST_PROBE(probe_write_kernel, double2_t, true, wg, B, wr);
    }
    for (int wg : {0, 4, 8, 16}) {
        PHAST_PROBE(probe_copy_kernel, double, true, wg, 2 * B, cp);
        PHAST_PROBE(probe_copy_kernel, double2_t, true, wg, 2 * B, cp);
        PHAST_PROBE(probe_copy_kernel, double, false, wg, 2 * B, cp);
    }
#undef PHAST_PROBE
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    out_gbps[0] = rd;
    out_gbps[1] = wr;
    out_gbps[2] = cp;
    return e;
}

}  // namespace phast
