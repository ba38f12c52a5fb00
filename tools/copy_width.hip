// copy_width.hip -- what does a plain streaming copy reach on MI355X as a function of bytes per lane per access
// (4 / 8 / 16), non-temporal hints and grid size?  Sets the ceiling the FFT passes are compared with.
//   hipcc --offload-arch=gfx950 -O3 tools/copy_width.hip -o tools/copy_width.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename V, bool NT> __global__ void __launch_bounds__(256) copy_kernel(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
        else out[i] = in[i];
    }
}
template <typename V, bool NT> __global__ void __launch_bounds__(256) read_kernel(const V* __restrict__ in, V* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    V acc = in[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        acc = acc + v;
    }
    if (acc == (V)12345.678) out[0] = acc;
}
template <typename V, bool NT> __global__ void __launch_bounds__(256) write_kernel(const V* __restrict__, V* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (NT) __builtin_nontemporal_store((V)1, out + i);
        else out[i] = (V)1;
    }
}

template <typename K> float timeit(K k, const void* a, void* b, size_t n, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, (decltype(nullptr))nullptr, nullptr, 0);
    (void)hipGetLastError();
    return 0;
}

#define RUN(KERNEL, V, NT, BYTES_MOVED, LABEL)                                                                  \
    for (int wg = 4; wg <= 32; wg *= 2) {                                                                        \
        const size_t n = bytes / sizeof(V);                                                                      \
        hipLaunchKernelGGL((KERNEL<V, NT>), dim3(256 * wg), dim3(256), 0, 0, (const V*)a, (V*)b, n);             \
        (void)hipEventRecord(e0);                                                                                \
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((KERNEL<V, NT>), dim3(256 * wg), dim3(256), 0, 0, (const V*)a, (V*)b, n); \
        (void)hipEventRecord(e1);                                                                                \
        (void)hipEventSynchronize(e1);                                                                           \
        float ms;                                                                                                \
        (void)hipEventElapsedTime(&ms, e0, e1);                                                                  \
        printf("%-6s %2zu B/lane nt=%d wg/cu=%2d: %.0f GB/s\n", LABEL, sizeof(V), (int)NT, wg, BYTES_MOVED / (ms / 5) / 1e6); \
    }

typedef double double2_t __attribute__((ext_vector_type(2)));
int main() {
    const size_t bytes = (size_t)2 << 30;
    void *a, *b;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    RUN(copy_kernel, float, false, 2.0 * bytes, "copy")
    RUN(copy_kernel, double, false, 2.0 * bytes, "copy")
    RUN(copy_kernel, double, true, 2.0 * bytes, "copy")
    RUN(copy_kernel, double2_t, false, 2.0 * bytes, "copy")
    RUN(copy_kernel, double2_t, true, 2.0 * bytes, "copy")
    RUN(read_kernel, double, false, 1.0 * bytes, "read")
    RUN(read_kernel, double, true, 1.0 * bytes, "read")
    RUN(write_kernel, double, false, 1.0 * bytes, "write")
    RUN(write_kernel, double, true, 1.0 * bytes, "write")
    return 0;
}
