// pass_floor.hip -- microbenchmark: what ONE pass over a single 2^20-point f64 transform (16 MiB in, 16 MiB out)
// can cost on MI355X as a function of the work decomposition, with the FFT replaced by a dummy FMA chain.
// A tile is ROWS x 16 columns (128-byte rows, the pass-B/C pattern: rows N/ROWS elements apart); every thread
// loads P (re, im) pairs, optionally burns `fma` dependent FMAs per value, and stores them in the same pattern.
//   variants: threads per workgroup (64 = one wave per tile ... 512), P, plain / nt stores, cold ring vs warm,
//   single kernel (HIP events) and chains of 3 dependent kernels replayed from a hipGraph (per-transform time).
//   hipcc --offload-arch=gfx950 -O3 tools/pass_floor.hip -o tools/pass_floor.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

constexpr unsigned LOG_N = 20;  // rows are N/ROWS elements apart: a tile holds ALL rows of its 16 columns

template <int NT, int P, bool NTS>
__global__ void __launch_bounds__(NT) pass_kernel(const double *__restrict__ in_re, const double *__restrict__ in_im,
                                                  double *__restrict__ out_re, double *__restrict__ out_im, int fma,
                                                  unsigned tiles) {
    constexpr int ROWS = NT * P / 16;  // tile = ROWS x 16 columns
    constexpr int M = ROWS / P;        // threads per column
    constexpr unsigned LOG_STRIDE = LOG_N - (unsigned)__builtin_ctz((unsigned)ROWS);
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned tile = (t & 7u) * (tiles >> 3) + (t >> 3);
        const size_t base = (size_t)tile * 16u;  // first column of the tile
        const unsigned col = threadIdx.x & 15u, tau = threadIdx.x >> 4;
        double r[P], m[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const size_t off = base + ((size_t)(j * M + tau) << LOG_STRIDE) + col;
            r[j] = __builtin_nontemporal_load(in_re + off);
            m[j] = __builtin_nontemporal_load(in_im + off);
        }
        for (int k = 0; k < fma; ++k) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                r[j] = r[j] * 0.999999 + m[j];
                m[j] = m[j] * 0.999999 - r[j];
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const size_t off = base + ((size_t)(j * M + tau) << LOG_STRIDE) + col;
            if (NTS) {
                __builtin_nontemporal_store(r[j], out_re + off);
                __builtin_nontemporal_store(m[j], out_im + off);
            } else {
                out_re[off] = r[j];
                out_im[off] = m[j];
            }
        }
    }
}

struct Variant {
    const char *name;
    void (*launch)(const double *, const double *, double *, double *, int, unsigned grid, hipStream_t);
    unsigned tiles;
};

template <int NT, int P, bool NTS>
void launch(const double *a, const double *b, double *c, double *d, int fma, unsigned grid, hipStream_t s) {
    constexpr unsigned tiles = (1u << LOG_N) / (NT * P);
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL((pass_kernel<NT, P, NTS>), dim3(grid), dim3(NT), 0, s, a, b, c, d, fma, tiles);
}

int main() {
    const size_t n = (size_t)1 << LOG_N;
    const int RING = 48;  // 48 x 16 MiB = 768 MiB per side: colder than the 256 MiB Infinity Cache
    double *in, *out, *tmp;
    CK(hipMalloc(&in, RING * 2 * n * 8));
    CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMalloc(&tmp, 4 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8));
    CK(hipMemset(out, 0, RING * 2 * n * 8));
    CK(hipMemset(tmp, 0, 4 * n * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

#define V(NT, P) \
    {#NT "thr x p" #P " plain", launch<NT, P, false>, (1u << LOG_N) / (NT * P)}, {#NT "thr x p" #P " nt", launch<NT, P, true>, (1u << LOG_N) / (NT * P)}
    const Variant vs[] = {V(64, 8), V(64, 16), V(128, 8), V(128, 16), V(256, 8), V(256, 16), V(512, 8), V(512, 16), V(64, 32)};
    const int fmas[] = {0, 8, 24};
    for (const Variant &v : vs) {
        for (int fma : fmas) {
            // (1) one kernel, cold ring in and out, eager launches timed as a block (includes launch gaps)
            float ms_cold = 0, ms_chain_graph = 0, ms_one_graph = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < RING; ++i)
                    v.launch(in + (size_t)i * 2 * n, in + (size_t)i * 2 * n + n, out + (size_t)i * 2 * n,
                             out + (size_t)i * 2 * n + n, fma, 4096, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_cold, e0, e1));
            }
            // (2) graph of RING independent single passes (kernel + boundary)
            {
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                for (int i = 0; i < RING; ++i)
                    v.launch(in + (size_t)i * 2 * n, in + (size_t)i * 2 * n + n, out + (size_t)i * 2 * n,
                             out + (size_t)i * 2 * n + n, fma, 4096, s);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_one_graph, e0, e1));
                hipGraphExecDestroy(ge);
                hipGraphDestroy(g);
            }
            // (3) graph of RING transforms, each a chain of 3 dependent passes: ring -> tmp0 -> tmp1 -> ring
            {
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                for (int i = 0; i < RING; ++i) {
                    double *x = in + (size_t)i * 2 * n, *y = out + (size_t)i * 2 * n;
                    v.launch(x, x + n, tmp, tmp + n, fma, 4096, s);
                    v.launch(tmp, tmp + n, tmp + 2 * n, tmp + 3 * n, fma, 4096, s);
                    v.launch(tmp + 2 * n, tmp + 3 * n, y, y + n, fma, 4096, s);
                }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_chain_graph, e0, e1));
                hipGraphExecDestroy(ge);
                hipGraphDestroy(g);
            }
            printf("%-22s tiles=%5u fma=%2d | one pass eager %6.2f us  graph %6.2f us | 3-pass chain (graph) %6.2f us per transform\n",
                   v.name, v.tiles, fma, 1e3 * ms_cold / RING, 1e3 * ms_one_graph / RING, 1e3 * ms_chain_graph / RING);
            fflush(stdout);
        }
    }
    return 0;
}
