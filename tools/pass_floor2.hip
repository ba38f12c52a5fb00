// pass_floor2.hip -- microbenchmark: copy floor of the access patterns of a TWO-pass plan for one 2^20-point f64
// transform (N = 1024 x 1024), FFT replaced by nothing.  A tile is 1024 rows x COLS columns (rows N/1024 apart);
// COLS = 4 gives 256 tiles of 4096 points with 32-byte rows -- narrow, but neighbouring tiles run on the same XCD
// at the same time, so the XCD's L2 (4 MiB for 2 MiB in + 2 MiB out) can merge them into full lines.
//   read  pattern: 'rows' (1024 x COLS strided rows)
//   write pattern: 'rows' | 'runs' (COLS contiguous 1024-element runs, pass A's store) | 'blk' (blocked intermediate:
//                  for each of 256 q-blocks one COLS*4-element segment)
//   hipcc --offload-arch=gfx950 -O3 tools/pass_floor2.hip -o tools/pass_floor2.bin
#include <hip/hip_runtime.h>

#include <cstdio>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

constexpr unsigned LOG_N = 20, ROWS = 1024, LOG_STRIDE = 10;

// RD: 0 rows, 1 contiguous tile (blocked intermediate read).  WR: 0 rows, 1 runs, 2 blocked segments of COLS*4
template <int NT, int P, int LC, int RD, int WR, bool XCD>
__global__ void __launch_bounds__(NT) pass_kernel(const double *__restrict__ in_re, const double *__restrict__ in_im,
                                                  double *__restrict__ out_re, double *__restrict__ out_im,
                                                  unsigned tiles) {
    constexpr int COLS = 1 << LC;
    static_assert(NT * P == ROWS * COLS, "tile = 1024 x COLS");
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        const unsigned tile = XCD ? (t & 7u) * (tiles >> 3) + (t >> 3) : t;
        double r[P], m[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const unsigned e = j * NT + threadIdx.x;
            size_t off;
            if (RD == 0) off = ((size_t)(e >> LC) << LOG_STRIDE) + tile * COLS + (e & (COLS - 1));
            else off = (size_t)tile * (ROWS * COLS) + e;
            r[j] = in_re[off];
            m[j] = in_im[off];
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const unsigned e = j * NT + threadIdx.x;
            size_t off;
            if (WR == 0) off = ((size_t)(e >> LC) << LOG_STRIDE) + tile * COLS + (e & (COLS - 1));
            else if (WR == 1) off = (size_t)tile * (ROWS * COLS) + e;
            else {  // segment s (of 256) = 4*COLS contiguous elements at block s, position tile
                const unsigned seg = e / (4 * COLS), in = e % (4 * COLS);
                off = (size_t)seg * 4096 + (size_t)tile * (4 * COLS) + in;
            }
            out_re[off] = r[j] + 1.0;
            out_im[off] = m[j] + 1.0;
        }
    }
}

typedef void (*Launch)(const double *, const double *, double *, double *, hipStream_t);
template <int NT, int P, int LC, int RD, int WR, bool XCD>
void launch(const double *a, const double *b, double *c, double *d, hipStream_t s) {
    constexpr unsigned tiles = (1u << LOG_N) / (NT * P);
    hipLaunchKernelGGL((pass_kernel<NT, P, LC, RD, WR, XCD>), dim3(tiles), dim3(NT), 0, s, a, b, c, d, tiles);
}

int main() {
    const size_t n = (size_t)1 << LOG_N;
    const int RING = 48;
    double *in, *out, *tmp;
    CK(hipMalloc(&in, RING * 2 * n * 8));
    CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMalloc(&tmp, 2 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8));
    CK(hipMemset(out, 0, RING * 2 * n * 8));
    CK(hipMemset(tmp, 0, 2 * n * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V {
        const char *name;
        Launch a, b;
    };
#define PAIR(NT, P, LC, X) \
    {#NT "thr p" #P " cols=2^" #LC " xcd=" #X ": A rows->runs, B rows->rows", launch<NT, P, LC, 0, 1, X>, launch<NT, P, LC, 0, 0, X>}, \
    {#NT "thr p" #P " cols=2^" #LC " xcd=" #X ": A rows->blk,  B tile->rows", launch<NT, P, LC, 0, 2, X>, launch<NT, P, LC, 1, 0, X>}
    const V vs[] = {PAIR(512, 8, 2, true), PAIR(512, 8, 2, false), PAIR(256, 16, 2, true), PAIR(1024, 8, 3, true),
                    PAIR(512, 16, 3, true), PAIR(1024, 16, 4, true), PAIR(512, 32, 4, true)};
    for (const V &v : vs) {
        float ms[3] = {0, 0, 0};
        for (int which = 0; which < 3; ++which) {  // 0: A alone, 1: B alone, 2: chain A;B through tmp
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < RING; ++i) {
                double *x = in + (size_t)i * 2 * n, *y = out + (size_t)i * 2 * n;
                if (which == 0) v.a(x, x + n, y, y + n, s);
                if (which == 1) v.b(x, x + n, y, y + n, s);
                if (which == 2) {
                    v.a(x, x + n, tmp, tmp + n, s);
                    v.b(tmp, tmp + n, y, y + n, s);
                }
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                best = t < best ? t : best;
            }
            ms[which] = best;
            (void)hipGraphExecDestroy(ge);
            (void)hipGraphDestroy(g);
        }
        printf("%-60s | A %6.2f us  B %6.2f us  chain A;B %6.2f us per transform\n", v.name, 1e3 * ms[0] / RING,
               1e3 * ms[1] / RING, 1e3 * ms[2] / RING);
        fflush(stdout);
    }
    return 0;
}
