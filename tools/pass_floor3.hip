// pass_floor3.hip -- microbenchmark: a 3-pass transform of 2^20 f64 points (copy kernels, 128-byte rows) as a DAG of
// half-/quarter-pass kernels on two streams, so that the tail of one pass overlaps the head of the next:
//   A0 -> B00, B01 ; A1 -> B10, B11 ; C0 <- B00, B10 ; C1 <- B01, B11        (the real dependencies of the 3-pass FFT
//   when pass A is split by u-halves, pass B by (u-half, q-half) and pass C by q-halves)
// against the plain chain A; B; C.  Captured into a hipGraph, replayed; us per transform.
//   hipcc --offload-arch=gfx950 -O3 tools/pass_floor3.hip -o tools/pass_floor3.bin
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

constexpr unsigned LOG_N = 20, NT = 128, P = 8, ROWS = 64, LOG_STRIDE = LOG_N - 6, TILES = (1u << LOG_N) / (NT * P);

__global__ void __launch_bounds__(NT) pass_kernel(const double *__restrict__ in_re, const double *__restrict__ in_im,
                                                  double *__restrict__ out_re, double *__restrict__ out_im, unsigned tile0,
                                                  unsigned ntiles) {
    const unsigned t = blockIdx.x;
    const unsigned tile = tile0 + (((ntiles & 7u) == 0u) ? (t & 7u) * (ntiles >> 3) + (t >> 3) : t);
    const size_t base = (size_t)tile * 16u;
    const unsigned col = threadIdx.x & 15u, tau = threadIdx.x >> 4;
    double r[P], m[P];
#pragma unroll
    for (unsigned j = 0; j < P; ++j) {
        const size_t off = base + ((size_t)(j * (ROWS / P) + tau) << LOG_STRIDE) + col;
        r[j] = __builtin_nontemporal_load(in_re + off);
        m[j] = __builtin_nontemporal_load(in_im + off);
    }
#pragma unroll
    for (unsigned j = 0; j < P; ++j) {
        const size_t off = base + ((size_t)(j * (ROWS / P) + tau) << LOG_STRIDE) + col;
        __builtin_nontemporal_store(r[j] + 1.0, out_re + off);
        __builtin_nontemporal_store(m[j] + 1.0, out_im + off);
    }
}

static void pass(hipStream_t s, const double *x, double *y, size_t n, unsigned tile0, unsigned ntiles) {
    hipLaunchKernelGGL(pass_kernel, dim3(ntiles), dim3(NT), 0, s, x, x + n, y, y + n, tile0, ntiles);
}

int main() {
    const size_t n = (size_t)1 << LOG_N;
    const int RING = 48;
    double *in, *out, *tmp;
    CK(hipMalloc(&in, RING * 2 * n * 8));
    CK(hipMalloc(&out, RING * 2 * n * 8));
    CK(hipMalloc(&tmp, 4 * n * 8));
    CK(hipMemset(in, 0, RING * 2 * n * 8));
    CK(hipMemset(out, 0, RING * 2 * n * 8));
    CK(hipMemset(tmp, 0, 4 * n * 8));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ev[8];
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        for (int i = 0; i < RING; ++i) {
            const double *x = in + (size_t)i * 2 * n;
            double *y = out + (size_t)i * 2 * n, *t1 = tmp, *t2 = tmp + 2 * n;
            const unsigned H = TILES / 2, Q = TILES / 4;
            if (mode == 0) {  // plain chain
                pass(s1, x, t1, n, 0, TILES);
                pass(s1, t1, t2, n, 0, TILES);
                pass(s1, t2, y, n, 0, TILES);
            } else if (mode == 1) {  // halves on two streams, joined after every pass (no overlap across passes)
                for (int p = 0; p < 3; ++p) {
                    const double *src = p == 0 ? x : p == 1 ? t1 : t2;
                    double *dst = p == 0 ? t1 : p == 1 ? t2 : y;
                    CK(hipEventRecord(ev[0], s1));
                    CK(hipStreamWaitEvent(s2, ev[0], 0));
                    pass(s1, src, dst, n, 0, H);
                    pass(s2, src, dst, n, H, H);
                    CK(hipEventRecord(ev[1], s2));
                    CK(hipStreamWaitEvent(s1, ev[1], 0));
                }
            } else {  // the DAG
                CK(hipEventRecord(ev[0], s1));
                CK(hipStreamWaitEvent(s2, ev[0], 0));
                pass(s1, x, t1, n, 0, H);      // A0
                pass(s2, x, t1, n, H, H);      // A1
                pass(s1, t1, t2, n, 0, Q);     // B00
                CK(hipEventRecord(ev[1], s1));
                pass(s1, t1, t2, n, Q, Q);     // B01
                CK(hipEventRecord(ev[2], s1));
                pass(s2, t1, t2, n, 2 * Q, Q); // B10
                CK(hipEventRecord(ev[3], s2));
                pass(s2, t1, t2, n, 3 * Q, Q); // B11
                CK(hipStreamWaitEvent(s1, ev[3], 0));  // C0 <- B00 (s1 order), B10
                pass(s1, t2, y, n, 0, H);      // C0
                CK(hipStreamWaitEvent(s2, ev[2], 0));  // C1 <- B01, B11 (s2 order)
                pass(s2, t2, y, n, H, H);      // C1
                CK(hipEventRecord(ev[4], s2));
                CK(hipStreamWaitEvent(s1, ev[4], 0));  // join
            }
        }
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s1));
        CK(hipStreamSynchronize(s1));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s1));
            CK(hipGraphLaunch(ge, s1));
            CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            best = t < best ? t : best;
        }
        printf("%s: %6.2f us per transform\n", mode == 0 ? "chain A;B;C              " : mode == 1 ? "halves, joined per pass  " : "DAG (tails overlap heads)",
               1e3 * best / RING);
        fflush(stdout);
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    }
    return 0;
}
