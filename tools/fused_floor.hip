// fused_floor.hip -- microbenchmark for a FUSED first+second pass of a single 2^20-point f64 transform, FFT math
// replaced by "+1".  Structure under test ("XCD-local fusion"):
//   N = 2^20 = [p:7][r:6][u:7].  Pass A (FFT over p) for the columns (r, u) with u in one 16-wide block writes
//   S[u][r][q]; pass B (FFT over r) for (u, q) needs exactly the S[u][*][*] of the SAME u -- so with the u-blocks
//   dealt to the 8 XCDs, A -> B is an XCD-local exchange: 2 MiB per XCD that can stay in that XCD's 4 MiB L2.
//   One kernel: every workgroup reads its XCC_ID, pulls A tiles of its XCD's group from a per-XCD queue, signals,
//   waits until the group's A tiles are all stored (XCD-local counter), then pulls B tiles.  Fabric traffic of the
//   fused kernel = 16 MiB in + 16 MiB out (B overwrites S in place) instead of 2 x (16 + 16).
// Measured against two separate pass kernels of the same access patterns (tools/pass_floor.hip).
// Validation: the scratch is filled with NaN before every launch; a B tile that read S before A wrote it would
// carry the NaN to the output; the checksum must be sum(input) + 2 N.
//   hipcc --offload-arch=gfx950 -O3 tools/fused_floor.hip -o tools/fused_floor.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

constexpr unsigned LOG_N = 20, LA = 7, LB = 6, LC_ = 7;  // [p][r][u]
constexpr unsigned NT = 256, P = 8;                      // 2048-point tiles
constexpr unsigned TA = 64;                              // A tiles per group: one per r (128 p x 16 u)
constexpr unsigned TB = 64;                              // B tiles per group: 16 u x 4 q-blocks of 32 (64 r x 32 q)

// every counter on its own 128-byte line, one set per XCD: device-scope atomics on ONE line serialise at ~11 ns each
// (MI355X_MICROARCH.md "dequeue": one word saturates at ~88 atomics/us, per-XCD heads do not)
struct Line {
    unsigned v, pad[31];
};
struct Ctl {
    Line a_next[8], a_done[8], b_next[8], xcd_seen[8];
    Line exit_count, error;
};

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

template <bool NTS>
__device__ inline void tile_a(const double *__restrict__ in_re, const double *__restrict__ in_im, double *s_re, double *s_im,
                              unsigned g, unsigned t) {
    // load: rows p = j*16 + tau (j < 8), 16 columns u = g*16 + col, fixed r = t
    const unsigned col = threadIdx.x & 15u, tau = threadIdx.x >> 4;
    double r[P], m[P];
#pragma unroll
    for (unsigned j = 0; j < P; ++j) {
        const size_t off = ((size_t)(j * 16 + tau) << (LB + LC_)) + ((size_t)t << LC_) + g * 16 + col;
        r[j] = __builtin_nontemporal_load(in_re + off);
        m[j] = __builtin_nontemporal_load(in_im + off);
    }
    // store: S[u][r][q], per column u a contiguous run of 128 q: element e -> (c = e / 128, q = e % 128)
#pragma unroll
    for (unsigned j = 0; j < P; ++j) {
        const unsigned e = j * NT + threadIdx.x, c = e >> LA, q = e & ((1u << LA) - 1u);
        const size_t off = ((size_t)(g * 16 + c) << (LA + LB)) + ((size_t)t << LA) + q;
        s_re[off] = r[j] + 1.0;
        s_im[off] = m[j] + 1.0;
    }
}

template <bool NTS>
__device__ inline void tile_b(double *s_re, double *s_im, double *o_re, double *o_im, unsigned g, unsigned t) {
    // tile t of group g: u = g*16 + t/4, q-block (t%4)*32; rows r = j*8 + tau (j < 8), 32 columns
    const unsigned u = g * 16 + (t >> 2), q0 = (t & 3u) * 32u;
    const unsigned col = threadIdx.x & 31u, tau = threadIdx.x >> 5;
    double r[P], m[P];
#pragma unroll
    for (unsigned j = 0; j < P; ++j) {
        const size_t off = ((size_t)u << (LA + LB)) + ((size_t)(j * 8 + tau) << LA) + q0 + col;
        r[j] = __builtin_nontemporal_load(s_re + off);  // nt: L1 bypass, served by the XCD's L2
        m[j] = __builtin_nontemporal_load(s_im + off);
    }
#pragma unroll
    for (unsigned j = 0; j < P; ++j) {
        const size_t off = ((size_t)u << (LA + LB)) + ((size_t)(j * 8 + tau) << LA) + q0 + col;
        if (NTS) {
            __builtin_nontemporal_store(r[j] + 1.0, o_re + off);
            __builtin_nontemporal_store(m[j] + 1.0, o_im + off);
        } else {
            o_re[off] = r[j] + 1.0;
            o_im[off] = m[j] + 1.0;
        }
    }
}

// the fused kernel: XCD-local A -> B
template <bool NTS>
__global__ void __launch_bounds__(NT) fused_kernel(const double *in_re, const double *in_im, double *s_re, double *s_im,
                                                   Ctl *ctl, int stage = 9) {
    __shared__ unsigned sh_t;
    const unsigned g = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&ctl->xcd_seen[g].v, 1u);
    if (stage < 1) return;
    for (unsigned it = 0; it < 4096; ++it) {  // phase A: this XCD's group
        if (threadIdx.x == 0) sh_t = atomicAdd(&ctl->a_next[g].v, 1u);
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(sh_t);  // wave-uniform: scalar branches around the barriers
        __syncthreads();
        if (t >= TA) break;
        tile_a<NTS>(in_re, in_im, s_re, s_im, g, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are in the L2
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&ctl->a_done[g].v, 1u);
    }
    if (stage < 2) return;
    if (threadIdx.x == 0) {  // XCD-local barrier: all A tiles of the group stored
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->a_done[g].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TA) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 14)) {
                atomicExch(&ctl->error.v, 1u);
                break;
            }
        }
    }
    __syncthreads();
    if (stage < 3) return;
    for (unsigned it = 0; it < 4096; ++it) {  // phase B
        if (threadIdx.x == 0) sh_t = atomicAdd(&ctl->b_next[g].v, 1u);
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(sh_t);  // wave-uniform: scalar branches around the barriers
        __syncthreads();
        if (t >= TB) break;
        tile_b<NTS>(s_re, s_im, s_re, s_im, g, t);
    }
    if (threadIdx.x == 0) {  // the last workgroup out resets the control block for the next launch
        if (atomicAdd(&ctl->exit_count.v, 1u) == gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) ctl->a_next[i].v = ctl->a_done[i].v = ctl->b_next[i].v = 0u;
            __threadfence();
            ctl->exit_count.v = 0u;
        }
    }
}

// static assignment (workgroup b: group b % 8 -- the observed XCD placement, VERIFIED against XCC_ID -- tile b / 8): the
// only device-scope traffic on the critical path is one fire-and-forget add and the poll
template <bool NTS>
__global__ void __launch_bounds__(NT) fused_static_kernel(const double *in_re, const double *in_im, double *s_re,
                                                          double *s_im, Ctl *ctl) {
    const unsigned g = blockIdx.x & 7u, t = blockIdx.x >> 3;
    if (xcc_id() != g && threadIdx.x == 0) atomicExch(&ctl->error.v, 2u);  // placement is not what the plan assumes
    tile_a<NTS>(in_re, in_im, s_re, s_im, g, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&ctl->a_done[g].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->a_done[g].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < TA) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 16)) {
                atomicExch(&ctl->error.v, 1u);
                break;
            }
        }
    }
    __syncthreads();
    tile_b<NTS>(s_re, s_im, s_re, s_im, g, t);
    if (threadIdx.x == 0) {  // last one through the barrier of its group re-arms it (nobody waits for this)
        if (__hip_atomic_fetch_add(&ctl->b_next[g].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == TA - 1) {
            __hip_atomic_store(&ctl->a_done[g].v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->b_next[g].v, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the two passes as separate kernels (same tile code, static tile -> workgroup map, XCD-aware order)
template <bool NTS>
__global__ void __launch_bounds__(NT) pass_a_kernel(const double *in_re, const double *in_im, double *s_re, double *s_im) {
    const unsigned b = blockIdx.x, g = b & 7u, t = b >> 3;
    tile_a<NTS>(in_re, in_im, s_re, s_im, g, t);
}
template <bool NTS> __global__ void __launch_bounds__(NT) pass_b_kernel(double *s_re, double *s_im) {
    const unsigned b = blockIdx.x, g = b & 7u, t = b >> 3;
    tile_b<NTS>(s_re, s_im, s_re, s_im, g, t);
}

__global__ void fill_nan(double *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = NAN;
}
__global__ void checksum(const double *p, size_t n, double *out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    atomicAdd(out, acc);
}

int main(int argc, char **argv) {
    const int stage = argc > 1 ? atoi(argv[1]) : 9;
    const size_t n = (size_t)1 << LOG_N;
    const int RING = 48;
    double *in, *scr, *sum;
    Ctl *ctl;
    CK(hipMalloc(&in, RING * 2 * n * 8));
    CK(hipMalloc(&scr, RING * 2 * n * 8));  // a ring of scratch/out buffers too: the output is cold like the real thing
    CK(hipMalloc(&sum, 8));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    std::vector<double> h(2 * n);
    for (size_t i = 0; i < 2 * n; ++i) h[i] = (double)(i % 7);
    double want = 0;
    for (size_t i = 0; i < 2 * n; ++i) want += h[i] + 2.0;
    for (int i = 0; i < RING; ++i) CK(hipMemcpy(in + (size_t)i * 2 * n, h.data(), 2 * n * 8, hipMemcpyHostToDevice));
    fprintf(stderr, "inputs uploaded\n");
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    {  // one plain launch first: census and sanity before anything is timed
        fill_nan<<<1024, 256, 0, s>>>(scr, 2 * n);
        CK(hipStreamSynchronize(s));
        fprintf(stderr, "scratch filled; launching the probe\n");
        fused_kernel<false><<<512, NT, 0, s>>>(in, in + n, scr, scr + n, ctl, stage);
        CK(hipStreamSynchronize(s));
        fprintf(stderr, "probe done\n");
        Ctl hc;
        CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("probe launch: error=%u exit_count=%u  xcd census:", hc.error.v, hc.exit_count.v);
        for (int i = 0; i < 8; ++i) printf(" %u", hc.xcd_seen[i].v);
        printf("  a_next:");
        for (int i = 0; i < 8; ++i) printf(" %u", hc.a_next[i].v);
        printf("  a_done:");
        for (int i = 0; i < 8; ++i) printf(" %u", hc.a_done[i].v);
        printf("\n");
        fflush(stdout);
        if (hc.error.v) return 2;
        if (stage < 9) return 0;
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
    }
    for (int wgcu = 1; wgcu <= 3; ++wgcu)
        for (int nts = 0; nts < 2; ++nts)
            for (int mode = 0; mode < 3; ++mode) {  // 0 = two kernels, 1 = fused (dynamic queues), 2 = fused, static
                const unsigned grid = 256u * wgcu;
                if ((mode == 0 || mode == 2) && wgcu != 2) continue;  // the separate kernels always launch 512 workgroups (one tile each)
                fill_nan<<<1024, 256, 0, s>>>(scr, RING * 2 * n);
                hipGraph_t gph;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                for (int i = 0; i < RING; ++i) {
                    const double *x = in + (size_t)i * 2 * n;
                    double *y = scr + (size_t)i * 2 * n;
                    if (mode == 0) {
                        if (nts) {
                            pass_a_kernel<true><<<512, NT, 0, s>>>(x, x + n, y, y + n);
                            pass_b_kernel<true><<<512, NT, 0, s>>>(y, y + n);
                        } else {
                            pass_a_kernel<false><<<512, NT, 0, s>>>(x, x + n, y, y + n);
                            pass_b_kernel<false><<<512, NT, 0, s>>>(y, y + n);
                        }
                    } else if (mode == 2) {
                        if (nts) fused_static_kernel<true><<<512, NT, 0, s>>>(x, x + n, y, y + n, ctl);
                        else fused_static_kernel<false><<<512, NT, 0, s>>>(x, x + n, y, y + n, ctl);
                    } else {
                        if (nts) fused_kernel<true><<<grid, NT, 0, s>>>(x, x + n, y, y + n, ctl);
                        else fused_kernel<false><<<grid, NT, 0, s>>>(x, x + n, y, y + n, ctl);
                    }
                }
                CK(hipStreamEndCapture(s, &gph));
                CK(hipGraphInstantiate(&ge, gph, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                // validate the first replay (scratch was NaN before it)
                double got = 0;
                CK(hipMemsetAsync(sum, 0, 8, s));
                checksum<<<512, 256, 0, s>>>(scr + (size_t)(RING - 1) * 2 * n, 2 * n, sum);
                CK(hipMemcpyAsync(&got, sum, 8, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                Ctl hc;
                CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
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
                printf("%s wg/cu=%d %-5s: %6.2f us per transform (A+B)   checksum %s (%.1f vs %.1f)  error=%u  xcd census:",
                       mode == 0 ? "two kernels " : mode == 1 ? "fused queues" : "fused static", wgcu, nts ? "nt" : "plain", 1e3 * best / RING,
                       got == want ? "ok" : "MISMATCH", got, want, hc.error.v);
                for (int i = 0; i < 8; ++i) printf(" %u", hc.xcd_seen[i].v);
                printf("\n");
                fflush(stdout);
                CK(hipMemset(ctl, 0, sizeof(Ctl)));
                (void)hipGraphExecDestroy(ge);
                (void)hipGraphDestroy(gph);
            }
    return 0;
}
