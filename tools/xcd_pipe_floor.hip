// xcd_pipe_floor.hip -- copy model ("+1" instead of the FFT) of an L2-RESIDENT DOUBLE PASS for N = 2^26 f64.
//
// Today N = 2^26 takes three passes over HBM (plan 512 . 512 . 256: 3 x 2 GiB read + 3 x 2 GiB written, ~435 us each).
// The only way below three sweeps with 160 KiB of LDS per CU is to keep an intermediate in the per-XCD L2 (4 MiB):
//   N = 2^13 x 2^13.  An HBM pass = 8192-point FFTs along one axis for a block of 16 adjacent columns (128-byte rows):
//   a SUPER-TILE of 8192 rows x 16 columns = 2 MiB (re + im).  The 8192-point FFT is done as 128 x 64 by two sub-passes
//   over the super-tile that meet in a 2 MiB scratch owned by ONE XCD:
//       phase 1: 64 tiles of 128 rows x 16 cols   HBM (128-B segments, 64 KiB apart) -> scratch [row][16] (contiguous)
//       -- XCD barrier (32 workgroups of the team) --
//       phase 2: 128 wave tiles of 64 rows x 16   scratch (L2) -> HBM, in place over the super-tile's own rows
//       -- XCD barrier (the scratch may be overwritten) --
//   HBM traffic of the double pass = ONE read + ONE write of the data: two of them make the whole transform, 2 sweeps
//   instead of 3 -- IF the double pass runs at better than ~2/3 of a plain pass's rate (T < 650 us).
// Variants: teams per XCD (1 or 2: two super-tiles per XCD in flight, out of phase, 2 x 2 MiB of scratch), workgroups per
// CU; against the same two phases as two separate kernels through a full-size HBM scratch.
// Validation: scratch poisoned with NaN before the run; checksum must be sum(input) + 2 * (2 N).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_pipe_floor.hip -o tools/xcd_pipe_floor.bin
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

constexpr unsigned LOG_N = 26, L1 = 13, L2B = 13;         // N = [n1 : 13][n2 : 13], pass over n1 for 16-column blocks of n2
constexpr unsigned ROWS = 1u << L1, N2 = 1u << L2B;       // 8192 rows, row stride N2 elements
constexpr unsigned COLS = 16, NT = 256;
constexpr unsigned SUPER = (1u << LOG_N) / (ROWS * COLS); // 512 super-tiles
constexpr unsigned TEAM = 32;                             // workgroups per team (one per CU of an XCD)

struct Line {
    unsigned v, pad[31];
};
struct Ctl {
    Line slot[8];        // workgroup census per XCD -> (team, index)
    Line done1[8][2];    // phase-1 arrivals per (XCD, team), monotone
    Line done2[8][2];    // phase-2 arrivals
    Line error;
};

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ inline void team_barrier(unsigned *counter, unsigned target, unsigned *err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {
                atomicExch(err, 1u);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: no stale L1 lines of the scratch
}

// phase 1 tile t (0..63) of super-tile st: rows [128 t, 128 t + 128) x 16 columns, HBM -> sc[row][16]
__device__ inline void phase1_tile(const double *in_re, const double *in_im, double *sc_re, double *sc_im, unsigned st, unsigned t) {
    const unsigned col = threadIdx.x & 15u, tau = threadIdx.x >> 4;  // 16 taus, 8 rows each
    double r[8], m[8];
#pragma unroll
    for (unsigned j = 0; j < 8; ++j) {
        const size_t off = ((size_t)(128u * t + tau + 16u * j) << L2B) + (size_t)st * COLS + col;
        r[j] = __builtin_nontemporal_load(in_re + off);
        m[j] = __builtin_nontemporal_load(in_im + off);
    }
#pragma unroll
    for (unsigned j = 0; j < 8; ++j) {
        const size_t off = (size_t)(128u * t + tau + 16u * j) * COLS + col;
        sc_re[off] = r[j] + 1.0;
        sc_im[off] = m[j] + 1.0;
    }
}
// phase 2 wave tile w (0..127): rows w + 128 j (j < 64) x 16 columns, sc -> HBM (same rows of the super-tile)
__device__ inline void phase2_tile(const double *sc_re, const double *sc_im, double *out_re, double *out_im, unsigned st, unsigned w) {
    const unsigned lane = threadIdx.x & 63u, col = lane & 15u, tau = lane >> 4;
    double r[16], m[16];
#pragma unroll
    for (unsigned j = 0; j < 16; ++j) {
        const size_t off = (size_t)(w + 128u * (tau + 4u * j)) * COLS + col;
        r[j] = sc_re[off];
        m[j] = sc_im[off];
    }
#pragma unroll
    for (unsigned j = 0; j < 16; ++j) {
        const size_t off = ((size_t)(w + 128u * (tau + 4u * j)) << L2B) + (size_t)st * COLS + col;
        __builtin_nontemporal_store(r[j] + 1.0, out_re + off);
        __builtin_nontemporal_store(m[j] + 1.0, out_im + off);
    }
}

// the fused double pass: persistent, XCD-local
__global__ void __launch_bounds__(NT) fused_kernel(const double *in_re, const double *in_im, double *out_re, double *out_im,
                                                   double *scratch, Ctl *ctl, unsigned teams) {
    __shared__ unsigned sh_slot;
    const unsigned g = xcc_id();
    if (threadIdx.x == 0) sh_slot = atomicAdd(&ctl->slot[g].v, 1u);
    __syncthreads();
    const unsigned slot = __builtin_amdgcn_readfirstlane(sh_slot);
    const unsigned team = slot / TEAM, idx = slot % TEAM;
    if (team >= teams) return;  // more workgroups landed on this XCD than the plan uses
    double *sc_re = scratch + ((size_t)(g * 2 + team) * 2) * ROWS * COLS, *sc_im = sc_re + (size_t)ROWS * COLS;
    unsigned it = 0;
    for (unsigned st = g + 8u * team; st < SUPER; st += 8u * teams, ++it) {
        phase1_tile(in_re, in_im, sc_re, sc_im, st, idx);
        phase1_tile(in_re, in_im, sc_re, sc_im, st, idx + TEAM);
        team_barrier(&ctl->done1[g][team].v, TEAM * (it + 1), &ctl->error.v);
        phase2_tile(sc_re, sc_im, out_re, out_im, st, idx * 4u + (threadIdx.x >> 6));
        team_barrier(&ctl->done2[g][team].v, TEAM * (it + 1), &ctl->error.v);
    }
}

// the same two phases as two kernels through a full-size scratch in HBM (what two plain passes cost)
__global__ void __launch_bounds__(NT) phase1_kernel(const double *in_re, const double *in_im, double *sc_re, double *sc_im) {
    for (unsigned w = blockIdx.x; w < SUPER * 64u; w += gridDim.x) {
        const unsigned st = w >> 6, t = w & 63u;
        phase1_tile(in_re, in_im, sc_re + (size_t)st * ROWS * COLS, sc_im + (size_t)st * ROWS * COLS, st, t);
    }
}
__global__ void __launch_bounds__(NT) phase2_kernel(const double *sc_re, const double *sc_im, double *out_re, double *out_im) {
    for (unsigned w = blockIdx.x; w < SUPER * 32u; w += gridDim.x) {
        const unsigned st = w >> 5, q = w & 31u;
        phase2_tile(sc_re + (size_t)st * ROWS * COLS, sc_im + (size_t)st * ROWS * COLS, out_re, out_im, st, q * 4u + (threadIdx.x >> 6));
    }
}

__global__ void fill_val(double *p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void checksum(const double *p, size_t n, double *out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    atomicAdd(out, acc);
}

int main() {
    const size_t n = (size_t)1 << LOG_N;
    double *in, *out, *big, *small, *sum;
    Ctl *ctl;
    CK(hipMalloc(&in, 2 * n * 8));
    CK(hipMalloc(&out, 2 * n * 8));
    CK(hipMalloc(&big, 2 * n * 8));                               // full-size scratch of the two-kernel form
    CK(hipMalloc(&small, (size_t)8 * 2 * 2 * ROWS * COLS * 8));   // 8 XCDs x 2 teams x (re, im) x 1 MiB
    CK(hipMalloc(&sum, 8));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    fill_val<<<2048, 256, 0, s>>>(in, 2 * n, 3.0);
    const double want = (3.0 + 2.0) * 2.0 * (double)n;

    auto report = [&](const char *name, float ms) -> int {
        double got = 0;
        CK(hipMemsetAsync(sum, 0, 8, s));
        checksum<<<2048, 256, 0, s>>>(out, 2 * n, sum);
        CK(hipMemcpyAsync(&got, sum, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        Ctl hc;
        CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("%-44s %8.1f us   HBM-side %.2f TB/s (read + write of 2 GiB each)   checksum %s  error=%u\n", name, 1e3 * ms,
               2.0 * 2.0 * n * 8 / (ms * 1e-3) / 1e12, got == want ? "ok" : "MISMATCH", hc.error.v);
        fflush(stdout);
        return 0;
    };

    // (a) two kernels through HBM
    for (int wgcu = 2; wgcu <= 8; wgcu *= 2) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            fill_val<<<2048, 256, 0, s>>>(out, 2 * n, NAN);
            CK(hipEventRecord(e0, s));
            phase1_kernel<<<256 * wgcu, NT, 0, s>>>(in, in + n, big, big + n);
            phase2_kernel<<<256 * wgcu, NT, 0, s>>>(big, big + n, out, out + n);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (rep && t < best) best = t;
        }
        char name[96];
        snprintf(name, sizeof name, "two kernels through HBM, %d wg/cu", wgcu);
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        if (report(name, best)) return 1;
    }
    // (b) fused, XCD-local
    for (unsigned teams = 1; teams <= 2; ++teams)
        for (unsigned extra = 0; extra <= 1; ++extra) {  // extra: over-subscribe the grid (placement slack)
            const unsigned grid = 256u * teams + (extra ? 64u : 0u);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                fill_val<<<2048, 256, 0, s>>>(out, 2 * n, NAN);
                fill_val<<<64, 256, 0, s>>>(small, (size_t)8 * 2 * 2 * ROWS * COLS, NAN);
                CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s));
                CK(hipEventRecord(e0, s));
                fused_kernel<<<grid, NT, 0, s>>>(in, in + n, out, out + n, small, ctl, teams);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (rep && t < best) best = t;
            }
            char name[96];
            snprintf(name, sizeof name, "fused XCD-local, %u team(s)/XCD, grid %u", teams, grid);
            if (report(name, best)) return 1;
            Ctl hc;
            CK(hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            printf("    workgroups per XCD:");
            for (int i = 0; i < 8; ++i) printf(" %u", hc.slot[i].v);
            printf("\n");
        }
    return 0;
}
