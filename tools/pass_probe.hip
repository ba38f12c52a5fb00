// pass_probe.hip -- ONE parameterised copy model of the transform's passes (round 4: replaces tools/pass_floor{,2..10}.hip,
// strided_copy{,_nt,_pf}.hip, strided_rw.hip and copy_width.hip, whose results live on in profiles/r01_* .. r03_*).
//
// What would a pass cost if the arithmetic were free?  Every pass of an N = 2^(a+b+c)-point transform reads each element of
// the planar re / im arrays once and writes it once, tile by tile, with the access pattern its place in the plan gives it:
//   A: rows 2^(b+c) elements apart, each column leaves as one contiguous run of 2^a elements (the transposing first pass)
//   B: rows 2^a apart inside a block of 2^(a+b), same pattern out        C: rows 2^(a+b) apart, same pattern out
// A workgroup takes tiles of ROWS x COLS points, P per thread, exactly as TileBody does (lane = column fastest, rows j M + tau;
// pass A transposes through the LDS so that its stores are contiguous runs), optionally runs R rounds of dependent FMAs and L LDS round trips (write, barrier, read, barrier) and stores.  The probe
// times every pass alone (execution time: events bound to the dispatch, median over a cold ring) and the chain of the
// three from a HIP graph -- the floor the real plan is to be read against WHEN A WORKGROUP HAS ONE TILE (single transforms of
// 2^19..2^21 points).  With several tiles per workgroup the library's kernels issue the next tile's loads behind the current
// tile's stores and beat this loop nest, which does not (f32 2^23 as 256 x 256 x 128 at 16 points per thread: 33 / 37 / 32 us per
// pass here, 25-28 in the library) -- there the box's copy kernel (csrc/probe.hip, bench.py: stream_probe) is the reference.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pass_probe.hip -o tools/pass_probe.bin
//   tools/pass_probe.bin f32|f64 a b c colsA colsB colsC P [fma_rounds] [lds_trips] [wg_per_cu]      (P = 8 or 16 points per thread;
//   a 32-point instantiation of this generic loop nest spills -- the library's 32-point kernels needed their own care)
//   e.g. the f32 2^20 single-transform plan 64x64 / 256x16 / 64x64 at 8 points per thread:   f32 6 8 6 64 16 64 8
//        its middle pass with 128-byte rows:                                                  f32 6 8 6 64 32 64 8
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

struct PassArgs {
    unsigned lr, lc;                 // tile: 2^lr rows x 2^lc columns
    unsigned lo_bits;                // columns are contiguous within 2^lo_bits ...
    unsigned long long hi_stride;    // ... and blocks of them this far apart
    unsigned long long row_stride;   // elements between consecutive rows
    unsigned transpose;              // 1: column g leaves as the contiguous run [g 2^lr, (g + 1) 2^lr)
    unsigned tiles;                  // columns / 2^lc
    int rounds, trips;
    unsigned rev_loads;              // 1: lane c loads column COLS - 1 - c (descending addresses across the lanes, as mirrored loads do)
};

// (threads per workgroup = rows x cols / P: up to 1024 at 8 points per thread, 512 at 16 / 32 -- with 1024 allowed the 16- and
// 32-point instantiations are held to 128 VGPRs and spill their points to scratch memory: a copy model of nothing)
template <int P> constexpr int probe_max_threads() { return P <= 8 ? 1024 : 512; }
template <typename T, int P> __global__ void __launch_bounds__(probe_max_threads<P>()) probe_pass(const T *in_re, const T *in_im, T *out_re, T *out_im, PassArgs a) {
    extern __shared__ unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    const unsigned nt = blockDim.x, tid = threadIdx.x, cols = 1u << a.lc, m = nt >> a.lc;  // m = threads per column
    const unsigned col = tid & (cols - 1), tau = tid >> a.lc;
    for (unsigned t = blockIdx.x; t < a.tiles; t += gridDim.x) {
        const unsigned tile = ((a.tiles & 7u) == 0u) ? (t & 7u) * (a.tiles >> 3) + (t >> 3) : t;  // XCD-aware, as TileBody::locate
        const unsigned g = (tile << a.lc) + col, gl = (tile << a.lc) + (a.rev_loads ? cols - 1u - col : col);
        const size_t cbase = (size_t)(g >> a.lo_bits) * a.hi_stride + (g & ((1u << a.lo_bits) - 1u));
        const size_t lbase = (size_t)(gl >> a.lo_bits) * a.hi_stride + (gl & ((1u << a.lo_bits) - 1u));
        T re[P], im[P];
        {
            const size_t step = (size_t)m * a.row_stride;  // running pointers: one 64-bit address per plane, not one per element
            const T *pr = in_re + lbase + (size_t)tau * a.row_stride, *pi = in_im + lbase + (size_t)tau * a.row_stride;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                re[j] = __builtin_nontemporal_load(pr);
                im[j] = __builtin_nontemporal_load(pi);
                pr += step;
                pi += step;
            }
        }
        for (int k = 0; k < a.rounds; ++k) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                re[j] = __builtin_fma(re[j], (T)1.0000001, (T)1e-9);
                im[j] = __builtin_fma(im[j], (T)1.0000001, (T)1e-9);
            }
        }
        for (int k = 0; k < a.trips; ++k) {  // one plane at a time, as the wide tiles exchange
#pragma unroll
            for (int j = 0; j < P; ++j) smem[j * nt + tid] = re[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < P; ++j) re[j] = smem[j * nt + (tid ^ 1u)];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < P; ++j) smem[j * nt + tid] = im[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < P; ++j) im[j] = smem[j * nt + (tid ^ 1u)];
            __syncthreads();
        }
        if (a.transpose) {
            // as the real first pass: through the LDS as [col][row], so that every store instruction of a wave writes one
            // contiguous run of a column (flat element f = j nt + tid -> column f >> lr, row f & (rows - 1))
            const unsigned rows = 1u << a.lr;
            const size_t obase = (size_t)(tile << a.lc) << a.lr;
            auto one_plane = [&](const T(&v)[P], T *out) {  // (the plane is fixed at compile time: a run-time plane index would
                __syncthreads();                              // make re / im one dynamically indexed array, i.e. scratch memory)
#pragma unroll
                for (int j = 0; j < P; ++j) smem[col * (rows + 1) + (j * m + tau)] = v[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const unsigned f = (unsigned)j * nt + tid, c = f >> a.lr, r = f & (rows - 1);
                    __builtin_nontemporal_store(smem[c * (rows + 1) + r], out + obase + f);
                }
            };
            one_plane(re, out_re);
            one_plane(im, out_im);
        } else {
            const size_t step = (size_t)m * a.row_stride;
            T *qr = out_re + cbase + (size_t)tau * a.row_stride, *qi = out_im + cbase + (size_t)tau * a.row_stride;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                __builtin_nontemporal_store(re[j], qr);
                __builtin_nontemporal_store(im[j], qi);
                qr += step;
                qi += step;
            }
        }
    }
}

template <typename T, int P> static void launch(const T *ir, const T *ii, T *orr, T *oi, const PassArgs &a, unsigned grid, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const unsigned nt = (1u << (a.lr + a.lc)) / P;
    if ((int)nt > probe_max_threads<P>() || nt < 64) {
        std::printf("tile of %u x %u points at %d per thread needs %u threads: not a shape of this probe\n", 1u << a.lr, 1u << a.lc, P, nt);
        std::exit(2);
    }
    const size_t lds = std::max<size_t>(a.trips ? (size_t)P * nt * sizeof(T) : 0, a.transpose ? (((size_t)1 << a.lr) + 1) * ((size_t)1 << a.lc) * sizeof(T) : 0);
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe_pass<T, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (e0) hipExtLaunchKernelGGL((probe_pass<T, P>), dim3(grid), dim3(nt), (uint32_t)lds, s, e0, e1, 0, ir, ii, orr, oi, a);
    else hipLaunchKernelGGL((probe_pass<T, P>), dim3(grid), dim3(nt), lds, s, ir, ii, orr, oi, a);
}

template <typename T> static int run(int argc, char **argv) {
    const unsigned a = std::atoi(argv[2]), b = std::atoi(argv[3]), c = std::atoi(argv[4]);
    const unsigned cols[3] = {(unsigned)std::atoi(argv[5]), (unsigned)std::atoi(argv[6]), (unsigned)std::atoi(argv[7])};
    const int P = std::atoi(argv[8]) == 8 ? 8 : 16, rounds = argc > 9 ? std::atoi(argv[9]) : 0, trips = argc > 10 ? std::atoi(argv[10]) : 0;
    const unsigned wg_per_cu = argc > 11 ? std::atoi(argv[11]) : 4, L = a + b + c;
    const size_t n = (size_t)1 << L;
    const int ring = (int)std::max<size_t>(4, std::min<size_t>(48, ((size_t)768 << 20) / (2 * n * sizeof(T))));
    T *x, *y;
    CK(hipMalloc((void **)&x, (size_t)ring * 2 * n * sizeof(T)));
    CK(hipMalloc((void **)&y, (size_t)ring * 2 * n * sizeof(T)));
    CK(hipMemset(x, 0, (size_t)ring * 2 * n * sizeof(T)));
    CK(hipMemset(y, 0, (size_t)ring * 2 * n * sizeof(T)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    PassArgs ps[3];
    const unsigned lrs[3] = {a, b, c};
    for (int i = 0; i < 3; ++i) {
        unsigned lc = 0;
        while ((1u << lc) < cols[i]) ++lc;
        ps[i].lr = lrs[i];
        ps[i].lc = lc;
        ps[i].rounds = rounds;
        ps[i].trips = trips;
        ps[i].rev_loads = std::getenv("PHAST_PROBE_REV") ? 1u : 0u;
        ps[i].transpose = i == 0;
        ps[i].row_stride = i == 0 ? (1ull << (b + c)) : i == 1 ? (1ull << a) : (1ull << (a + b));
        ps[i].lo_bits = i == 0 ? b + c : i == 1 ? a : a + b;
        ps[i].hi_stride = i == 1 ? (1ull << (a + b)) : 0;
        ps[i].tiles = (unsigned)((n >> lrs[i]) >> lc);
    }
    auto go = [&](int i, const T *in, T *out, hipEvent_t e0, hipEvent_t e1) {
        const unsigned grid = std::min(ps[i].tiles, 256u * wg_per_cu) & (ps[i].tiles >= 8 ? ~7u : ~0u);
        if (P == 8) launch<T, 8>(in, in + n, out, out + n, ps[i], grid, s, e0, e1);
        else launch<T, 16>(in, in + n, out, out + n, ps[i], grid, s, e0, e1);
    };
    std::printf("%s N = 2^%u = 2^%u x 2^%u x 2^%u, %d points per thread, %d FMA rounds, %d LDS trips, %u workgroups per CU, ring of %d transforms\n",
                sizeof(T) == 4 ? "f32" : "f64", L, a, b, c, P, rounds, trips, wg_per_cu, ring);
    std::vector<hipEvent_t> ev(2 * ring);
    for (auto &e : ev) CK(hipEventCreate(&e));
    double sum = 0;
    for (int i = 0; i < 3; ++i) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            for (int k = 0; k < ring; ++k) go(i, x + (size_t)k * 2 * n, y + (size_t)k * 2 * n, ev[2 * k], ev[2 * k + 1]);
            CK(hipStreamSynchronize(s));
            std::vector<float> ts;
            for (int k = 0; k < ring; ++k) {
                float t;
                CK(hipEventElapsedTime(&t, ev[2 * k], ev[2 * k + 1]));
                ts.push_back(t);
            }
            std::sort(ts.begin(), ts.end());
            best = std::min(best, 1e3f * ts[ring / 2]);
        }
        const double bytes = 4.0 * n * sizeof(T);
        std::printf("  pass %c: %4u x %-4u tiles (%5u of them, rows of %4zu bytes, %4u threads): %7.2f us = %5.2f TB/s\n", "ABC"[i], 1u << ps[i].lr,
                    1u << ps[i].lc, ps[i].tiles, ((size_t)1 << ps[i].lc) * sizeof(T), (1u << (ps[i].lr + ps[i].lc)) / P, best, bytes / best / 1e6);
        sum += best;
    }
    // the chain x -> y -> x -> y of every transform of the ring, from one graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < ring; ++k) {
        T *xx = x + (size_t)k * 2 * n, *yy = y + (size_t)k * 2 * n;
        go(0, xx, yy, nullptr, nullptr);
        go(1, yy, yy, nullptr, nullptr);
        go(2, yy, xx, nullptr, nullptr);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t w0, w1;
    CK(hipEventCreate(&w0));
    CK(hipEventCreate(&w1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(w0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(w1, s));
        CK(hipEventSynchronize(w1));
        float t;
        CK(hipEventElapsedTime(&t, w0, w1));
        best = std::min(best, 1e3f * t / ring);
    }
    std::printf("  sum of the three executions %7.2f us; chained from a graph %7.2f us per transform = %6.1f GSamples/s\n", sum, best, n / best / 1e3);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 9) {
        std::printf("usage: %s f32|f64 a b c colsA colsB colsC P [fma_rounds] [lds_trips] [wg_per_cu]\n", argv[0]);
        return 2;
    }
    return std::strcmp(argv[1], "f32") == 0 ? run<float>(argc, argv) : run<double>(argc, argv);
}
