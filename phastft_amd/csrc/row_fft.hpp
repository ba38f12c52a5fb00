// row_fft.hpp -- batches of SMALL contiguous transforms (N = 2 .. 2048): one pass, whole transforms on chip.
//
// GPU counterpart of the reference's L1-resident leaf (algorithms/dit.rs:44-65: codelet + stages on a block
// that fits L1; algorithms/bravo.rs:225-251 for its bit reversal).  A workgroup owns COLS whole transforms of
// ROWS = N points (4096 points for N >= 64) and runs them through the SAME register/LDS machinery as a first
// pass of the large transforms -- TileBody<T, LR, LC, LP, PRE_TW = false, TRANSPOSE = true> (tile_fft.hpp):
// the DIF digit chain in registers, and the transposing exchange that puts every transform's outputs in
// natural order so that they leave as one contiguous run.  What differs is the way in: the transforms are
// contiguous in memory (a "column" of the tile is contiguous, the FFT axis is the fast one), so the tile is
// read with fully coalesced flat loads, parked in LDS as [transform][point] with an odd row pitch, and picked up
// from there in the (column-fastest) register layout of the chain -- bank-conflict free both ways.
//
// One pass over the data: 4*sizeof(T) bytes per complex sample, the algorithmic minimum.  N <= 32 runs as one
// radix-N butterfly per thread (no exchange besides the transposing one).
#pragma once

#include "tile_fft.hpp"

namespace phast {

struct RowArgs {
    const void *in_re;
    const void *in_im;  // unused when in_interleaved
    void *out_re;
    void *out_im;       // unused when out_interleaved
    const void *twr;    // [32 + max(64, N/32)] complex: W_N^j, W_N^(32 j)   (plan.hpp: host_twr)
    unsigned long long in_dist;   // elements between consecutive transforms
    unsigned long long out_dist;
    unsigned long long batch;
    unsigned tiles_total;
    unsigned in_interleaved;   // 0 planar, 1 (re, im) pairs, 2 (im, re) pairs
    unsigned out_interleaved;
    double scale;
};

template <typename T, int LR, int LC, int LP> struct RowBody {
    using Body = TileBody<T, LR, LC, LP, false, true, false>;
    using Regs = typename Body::Regs;
    using Shared = typename Body::Shared;
    using cx = cx_t<T>;
    static constexpr int ROWS = Body::ROWS, COLS = Body::COLS, P = Body::P, NT = Body::NT, M = Body::M;
    // row pitch of the parked tile: a 32-lane group of pick() covers 32/G transforms x G consecutive points, so
    // consecutive transforms must land G banks apart (N >= 32); below that one pad word per transform keeps the
    // pitch odd (2-way conflicts remain on the flat side for N <= 16 -- a pad every N < 32 words cannot avoid them)
    static constexpr int PITCH = ROWS >= 32 ? ROWS + Body::G : ROWS + 1;
    static constexpr int STAGE = COLS * PITCH;          // one plane of the parked tile
    static constexpr int PLANE = STAGE > Body::EXCH ? STAGE : Body::EXCH;  // the parked tile and the exchanges share LDS
    static constexpr int TWR = 32 + (ROWS / 32 > 64 ? ROWS / 32 : 64);  // plan.hpp: twr_entries

    static size_t lds_bytes() { return (size_t)2 * PLANE * sizeof(T) + TWR * sizeof(cx); }

    // flat element f = i*NT + tid of the tile (i < P): transform f >> LR of the tile, point f & (ROWS - 1)
    PHAST_HD static void load_flat(const RowArgs &a, unsigned tile, int tid, Regs &r) {
        const unsigned long long xf0 = (unsigned long long)tile << LC;
        static_for<0, P>([&](auto i) {
            const unsigned f = (unsigned)(decltype(i)::value * NT + tid);
            const unsigned long long xf = xf0 + (f >> LR);
            T re = (T)0, im = (T)0;
            if (xf < a.batch) {
                const size_t off = (size_t)xf * a.in_dist + (f & (ROWS - 1));
                if (!a.in_interleaved) {
                    re = reinterpret_cast<const T *>(a.in_re)[off];
                    im = reinterpret_cast<const T *>(a.in_im)[off];
                } else {
                    const cx v = reinterpret_cast<const cx *>(a.in_re)[off];
                    re = a.in_interleaved == 2 ? v.y : v.x;
                    im = a.in_interleaved == 2 ? v.x : v.y;
                }
            }
            r.re[i] = re;
            r.im[i] = im;
        });
    }
    PHAST_HD static void park(T *st_re, T *st_im, int tid, const Regs &r) {
        static_for<0, P>([&](auto i) {
            const int f = decltype(i)::value * NT + tid;
            const int at = (f >> LR) * PITCH + (f & (ROWS - 1));
            st_re[at] = r.re[i];
            st_im[at] = r.im[i];
        });
    }
    // register j of thread (col, tau) <- point j*M + tau of transform col (TileBody's load layout)
    PHAST_HD static void pick(const T *st_re, const T *st_im, int tid, Regs &r) {
        const int at0 = Body::col_of(tid) * PITCH + Body::tau_of(tid);
        static_for<0, P>([&](auto j) {
            r.re[j] = st_re[at0 + decltype(j)::value * M];
            r.im[j] = st_im[at0 + decltype(j)::value * M];
        });
    }
    // after the transposing exchange register Q holds flat element Q*NT + tid of the [transform][frequency] tile
    PHAST_HD static void store_flat(const RowArgs &a, unsigned tile, int tid, const Regs &r) {
        const unsigned long long xf0 = (unsigned long long)tile << LC;
        const T scale = (T)a.scale;
        static_for<0, P>([&](auto Q) {
            const unsigned f = (unsigned)(decltype(Q)::value * NT + tid);
            const unsigned long long xf = xf0 + (f >> LR);
            if (xf < a.batch) {
                const size_t off = (size_t)xf * a.out_dist + (f & (ROWS - 1));
                const T re = r.re[Q] * scale, im = r.im[Q] * scale;
                if (!a.out_interleaved) {
                    reinterpret_cast<T *>(a.out_re)[off] = re;
                    reinterpret_cast<T *>(a.out_im)[off] = im;
                } else {
                    cx v;
                    v.x = a.out_interleaved == 2 ? im : re;
                    v.y = a.out_interleaved == 2 ? re : im;
                    reinterpret_cast<cx *>(a.out_re)[off] = v;
                }
            }
        });
    }
};

template <typename T, int LR, int LC, int LP>
__global__ void __launch_bounds__(1 << (LR + LC - LP)) row_fft_kernel(const RowArgs a) {
    using RB = RowBody<T, LR, LC, LP>;
    using Body = typename RB::Body;
    using cx = cx_t<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *buf = reinterpret_cast<T *>(smem);
    cx *l_twr = reinterpret_cast<cx *>(smem + (size_t)2 * RB::PLANE * sizeof(T));
    const typename Body::Shared sh{buf, buf + RB::PLANE, nullptr, l_twr};

    int tid = threadIdx.x;
    typename Body::Regs r;
    unsigned t = blockIdx.x;
    if (t < a.tiles_total) RB::load_flat(a, t, tid, r);  // in flight while the table arrives
    for (int i = tid; i < RB::TWR; i += RB::NT) l_twr[i] = reinterpret_cast<const cx *>(a.twr)[i];

    auto exchange = [&](auto e) {
        constexpr int E = decltype(e)::value;
        __syncthreads();
        Body::template ex_write<E>(sh, tid, r, 0);
        Body::template ex_write<E>(sh, tid, r, 1);
        __syncthreads();
        Body::template ex_read<E>(sh, tid, r, 0);
        Body::template ex_read<E>(sh, tid, r, 1);
    };
    auto do_step = [&](auto i) { Body::template step<decltype(i)::value>(sh, tid, r); };

    while (t < a.tiles_total) {
        asm volatile("" : "+v"(tid));  // per-tile address recomputation instead of ~100 hoisted values (tile_fft.hpp)
        __syncthreads();               // the previous tile's readers are done with the LDS buffer
        RB::park(buf, buf + RB::PLANE, tid, r);
        __syncthreads();
        RB::pick(buf, buf + RB::PLANE, tid, r);
        Body::chain(do_step, exchange);
        RB::store_flat(a, t, tid, r);
        t += gridDim.x;
        if (t < a.tiles_total) RB::load_flat(a, t, tid, r);
    }
}

template <typename T, int LR, int LC, int LP>
hipError_t launch_row_inst(const RowArgs &a, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    using RB = RowBody<T, LR, LC, LP>;
    auto kern = row_fft_kernel<T, LR, LC, LP>;
    const size_t lds = RB::lds_bytes();
    static bool raised = false;  // once per instantiation: nothing but the launch in the steady state (graph capture)
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        raised = true;
    }
    // persistent workgroups: enough to fill the chip several times over, grid-stride over the tiles
    const unsigned per_cu = (unsigned)((160 * 1024) / lds) < 8u ? (unsigned)((160 * 1024) / lds) : 8u;
    unsigned grid = 256u * (per_cu ? per_cu : 1u);
    if (grid > a.tiles_total) grid = a.tiles_total;
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(RB::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RB::NT), lds, stream, a);
    return hipGetLastError();
}

// log2 N -> (log2 transforms per tile, log2 points per thread): 4096-point tiles from N = 64 up (8 points per
// thread below 256, 16 from there); N <= 32 is one butterfly per thread, 256 (N = 32: 128) threads; N = 8192 is
// one transform per workgroup with 32 points per thread.
#define PHAST_ROW_SHAPES(X)                                                                                     \
    X(1, 8, 1) X(2, 8, 2) X(3, 8, 3) X(4, 8, 4) X(5, 7, 5) X(6, 6, 3) X(7, 5, 3) X(8, 4, 4) X(9, 3, 4) X(10, 2, 4) \
    X(11, 1, 4) X(12, 0, 4) X(13, 0, 5)

inline unsigned row_tile_cols_log(unsigned log_n) {
#define PHAST_ROW_LC(LR_, LC_, LP_) \
    if (log_n == LR_) return LC_;
    PHAST_ROW_SHAPES(PHAST_ROW_LC)
#undef PHAST_ROW_LC
    return 0;
}

// Thread-by-thread host execution (tests/test_emulator.py): the same RowBody / TileBody phases.
template <typename T, int LR, int LC, int LP> void emulate_row_fft(const RowArgs &a) {
    using RB = RowBody<T, LR, LC, LP>;
    using Body = typename RB::Body;
    using Regs = typename Body::Regs;
    constexpr int NT = RB::NT;
    T *buf = new T[(size_t)2 * RB::PLANE];
    const typename Body::Shared sh{buf, buf + RB::PLANE, nullptr, reinterpret_cast<const cx_t<T> *>(a.twr)};
    Regs *regs = new Regs[NT];
    auto exchange = [&](auto e) {
        constexpr int E = decltype(e)::value;
        for (int t = 0; t < NT; ++t) {
            Body::template ex_write<E>(sh, t, regs[t], 0);
            Body::template ex_write<E>(sh, t, regs[t], 1);
        }
        for (int t = 0; t < NT; ++t) {
            Body::template ex_read<E>(sh, t, regs[t], 0);
            Body::template ex_read<E>(sh, t, regs[t], 1);
        }
    };
    auto do_step = [&](auto i) {
        for (int t = 0; t < NT; ++t) Body::template step<decltype(i)::value>(sh, t, regs[t]);
    };
    for (unsigned tile = 0; tile < a.tiles_total; ++tile) {
        for (int t = 0; t < NT; ++t) RB::load_flat(a, tile, t, regs[t]);
        for (int t = 0; t < NT; ++t) RB::park(buf, buf + RB::PLANE, t, regs[t]);
        for (int t = 0; t < NT; ++t) RB::pick(buf, buf + RB::PLANE, t, regs[t]);
        Body::chain(do_step, exchange);
        for (int t = 0; t < NT; ++t) RB::store_flat(a, tile, t, regs[t]);
    }
    delete[] regs;
    delete[] buf;
}

}  // namespace phast
