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
    // Real transforms of 2N points whose N-point complex core this kernel runs (REAL instantiations only):
    //   1 = R2C: the untangle (algorithms/r2c.rs:150-242) is the epilogue -- input = the real signal read as N
    //       (even, odd) pairs (in_interleaved = 1), output = planar X[0..N] (N + 1 values, out_dist apart);
    //   2 = C2R: the preprocess (algorithms/r2c.rs:263-433) is the prologue -- input = planar X[0..N], output =
    //       the real signal written as N (im, re) pairs of the swap-trick inverse (out_interleaved = 2).
    unsigned real_mode;
    unsigned rtw_bits;
    const void *rtw3;          // [3][1 << rtw_bits] complex: W_{2N}^e
};

template <typename T, int LR, int LC, int LP> struct RowBody {
    using Body = TileBody<T, LR, LC, LP, false, true, false, false>;  // its own park / store: the classic transposing exchange
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
    // (The layout is decided ONCE, outside the unrolled loop, and the pairs' swap after all loads of a chunk are out: with
    //  the branch inside the loop every 16-byte pair load was followed by s_waitcnt vmcnt(0) -- sixteen serialised round trips
    //  to memory per thread, one 4096-point transform on Complex<T> pairs 11.7 us where the planar one takes 8.6,
    //  profiles/r04_interleaved_ladder.log.)
    PHAST_HD static void load_flat(const RowArgs &a, unsigned tile, int tid, Regs &r) {
        const unsigned long long xf0 = (unsigned long long)tile << LC;
        if (!a.in_interleaved) {
            static_for<0, P>([&](auto i) {
                const unsigned f = (unsigned)(decltype(i)::value * NT + tid);
                const unsigned long long xf = xf0 + (f >> LR);
                T re = (T)0, im = (T)0;
                if (xf < a.batch) {
                    const size_t off = (size_t)xf * a.in_dist + (f & (ROWS - 1));
                    re = reinterpret_cast<const T *>(a.in_re)[off];
                    im = reinterpret_cast<const T *>(a.in_im)[off];
                }
                r.re[i] = re;
                r.im[i] = im;
            });
            return;
        }
        const bool swapped = a.in_interleaved == 2;
        constexpr int CH = P < 8 ? P : 8;  // pairs in flight per chunk (their registers come on top of the tile's own)
        static_for<0, P / CH>([&](auto c) {
            constexpr int C0 = decltype(c)::value * CH;
            cx v[CH];
            static_for<0, CH>([&](auto i) {
                const unsigned f = (unsigned)((C0 + decltype(i)::value) * NT + tid);
                const unsigned long long xf = xf0 + (f >> LR);
                cx w;
                w.x = w.y = (T)0;
                if (xf < a.batch) w = reinterpret_cast<const cx *>(a.in_re)[(size_t)xf * a.in_dist + (f & (ROWS - 1))];
                v[decltype(i)::value] = w;
            });
            static_for<0, CH>([&](auto i) {
                constexpr int I = decltype(i)::value;
                r.re[C0 + I] = swapped ? v[I].y : v[I].x;
                r.im[C0 + I] = swapped ? v[I].x : v[I].y;
            });
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

    // ---- R2C epilogue: the transposing exchange has been WRITTEN (plane [transform][k], pitch CS); pairs
    // (k, N - k) of one transform are untangled straight from LDS into the planar output (r2c.rs:150-242) ----
    PHAST_HD static void untangle_store(const RowArgs &a, unsigned tile, int tid, const T *z_re, const T *z_im,
                                        const cx *rtab) {
        constexpr int Q = ROWS / 2, CS = Body::CS;
        const unsigned long long xf0 = (unsigned long long)tile << LC;
        T *out_re = reinterpret_cast<T *>(a.out_re), *out_im = reinterpret_cast<T *>(a.out_im);
        for (int g = tid; g < COLS * Q; g += NT) {  // k in [0, Q): k = 0 is the DC / Nyquist pair
            const int col = g / Q, k = g - col * Q;
            const unsigned long long xf = xf0 + col;
            if (xf >= a.batch) continue;
            const size_t o = (size_t)xf * a.out_dist;
            const T x = z_re[col * CS + k], y = z_im[col * CS + k];
            if (k == 0) {  // r2c.rs:161-166
                out_re[o] = x + y;
                out_im[o] = (T)0;
                out_re[o + ROWS] = x - y;
                out_im[o + ROWS] = (T)0;
                continue;
            }
            T wr, wi;
            tw3_lookup<T>(rtab, a.rtw_bits, (unsigned)k, wr, wi);
            wr *= (T)0.5;
            wi *= (T)0.5;
            const T c = z_re[col * CS + ROWS - k], d = z_im[col * CS + ROWS - k];
            const T s_re = (T)0.5 * (x + c), s_im = (T)0.5 * (y - d), t_re = y + d, t_im = c - x;
            const T wzr = wr * t_re - wi * t_im, wzi = wr * t_im + wi * t_re;
            out_re[o + k] = s_re + wzr;
            out_im[o + k] = s_im + wzi;
            out_re[o + ROWS - k] = s_re - wzr;
            out_im[o + ROWS - k] = wzi - s_im;
        }
        for (int col = tid; col < COLS; col += NT) {  // k = Q: r2c.rs:233-236
            const unsigned long long xf = xf0 + col;
            if (xf >= a.batch) continue;
            T wr, wi;
            tw3_lookup<T>(rtab, a.rtw_bits, (unsigned)Q, wr, wi);
            const T x = z_re[col * CS + Q], y = z_im[col * CS + Q];
            if (Q == 0) continue;
            out_re[(size_t)xf * a.out_dist + Q] = x + wr * y;  // 2 * (0.5 w) = w
            out_im[(size_t)xf * a.out_dist + Q] = wi * y;
        }
    }
    // ---- C2R prologue: z[k] from X[k], X[N - k] (r2c.rs:263-433), parked with re and im swapped -- the
    // swap-trick inverse (algorithms/dit.rs:297-300) runs the forward chain on (z_im, z_re) ----
    PHAST_HD static void c2r_park(const RowArgs &a, unsigned tile, int tid, T *st_re, T *st_im, const cx *rtab) {
        const unsigned long long xf0 = (unsigned long long)tile << LC;
        const T *in_re = reinterpret_cast<const T *>(a.in_re), *in_im = reinterpret_cast<const T *>(a.in_im);
        for (int g = tid; g < COLS * ROWS; g += NT) {
            const int col = g >> LR, k = g & (ROWS - 1);
            const unsigned long long xf = xf0 + col;
            T zr = (T)0, zi = (T)0;
            if (xf < a.batch) {
                const size_t o = (size_t)xf * a.in_dist;
                T c_h, s_h;
                tw3_lookup<T>(rtab, a.rtw_bits, (unsigned)k, c_h, s_h);
                c_h *= (T)0.5;
                s_h *= (T)0.5;
                const T re_first = in_re[o + k], im_first = in_im[o + k];
                const T re_second = in_re[o + ROWS - k], im_second = -in_im[o + ROWS - k];
                const T zx_re = (T)0.5 * (re_first + re_second), zx_im = (T)0.5 * (im_first + im_second);
                const T dr = re_first - re_second, di = im_first - im_second;
                const T zy_re = c_h * dr + s_h * di, zy_im = c_h * di - s_h * dr;
                zr = zx_re - zy_im;
                zi = zx_im + zy_re;
            }
            st_re[col * PITCH + k] = zi;  // swapped on purpose
            st_im[col * PITCH + k] = zr;
        }
    }
};

template <typename T, int LR, int LC, int LP, bool REAL>
__global__ void __launch_bounds__(1 << (LR + LC - LP)) row_fft_kernel(const RowArgs a) {
    using RB = RowBody<T, LR, LC, LP>;
    using Body = typename RB::Body;
    using cx = cx_t<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *buf = reinterpret_cast<T *>(smem);
    cx *l_twr = reinterpret_cast<cx *>(smem + (size_t)2 * RB::PLANE * sizeof(T));
    cx *l_rtw = l_twr + RB::TWR;  // REAL only: W_{2N} three-level tables
    const typename Body::Shared sh{buf, buf + RB::PLANE, nullptr, l_twr};
    const unsigned mode = REAL ? a.real_mode : 0u;

    int tid = threadIdx.x;
    typename Body::Regs r;
    unsigned t = blockIdx.x;
    if (mode != 2 && t < a.tiles_total) RB::load_flat(a, t, tid, r);  // in flight while the tables arrive
    for (int i = tid; i < RB::TWR; i += RB::NT) l_twr[i] = reinterpret_cast<const cx *>(a.twr)[i];
    if constexpr (REAL)
        for (unsigned i = tid; i < (3u << a.rtw_bits); i += RB::NT) l_rtw[i] = reinterpret_cast<const cx *>(a.rtw3)[i];

    auto exchange = [&](auto e) {
        constexpr int E = decltype(e)::value;
        __syncthreads();
        Body::template ex_write<E>(sh, tid, r, 0);
        Body::template ex_write<E>(sh, tid, r, 1);
        __syncthreads();
        if (REAL && E == Body::S && mode == 1) return;  // R2C: the untangle reads the transposed tile from LDS itself
        Body::template ex_read<E>(sh, tid, r, 0);
        Body::template ex_read<E>(sh, tid, r, 1);
    };
    auto do_step = [&](auto i) { Body::template step<decltype(i)::value>(sh, tid, r); };

    while (t < a.tiles_total) {
        asm volatile("" : "+v"(tid));  // per-tile address recomputation instead of ~100 hoisted values (tile_fft.hpp)
        __syncthreads();               // the previous tile's readers are done with the LDS buffer (and the tables are in)
        if (REAL && mode == 2) RB::c2r_park(a, t, tid, buf, buf + RB::PLANE, l_rtw);
        else RB::park(buf, buf + RB::PLANE, tid, r);
        __syncthreads();
        RB::pick(buf, buf + RB::PLANE, tid, r);
        Body::chain(do_step, exchange);
        if (REAL && mode == 1) RB::untangle_store(a, t, tid, buf, buf + RB::PLANE, l_rtw);
        else RB::store_flat(a, t, tid, r);
        t += gridDim.x;
        if (mode != 2 && t < a.tiles_total) RB::load_flat(a, t, tid, r);
    }
}

template <typename T, int LR, int LC, int LP, bool REAL>
hipError_t launch_row_inst(const RowArgs &a, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    using RB = RowBody<T, LR, LC, LP>;
    auto kern = row_fft_kernel<T, LR, LC, LP, REAL>;
    const size_t lds = RB::lds_bytes() + (REAL ? ((size_t)3 << a.rtw_bits) * sizeof(cx_t<T>) : 0);
    static PerDeviceLimit lds_limit;  // raised once per instantiation and device: nothing but the launch in the steady state (graph capture)
    if (hipError_t e = raise_lds_limit(lds_limit, reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
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
#ifndef PHAST_ROW13_LP
// 8192 points in one workgroup: 16 points per thread on 512 threads (round 4; 32 per thread on 256 threads saved an LDS exchange
// but left one wave per SIMD: batches of 2^13 f64 1403 -> 1276 us per 2^27 points, f32 875 -> 718; profiles/r04_row13_ab.log)
#define PHAST_ROW13_LP 4
#endif
#define PHAST_ROW_SHAPES(X)                                                                                     \
    X(1, 8, 1) X(2, 8, 2) X(3, 8, 3) X(4, 8, 4) X(5, 7, 5) X(6, 6, 3) X(7, 5, 3) X(8, 4, 4) X(9, 3, 4) X(10, 2, 4) \
    X(11, 1, 4) X(12, 0, 4) X(13, 0, PHAST_ROW13_LP)

// Real transforms of 64 points (the 32-point core with the untangle as epilogue / the preprocess as prologue) run 16 points per
// thread on 256 threads -- radix 16 x 2, one more LDS exchange -- instead of one radix-32 butterfly per thread on 128: the
// epilogue / prologue is per-point work, and 128 threads per 4096-point tile left one wave per SIMD to hide it.  Same-box A/B
// (profiles/r05_real64_lp4_ab.log, 2^27 / 2^22 real points in flight): c2r_fft_f64 132 -> 199 / 117 -> 167 GSamples/s,
// c2r_fft_f32 264 -> 392 / 136 -> 201, r2c_fft_f32 397 -> 389 (noise) / 213 -> 263, r2c_fft_f64 225 -> 234 / 163 -> 175.
// Plain C2C of 32 points keeps the single butterfly (no epilogue to hide).
constexpr int kRealRow5LP = 4;

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
    using cx = cx_t<T>;
    constexpr int NT = RB::NT;
    T *buf = new T[(size_t)2 * RB::PLANE];
    const typename Body::Shared sh{buf, buf + RB::PLANE, nullptr, reinterpret_cast<const cx *>(a.twr)};
    const cx *rtab = reinterpret_cast<const cx *>(a.rtw3);
    const unsigned mode = a.real_mode;
    Regs *regs = new Regs[NT];
    auto exchange = [&](auto e) {
        constexpr int E = decltype(e)::value;
        for (int t = 0; t < NT; ++t) {
            Body::template ex_write<E>(sh, t, regs[t], 0);
            Body::template ex_write<E>(sh, t, regs[t], 1);
        }
        if (E == Body::S && mode == 1) return;
        for (int t = 0; t < NT; ++t) {
            Body::template ex_read<E>(sh, t, regs[t], 0);
            Body::template ex_read<E>(sh, t, regs[t], 1);
        }
    };
    auto do_step = [&](auto i) {
        for (int t = 0; t < NT; ++t) Body::template step<decltype(i)::value>(sh, t, regs[t]);
    };
    for (unsigned tile = 0; tile < a.tiles_total; ++tile) {
        if (mode == 2) {
            for (int t = 0; t < NT; ++t) RB::c2r_park(a, tile, t, buf, buf + RB::PLANE, rtab);
        } else {
            for (int t = 0; t < NT; ++t) RB::load_flat(a, tile, t, regs[t]);
            for (int t = 0; t < NT; ++t) RB::park(buf, buf + RB::PLANE, t, regs[t]);
        }
        for (int t = 0; t < NT; ++t) RB::pick(buf, buf + RB::PLANE, t, regs[t]);
        Body::chain(do_step, exchange);
        if (mode == 1) {
            for (int t = 0; t < NT; ++t) RB::untangle_store(a, tile, t, buf, buf + RB::PLANE, rtab);
        } else {
            for (int t = 0; t < NT; ++t) RB::store_flat(a, tile, t, regs[t]);
        }
    }
    delete[] regs;
    delete[] buf;
}

}  // namespace phast
