// r2c_fused.hpp -- the LAST pass of the inner N/2-point transform of a real FFT with the untangle fused into it
// (algorithms/r2c.rs:150-242: simd_untangle_inplace_*; round 3).  Round 2 ran the untangle as a sweep of its own over
// the N/2+1 outputs -- a quarter of BASELINE configs[3] (R2C f32, N = 2^24: three passes + the untangle, 4 x 128 MiB).
//
// The untangle pairs Z[k] with Z[h - k], h = N/2.  In the last pass a tile holds ALL rows kc of its columns g
// (k = kc M + g, M = 2^log_s_in columns, R = 2^LR rows), and h - k = (R - 1 - kc) M + (M - g): the partner lives in the
// MIRRORED column M - g, so no aligned tiling of the columns is closed under the pairing (an interval [g0, g0 + C) mirrors to
// (M - g0 - C, M - g0]: always off by one; profiles/HISTORY.md section 10 has the dead ends this leads to: orphan columns, partial
// lines).  What makes the fusion cheap is an identity:
//
//     conj Z[R - 1 - kc][M - g]  =  sum_u conj(s[u][M - g]) W_N'^(u g) W_R^(u kc)           (N' = R M = h)
//
// i.e. the partner value -- conjugated, and with the rows already reversed -- is the SAME pass computation (same
// pre-twiddle exponent u g, same R-point FFT) applied to the conjugated input of the mirrored column: the two phase ramps
// W_R^(+u) (row reversal R - 1 - kc) and W_R^(-u) (W_N'^(-u M)) cancel.  So a thread runs the tile pipeline twice, once on
// column g and once on the conjugate of column M - g (for g = 0: of column 0 itself, whose partner row is R - kc), and
// finds Z[k] and conj Z[h - k] in the same register of the same thread: the untangle is thread-local -- no LDS parking, no
// exchange between workgroups, no state carried between tiles.  The thread then stores BOTH X[k] and X[h - k].
//
//   * tiles: only the column blocks g0 < M/2 are processed (each writes itself and its mirror), plus the block at M/2
//     for the self-mirrored column M/2 alone (lane c = 0 stores, nothing mirrored): M / (2 COLS) + 1 tiles per transform;
//   * the mirrored side is read and written as the columns (M - g0 - COLS, M - g0], one element off the 128-byte
//     alignment: a row is 124 + 4 bytes (f32) in two lines.  The other part of each line belongs to the neighbouring
//     tile, which the XCD-aware order runs on the same XCD at about the same time: plain (not non-temporal) stores let
//     the XCD's L2 merge them;
//   * g = 0: X[0] = (a + b, 0) and X[h] = (a - b, 0) are what the general formula gives for k = 0 (exactly: t_im = 0
//     and w = 0.5), and k = h/2 (column 0, row R/2) pairs with itself; the thread of (kc = 0, g = 0) also stores X[h],
//     the other rows of column 0 are each other's partners and store only themselves;
//   * twiddle 0.5 W_N^k, N = 2h: W_N^(kc M + g) = W_N^g W_2R^kc -- one three-level look-up per thread and tile and one
//     entry of a small table (R entries, read through the L2) per element; no rotation recurrence, no drift.
//
// Registers: two tiles' worth of points per thread, so the fusion exists for last passes with <= 16 points per thread
// (every single-transform plan: the latency / single plans; batched throughput plans with 32 points per thread keep the
// separate untangle sweep).
#pragma once

#include "tile_fft.hpp"

namespace phast {

struct R2cFuseArgs {
    const void *tw3n;   // [3][1 << twn_bits] complex: W_N^e, N = 2 h (the R2C planner's table)
    const void *twu;    // [R] complex: W_{2R}^kc
    unsigned twn_bits;
    unsigned tiles_per_xform;  // M / (2 COLS) + 1
    unsigned tiles_total;      // batch * tiles_per_xform
    unsigned pair_tiles;       // batch * (tiles_per_xform - 1): the tiles with g0 < M/2 (locate)
};

template <typename T, int LR, int LC, int LP, bool SEQ> struct R2cLastBody {
    using Body = TileBody<T, LR, LC, LP, true, false, SEQ>;
    using Regs = typename Body::Regs;
    using cx = cx_t<T>;
    static constexpr int P = Body::P, M = Body::M, COLS = Body::COLS, ROWS = Body::ROWS;

    // position t of the launch -> (transform, first column, is it the self-mirrored block at M/2).
    // The pair tiles (g0 < M/2: M / (2 COLS) per transform, a power of two) come first, in the XCD-aware order of
    // TileBody::locate: workgroup b runs on XCD b % 8 and XCD x gets the contiguous run [x pair_tiles / 8, ...) of them, so
    // that a tile and its neighbours -- which share the split lines of the mirrored side, 124 + 4 bytes -- are in flight in
    // ONE L2 at the same time (the 4-byte piece of a line is then an L2 hit on the way in and merges with the 124-byte
    // piece on the way out).  The self-mirrored blocks (one per transform) follow: they share nothing.  Round 3 applied the
    // XCD order to all tiles_total = batch (M / (2 COLS) + 1) tiles -- an ODD number for one transform, for which the order
    // falls back to t itself: neighbouring tiles on DIFFERENT XCDs, every split line fetched twice and written as two
    // partial lines (PMC: 1.32 x the algorithmic traffic at f32 2^24, profiles/r03_pmc_hbm_traffic_r2c_f32_2p24.txt).
    PHAST_HD static bool locate(const TileArgs &a, const R2cFuseArgs &f, unsigned t, Regs &r) {
        const unsigned pairs_per = f.tiles_per_xform - 1u;
        if (t >= f.pair_tiles) {
            r.xform = t - f.pair_tiles;
            r.g0 = pairs_per << LC;  // M/2
            return true;
        }
        const unsigned tile = ((f.pair_tiles & 7u) == 0u) ? (t & 7u) * (f.pair_tiles >> 3) + (t >> 3) : t;
        r.xform = tile >> (unsigned)__builtin_ctz(pairs_per);
        r.g0 = (tile & (pairs_per - 1u)) << LC;
        return false;
    }

    // the conjugate of the mirrored columns: lane (col, tau) reads column M - (g0 + col) (column 0 for g = 0) of the rows
    // the plain load reads
    PHAST_HD static void load_mirror(const TileArgs &a, int tid, Regs &r) {
        const int col = Body::col_of(tid), tau = Body::tau_of(tid);
        const unsigned mcols = 1u << a.log_s_in, g = r.g0 + (unsigned)col, gm = g ? mcols - g : 0u;
        const unsigned lo = gm & ((1u << a.in_lo_bits) - 1u), mid = (gm >> a.in_lo_bits) & ((1u << (a.log_s_in - a.in_lo_bits)) - 1u);
        const size_t ubase = (size_t)r.xform * a.in_dist;
        const unsigned voff = (unsigned)tau * (unsigned)a.in_row_stride + mid * (unsigned)a.in_mid_stride + lo;
        const T *pr = reinterpret_cast<const T *>(a.in_re) + ubase;
        const T *pi = reinterpret_cast<const T *>(a.in_im) + ubase;
        static_for<0, P>([&](auto j) {
            const size_t urow = (size_t)(decltype(j)::value * M) * a.in_row_stride;
            r.re[j] = (pr + urow)[voff];
            r.im[j] = -(pi + urow)[voff];
        });
    }

    // 0.5 W_N^g of this thread's column (three table entries, multiplied by the caller)
    PHAST_HD static Tw3Raw<T> col_twiddle_fetch(const R2cFuseArgs &f, int tid, const Regs &r) {
        return tw3_fetch<T>(reinterpret_cast<const cx *>(f.tw3n), f.twn_bits, r.g0 + (unsigned)Body::col_last(tid));
    }

    // z = Z[k] of this thread's registers, p = conj Z[h - k]: X[k] -> z, X[h - k] -> p (algorithms/r2c.rs:177-231)
    PHAST_HD static void untangle(const R2cFuseArgs &f, int tid, const Tw3Raw<T> &wg_raw, Regs &z, Regs &p) {
        T gr, gi;
        tw3_combine<T>(wg_raw, gr, gi);
        gr *= (T)0.5;
        gi *= (T)0.5;
        const cx *twu = reinterpret_cast<const cx *>(f.twu);
        const unsigned kl = Body::krow_lane(tid);
        // the table entries FOUR at a time: asked for one by one, next to their use, every one of the P loads (L1 / L2 hits, but
        // a few hundred cycles each) was waited for before the next went out -- 16 waits in a row per tile pair in the ISA of
        // the 16-point f64 kernels (round 4); two tiles' points leave no room for all P entries at once
        constexpr int UC = P < 4 ? P : 4;
        static_for<0, P / UC>([&](auto c) {
            constexpr int C0 = decltype(c)::value * UC;
            cx u[UC];
            static_for<0, UC>([&](auto i) { u[decltype(i)::value] = twu[kl + Body::template krow_const<C0 + decltype(i)::value>()]; });
            static_for<0, UC>([&](auto i) {
                constexpr int I = decltype(i)::value, Q = C0 + I;
                const T wr = gr * u[I].x - gi * u[I].y, wi = gr * u[I].y + gi * u[I].x;  // 0.5 W_N^(kc M + g)
                const T a_ = z.re[Q], b_ = z.im[Q], c_ = p.re[Q], d_ = -p.im[Q];
                const T s_re = (T)0.5 * (a_ + c_), s_im = (T)0.5 * (b_ - d_);
                const T t_re = b_ + d_, t_im = c_ - a_;
                const T wzr = wr * t_re - wi * t_im, wzi = wr * t_im + wi * t_re;
                z.re[Q] = s_re + wzr;
                z.im[Q] = s_im + wzi;
                p.re[Q] = s_re - wzr;
                p.im[Q] = wzi - s_im;
            });
        });
    }

    // X[k] rows of the tile's own columns (aligned, as TileBody::store); `only_col0`: the block at M/2 stores column M/2 alone
    PHAST_HD static void store_own(const TileArgs &a, int tid, const Regs &z, bool only_col0) {
        if (only_col0 && Body::col_last(tid) != 0) return;
        const size_t base = (size_t)z.xform * a.out_dist + z.g0;
        const unsigned voff = (unsigned)Body::col_last(tid) + Body::krow_lane(tid) * (unsigned)a.out_row_stride;
        T *ore = reinterpret_cast<T *>(a.out_re) + base, *oim = reinterpret_cast<T *>(a.out_im) + base;
        static_for<0, P>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const size_t urow = (size_t)Body::template krow_const<Q>() * a.out_row_stride;
            (ore + urow)[voff] = z.re[Q];
            (oim + urow)[voff] = z.im[Q];
        });
    }
    // X[h - k]: row R - 1 - kc, column M - g (g >= 1); for g = 0 only X[h] (kc = 0) -- the other rows of column 0 are owned
    PHAST_HD static void store_mirror(const TileArgs &a, int tid, const Regs &p) {
        const unsigned mcols = 1u << a.log_s_in, g = p.g0 + (unsigned)Body::col_last(tid);
        const unsigned kl = Body::krow_lane(tid);
        T *ore = reinterpret_cast<T *>(a.out_re) + (size_t)p.xform * a.out_dist;
        T *oim = reinterpret_cast<T *>(a.out_im) + (size_t)p.xform * a.out_dist;
        static_for<0, P>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const unsigned kc = kl + Body::template krow_const<Q>();
            if (g != 0u || kc == 0u) {
                const size_t at = (size_t)((unsigned)ROWS - 1u - kc) * a.out_row_stride + (mcols - g);  // g = 0, kc = 0: R M = h
                ore[at] = p.re[Q];
                oim[at] = p.im[Q];
            }
        });
    }
};

template <typename T, int LR, int LC, int LP, bool SEQ>
__global__ void __launch_bounds__(1 << (LR + LC - LP)) r2c_last_pass_kernel(const TileArgs a, const R2cFuseArgs f) {
    using RB = R2cLastBody<T, LR, LC, LP, SEQ>;
    using Body = typename RB::Body;
    using cx = cx_t<T>;
    constexpr int NT = Body::NT;
    pin_tile_args(a);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *ex_re = reinterpret_cast<T *>(smem);
    cx *l_tw3 = reinterpret_cast<cx *>(smem + (size_t)Body::EXCH * sizeof(T) * (Body::PLANE_SEQ ? 1 : 2));
    cx *l_twr = l_tw3 + (3u << a.tw_bits);
    const typename Body::Shared sh{ex_re, Body::PLANE_SEQ ? ex_re : ex_re + Body::EXCH, l_tw3, l_twr};

    int tid = threadIdx.x;
    unsigned wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    auto fresh_tid = [&]() {  // see tile_fft_kernel
        asm volatile("" : "+s"(wave_base));
        return (int)(wave_base | __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    };
    for (int i = tid; i < Body::TWR; i += NT) l_twr[i] = reinterpret_cast<const cx *>(a.twr)[i];
    for (unsigned i = tid; i < (3u << a.tw_bits); i += NT) l_tw3[i] = reinterpret_cast<const cx *>(a.tw3)[i];
    __syncthreads();

    typename Body::Regs z, p;
    auto run = [&](typename Body::Regs &r) {  // pre-twiddle + the tile's radix chain, exactly as tile_fft_kernel runs them
        auto exchange = [&](auto e) {
            constexpr int E = decltype(e)::value;
            tid = fresh_tid();
            if constexpr (!Body::PLANE_SEQ) {
                __syncthreads();
                Body::template ex_write<E>(sh, tid, r, 0);
                Body::template ex_write<E>(sh, tid, r, 1);
                __syncthreads();
                Body::template ex_read<E>(sh, tid, r, 0);
                Body::template ex_read<E>(sh, tid, r, 1);
            } else {
                __syncthreads();
                Body::template ex_write<E>(sh, tid, r, 0);
                __syncthreads();
                Body::template ex_read<E>(sh, tid, r, 0);
                __syncthreads();
                Body::template ex_write<E>(sh, tid, r, 1);
                __syncthreads();
                Body::template ex_read<E>(sh, tid, r, 1);
            }
        };
        auto do_step = [&](auto i) {
            tid = fresh_tid();
            Body::template step<decltype(i)::value>(sh, tid, r);
        };
        tid = fresh_tid();
        Body::pre_twiddle(a, sh, tid, r);
        Body::chain(do_step, exchange);
    };

    for (unsigned t = blockIdx.x; t < f.tiles_total; t += gridDim.x) {
        tid = fresh_tid();
        const bool self_block = RB::locate(a, f, t, z);
        p.xform = z.xform;
        p.g0 = z.g0;
        Body::load_raw(a, tid, z);
        RB::load_mirror(a, tid, p);
        const Tw3Raw<T> wg = RB::col_twiddle_fetch(f, tid, z);  // in flight under the two transforms
        run(z);
        run(p);
        tid = fresh_tid();
        RB::untangle(f, tid, wg, z, p);
        RB::store_own(a, tid, z, self_block);
        if (!self_block) RB::store_mirror(a, tid, p);
    }
}

template <typename T, int LR, int LC, int LP, bool SEQ>
hipError_t launch_r2c_last_inst(unsigned grid, hipStream_t stream, const TileArgs &a, const R2cFuseArgs &f, bool query_only,
                                int *blocks_per_cu, hipEvent_t ev_start, hipEvent_t ev_stop) {
    using Body = TileBody<T, LR, LC, LP, true, false, SEQ>;
    auto kern = r2c_last_pass_kernel<T, LR, LC, LP, SEQ>;
    const size_t lds = (size_t)Body::EXCH * sizeof(T) * (Body::PLANE_SEQ ? 1 : 2) + (size_t)(3u << a.tw_bits) * sizeof(cx_t<T>) +
                       Body::TWR * sizeof(cx_t<T>);
    if (lds > (size_t)160 * 1024) {
        if (query_only && blocks_per_cu) *blocks_per_cu = 0;
        return query_only ? hipSuccess : hipErrorInvalidValue;
    }
    static PerDeviceLimit lds_limit;
    if (hipError_t e = raise_lds_limit(lds_limit, reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    if (query_only) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
        if (e != hipSuccess) return e;
        const int alloc = ((fa.numRegs + 7) / 8) * 8, waves_per_wg = Body::NT / 64;
        int waves_per_simd = alloc > 0 ? 512 / alloc : 8;
        if (waves_per_simd > 8) waves_per_simd = 8;
        int b = waves_per_simd * 4 / waves_per_wg;
        const int by_lds = (int)((160 * 1024) / lds), by_waves = 32 / waves_per_wg;
        if (by_lds < b) b = by_lds;
        if (by_waves < b) b = by_waves;
        *blocks_per_cu = b < 1 ? 1 : b;
        return hipSuccess;
    }
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a, f);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), lds, stream, a, f);
    return hipGetLastError();
}

// thread-by-thread host execution (tests/emu): the same phase functions
template <typename T, int LR, int LC, int LP, bool SEQ> void emulate_r2c_last_pass(const TileArgs &a, const R2cFuseArgs &f) {
    using RB = R2cLastBody<T, LR, LC, LP, SEQ>;
    using Body = typename RB::Body;
    using Regs = typename Body::Regs;
    constexpr int NT = Body::NT;
    std::vector<T> ex((size_t)Body::EXCH * 2);
    const typename Body::Shared sh{ex.data(), Body::PLANE_SEQ ? ex.data() : ex.data() + Body::EXCH,
                                   reinterpret_cast<const cx_t<T> *>(a.tw3), reinterpret_cast<const cx_t<T> *>(a.twr)};
    std::vector<Regs> zs(NT), ps(NT);
    auto run = [&](std::vector<Regs> &regs) {
        auto exchange = [&](auto e) {
            constexpr int E = decltype(e)::value;
            if constexpr (!Body::PLANE_SEQ) {
                for (int t = 0; t < NT; ++t) {
                    Body::template ex_write<E>(sh, t, regs[t], 0);
                    Body::template ex_write<E>(sh, t, regs[t], 1);
                }
                for (int t = 0; t < NT; ++t) {
                    Body::template ex_read<E>(sh, t, regs[t], 0);
                    Body::template ex_read<E>(sh, t, regs[t], 1);
                }
            } else {
                for (int plane = 0; plane < 2; ++plane) {
                    for (int t = 0; t < NT; ++t) Body::template ex_write<E>(sh, t, regs[t], plane);
                    for (int t = 0; t < NT; ++t) Body::template ex_read<E>(sh, t, regs[t], plane);
                }
            }
        };
        auto do_step = [&](auto i) {
            for (int t = 0; t < NT; ++t) Body::template step<decltype(i)::value>(sh, t, regs[t]);
        };
        for (int t = 0; t < NT; ++t) Body::pre_twiddle(a, sh, t, regs[t]);
        Body::chain(do_step, exchange);
    };
    for (unsigned tile = 0; tile < f.tiles_total; ++tile) {
        bool self_block = false;
        for (int t = 0; t < NT; ++t) {
            self_block = RB::locate(a, f, tile, zs[t]);
            ps[t].xform = zs[t].xform;
            ps[t].g0 = zs[t].g0;
            Body::load_raw(a, t, zs[t]);
            RB::load_mirror(a, t, ps[t]);
        }
        run(zs);
        run(ps);
        for (int t = 0; t < NT; ++t) {
            RB::untangle(f, t, RB::col_twiddle_fetch(f, t, zs[t]), zs[t], ps[t]);
            RB::store_own(a, t, zs[t], self_block);
            if (!self_block) RB::store_mirror(a, t, ps[t]);
        }
    }
}

}  // namespace phast
