// c2r_fused.hpp -- the FIRST pass of the inner N/2-point transform of an inverse real FFT with the preprocess fused into
// its load (algorithms/r2c.rs:263-433: simd_c2r_preprocess_*; round 3).  Until now the preprocess ran as a sweep of its
// own: half-spectrum in, z out into a workspace of N values per transform, which the first pass read back -- a quarter of
// a C2R of 2^24 points (three passes + the preprocess).
//
// z[k] needs X[k] and X[h - k], h = N/2.  A first-pass tile holds ALL rows n of its columns g (k = n M + g, M = 2^log_s_in
// columns, R = 2^LR rows) and h - k = (R - 1 - n) M + (M - g): the partner lives in the MIRRORED column, rows reversed.
// Unlike the store side (r2c_fused.hpp) nothing has to be computed twice here -- the partner is INPUT: a thread simply
// loads it next to its own element (four loads per point instead of two) and forms z in registers:
//
//   * every element is then read twice per pass, once as itself and once as the partner of its mirror.  The tile order
//     makes the second read an L2 hit instead of HBM traffic: tiles are taken in PAIRS (q, tiles - 1 - q) at consecutive
//     positions of the XCD-aware order, i.e. by neighbouring workgroups of one XCD at the same time -- each is (up to the
//     one-element shift of the mirror: columns (M - g0 - C, M - g0]) the other's partner;
//   * twiddle 0.5 W_N^k, N = 2h: W_N^(n M + g) = W_N^g W_2R^n -- one three-level look-up per thread and tile and one entry of
//     a small table (R entries, read through the L1) per element, as in r2c_fused.hpp; no rotation recurrence, no drift;
//   * k = 0 pairs with X[h] (row R, column 0: one past the R x M block, the half-spectrum has h + 1 entries), k = h/2 with
//     itself: both fall out of the address arithmetic, no special cases;
//   * the inverse runs as a forward transform of (z_im, z_re) (algorithms/dit.rs:297-300): the thread hands (z_im, z_re)
//     to the radix chain; the last pass stores (im, re) pairs scaled by 1/h as before.
//
// No workspace any more, and the half-spectrum is read-only as the reference's (r2c.rs:740-790 takes `&[T]`).
#pragma once

#include "tile_fft.hpp"

#ifndef PHAST_C2R_PAIRS
#define PHAST_C2R_PAIRS 1
#endif
#ifndef PHAST_C2R_NT
#define PHAST_C2R_NT 0
#endif
#ifndef PHAST_C2R_DUAL  // two-tile workgroups (a tile and its mirror side by side): built, parity-green, measured, NOT faster --
#define PHAST_C2R_DUAL 0  // profiles/r06_c2r_first_pass_ab.log; -DPHAST_C2R_DUAL=1 (tools/build_variant.py) instantiates them
#endif

namespace phast {

inline bool c2r_dual_enabled() {  // PHAST_C2R_DUAL=0 (environment): single-tile workgroups, for A/B runs
    static const bool on = [] {
        const char *e = getenv("PHAST_C2R_DUAL");
        return !(e && *e == '0');
    }();
    return on;
}

struct C2rFuseArgs {
    const void *tw3n;   // [3][1 << twn_bits] complex: W_N^e, N = 2 h (the R2C planner's table)
    const void *twu;    // [R] complex: W_{2R}^n
    unsigned twn_bits;
};

template <typename T, int LR, int LC, int LP, bool SEQ> struct C2rFirstBody {
    using Body = TileBody<T, LR, LC, LP, false, true, SEQ>;
    using Regs = typename Body::Regs;
    using cx = cx_t<T>;
    static constexpr int P = Body::P, M = Body::M, COLS = Body::COLS, ROWS = Body::ROWS;
    // rows loaded at a time (load_pre): registers per lane the shape may use, minus the 2 P values of z and ~40 of
    // addresses and twiddles, over the 4 values a row brings
    static constexpr int W = (int)sizeof(T) / 4, BUDGET = Body::NT > 512 ? 128 : 256, FREE = BUDGET - 2 * P * W - 40;
    static constexpr int chunk_rows() {
        int ch = 1;
        while (2 * ch <= P && 4 * W * 2 * ch <= FREE) ch *= 2;
        return ch;
    }
    static constexpr int CH = chunk_rows();

    // position t of the launch -> (transform, first column): XCD-aware as TileBody::locate (workgroup b runs on XCD b % 8
    // and gets one contiguous run of positions); inside a transform positions 2q and 2q + 1 are tile q and its mirror
    PHAST_HD static void locate(const TileArgs &a, unsigned t, Regs &r) {
        locate_pos(a, ((a.tiles_total & 7u) == 0u) ? (t & 7u) * (a.tiles_total >> 3) + (t >> 3) : t, r);
    }
    // two-tile workgroups: workgroup b takes the positions 2 j and 2 j + 1 (a tile and its mirror) of ITS XCD's run
    PHAST_HD static unsigned dual_pos(const TileArgs &a, unsigned b, unsigned half) {
        return ((a.tiles_total & 15u) == 0u) ? (b & 7u) * (a.tiles_total >> 3) + 2u * (b >> 3) + half : 2u * b + half;
    }
    PHAST_HD static void locate_pos(const TileArgs &a, unsigned pos, Regs &r) {
        r.xform = pos >> (unsigned)__builtin_ctz(a.tiles_per_xform);
        const unsigned ti = pos & (a.tiles_per_xform - 1u), q = ti >> 1;
#if PHAST_C2R_PAIRS
        r.g0 = ((ti & 1u) ? a.tiles_per_xform - 1u - q : q) << LC;
#else  // tools only: plain column order, the partner tile runs on another XCD at another time
        (void)q;
        r.g0 = ti << LC;
#endif
    }

    // rows n = j M + tau of column g = g0 + col: X[k] from (row n, column g), X[h - k] from (row R - 1 - n, column M - g)
    //   = row (P - 1 - j) M + (M - 1 - tau), column (M - g0 - COLS) + (COLS - col): uniform base + non-negative lane offset
    PHAST_HD static void load_pre(const TileArgs &a, const C2rFuseArgs &f, int tid, Regs &r) {
        const int col = Body::col_of(tid), tau = Body::tau_of(tid);
        const unsigned mcols = 1u << a.log_s_in;
        const size_t xbase = (size_t)r.xform * a.in_dist;
        const T *pr = reinterpret_cast<const T *>(a.in_re) + xbase + r.g0;
        const T *pi = reinterpret_cast<const T *>(a.in_im) + xbase + r.g0;
        const T *qr = reinterpret_cast<const T *>(a.in_re) + xbase + (mcols - r.g0 - (unsigned)COLS);
        const T *qi = reinterpret_cast<const T *>(a.in_im) + xbase + (mcols - r.g0 - (unsigned)COLS);
        const unsigned voff = (unsigned)tau * mcols + (unsigned)col;
        const unsigned moff = (unsigned)(M - 1 - tau) * mcols + (unsigned)(COLS - col);
        T gr, gi;  // 0.5 W_N^g
        tw3_lookup<T>(reinterpret_cast<const cx *>(f.tw3n), f.twn_bits, r.g0 + (unsigned)col, gr, gi);
        gr *= (T)0.5;
        gi *= (T)0.5;
        const cx *twu = reinterpret_cast<const cx *>(f.twu) + tau;
        // CH rows at a time: the loaded values of a chunk (4 per point) live next to the 2 P values of z, so the chunk is
        // what the register budget of the shape leaves (all P rows at once for the 8- and most 16-point shapes)
        static_for<0, P / CH>([&](auto c) {
            constexpr int C0 = decltype(c)::value * CH;
            T x_re[CH], x_im[CH], m_re[CH], m_im[CH];
            static_for<0, CH>([&](auto i) {
                constexpr int I = decltype(i)::value, J = C0 + I;
                const size_t urow = (size_t)(J * M) << a.log_s_in, mrow = (size_t)((P - 1 - J) * M) << a.log_s_in;
                // (cache hints on the two streams -- PHAST_C2R_NT: 1 = the mirrored partner stream non-temporal (its second and
                //  last use), 2 = the tile's own stream, 3 = both; measured in profiles/r06_c2r_first_pass_ab.log)
#if PHAST_C2R_NT & 2
                x_re[I] = __builtin_nontemporal_load(pr + urow + voff);
                x_im[I] = __builtin_nontemporal_load(pi + urow + voff);
#else
                x_re[I] = (pr + urow)[voff];
                x_im[I] = (pi + urow)[voff];
#endif
#if PHAST_C2R_NT & 1
                m_re[I] = __builtin_nontemporal_load(qr + mrow + moff);
                m_im[I] = __builtin_nontemporal_load(qi + mrow + moff);
#else
                m_re[I] = (qr + mrow)[moff];
                m_im[I] = (qi + mrow)[moff];
#endif
            });
            // (the rows' table entries four at a time -- as r2c_fused.hpp: untangle; all CH of them with the data loads cost the
            //  16-point f64 kernels a wave of occupancy, 139 -> 211 VGPRs: measured in kernel_resources.json, round 4)
            constexpr int UC = CH < 4 ? CH : 4;
            static_for<0, CH / UC>([&](auto g) {
                constexpr int G0 = decltype(g)::value * UC;
                cx tu[UC];
                static_for<0, UC>([&](auto i) { tu[decltype(i)::value] = twu[(C0 + G0 + decltype(i)::value) * M]; });
                static_for<0, UC>([&](auto i) {
                    constexpr int I = G0 + decltype(i)::value, J = C0 + I;
                    const cx u = tu[decltype(i)::value];
                    const T c_h = gr * u.x - gi * u.y, s_h = gr * u.y + gi * u.x;  // 0.5 W_N^(n M + g)
                    // algorithms/r2c.rs:263-433 (the arithmetic of r2c.hip: c2r_preprocess_kernel)
                    const T re_first = x_re[I], im_first = x_im[I], re_second = m_re[I], im_second = -m_im[I];
                    const T zx_re = (T)0.5 * (re_first + re_second), zx_im = (T)0.5 * (im_first + im_second);
                    const T dr = re_first - re_second, di = im_first - im_second;
                    const T zy_re = c_h * dr + s_h * di, zy_im = c_h * di - s_h * dr;
                    r.re[J] = zx_im + zy_re;  // positional re = z_im, positional im = z_re: the swap-trick inverse
                    r.im[J] = zx_re - zy_im;
                });
            });
        });
    }
};

// DUAL (round 6, measured and not adopted): one workgroup of 2 NT threads runs a tile AND its mirror tile side by side
// (positions 2 j and 2 j + 1 of its XCD's run, each half with its own LDS region and its own copy of the tables; the barriers
// are shared, both halves run the same instruction stream).  The idea: every line of the half-spectrum is wanted twice -- as the
// tile's own element and as the mirror tile's partner -- and with both on ONE CU at the same time the second request would be
// a hit in that CU's vector L1.  Measured (profiles/r06_c2r_first_pass_ab.log): f32 2^24 40.3 -> 39.7 us, f64 2^24 71 -> 81,
// f32 2^26 157 -> 173: the 512-thread workgroups cost more than the L1 hits give.
template <typename T, int LR, int LC, int LP, bool SEQ, bool DUAL>
__global__ void __launch_bounds__((DUAL ? 2 : 1) << (LR + LC - LP)) c2r_first_pass_kernel(const TileArgs a, const C2rFuseArgs f, unsigned lds_half) {
    using CB = C2rFirstBody<T, LR, LC, LP, SEQ>;
    using Body = typename CB::Body;
    using cx = cx_t<T>;
    constexpr int NT = Body::NT;
    pin_tile_args(a);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const unsigned half = DUAL ? (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >= (unsigned)NT)) : 0u;  // wave-uniform
    unsigned char *smem = smem_all + (size_t)half * lds_half;
    T *ex_re = reinterpret_cast<T *>(smem);
    cx *l_twr = reinterpret_cast<cx *>(smem + (size_t)Body::EXCH * sizeof(T) * (Body::PLANE_SEQ ? 1 : 2));
    const typename Body::Shared sh{ex_re, Body::PLANE_SEQ ? ex_re : ex_re + Body::EXCH, l_twr, l_twr};

    int tid = (int)threadIdx.x - (int)half * NT;
    unsigned wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u) - (int)half * NT);
    auto fresh_tid = [&]() {  // see tile_fft_kernel
        asm volatile("" : "+s"(wave_base));
        return (int)(wave_base | __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    };
    typename Body::Regs r;
    // t counts workgroup rounds: a tile per round (DUAL: a tile and its mirror -- tiles_total is even, both halves make the
    // same number of rounds and meet at the same barriers)
    unsigned t = blockIdx.x;
    const unsigned t_end = DUAL ? a.tiles_total / 2u : a.tiles_total;
    auto place = [&]() {
        if constexpr (DUAL) CB::locate_pos(a, CB::dual_pos(a, t, half), r);
        else CB::locate(a, t, r);
    };
    if (t < t_end) {  // the first tile's loads go out before the table is staged
        place();
        CB::load_pre(a, f, tid, r);
    }
    for (int i = tid; i < Body::TWR; i += NT) l_twr[i] = reinterpret_cast<const cx *>(a.twr)[i];
    __syncthreads();

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
    while (t < t_end) {
        Body::chain(do_step, exchange);
        tid = fresh_tid();
        Body::store(a, tid, r);
        t += gridDim.x;
        if (t < t_end) {
            tid = fresh_tid();
            place();
            CB::load_pre(a, f, tid, r);
        }
    }
}

template <typename T, int LR, int LC, int LP, bool SEQ>
hipError_t launch_c2r_first_inst(unsigned grid, hipStream_t stream, const TileArgs &a, const C2rFuseArgs &f, bool query_only,
                                 int *blocks_per_cu, hipEvent_t ev_start, hipEvent_t ev_stop) {
    using Body = TileBody<T, LR, LC, LP, false, true, SEQ>;
    const size_t lds_one = (Body::lds_bytes(a.tw_bits) + 15) & ~(size_t)15;
    if (lds_one > (size_t)160 * 1024) {
        if (query_only && blocks_per_cu) *blocks_per_cu = 0;
        return query_only ? hipSuccess : hipErrorInvalidValue;
    }
    // two tiles (a tile and its mirror) per workgroup where 2 NT threads and twice the LDS fit -- the launch form, not the
    // occupancy query (which stays in single-tile workgroups: the grid below is derived from it).  PHAST_C2R_DUAL=0: tools.
    constexpr bool kCanDual = PHAST_C2R_DUAL && Body::NT <= 256;  // (512-thread workgroups: 256 registers per lane; 1024 would halve them and spill)
    const bool dual = kCanDual && !query_only && 2 * lds_one <= (size_t)160 * 1024 && (a.tiles_total & 1u) == 0u && c2r_dual_enabled();
    if constexpr (kCanDual) {
        if (dual) {
            auto kern2 = c2r_first_pass_kernel<T, LR, LC, LP, SEQ, true>;
            static PerDeviceLimit lds_limit2;
            if (hipError_t e = raise_lds_limit(lds_limit2, reinterpret_cast<const void *>(kern2), 2 * lds_one); e != hipSuccess) return e;
            unsigned g2 = (grid + 1) / 2;
            if (g2 > a.tiles_total / 2) g2 = a.tiles_total / 2;
            if (g2 >= 8 && (a.tiles_total & 15u) == 0u) g2 &= ~7u;   // workgroup b stays in the run of XCD b % 8 (dual_pos)
            if (g2 == 0) g2 = 1;
            if (ev_start && ev_stop)
                hipExtLaunchKernelGGL(kern2, dim3(g2), dim3(2 * Body::NT), (uint32_t)(2 * lds_one), stream, ev_start, ev_stop, 0, a, f, (unsigned)lds_one);
            else
                hipLaunchKernelGGL(kern2, dim3(g2), dim3(2 * Body::NT), 2 * lds_one, stream, a, f, (unsigned)lds_one);
            return hipGetLastError();
        }
    }
    auto kern = c2r_first_pass_kernel<T, LR, LC, LP, SEQ, false>;
    const size_t lds = lds_one;
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
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a, f, (unsigned)lds);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), lds, stream, a, f, (unsigned)lds);
    return hipGetLastError();
}

// thread-by-thread host execution (tests/emu): the same phase functions
template <typename T, int LR, int LC, int LP, bool SEQ> void emulate_c2r_first_pass(const TileArgs &a, const C2rFuseArgs &f) {
    using CB = C2rFirstBody<T, LR, LC, LP, SEQ>;
    using Body = typename CB::Body;
    using Regs = typename Body::Regs;
    constexpr int NT = Body::NT;
    std::vector<T> ex((size_t)Body::EXCH * 2);
    const typename Body::Shared sh{ex.data(), Body::PLANE_SEQ ? ex.data() : ex.data() + Body::EXCH,
                                   reinterpret_cast<const cx_t<T> *>(a.twr), reinterpret_cast<const cx_t<T> *>(a.twr)};
    std::vector<Regs> regs(NT);
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
    for (unsigned tile = 0; tile < a.tiles_total; ++tile) {
        for (int t = 0; t < NT; ++t) {
            CB::locate(a, tile, regs[t]);
            CB::load_pre(a, f, t, regs[t]);
        }
        Body::chain(do_step, exchange);
        for (int t = 0; t < NT; ++t) Body::store(a, t, regs[t]);
    }
}

}  // namespace phast
