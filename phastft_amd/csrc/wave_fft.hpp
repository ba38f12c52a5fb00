// wave_fft.hpp -- one pass of the multi-pass FFT with ONE WAVE PER TILE and the butterfly exchange done by
// cross-lane swaps instead of LDS + barriers (BASELINE.json north_star: "cross-lane shuffles via
// DS_PERMUTE/ds_swizzle wavefront primitives" -- on gfx950 the cheapest member of that family for this pattern is
// v_permlane16_swap / v_permlane32_swap, one VALU instruction per swapped dword pair, no LDS round trip).
//
// Same pass algebra as tile_fft.hpp (see there for the reference citations: kernels/dit.rs, algorithms/dit.rs,
// algorithms/bravo.rs): a tile is ROWS = 64 rows x COLS adjacent columns with 128-BYTE ROWS in either type, held by the
// 64 lanes of one wave:
//     f64: COLS = 16, TAUS = 4 lanes per column, P = 16 points per lane  (1024-point tiles; round 2)
//     f32: COLS = 32, TAUS = 2 lanes per column, P = 32 points per lane  (2048-point tiles; round 3: the f32 twin)
//     lane = (col = lane & (COLS - 1), tau = lane >> LC),   register j holds row n = TAUS j + tau        (j < P)
//   1. radix-P DIF over j in registers (literal twiddles)             -> register p holds digit k1 = bitrev(p)
//   2. inter-digit twiddle W_64^(tau * k1)
//   3. EXCHANGE: the transposition between the lane bits above the column and the low register bits:
//         f64: v_permlane32_swap on the register pairs (p, p | 2)   -- lane bit 5 <-> register bit 1
//              v_permlane16_swap on the register pairs (p, p | 1)   -- lane bit 4 <-> register bit 0
//         f32: v_permlane32_swap on the register pairs (p, p | 1)   -- lane bit 5 <-> register bit 0
//      afterwards lane tau', register (g, s) holds the value of tau = s, k1 = bitrev((g << LT) | tau')
//   4. radix-TAUS DIF over tau in registers, P / TAUS independent groups -> register TAUS g + s holds k2 = bitrev(s)
//      output row k = k1 + P k2
// No workgroup barrier anywhere: the four waves of a 256-thread block are four independent tiles (own copy of the W_64
// table in LDS, inter-pass tables read from global memory), and their loads are deliberately issued a little apart
// (see the kernel: staggered waves).  Pass A's transposition to contiguous output runs goes through
// a WAVE-PRIVATE LDS buffer (written and read by the same wave: ordered by the LDS queue, no barrier).
//
// Phase functions are __host__ __device__ (per lane); tests/emu/emu.hip runs them lane by lane with the swaps emulated.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tile_fft.hpp"

namespace phast {

// tiles (= waves) per workgroup: 4 measured best for the single 2^20 f64 transform (26.5 us; 27.2 / 27.5 / 29.2 at 1 / 2 / 8);
// the f32 tile is twice the points: 2 per workgroup keeps one workgroup per CU at 2^20
#ifndef PHAST_WAVE_TILES_PER_BLOCK
#define PHAST_WAVE_TILES_PER_BLOCK 4
#endif
#ifndef PHAST_WAVE_TILES_PER_BLOCK_F32
#define PHAST_WAVE_TILES_PER_BLOCK_F32 2
#endif
template <typename T> constexpr int wave_block_threads() { return 64 * (sizeof(T) == 4 ? PHAST_WAVE_TILES_PER_BLOCK_F32 : PHAST_WAVE_TILES_PER_BLOCK); }

template <typename T, bool PRE_TW, bool TRANSPOSE> struct WaveBody {
    using cx = cx_t<T>;
    // The register type of a lane (common.hpp): one f64, or -- round 6 -- TWO ADJACENT f32 COLUMNS in 8 bytes.  Either way a lane
    // moves 8 bytes per access, a tile row is 128 bytes per plane, lane = (16 lane-columns, 4 taus) and a lane holds 16 registers
    // per plane: the f32 tile is the f64 tile with every register carrying a column pair (64 x 32 = 2048 points per wave, 32
    // per lane).  Round 3's f32 twin (32 lane-columns of 4 bytes, 2 taus, 32 registers: 64 four-byte loads and 142 VGPRs per
    // lane, profiles/r03_sweep_wave_f32.log) is replaced by this.
    using V = lane_vec_t<T>;
    static constexpr int VW = ScalarOf<V>::W, LV = VW == 2 ? 1 : 0;
    static constexpr int LR = 6, LCL = 4;             // 64 rows, 16 lane-columns
    static constexpr int LC = LCL + LV;               // log2 of the tile's (scalar) columns: 16 f64 / 32 f32
    static constexpr int LT = 2, LP = 4;              // lanes per column 2^LT, registers per lane and plane 2^LP
    static constexpr int ROWS = 64, COLS = 1 << LC, P = 1 << LP, TAUS = 1 << LT;
    static constexpr int NT = wave_block_threads<T>(), WAVES = NT / 64;  // tiles (= waves) per workgroup
    static constexpr int CS = ROWS + 1;               // column pitch of the transposing buffer: odd => conflict-free
    static constexpr int XP = TRANSPOSE ? COLS * CS : 0;  // scalar elements per plane per wave
#ifndef PHAST_WQ_NT_LOADS
#define PHAST_WQ_NT_LOADS 1
#endif
#ifndef PHAST_WQ_NT_STORES
#define PHAST_WQ_NT_STORES 1
#endif
    // cache policy per pass position (tools: -DPHAST_WAVE_NT_A_STORES=0 etc.; measured in profiles/r06_cache_policy_ab.log):
    // the first pass reads the caller's (cold) array and writes the scratch, the later passes read the scratch
#ifndef PHAST_WAVE_NT_A_LOADS
#define PHAST_WAVE_NT_A_LOADS PHAST_WQ_NT_LOADS
#endif
#ifndef PHAST_WAVE_NT_A_STORES
#define PHAST_WAVE_NT_A_STORES PHAST_WQ_NT_STORES
#endif
#ifndef PHAST_WAVE_NT_C_LOADS
#define PHAST_WAVE_NT_C_LOADS PHAST_WQ_NT_LOADS
#endif
#ifndef PHAST_WAVE_NT_C_STORES
#define PHAST_WAVE_NT_C_STORES PHAST_WQ_NT_STORES
#endif
    static constexpr bool NT_LOAD = TRANSPOSE ? PHAST_WAVE_NT_A_LOADS : PHAST_WAVE_NT_C_LOADS;
    static constexpr bool NT_STORE = TRANSPOSE ? PHAST_WAVE_NT_A_STORES : PHAST_WAVE_NT_C_STORES;
    // a register's worth in the caller's memory: aligned to ONE ELEMENT only (`&mut v[1..]` is a legal slice: no more than
    // element alignment may be assumed of a caller's pointer; the hardware takes unaligned dword-multiple accesses)
    typedef V VU __attribute__((aligned(sizeof(T))));

    struct Regs {
        V re[P], im[P];
        unsigned xform, g0;
    };

    static size_t lds_bytes(unsigned tw_bits) {
        (void)tw_bits;  // the inter-pass tables are read from global memory (six entries per lane and column)
        return (size_t)WAVES * (64 * sizeof(cx) + (size_t)2 * XP * sizeof(T));  // per wave: W_64 table, transposing buffer
    }

    PHAST_HD static int lcol_of(int lane) { return lane & 15; }
    PHAST_HD static int col_of(int lane) { return (lane & 15) << LV; }  // first (scalar) column of the lane
    PHAST_HD static int tau_of(int lane) { return lane >> LCL; }

    // tile index -> (transform, first column); XCD-aware: workgroup b (on XCD b % 8) owns WAVES adjacent tiles and each XCD
    // gets one contiguous run of workgroups (cf. TileBody::locate)
    PHAST_HD static void locate(const TileArgs &a, unsigned block, unsigned blocks_total, unsigned wave, Regs &r) {
        const unsigned b = ((blocks_total & 7u) == 0u) ? (block & 7u) * (blocks_total >> 3) + (block >> 3) : block;
        locate_tile(a, b * WAVES + wave, r);
    }
    PHAST_HD static void locate_tile(const TileArgs &a, unsigned tile, Regs &r) {
        r.xform = tile >> (unsigned)__builtin_ctz(a.tiles_per_xform);  // a power of two (plan.hpp: geom_to_args)
        const unsigned ti = tile & (a.tiles_per_xform - 1u);
        r.g0 = a.cs_bits ? (((ti >> a.cb_bits) << a.cs_bits) | ((ti & ((1u << a.cb_bits) - 1u)) << LC)) : (ti << LC);
    }

    // rows n = 4 j + tau: one VGPR offset for all 16 loads, the row part is wave-uniform
    PHAST_HD static void load_raw(const TileArgs &a, int lane, Regs &r) {
        const int col = col_of(lane), tau = tau_of(lane);
        const size_t ubase = in_tile_base(a, r.xform, r.g0);
        const unsigned voff = (unsigned)tau * (unsigned)a.in_row_stride + (unsigned)col;
        if (PRE_TW || !a.in_interleaved) {
            const T *pr = reinterpret_cast<const T *>(a.in_re) + ubase;
            const T *pi = reinterpret_cast<const T *>(a.in_im) + ubase;
            // the lane's part of the address as a 32-bit BYTE offset: with the tile's base wave-uniform (the kernel takes the
            // wave index through readfirstlane) every load is  scalar row base + one shared VGPR offset  -- no 64-bit
            // vector address arithmetic per load (it was 4-6 VALU instructions of each of the 32 loads and 32 stores of a
            // tile, on a wave that has its SIMD to itself: nothing hides them).  launch_wave_inst checks the range.
            const unsigned vbyte = voff * (unsigned)sizeof(T);
            static_for<0, P>([&](auto j) {
                const size_t urow = (size_t)(decltype(j)::value * TAUS) * a.in_row_stride;
                const VU *qr = reinterpret_cast<const VU *>(reinterpret_cast<const char *>(pr + urow) + vbyte);
                const VU *qi = reinterpret_cast<const VU *>(reinterpret_cast<const char *>(pi + urow) + vbyte);
                if constexpr (NT_LOAD) {
                    r.re[j] = __builtin_nontemporal_load(qr);
                    r.im[j] = __builtin_nontemporal_load(qi);
                } else {
                    r.re[j] = *qr;
                    r.im[j] = *qi;
                }
            });
        } else {  // first pass of an interleaved / real transform: (re, im) or (im, re) pairs, VW of them side by side
            const cx *pz = reinterpret_cast<const cx *>(a.in_re) + ubase;
            static_for<0, P>([&](auto j) {
                const size_t urow = (size_t)(decltype(j)::value * TAUS) * a.in_row_stride;
                if constexpr (VW == 1) {
                    cx v = (pz + urow)[voff];
                    r.re[j] = a.in_interleaved == 2 ? v.y : v.x;
                    r.im[j] = a.in_interleaved == 2 ? v.x : v.y;
                } else {
                    const cx v0 = (pz + urow)[voff], v1 = (pz + urow)[voff + 1];
                    const V x{v0.x, v1.x}, y{v0.y, v1.y};
                    r.re[j] = a.in_interleaved == 2 ? y : x;
                    r.im[j] = a.in_interleaved == 2 ? x : y;
                }
            });
        }
    }

    // inter-pass twiddle W_{64 S}^{row * lo}: row = tau + TAUS j  =>  W^(tau lo) * (W^(TAUS lo))^j.  Two table look-ups and a
    // geometric progression (tw_progression: 4 running values stepped by D^4, 17 products where a table of the powers costs
    // 30) instead of 16 look-ups (48 LDS reads): its rounding (<= 7 extra complex products, ~1.1e-16 each) is inside the gates
    // of tests/tolerances.py (measured rel-L2 unchanged).  Per COLUMN: the packed f32 lane looks up two sets and runs the
    // progression on the pair.
    struct TwRaw {
        Tw3Raw<T> b[VW], d[VW];
    };
    PHAST_HD static TwRaw pre_twiddle_fetch(const TileArgs &a, const cx *tw3, int lane, const Regs &r) {
        TwRaw t;
        static_for<0, VW>([&](auto e) {
            const unsigned lo = ((r.g0 + (unsigned)col_of(lane) + (unsigned)decltype(e)::value) >> a.tw_shift) & a.tw_mask;
            t.b[e] = tw3_fetch<T>(tw3, a.tw_bits, (unsigned)tau_of(lane) * lo);
            t.d[e] = tw3_fetch<T>(tw3, a.tw_bits, (unsigned)TAUS * lo);
        });
        return t;
    }
    PHAST_HD static void pre_twiddle_apply(const TwRaw &t, Regs &r) {
        V br, bi, dr, di;
        if constexpr (VW == 1) {
            tw3_combine<T>(t.b[0], br, bi);
            tw3_combine<T>(t.d[0], dr, di);
        } else {
            T b0r, b0i, b1r, b1i, d0r, d0i, d1r, d1i;
            tw3_combine<T>(t.b[0], b0r, b0i);
            tw3_combine<T>(t.b[1], b1r, b1i);
            tw3_combine<T>(t.d[0], d0r, d0i);
            tw3_combine<T>(t.d[1], d1r, d1i);
            br = V{b0r, b1r};
            bi = V{b0i, b1i};
            dr = V{d0r, d1r};
            di = V{d0i, d1i};
        }
        tw_progression<V, P, 4>(br, bi, dr, di, [&](auto j, V wr, V wi) {
            cmul(r.re[decltype(j)::value], r.im[decltype(j)::value], wr, wi);
        });
    }
    PHAST_HD static void pre_twiddle(const TileArgs &a, const cx *tw3, int lane, Regs &r) {
        if constexpr (PRE_TW) pre_twiddle_apply(pre_twiddle_fetch(a, tw3, lane, r), r);
    }

    // W_64^e, e < 64.  The kernel stages all 64 entries in LDS (the second half negated once, when the table is staged);
    // the host emulator reads the planner's 32-entry table, W^(e + 32) = -W^e -- bitwise the same values
    PHAST_HD static void w64(const cx *twr, unsigned e, T &wr, T &wi) {
#if defined(__HIP_DEVICE_COMPILE__)
        const cx w = twr[e];
        wr = w.x;
        wi = w.y;
#else
        const cx w = twr[e & 31u];
        wr = (e & 32u) ? -w.x : w.x;
        wi = (e & 32u) ? -w.y : w.y;
#endif
    }

    PHAST_HD static void step1(const cx *twr, int lane, Regs &r) {
        fft_reg_dif<V, P, 0, P>(r.re, r.im);
        const unsigned tau = (unsigned)tau_of(lane);
        static_for<1, P>([&](auto p) {
            constexpr int K1 = bitrev_c(decltype(p)::value, LP);
            T wr, wi;
            w64(twr, tau * K1, wr, wi);   // one scalar twiddle for the lane's column pair
            cmul(r.re[p], r.im[p], wr, wi);
        });
    }

    PHAST_HD static void step2(Regs &r) {
        static_for<0, P / TAUS>([&](auto g) { fft_reg_dif<V, TAUS, decltype(g)::value * TAUS, P>(r.re, r.im); });
    }

    // output row held by register Q = TAUS g + s of lane tau' after step2:  k1 + P k2,
    //   k1 = bitrev_LP((g << LT) | tau') = (bitrev_LT(tau') << (LP - LT)) + bitrev_(LP-LT)(g),   k2 = bitrev_LT(s)
    //   (k1 = bitrev4(p3 p2 b5 b4) = 8 b4 + 4 b5 + 2 p2 + p3)
    PHAST_HD static unsigned krow_lane(int lane) {
        const unsigned t = (unsigned)tau_of(lane);
        return ((t & 1u) << 3) | ((t >> 1) << 2);
    }
    template <int Q> PHAST_HD static constexpr unsigned krow_const() {
        constexpr int G = Q >> LT, S = Q & (TAUS - 1);
        return (unsigned)bitrev_c(G, LP - LT) + (unsigned)P * (unsigned)bitrev_c(S, LT);
    }

    PHAST_HD static size_t out_base(const TileArgs &a, const Regs &r) {
        return (size_t)(r.g0 & ((1u << a.out_lo_bits) - 1u)) * a.out_s1 + (size_t)(r.g0 >> a.out_lo_bits) * a.out_s2 +
               (size_t)r.xform * a.out_dist;
    }
    // ubase: wave-uniform element offset; vbyte: the lane's part as a 32-bit byte offset (see load_raw)
    template <bool PAIRS, bool SCALE> PHAST_HD static void put(const TileArgs &a, size_t ubase, unsigned vbyte, V re, V im, T scale) {
        if constexpr (SCALE) {
            re *= scale;
            im *= scale;
        }
        if constexpr (!PAIRS) {
            VU *qr = reinterpret_cast<VU *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out_re) + ubase) + vbyte);
            VU *qi = reinterpret_cast<VU *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out_im) + ubase) + vbyte);
            if constexpr (NT_STORE) {
                __builtin_nontemporal_store(re, qr);
                __builtin_nontemporal_store(im, qi);
            } else {
                *qr = re;
                *qi = im;
            }
        } else {
            const V x = a.out_interleaved == 2 ? im : re, y = a.out_interleaved == 2 ? re : im;
            cx *q = reinterpret_cast<cx *>(reinterpret_cast<char *>(reinterpret_cast<cx *>(a.out_re) + ubase) + 2u * vbyte);
            if constexpr (VW == 1) {
                cx v;
                v.x = x;
                v.y = y;
                *q = v;
            } else {
                cx v0, v1;
                v0.x = x[0];
                v0.y = y[0];
                v1.x = x[1];
                v1.y = y[1];
                q[0] = v0;
                q[1] = v1;
            }
        }
    }
    // later passes: same column-wide pattern out as in (the lane's columns are adjacent in memory: out_s1 = 1, checked by
    // launch_wave_inst for the packed tile).  The planar / pair decision and the multiplication by `scale` (1/N of an inverse
    // transform's last pass, else exactly 1) are wave-uniform branches taken ONCE, outside the sixteen stores: a forward
    // pass does not carry 32 multiplications by one.
    template <bool PAIRS, bool SCALE> PHAST_HD static void store_rows_as(const TileArgs &a, int lane, const Regs &r) {
        const size_t base = out_base(a, r);
        const unsigned vbyte =
            ((unsigned)col_of(lane) * (unsigned)a.out_s1 + krow_lane(lane) * (unsigned)a.out_row_stride) * (unsigned)sizeof(T);
        const T scale = (T)a.scale;
        static_for<0, P>([&](auto Q) {
            put<PAIRS, SCALE>(a, base + (size_t)krow_const<decltype(Q)::value>() * a.out_row_stride, vbyte, r.re[Q], r.im[Q], scale);
        });
    }
    PHAST_HD static void store_rows(const TileArgs &a, int lane, const Regs &r) {
        const bool scaled = a.scale != 1.0;
        if (!a.out_interleaved) {
            if (scaled) store_rows_as<false, true>(a, lane, r);
            else store_rows_as<false, false>(a, lane, r);
        } else {
            if (scaled) store_rows_as<true, true>(a, lane, r);
            else store_rows_as<true, false>(a, lane, r);
        }
    }
    // pass A: through the wave-private buffer [column][k] (scalars): afterwards every store instruction writes whole runs --
    // f64: register Q = column Q, lane = row k (one 512-byte run); f32: register Q = columns 2 Q and 2 Q + 1 on the two half-waves,
    // lane & 31 = the row pair (2 l, 2 l + 1) (two 256-byte runs)
    PHAST_HD static void xp_park(T *xp, int lane, const Regs &r) {
        static_for<0, P>([&](auto Q) {
            const int at = col_of(lane) * CS + (int)krow_lane(lane) + (int)krow_const<decltype(Q)::value>();
            if constexpr (VW == 1) {
                xp[at] = r.re[Q];
                xp[XP + at] = r.im[Q];
            } else {
                xp[at] = r.re[Q][0];
                xp[at + CS] = r.re[Q][1];
                xp[XP + at] = r.im[Q][0];
                xp[XP + at + CS] = r.im[Q][1];
            }
        });
    }
    PHAST_HD static void xp_pick(const T *xp, int lane, Regs &r) {
        static_for<0, P>([&](auto Q) {
            if constexpr (VW == 1) {
                r.re[Q] = xp[decltype(Q)::value * CS + lane];
                r.im[Q] = xp[XP + decltype(Q)::value * CS + lane];
            } else {
                const int at = (2 * decltype(Q)::value + (lane >> 5)) * CS + 2 * (lane & 31);
                r.re[Q] = V{xp[at], xp[at + 1]};
                r.im[Q] = V{xp[XP + at], xp[XP + at + 1]};
            }
        });
    }
    PHAST_HD static void store_runs(const TileArgs &a, int lane, const Regs &r) {  // a first pass: never the scaled last one
        const size_t base = out_base(a, r);
        const unsigned vbyte = VW == 1 ? (unsigned)lane * (unsigned)a.out_row_stride * (unsigned)sizeof(T)
                                       : ((unsigned)(lane >> 5) * (unsigned)a.out_s1 + 2u * (unsigned)(lane & 31) * (unsigned)a.out_row_stride) *
                                             (unsigned)sizeof(T);
        static_for<0, P>([&](auto Q) {
            put<false, false>(a, base + (size_t)(VW * decltype(Q)::value) * a.out_s1, vbyte, r.re[Q], r.im[Q], (T)1);
        });
    }
};

// ---- the exchange on the GPU: lane bits (5, 4) <-> register bits (1, 0), one VALU swap per dword pair ----
// -DPHAST_WAVE_XCHG_BPERMUTE builds the same exchange from ds_bpermute_b32 (the LDS crossbar without the memory: two
// selects + one DS instruction + an lgkmcnt wait per dword pair) for comparison -- tools/ and profiles/r02_wave_xchg.log:
// the swap instructions are what the product uses.
#ifdef PHAST_WAVE_XCHG_BPERMUTE
template <int BIT> __device__ __forceinline__ void swap_via_bpermute(unsigned &a, unsigned &b) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool up = (lane & BIT) != 0;
    const unsigned send = up ? a : b;  // the upper lane gives its `a`, the lower lane its `b`
    const unsigned recv = (unsigned)__builtin_amdgcn_ds_bpermute((int)((lane ^ BIT) << 2), (int)send);
    if (up) a = recv;
    else b = recv;
}
__device__ __forceinline__ void swap_lane_halves(unsigned &a, unsigned &b) { swap_via_bpermute<32>(a, b); }
__device__ __forceinline__ void swap_lane_rows(unsigned &a, unsigned &b) { swap_via_bpermute<16>(a, b); }
#else
__device__ __forceinline__ void swap_lane_halves(unsigned &a, unsigned &b) {  // a.lanes[32..63] <-> b.lanes[0..31]
    auto v = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = v[0];
    b = v[1];
}
__device__ __forceinline__ void swap_lane_rows(unsigned &a, unsigned &b) {  // a.rows{1,3} <-> b.rows{0,2} (16-lane rows)
    auto v = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = v[0];
    b = v[1];
}
#endif
template <bool HALVES> __device__ __forceinline__ void swap_pair(double &a, double &b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    if constexpr (HALVES) {
        swap_lane_halves(alo, blo);
        swap_lane_halves(ahi, bhi);
    } else {
        swap_lane_rows(alo, blo);
        swap_lane_rows(ahi, bhi);
    }
    a = __hiloint2double((int)ahi, (int)alo);
    b = __hiloint2double((int)bhi, (int)blo);
}
template <bool HALVES> __device__ __forceinline__ void swap_pair(float &a, float &b) {
    unsigned x = __float_as_uint(a), y = __float_as_uint(b);
    if constexpr (HALVES) swap_lane_halves(x, y);
    else swap_lane_rows(x, y);
    a = __uint_as_float(x);
    b = __uint_as_float(y);
}
template <bool HALVES> __device__ __forceinline__ void swap_pair(f32x2 &a, f32x2 &b) {  // a packed column pair: 8 bytes, as a double
    unsigned a0 = __float_as_uint(a[0]), a1 = __float_as_uint(a[1]), b0 = __float_as_uint(b[0]), b1 = __float_as_uint(b[1]);
    if constexpr (HALVES) {
        swap_lane_halves(a0, b0);
        swap_lane_halves(a1, b1);
    } else {
        swap_lane_rows(a0, b0);
        swap_lane_rows(a1, b1);
    }
    a = f32x2{__uint_as_float(a0), __uint_as_float(a1)};
    b = f32x2{__uint_as_float(b0), __uint_as_float(b1)};
}
template <typename T> __device__ __forceinline__ void wave_exchange(T (&re)[16], T (&im)[16]) {  // two lane bits
    static_for<0, 16>([&](auto p) {  // round 1: lane bit 5 <-> register bit 1
        constexpr int Pp = decltype(p)::value;
        if constexpr ((Pp & 2) == 0) {
            swap_pair<true>(re[Pp], re[Pp | 2]);
            swap_pair<true>(im[Pp], im[Pp | 2]);
        }
    });
    static_for<0, 16>([&](auto p) {  // round 2: lane bit 4 <-> register bit 0
        constexpr int Pp = decltype(p)::value;
        if constexpr ((Pp & 1) == 0) {
            swap_pair<false>(re[Pp], re[Pp | 1]);
            swap_pair<false>(im[Pp], im[Pp | 1]);
        }
    });
}
// Stagger of the waves' loads (see the kernel): group g = ((block & 1) << 2 | wave) & mask sleeps g * units * 64 cycles.
// Packed as units | mask << 8; PHAST_WAVE_STAGGER="units,mask" overrides the default (tuning, tools/sweep_stagger.py).
#ifndef PHAST_WAVE_STAGGER_DEFAULT
#define PHAST_WAVE_STAGGER_DEFAULT (8u | (3u << 8))
#endif
inline unsigned wave_stagger_setting() {
    static const unsigned v = [] {
        const char *e = getenv("PHAST_WAVE_STAGGER");
        unsigned units = 0, mask = 0;
        if (e && sscanf(e, "%u,%u", &units, &mask) == 2) return (units & 255u) | ((mask & 7u) << 8);
        return (unsigned)PHAST_WAVE_STAGGER_DEFAULT;
    }();
    return v;
}
template <typename T, bool PRE_TW, bool TRANSPOSE>
__global__ void __launch_bounds__((wave_block_threads<T>())) wave_fft_kernel(const TileArgs a, unsigned blocks_total, unsigned stagger) {
    using Body = WaveBody<T, PRE_TW, TRANSPOSE>;
    using cx = cx_t<T>;
    pin_tile_args(a);
    pin_scalars(blocks_total, stagger);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and the compiler is told so: tile bases live in SGPRs
#else
    const int wave = tid >> 6;
#endif
    // everything in LDS is private to a wave: its copy of the W_64 step-twiddle table and its transposing buffer.  No
    // workgroup barrier anywhere: the four waves of a block are four independent tiles.
    cx *l_twr = reinterpret_cast<cx *>(smem + (size_t)wave * (64 * sizeof(cx) + (size_t)2 * Body::XP * sizeof(T)));
    T *xp = reinterpret_cast<T *>(l_twr + 64);
    if ((blockIdx.x * Body::WAVES + (unsigned)wave) >= a.tiles_total) return;

    // Twiddle tables: their global loads go out FIRST and the tile's loads right behind them.  Loads return in order
    // (one vmcnt counter), so the other way round the table values would sit behind all 32 tile loads.  The inter-pass
    // tables are not staged at all: a lane needs six entries (two three-level look-ups), read straight from global
    // memory (a few KiB, L2-resident).
    using V = typename Body::V;
    // phase stamps: -DPHAST_TRACE builds only (tools/trace_wave_quad.py); one row of 16 per WAVE, lane 0 writes
#ifdef PHAST_TRACE
    int stamp_i = 0;
    auto stamp = [&](bool drain) {
        if (a.trace != nullptr) {
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (lane == 0 && stamp_i < 16) a.trace[((size_t)blockIdx.x * Body::WAVES + (size_t)wave) * 16 + stamp_i] = now;
            ++stamp_i;
        }
    };
#define PHAST_STAMP(d) stamp(d)
#else
#define PHAST_STAMP(d)
#endif
    PHAST_STAMP(false);  // 0: entry
    typename Body::Regs r;
    Body::locate(a, blockIdx.x, blocks_total, (unsigned)wave, r);
    const cx twr_stage = reinterpret_cast<const cx *>(a.twr)[lane & 31];
    typename Body::TwRaw twraw;
    if constexpr (PRE_TW) twraw = Body::pre_twiddle_fetch(a, reinterpret_cast<const cx *>(a.tw3), lane, r);
    // Staggered waves.  With one tile per SIMD every wave of the chip would wait for HBM, compute and store in step, the
    // memory system idle while the arithmetic runs.  The waves of the later groups issue their loads a little later: the
    // early groups' arithmetic and stores then overlap the late groups' loads (tools/pass_floor7.hip: 7.3 -> 6.5 us per
    // pass for the copy model with this pass's arithmetic; nothing lost when there is no arithmetic).
    for (unsigned k = ((((blockIdx.x & 1u) << 2) | (unsigned)wave) & (stagger >> 8)) * (stagger & 255u); k > 0; --k)
        __builtin_amdgcn_s_sleep(1);
    PHAST_STAMP(false);  // 1: the stagger's sleep is over
    Body::load_raw(a, lane, r);
    PHAST_STAMP(false);  // 2: loads issued
    PHAST_STAMP(true);   // 3: loads back
#ifdef PHAST_WAVE_WAIT_ALL  // tools only: every load back before anything else happens
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if defined(PHAST_WAVE_DEBUG_SKIP) && PHAST_WAVE_DEBUG_SKIP >= 4  // tools only: no table staging
    if (a.tiles_total == 0xffffffffu) l_twr[lane] = twr_stage;
    if constexpr (TRANSPOSE) { } else { Body::store_rows(a, lane, r); return; }
#endif
    {  // all 64 powers: lanes 32..63 store W^(e + 32) = -W^e, so step1 needs no sign selects (74 v_cndmask per tile)
        cx w = twr_stage;
        if (lane & 32) {
            w.x = -w.x;
            w.y = -w.y;
        }
        l_twr[lane] = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if !defined(PHAST_WAVE_DEBUG_SKIP) || PHAST_WAVE_DEBUG_SKIP < 3   // tools only: 1 = no pre-twiddle, 2 = + no exchange, 3 = no arithmetic at all
#if !defined(PHAST_WAVE_DEBUG_SKIP) || PHAST_WAVE_DEBUG_SKIP < 1
    if constexpr (PRE_TW) Body::pre_twiddle_apply(twraw, r);
#endif
    Body::step1(l_twr, lane, r);
#if !defined(PHAST_WAVE_DEBUG_SKIP) || PHAST_WAVE_DEBUG_SKIP < 2
    wave_exchange<V>(r.re, r.im);
#endif
    Body::step2(r);
#endif
    PHAST_STAMP(true);   // 4: arithmetic done
    if constexpr (TRANSPOSE) {
        // wave-private transposition: this wave writes and then reads its own buffer; LDS operations of one wave
        // execute in order, and the compiler's s_waitcnt lgkmcnt covers the data dependency -- no barrier
        Body::xp_park(xp, lane, r);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Body::xp_pick(xp, lane, r);
        PHAST_STAMP(true);   // 5: through the transposing buffer
        Body::store_runs(a, lane, r);
    } else {
        PHAST_STAMP(false);  // 5: (no transposition)
        Body::store_rows(a, lane, r);
    }
    PHAST_STAMP(false);  // 6: stores issued
    PHAST_STAMP(true);   // 7: stores retired
#undef PHAST_STAMP
}

// ---- two tiles per wave, software-pipelined ("build it, do not model it again") ----
// Half as many waves, each owning the tiles (2 t, 2 t + 1): both tiles' loads go out back to back (64 loads in flight per
// lane, two register sets), tile 2 t is computed and stored while tile 2 t + 1's loads are still arriving, then tile 2 t + 1.
// One wave per SIMD leaves registers free (2 x 64 data VGPRs + temporaries), so the second set costs nothing in occupancy.
// What the copy model promised (profiles/r02_pipelining_floor.log: 7.3 -> 6.8 us per pass at this pass's arithmetic) assumed
// the arithmetic is what a pass waits for; measured on the real kernels -- PHAST_WAVE_PIPELINE=1, same-box A/B in
// profiles/r06_pipelined_tiles_ab.log -- half the SIMDs idle costs more than the overlap gives.  Kept behind the switch.
template <typename T, bool PRE_TW, bool TRANSPOSE>
__global__ void __launch_bounds__((wave_block_threads<T>())) wave_fft2_kernel(const TileArgs a, unsigned blocks_total, unsigned stagger) {
    using Body = WaveBody<T, PRE_TW, TRANSPOSE>;
    using cx = cx_t<T>;
    using V = typename Body::V;
    pin_tile_args(a);
    pin_scalars(blocks_total, stagger);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cx *l_twr = reinterpret_cast<cx *>(smem + (size_t)wave * (64 * sizeof(cx) + (size_t)2 * Body::XP * sizeof(T)));
    T *xp = reinterpret_cast<T *>(l_twr + 64);
    // slot s of the launch = tiles 2 s and 2 s + 1; slots are dealt to the workgroups in the XCD-aware order of Body::locate
    const unsigned b = ((blocks_total & 7u) == 0u) ? (blockIdx.x & 7u) * (blocks_total >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned slot = b * Body::WAVES + (unsigned)wave;
    if (2u * slot >= a.tiles_total) return;
    typename Body::Regs r0, r1;
    Body::locate_tile(a, 2u * slot, r0);
    Body::locate_tile(a, 2u * slot + 1u, r1);
    const cx twr_stage = reinterpret_cast<const cx *>(a.twr)[lane & 31];
    typename Body::TwRaw tw0, tw1;
    if constexpr (PRE_TW) {
        tw0 = Body::pre_twiddle_fetch(a, reinterpret_cast<const cx *>(a.tw3), lane, r0);
        tw1 = Body::pre_twiddle_fetch(a, reinterpret_cast<const cx *>(a.tw3), lane, r1);
    }
    for (unsigned k = ((((blockIdx.x & 1u) << 2) | (unsigned)wave) & (stagger >> 8)) * (stagger & 255u); k > 0; --k)
        __builtin_amdgcn_s_sleep(1);
    Body::load_raw(a, lane, r0);
    Body::load_raw(a, lane, r1);  // in flight while tile 0 is computed and stored (loads return in order: tile 0's first)
    {
        cx w = twr_stage;
        if (lane & 32) {
            w.x = -w.x;
            w.y = -w.y;
        }
        l_twr[lane] = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto finish = [&](typename Body::Regs &r, const typename Body::TwRaw &tw) {
        if constexpr (PRE_TW) Body::pre_twiddle_apply(tw, r);
        Body::step1(l_twr, lane, r);
        wave_exchange<V>(r.re, r.im);
        Body::step2(r);
        if constexpr (TRANSPOSE) {
            Body::xp_park(xp, lane, r);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            Body::xp_pick(xp, lane, r);
            Body::store_runs(a, lane, r);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the buffer is free again before the next tile parks
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            Body::store_rows(a, lane, r);
        }
    };
    finish(r0, tw0);
    finish(r1, tw1);
}
inline bool wave_pipeline_enabled() {  // PHAST_WAVE_PIPELINE=1: two tiles per wave (wave_fft2_kernel); tools, A/B
    static const bool on = [] {
        const char *e = getenv("PHAST_WAVE_PIPELINE");
        return e && *e == '1';
    }();
    return on;
}

// host-side launcher
template <typename T, bool PRE_TW, bool TRANSPOSE>
hipError_t launch_wave_inst(hipStream_t stream, const TileArgs &a, bool query_only, int *blocks_per_cu, size_t *lds_out,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    using Body = WaveBody<T, PRE_TW, TRANSPOSE>;
    auto kern = wave_fft_kernel<T, PRE_TW, TRANSPOSE>;
    const size_t lds = Body::lds_bytes(a.tw_bits);
    if (lds_out) *lds_out = lds;
    if (lds > (size_t)160 * 1024) {
        if (query_only && blocks_per_cu) *blocks_per_cu = 0;
        return query_only ? hipSuccess : hipErrorInvalidValue;
    }
    static PerDeviceLimit lds_limit;
    if (hipError_t e = raise_lds_limit(lds_limit, reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    if (query_only) {
        if (blocks_per_cu) *blocks_per_cu = (int)((160 * 1024) / lds) < 4 ? (int)((160 * 1024) / lds) : 4;
        return hipSuccess;
    }
    // what the kernel's addressing takes for granted: the lane's part of every address fits a 32-bit BYTE offset, and the
    // packed (f32) tile's two columns -- in a first pass its two rows -- are adjacent in memory
    {
        const unsigned long long esz = sizeof(T) * (a.in_interleaved || a.out_interleaved ? 2u : 1u);
        const unsigned long long in_span = ((unsigned long long)(Body::TAUS - 1) * a.in_row_stride + Body::COLS) * esz;
        const unsigned long long out_span = TRANSPOSE ? ((unsigned long long)a.out_s1 + 64ull * a.out_row_stride) * esz
                                                      : ((unsigned long long)Body::COLS * a.out_s1 + 16ull * a.out_row_stride) * esz;
        if (in_span >= (1ull << 32) || out_span >= (1ull << 32)) return hipErrorInvalidValue;
        if (Body::VW == 2 && (TRANSPOSE ? a.out_row_stride != 1 : a.out_s1 != 1)) return hipErrorInvalidValue;
    }
    if (wave_pipeline_enabled() && (a.tiles_total % (2u * Body::WAVES)) == 0u) {  // two tiles per wave: half the waves
        auto kern2 = wave_fft2_kernel<T, PRE_TW, TRANSPOSE>;
        static PerDeviceLimit lds_limit2;
        if (hipError_t e = raise_lds_limit(lds_limit2, reinterpret_cast<const void *>(kern2), lds); e != hipSuccess) return e;
        const unsigned blocks2 = a.tiles_total / (2u * Body::WAVES);
        const unsigned stagger2 = a.tiles_total <= 2048u ? wave_stagger_setting() : 0u;
        if (ev_start && ev_stop)
            hipExtLaunchKernelGGL(kern2, dim3(blocks2), dim3(Body::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a, blocks2, stagger2);
        else
            hipLaunchKernelGGL(kern2, dim3(blocks2), dim3(Body::NT), lds, stream, a, blocks2, stagger2);
        return hipGetLastError();
    }
    const unsigned blocks = (a.tiles_total + Body::WAVES - 1) / Body::WAVES;
    // the stagger pays when every wave of the launch is resident at once and they would all move in step: at most one
    // tile per SIMD (256 CUs x 4).  With two tiles per SIMD the waves already interleave (2^21: 43.4 us without, 45.0 with).
    const unsigned stagger = a.tiles_total <= 1024u ? wave_stagger_setting() : 0u;
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(Body::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a, blocks, stagger);
    else
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(Body::NT), lds, stream, a, blocks, stagger);
    return hipGetLastError();
}

// Lane-by-lane host execution of one pass (tests/emu): the same phase functions, the swaps replaced by their definition
//   new[lane (b5 b4)][register (p3 p2 s1 s0)] = old[lane (s1 s0)][register (p3 p2 b5 b4)]      (same column)
template <typename T, bool PRE_TW, bool TRANSPOSE> void emulate_wave_pass(const TileArgs &a) {
    using Body = WaveBody<T, PRE_TW, TRANSPOSE>;
    using Regs = typename Body::Regs;
    const unsigned blocks_total = (a.tiles_total + Body::WAVES - 1) / Body::WAVES;
    std::vector<T> xp((size_t)2 * Body::XP + 1);
    for (unsigned block = 0; block < blocks_total; ++block)
        for (unsigned wave = 0; wave < (unsigned)Body::WAVES; ++wave) {
            if (block * Body::WAVES + wave >= a.tiles_total) continue;
            Regs regs[64], nxt[64];
            for (int l = 0; l < 64; ++l) {
                Body::locate(a, block, blocks_total, wave, regs[l]);
                Body::load_raw(a, l, regs[l]);
                Body::pre_twiddle(a, reinterpret_cast<const cx_t<T> *>(a.tw3), l, regs[l]);
                Body::step1(reinterpret_cast<const cx_t<T> *>(a.twr), l, regs[l]);
            }
            for (int l = 0; l < 64; ++l) {  // new[lane tau'][register (g, s)] = old[lane tau = s][register (g, tau')]
                nxt[l] = regs[l];
                const int col = Body::lcol_of(l), b = Body::tau_of(l);
                for (int q = 0; q < Body::P; ++q) {
                    const int src_lane = ((q & (Body::TAUS - 1)) << Body::LCL) | col, src_reg = (q & ~(Body::TAUS - 1)) | b;
                    nxt[l].re[q] = regs[src_lane].re[src_reg];
                    nxt[l].im[q] = regs[src_lane].im[src_reg];
                }
            }
            for (int l = 0; l < 64; ++l) Body::step2(nxt[l]);
            if constexpr (TRANSPOSE) {
                for (int l = 0; l < 64; ++l) Body::xp_park(xp.data(), l, nxt[l]);
                for (int l = 0; l < 64; ++l) {
                    Body::xp_pick(xp.data(), l, nxt[l]);
                    Body::store_runs(a, l, nxt[l]);
                }
            } else {
                for (int l = 0; l < 64; ++l) Body::store_rows(a, l, nxt[l]);
            }
        }
}

}  // namespace phast
