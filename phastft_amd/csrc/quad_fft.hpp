// quad_fft.hpp -- a 256-row pass of the multi-pass FFT on FOUR waves: 256 rows x 16 columns (f64: 128-byte rows) per
// 256-thread workgroup, 16 points per lane, radix 4 x 4 x 4 x 4 with ONE LDS exchange (between the waves) and one
// cross-lane exchange (v_permlane32/16_swap, as wave_fft.hpp) -- where the generic tile kernel (tile_fft.hpp, shape
// 256 x 16 at 8 points per thread) takes three radix steps with two LDS exchanges of four barriers.  This is the middle
// pass of the single 2^20-point f64 transform (BASELINE configs[1]: plan 64 . 256 . 64), the one kernel of that plan the
// copy-floor microbenchmark (profiles/r02_pass_floor_686_patterns.log) showed 2 us above its access pattern's floor.
//
//   lane = (col = lane & 15, tau = lane >> 4), wave w;   row n = 64 n_hi + 16 jj + 4 w + tau,   register q = 4 n_hi + jj
//   0. inter-pass twiddle W_{256 S}^(n lo) = W^((4 w + tau) lo) (W^(16 lo))^q        two table look-ups + a power ladder
//   1. radix-4 over n_hi (registers, stride 4)      -> k_a = bitrev2(s) at register jj + 4 s;   x W_256^(n_lo k_a)
//   2. radix-4 over jj   (registers, contiguous)    -> k_b = bitrev2(t) at register 4 s + t;    x W_64^((4 w + tau) k_b)
//   3. LDS exchange [register][wave][lane]: wave w' collects t = w' from all four waves  -> register 4 s + ws
//      radix-4 over ws = w                          -> k_c = bitrev2(u) at register 4 s + u;    x W_16^(tau k_c)
//   4. cross-lane exchange: lane bits (5, 4) <-> register bits (3, 2)                    -> register 4 tt + u (tt = tau)
//      radix-4 over tt (registers, stride 4)        -> k_d = bitrev2(v) at register u + 4 v
//   output row k = k_a + 4 k_b + 16 k_c + 64 k_d with k_a = bitrev2(tau'), k_b = bitrev2(w') from the lane and wave
// Phase functions are __host__ __device__ per lane (tests/emu emulates both exchanges).
#pragma once

#include <vector>

#include "tile_fft.hpp"
#include "wave_fft.hpp"

namespace phast {

// radix-R DIF on registers OFF, OFF + STRIDE, ...: afterwards position OFF + STRIDE p holds X[bitrev(p)]
template <typename T, int R, int OFF, int STRIDE, int P> PHAST_HD void fft_reg_dif_s(T (&re)[P], T (&im)[P]) {
    static_for<0, ilog2_c(R)>([&](auto st) {
        constexpr int SPAN = R >> (decltype(st)::value + 1);
        static_for<0, R / 2>([&](auto bi) {
            constexpr int B = decltype(bi)::value;
            constexpr int J = B % SPAN;
            constexpr int I0 = OFF + STRIDE * ((B / SPAN) * 2 * SPAN + J);
            constexpr int I1 = I0 + STRIDE * SPAN;
            T ar = re[I0], ai = im[I0], br = re[I1], bi_ = im[I1];
            re[I0] = ar + br;
            im[I0] = ai + bi_;
            T dr = ar - br, di = ai - bi_;
            mul_w<T, 2 * SPAN, J>(dr, di);
            re[I1] = dr;
            im[I1] = di;
        });
    });
}

template <typename T> struct QuadBody {
    using cx = cx_t<T>;
    // the register type of a lane: one f64, or (round 6) two adjacent f32 columns in 8 bytes -- wave_fft.hpp: WaveBody::V.  The
    // f32 tile is 256 rows x 32 columns (8192 points, 128-byte rows per plane) on the same four waves, 32 points per lane.
    using V = lane_vec_t<T>;
    static constexpr int VW = ScalarOf<V>::W, LV = VW == 2 ? 1 : 0;
    static constexpr int LR = 8, LCL = 4, LC = LCL + LV, ROWS = 256, COLS = 1 << LC, P = 16, WAVES = 4, NT = 256;
    static constexpr int EXCH = P * NT;  // registers (V) per plane of the exchange buffer [register][wave][lane]
    typedef V VU __attribute__((aligned(sizeof(T))));  // a register's worth in the caller's memory: element alignment only
    static constexpr int TWQ = 128;      // staged entries of the step-twiddle table: W_256^j (j < 64), then W_64^j (j < 64)
#ifndef PHAST_WQ_NT_LOADS
#define PHAST_WQ_NT_LOADS 1
#endif
#ifndef PHAST_WQ_NT_STORES
#define PHAST_WQ_NT_STORES 1
#endif
    static constexpr bool NT_LOAD = PHAST_WQ_NT_LOADS, NT_STORE = PHAST_WQ_NT_STORES;

    struct Regs {
        V re[P], im[P];
        unsigned xform, g0;
    };

    static size_t lds_bytes(unsigned tw_bits) {
        return (size_t)(3u << tw_bits) * sizeof(cx) + TWQ * sizeof(cx) + (size_t)2 * EXCH * sizeof(V);
    }

    PHAST_HD static int lcol_of(int lane) { return lane & 15; }
    PHAST_HD static int col_of(int lane) { return (lane & 15) << LV; }  // first (scalar) column of the lane
    PHAST_HD static int tau_of(int lane) { return lane >> 4; }

    PHAST_HD static void locate(const TileArgs &a, unsigned t, Regs &r) {  // as TileBody::locate
        const unsigned tile = ((a.tiles_total & 7u) == 0u) ? (t & 7u) * (a.tiles_total >> 3) + (t >> 3) : t;
        r.xform = tile >> (unsigned)__builtin_ctz(a.tiles_per_xform);  // a power of two (plan.hpp: geom_to_args)
        const unsigned ti = tile & (a.tiles_per_xform - 1u);
        r.g0 = a.cs_bits ? (((ti >> a.cb_bits) << a.cs_bits) | ((ti & ((1u << a.cb_bits) - 1u)) << LC)) : (ti << LC);
    }

    // row of register q: 64 (q >> 2) + 16 (q & 3) + 4 w + tau
    PHAST_HD static void load_raw(const TileArgs &a, int wave, int lane, Regs &r) {
        const size_t ubase = in_tile_base(a, r.xform, r.g0);
        // the wave's rows start at 4 wave (wave-uniform: the kernel takes it through readfirstlane); the lane's part of the
        // address is ONE 32-bit byte offset shared by the 32 loads (wave_fft.hpp: load_raw)
        const unsigned vbyte = ((unsigned)tau_of(lane) * (unsigned)a.in_row_stride + (unsigned)col_of(lane)) * (unsigned)sizeof(T);
        const T *pr = reinterpret_cast<const T *>(a.in_re) + ubase + (size_t)(4 * wave) * a.in_row_stride;
        const T *pi = reinterpret_cast<const T *>(a.in_im) + ubase + (size_t)(4 * wave) * a.in_row_stride;
        static_for<0, P>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const size_t urow = (size_t)(64 * (Q >> 2) + 16 * (Q & 3)) * a.in_row_stride;
            const VU *qr = reinterpret_cast<const VU *>(reinterpret_cast<const char *>(pr + urow) + vbyte);
            const VU *qi = reinterpret_cast<const VU *>(reinterpret_cast<const char *>(pi + urow) + vbyte);
            if constexpr (NT_LOAD) {
                r.re[Q] = __builtin_nontemporal_load(qr);
                r.im[Q] = __builtin_nontemporal_load(qi);
            } else {
                r.re[Q] = *qr;
                r.im[Q] = *qi;
            }
        });
    }

    // W^(n lo), n = (4 w + tau) + 16 (jj + 4 n_hi): base W^((4w+tau) lo) times the power (jj + 4 n_hi) of D = W^(16 lo)
    PHAST_HD static void pre_twiddle(const TileArgs &a, const cx *tw3, int wave, int lane, Regs &r) {
        V br, bi, dr, di;
        if constexpr (VW == 1) {
            const unsigned lo = ((r.g0 + (unsigned)col_of(lane)) >> a.tw_shift) & a.tw_mask;
            tw3_lookup<T>(tw3, a.tw_bits, (unsigned)(4 * wave + tau_of(lane)) * lo, br, bi);
            tw3_lookup<T>(tw3, a.tw_bits, 16u * lo, dr, di);
        } else {  // per column of the pair
            T b_r[2], b_i[2], d_r[2], d_i[2];
            static_for<0, 2>([&](auto e) {
                const unsigned lo = ((r.g0 + (unsigned)col_of(lane) + (unsigned)decltype(e)::value) >> a.tw_shift) & a.tw_mask;
                tw3_lookup<T>(tw3, a.tw_bits, (unsigned)(4 * wave + tau_of(lane)) * lo, b_r[e], b_i[e]);
                tw3_lookup<T>(tw3, a.tw_bits, 16u * lo, d_r[e], d_i[e]);
            });
            br = V{b_r[0], b_r[1]};
            bi = V{b_i[0], b_i[1]};
            dr = V{d_r[0], d_r[1]};
            di = V{d_i[0], d_i[1]};
        }
        tw_progression<V, P, 4>(br, bi, dr, di, [&](auto q, V wr, V wi) {  // register q = jj + 4 n_hi holds row 16 q + (4 wave + tau)
            cmul(r.re[decltype(q)::value], r.im[decltype(q)::value], wr, wi);
        });
    }

    // steps 1 and 2 (both in registers) with their twiddles; twq = [W_256^j | W_64^j], j < 64 each
    PHAST_HD static void steps12(const cx *twq, int wave, int lane, Regs &r) {
        const unsigned m = (unsigned)(4 * wave + tau_of(lane));  // the part of n_lo that lives in (wave, lane)
        static_for<0, 4>([&](auto jj) { fft_reg_dif_s<V, 4, decltype(jj)::value, 4, P>(r.re, r.im); });
        static_for<1, 4>([&](auto s) {  // k_a = bitrev2(s) != 0: x W_256^((16 jj + m) k_a) = W_16^(jj k_a) W_256^(m k_a)
            constexpr int KA = bitrev_c(decltype(s)::value, 2);
            const cx w = twq[m * KA];
            static_for<0, 4>([&](auto jj) {
                constexpr int Q = decltype(jj)::value + 4 * decltype(s)::value;
                cmul(r.re[Q], r.im[Q], w.x, w.y);
                constexpr int J = decltype(jj)::value * KA;  // W_16^J, J <= 9: W_16^(J) = -W_16^(J - 8) beyond the half turn
                mul_w<V, 16, J % 8>(r.re[Q], r.im[Q]);
                if constexpr (J >= 8) {
                    r.re[Q] = -r.re[Q];
                    r.im[Q] = -r.im[Q];
                }
            });
        });
        static_for<0, 4>([&](auto s) { fft_reg_dif<V, 4, 4 * decltype(s)::value, P>(r.re, r.im); });
        static_for<1, 4>([&](auto t) {  // k_b = bitrev2(t): x W_64^(m k_b)
            constexpr int KB = bitrev_c(decltype(t)::value, 2);
            const cx w = twq[64 + m * KB];
            static_for<0, 4>([&](auto s) {
                constexpr int Q = 4 * decltype(s)::value + decltype(t)::value;
                cmul(r.re[Q], r.im[Q], w.x, w.y);
            });
        });
    }
    // exchange addresses: writer (register Q, wave, lane); reader wave w' takes t = w', slot 4 s + ws <- (Q = 4 s + w', ws)
    template <int Q> PHAST_HD static int waddr(int wave, int lane) { return (Q * WAVES + wave) * 64 + lane; }
    template <int Q> PHAST_HD static int raddr(int wave, int lane) {
        constexpr int S = Q >> 2, WS = Q & 3;
        return ((4 * S + wave) * WAVES + WS) * 64 + lane;
    }
    // step 3 (radix-4 over the source wave) and its twiddle W_16^(tau k_c) = W_64^(4 tau k_c)
    PHAST_HD static void step3(const cx *twq, int lane, Regs &r) {
        static_for<0, 4>([&](auto s) { fft_reg_dif<V, 4, 4 * decltype(s)::value, P>(r.re, r.im); });
        const unsigned tau = (unsigned)tau_of(lane);
        static_for<1, 4>([&](auto u) {
            constexpr int KC = bitrev_c(decltype(u)::value, 2);
            const cx w = twq[64 + 4 * tau * KC];
            static_for<0, 4>([&](auto s) {
                constexpr int Q = 4 * decltype(s)::value + decltype(u)::value;
                cmul(r.re[Q], r.im[Q], w.x, w.y);
            });
        });
    }
    PHAST_HD static void step4(Regs &r) {
        static_for<0, 4>([&](auto u) { fft_reg_dif_s<V, 4, decltype(u)::value, 4, P>(r.re, r.im); });
    }

    PHAST_HD static unsigned br2(unsigned v) { return ((v & 1u) << 1) | (v >> 1); }
    PHAST_HD static unsigned krow_lane(int wave, int lane) { return br2((unsigned)tau_of(lane)) + 4u * br2((unsigned)wave); }
    template <int Q> PHAST_HD static constexpr unsigned krow_const() {  // register Q = u + 4 v
        return 16u * (unsigned)bitrev_c(Q & 3, 2) + 64u * (unsigned)bitrev_c(Q >> 2, 2);
    }
    // planar or (re, im) pairs, scaled (the last pass of an inverse transform) or not: decided ONCE per tile, outside the
    // sixteen stores -- inside it was a scalar branch per store and 32 multiplications by one in every forward pass
    template <bool PAIRS, bool SCALE> PHAST_HD static void store_as(const TileArgs &a, int wave, int lane, const Regs &r) {
        const size_t base = (size_t)(r.g0 & ((1u << a.out_lo_bits) - 1u)) * a.out_s1 + (size_t)(r.g0 >> a.out_lo_bits) * a.out_s2 +
                            (size_t)r.xform * a.out_dist + (size_t)(4u * br2((unsigned)wave)) * a.out_row_stride;
        const unsigned vbyte = ((unsigned)col_of(lane) * (unsigned)a.out_s1 + br2((unsigned)tau_of(lane)) * (unsigned)a.out_row_stride) *
                               (unsigned)sizeof(T);
        const T scale = (T)a.scale;
        static_for<0, P>([&](auto Q) {
            const size_t at = base + (size_t)krow_const<decltype(Q)::value>() * a.out_row_stride;
            V re = r.re[Q], im = r.im[Q];
            if constexpr (SCALE) {
                re *= scale;
                im *= scale;
            }
            if constexpr (!PAIRS) {
                VU *qr = reinterpret_cast<VU *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out_re) + at) + vbyte);
                VU *qi = reinterpret_cast<VU *>(reinterpret_cast<char *>(reinterpret_cast<T *>(a.out_im) + at) + vbyte);
                if constexpr (NT_STORE) {
                    __builtin_nontemporal_store(re, qr);
                    __builtin_nontemporal_store(im, qi);
                } else {
                    *qr = re;
                    *qi = im;
                }
            } else {
                const V x = a.out_interleaved == 2 ? im : re, y = a.out_interleaved == 2 ? re : im;
                cx *q = reinterpret_cast<cx *>(reinterpret_cast<char *>(reinterpret_cast<cx *>(a.out_re) + at) + 2u * vbyte);
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
        });
    }
    PHAST_HD static void store(const TileArgs &a, int wave, int lane, const Regs &r) {
        const bool scaled = a.scale != 1.0;
        if (!a.out_interleaved) {
            if (scaled) store_as<false, true>(a, wave, lane, r);
            else store_as<false, false>(a, wave, lane, r);
        } else {
            if (scaled) store_as<true, true>(a, wave, lane, r);
            else store_as<true, false>(a, wave, lane, r);
        }
    }
};

// lane bits (5, 4) <-> register bits (3, 2)
template <typename T> __device__ __forceinline__ void quad_lane_exchange(T (&re)[16], T (&im)[16]) {
    static_for<0, 16>([&](auto p) {
        constexpr int Pp = decltype(p)::value;
        if constexpr ((Pp & 8) == 0) {
            swap_pair<true>(re[Pp], re[Pp | 8]);
            swap_pair<true>(im[Pp], im[Pp | 8]);
        }
    });
    static_for<0, 16>([&](auto p) {
        constexpr int Pp = decltype(p)::value;
        if constexpr ((Pp & 4) == 0) {
            swap_pair<false>(re[Pp], re[Pp | 4]);
            swap_pair<false>(im[Pp], im[Pp | 4]);
        }
    });
}

// (Staggering the workgroups' or the waves' first loads, as wave_fft.hpp does, buys nothing here: profiles/r02_stagger.log, and again
//  on this round's SGPR-base kernel with groups (block >> 3) & mask sleeping 2 .. 16 units: profiles/r06_quad_stagger_ab.log.)
// ONE: the grid has one workgroup per tile (a single transform: 256 tiles on 256 CUs) -- no tile loop, so the second
// load site, its hoisted 64-bit lane offsets and the SGPRs parked in VGPR lanes across the loop are gone.
template <typename T, bool ONE> __global__ void __launch_bounds__(256) quad_fft_kernel(const TileArgs a) {
    using Body = QuadBody<T>;
    using cx = cx_t<T>;
    pin_tile_args(a);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using V = typename Body::V;
    V *ex_re = reinterpret_cast<V *>(smem);
    V *ex_im = ex_re + Body::EXCH;
    cx *l_tw3 = reinterpret_cast<cx *>(ex_im + Body::EXCH);
    cx *l_twq = l_tw3 + (3u << a.tw_bits);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row bases and table rows live in SGPRs
#ifdef PHAST_TRACE  // phase stamps (tools/trace_wave_quad.py): one row of 16 per wave of the first tile, lane 0 writes
    int stamp_i = 0;
    auto stamp = [&](bool drain) {
        if (a.trace != nullptr) {
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (lane == 0 && stamp_i < 16) a.trace[((size_t)blockIdx.x * 4 + (size_t)wave) * 16 + stamp_i] = now;
            ++stamp_i;
        }
    };
#define PHAST_STAMP(d) stamp(d)
#else
#define PHAST_STAMP(d)
#endif
    PHAST_STAMP(false);  // 0: entry
    typename Body::Regs r;
    // tables: global loads first, the first tile's loads right behind them (loads return in order: see wave_fft.hpp).
    // Four named registers, not an array: an array here lands in scratch memory as soon as control flow separates
    // the loads from the LDS stores (tests/test_kernel_resources.py watches it).
    const unsigned n_tw3 = 3u << a.tw_bits;
    const cx *g_tw3 = reinterpret_cast<const cx *>(a.tw3);
    const unsigned i0 = (unsigned)tid, i1 = i0 + Body::NT, i2 = i1 + Body::NT, i3 = i2 + Body::NT;
    const cx twq_stage = reinterpret_cast<const cx *>(a.twr)[tid & (Body::TWQ - 1)];
    const cx ts0 = g_tw3[i0 < n_tw3 ? i0 : 0u], ts1 = g_tw3[i1 < n_tw3 ? i1 : 0u], ts2 = g_tw3[i2 < n_tw3 ? i2 : 0u],
             ts3 = g_tw3[i3 < n_tw3 ? i3 : 0u];
    unsigned t = blockIdx.x;
    if (t >= a.tiles_total) return;  // uniform over the workgroup
    Body::locate(a, t, r);
    Body::load_raw(a, wave, lane, r);
    PHAST_STAMP(false);  // 1: loads issued
    if (tid < Body::TWQ) l_twq[tid] = twq_stage;
    if (i0 < n_tw3) l_tw3[i0] = ts0;
    if (i1 < n_tw3) l_tw3[i1] = ts1;
    if (i2 < n_tw3) l_tw3[i2] = ts2;
    if (i3 < n_tw3) l_tw3[i3] = ts3;
    for (unsigned i = i3 + Body::NT; i < n_tw3; i += Body::NT) l_tw3[i] = g_tw3[i];
    for (;;) {
        __syncthreads();  // tables visible / the previous tile's exchange reads done
        PHAST_STAMP(false);  // 2: tables staged, first barrier passed
        PHAST_STAMP(true);   // 3: loads back
        Body::pre_twiddle(a, l_tw3, wave, lane, r);
        Body::steps12(l_twq, wave, lane, r);
        PHAST_STAMP(false);  // 4: pre-twiddle + two radix-4 steps
        static_for<0, 16>([&](auto Q) {
            ex_re[Body::template waddr<decltype(Q)::value>(wave, lane)] = r.re[Q];
            ex_im[Body::template waddr<decltype(Q)::value>(wave, lane)] = r.im[Q];
        });
        PHAST_STAMP(true);   // 5: exchange written
        __syncthreads();
        PHAST_STAMP(false);  // 6: barrier passed
        static_for<0, 16>([&](auto Q) {
            r.re[Q] = ex_re[Body::template raddr<decltype(Q)::value>(wave, lane)];
            r.im[Q] = ex_im[Body::template raddr<decltype(Q)::value>(wave, lane)];
        });
        PHAST_STAMP(true);   // 7: exchange read
        Body::step3(l_twq, lane, r);
        quad_lane_exchange<V>(r.re, r.im);
        Body::step4(r);
        PHAST_STAMP(false);  // 8: last two radix-4 steps
        Body::store(a, wave, lane, r);
        PHAST_STAMP(false);  // 9: stores issued
        PHAST_STAMP(true);   // 10: stores retired
        if constexpr (ONE) break;
        t += gridDim.x;
        if (t >= a.tiles_total) break;
        Body::locate(a, t, r);
        Body::load_raw(a, wave, lane, r);
    }
#undef PHAST_STAMP
}

template <typename T>
hipError_t launch_quad_inst(unsigned grid, hipStream_t stream, const TileArgs &a, bool query_only, int *blocks_per_cu,
                            size_t *lds_out, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    using Body = QuadBody<T>;
    const bool one = !query_only && grid == a.tiles_total;
    auto kern = one ? quad_fft_kernel<T, true> : quad_fft_kernel<T, false>;
    const size_t lds = Body::lds_bytes(a.tw_bits);
    if (lds_out) *lds_out = lds;
    if (lds > (size_t)160 * 1024) {
        if (query_only && blocks_per_cu) *blocks_per_cu = 0;
        return query_only ? hipSuccess : hipErrorInvalidValue;
    }
    static PerDeviceLimit lds_limit[2];
    if (hipError_t e = raise_lds_limit(lds_limit[one], reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    if (query_only) {
        if (blocks_per_cu) *blocks_per_cu = (int)((160 * 1024) / lds) < 2 ? (int)((160 * 1024) / lds) : 2;
        return hipSuccess;
    }
    {  // the lane's part of every address is a 32-bit byte offset; the packed (f32) tile's two columns are adjacent in memory
        const unsigned long long esz = sizeof(T) * (a.out_interleaved ? 2u : 1u);
        if (((unsigned long long)4 * a.in_row_stride + Body::COLS) * sizeof(T) >= (1ull << 32) ||
            ((unsigned long long)Body::COLS * a.out_s1 + 4ull * a.out_row_stride) * esz >= (1ull << 32))
            return hipErrorInvalidValue;
        if (Body::VW == 2 && a.out_s1 != 1) return hipErrorInvalidValue;
    }
    if (ev_start && ev_stop)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), lds, stream, a);
    return hipGetLastError();
}

// host emulation of one pass: the same phase functions, both exchanges by their definitions
template <typename T> void emulate_quad_pass(const TileArgs &a) {
    using Body = QuadBody<T>;
    using Regs = typename Body::Regs;
    std::vector<typename Body::V> ex_re(Body::EXCH), ex_im(Body::EXCH);
    std::vector<Regs> regs(Body::NT), nxt(Body::NT);
    const cx_t<T> *tw3 = reinterpret_cast<const cx_t<T> *>(a.tw3), *twq = reinterpret_cast<const cx_t<T> *>(a.twr);
    for (unsigned t = 0; t < a.tiles_total; ++t) {
        for (int tid = 0; tid < Body::NT; ++tid) {
            const int lane = tid & 63, wave = tid >> 6;
            Body::locate(a, t, regs[tid]);
            Body::load_raw(a, wave, lane, regs[tid]);
            Body::pre_twiddle(a, tw3, wave, lane, regs[tid]);
            Body::steps12(twq, wave, lane, regs[tid]);
            static_for<0, 16>([&](auto Q) {
                ex_re[Body::template waddr<decltype(Q)::value>(wave, lane)] = regs[tid].re[Q];
                ex_im[Body::template waddr<decltype(Q)::value>(wave, lane)] = regs[tid].im[Q];
            });
        }
        for (int tid = 0; tid < Body::NT; ++tid) {
            const int lane = tid & 63, wave = tid >> 6;
            static_for<0, 16>([&](auto Q) {
                regs[tid].re[Q] = ex_re[Body::template raddr<decltype(Q)::value>(wave, lane)];
                regs[tid].im[Q] = ex_im[Body::template raddr<decltype(Q)::value>(wave, lane)];
            });
            Body::step3(twq, lane, regs[tid]);
        }
        // lane bits (5, 4) <-> register bits (3, 2): new[lane (b5 b4)][reg (s1 s0 | u)] = old[lane (s1 s0)][reg (b5 b4 | u)]
        for (int tid = 0; tid < Body::NT; ++tid) {
            const int lane = tid & 63, wave = tid >> 6, col = Body::lcol_of(lane), b = lane >> 4;
            nxt[tid] = regs[tid];
            for (int q = 0; q < 16; ++q) {
                const int src = wave * 64 + (((q >> 2) << 4) | col), src_reg = (b << 2) | (q & 3);
                nxt[tid].re[q] = regs[src].re[src_reg];
                nxt[tid].im[q] = regs[src].im[src_reg];
            }
        }
        for (int tid = 0; tid < Body::NT; ++tid) {
            Body::step4(nxt[tid]);
            Body::store(a, tid >> 6, tid & 63, nxt[tid]);
        }
    }
}

}  // namespace phast
