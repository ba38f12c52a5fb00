// tile_fft.hpp -- one pass of the multi-pass power-of-two FFT for gfx950 (MI355X).
//
// Replaces, on the GPU, the reference's per-stage sweeps: kernels/dit.rs:971-1115
// (fft_dit_chunk_n_*), kernels/dit.rs:132-967 (chunk 8..64), kernels/codelets.rs:34-498, the
// recursion of algorithms/dit.rs:33-164 and -- in the first pass -- the bit-reversal permutation of
// algorithms/bravo.rs (the permutation is folded into the first pass's store pattern).
//
// Design.  N = 2^L is factored into 2 or 3 passes N = R_A * R_B (* R_C).  A pass
// runs ROWS-point FFTs along a strided axis; a workgroup owns a tile of ROWS x COLS points (COLS
// adjacent columns => every global access of a wave covers COLS*sizeof(T)-byte contiguous segments).
// Each thread holds 16 complex points in registers.  Inside the tile the ROWS-point FFT is a
// decimation-in-frequency Cooley-Tukey split 16 x R2 (x R3); the radix-16/8/4/2 butterflies run in
// registers with literal twiddles, and data moves between the radix steps through LDS:
//     exchange 1 layout [n'][k1]  (row = n'*(16 + PADR) + k1; one pad row per n' when a 32-lane LDS access
//                                  group spans G > 1 rows, so the G rows of a group fall on distinct banks)
//     exchange 2 layout [n3][k2][k1]
// Lanes are (column fastest, butterfly index); both exchanges are bank-conflict free for reads
// (32-lane groups see G = 32/COLS consecutive rows) and writes.
// Pass A (TRANSPOSE) additionally transposes through LDS (exchange 3, [col][k], padded) so that each
// column's ROWS outputs leave as one contiguous run: this is where the digit reversal happens.  Later
// passes (PRE_TW) multiply by the inter-pass twiddle W_{ROWS*S}^{row*lo} on load, looked up as a product
// of three small LDS tables.  No MFMA: 6 FMA-class ops per 64 B moved per radix-2 stage -- HBM-bound.
//
// The per-thread work is written as __host__ __device__ phase functions (TileBody) so that the very
// same index arithmetic is executed thread-by-thread on the CPU by csrc/emu.hip (tests/test_emulator.py)
// -- the build container has no GPU.
#pragma once

#include <hip/hip_ext.h>

#include "common.hpp"

namespace phast {

// ---- literal twiddles: (re, im) *= W_N^J = exp(-2*pi*i*J/N), N in {2,4,8,16}, 0 <= J < N/2 ----
template <typename T, int N, int J> PHAST_HD void mul_w(T &re, T &im) {
    constexpr T S = (T)0.70710678118654752440L;
    if constexpr (J == 0) {
    } else if constexpr (4 * J == N) {  // -i
        T t = re;
        re = im;
        im = -t;
    } else if constexpr (8 * J == N) {  // (1 - i)/sqrt2
        T r = (re + im) * S;
        T i = (im - re) * S;
        re = r;
        im = i;
    } else if constexpr (8 * J == 3 * N) {  // (-1 - i)/sqrt2
        T r = (im - re) * S;
        T i = -(re + im) * S;
        re = r;
        im = i;
    } else {
        static_assert(N == 16, "general twiddle only for N=16");
        constexpr T C1 = (T)0.92387953251128675613L;  // cos(pi/8)
        constexpr T S1 = (T)0.38268343236508977173L;  // sin(pi/8)
        constexpr T c = (J == 1) ? C1 : (J == 3) ? S1 : (J == 5) ? -S1 : -C1;
        constexpr T s = (J == 1 || J == 7) ? S1 : C1;
        T r = re * c + im * s;
        T i = im * c - re * s;
        re = r;
        im = i;
    }
}

// In-register radix-R decimation-in-frequency FFT on regs [OFF, OFF+R).
// Afterwards position OFF+p holds X[bitrev(p)].
template <typename T, int R, int OFF> PHAST_HD void fft_reg_dif(T (&re)[16], T (&im)[16]) {
    static_for<0, ilog2_c(R)>([&](auto st) {
        constexpr int SPAN = R >> (decltype(st)::value + 1);
        static_for<0, R / 2>([&](auto bi) {
            constexpr int B = decltype(bi)::value;
            constexpr int J = B % SPAN;
            constexpr int I0 = OFF + (B / SPAN) * 2 * SPAN + J;
            constexpr int I1 = I0 + SPAN;
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

template <typename T, int LR, int LC, bool PRE_TW, bool TRANSPOSE, bool SEQ> struct TileBody {
    using cx = cx_t<T>;
    static constexpr int ROWS = 1 << LR;
    static constexpr int COLS = 1 << LC;
    static constexpr int NT = ROWS * COLS / 16;  // threads per workgroup, 16 points each
    static constexpr int M = ROWS / 16;          // threads per column
    static constexpr int G = COLS >= 32 ? 1 : 32 / COLS;  // rows seen by one 32-lane LDS access group
    static constexpr bool THREE = LR > 8;        // 16 x 16 x R3, else 16 x R2
    static constexpr int R2 = THREE ? 16 : M;
    static constexpr int R3 = THREE ? M / 16 : 1;
    static constexpr bool PLANE_SEQ = SEQ;  // exchange re and im one after the other (half the LDS, twice the barriers)
    static constexpr int CS = ROWS + G;                // padded column stride of the transposing exchange
    static constexpr int E1S = 16 + (G > 1 ? 1 : 0);   // rows per n' in exchange 1 (16 used + padding)
    static constexpr int EXCH_E1 = M * E1S * COLS;
    static constexpr int EXCH_E3 = TRANSPOSE ? COLS * CS : 0;
    static constexpr int EXCH = EXCH_E1 > EXCH_E3 ? EXCH_E1 : EXCH_E3;
    // Non-temporal global accesses when a tile row is a whole 128-byte line or more: every byte is touched once
    // per pass, and the strided-copy microbenchmark gains 5-10 % (profiles/r01_strided_copy_nt.log).  Narrower
    // rows share their line with the neighbouring tile and NEED the L2 (nt loads cost 25 % there).
    static constexpr bool NT_HINT = COLS * sizeof(T) >= 128;
    static_assert(LR >= 6 && LR <= 10, "tile FFT length 64..1024");
    static_assert(NT <= 1024, "at most 1024 threads per workgroup");

    static size_t lds_bytes(unsigned tw_bits) {
        size_t exch = (size_t)EXCH * sizeof(T) * (PLANE_SEQ ? 1 : 2);
        size_t tw3 = PRE_TW ? (size_t)(3u << tw_bits) * sizeof(cx) : 0;
        return exch + tw3 + 64 * sizeof(cx);
    }

    // what a workgroup shares (LDS on the GPU, plain host arrays in the emulator);
    // ex_im == ex_re when PLANE_SEQ
    struct Shared {
        T *ex_re;
        T *ex_im;
        const cx *tw3;
        const cx *twr;
    };
    // what a thread keeps in registers across barriers
    struct Regs {
        T re[16], im[16];
        unsigned xform, g0;
    };

    PHAST_HD static int col_of(int tid) { return tid & (COLS - 1); }
    PHAST_HD static int tau_of(int tid) { return tid >> LC; }

    // tile index -> (transform, first column).  XCD-aware order: workgroup b runs on XCD b%8 (observed);
    // each XCD gets one contiguous run of tiles so that neighbouring column groups -- which share DRAM
    // pages and L2 lines -- meet in one L2.
    PHAST_HD static void locate(const TileArgs &a, unsigned t, Regs &r) {
        const unsigned tile = ((a.tiles_total & 7u) == 0u) ? (t & 7u) * (a.tiles_total >> 3) + (t >> 3) : t;
        r.xform = tile / a.tiles_per_xform;
        r.g0 = (tile - r.xform * a.tiles_per_xform) << LC;
    }

    // ---------------- load rows n = n1*M + tau, apply the inter-pass twiddle ----------------
    // Addresses are (wave-uniform 64-bit base) + (32-bit per-lane element offset): the tile's columns share
    // the high part of in_col (tiles are COLS-aligned and COLS <= 2^log_s_in), the row n1*M is uniform, and
    // only tau*2^log_s_in + col differs between lanes -- one VGPR for all 16 loads (saddr addressing).
    PHAST_HD static void load_raw(const TileArgs &a, int tid, Regs &r) {
        const int col = col_of(tid), tau = tau_of(tid);
        const unsigned lo0 = r.g0 & ((1u << a.log_s_in) - 1u);
        const size_t ubase = (size_t)r.xform * a.in_dist + (((size_t)(r.g0 >> a.log_s_in) << (a.log_s_in + LR)) | lo0);
        const unsigned voff = ((unsigned)tau << a.log_s_in) + (unsigned)col;
        if (!a.in_interleaved) {
            const T *pr = reinterpret_cast<const T *>(a.in_re) + ubase;
            const T *pi = reinterpret_cast<const T *>(a.in_im) + ubase;
            static_for<0, 16>([&](auto n1) {
                const size_t urow = (size_t)(decltype(n1)::value * M) << a.log_s_in;
                if constexpr (NT_HINT) {
                    r.re[n1] = __builtin_nontemporal_load(pr + urow + voff);
                    r.im[n1] = __builtin_nontemporal_load(pi + urow + voff);
                } else {
                    r.re[n1] = (pr + urow)[voff];
                    r.im[n1] = (pi + urow)[voff];
                }
            });
        } else {
            const cx *pz = reinterpret_cast<const cx *>(a.in_re) + ubase;
            static_for<0, 16>([&](auto n1) {
                const size_t urow = (size_t)(decltype(n1)::value * M) << a.log_s_in;
                cx v = (pz + urow)[voff];
                r.re[n1] = a.in_interleaved == 2 ? v.y : v.x;
                r.im[n1] = a.in_interleaved == 2 ? v.x : v.y;
            });
        }
    }
    // inter-pass twiddle W_{ROWS*S}^{row*lo} on the freshly loaded rows (needs the LDS tables)
    PHAST_HD static void pre_twiddle(const TileArgs &a, const Shared &sh, int tid, Regs &r) {
        if constexpr (PRE_TW) {
            const int col = col_of(tid), tau = tau_of(tid);
            const unsigned lo0 = r.g0 & ((1u << a.log_s_in) - 1u);
            const unsigned lo = lo0 + (unsigned)col;
            const unsigned e0 = (unsigned)tau * lo, de = (unsigned)M * lo;  // exponent of row n1*M + tau: e0 + n1*de
            static_for<0, 16>([&](auto n1) {
                T wr, wi;
                tw3_lookup<T>(sh.tw3, a.tw_bits, e0 + decltype(n1)::value * de, wr, wi);
                cmul(r.re[n1], r.im[n1], wr, wi);
            });
        }
    }

    PHAST_HD static void twr_lookup(const Shared &sh, unsigned e, T &wr, T &wi) {  // W_ROWS^e, e < ROWS
        cx w0 = sh.twr[e & 31u], w1 = sh.twr[32u + (e >> 5)];
        wr = w0.x * w1.x - w0.y * w1.y;
        wi = w0.x * w1.y + w0.y * w1.x;
    }

    // ---------------- step 1: radix-16 over n1; register p then holds k1 = bitrev4(p) ----------------
    PHAST_HD static void step1(const Shared &sh, int tid, Regs &r) {
        const int tau = tau_of(tid);
        fft_reg_dif<T, 16, 0>(r.re, r.im);
        static_for<1, 16>([&](auto p) {
            constexpr int K1 = bitrev_c(decltype(p)::value, 4);
            T wr, wi;
            twr_lookup(sh, (unsigned)tau * K1, wr, wi);  // W_ROWS^(n' * k1), n' = tau
            cmul(r.re[p], r.im[p], wr, wi);
        });
    }

    // ---------------- step 2 ----------------
    // two-level tiles: 16/R2 radix-R2 butterflies per thread, butterfly i has k1 = tau + R2*i (final step)
    // three-level tiles: one radix-16 over n2 for (k1, n3) = (tau & 15, tau >> 4), then W_{16*R3}^{n3*k2}
    PHAST_HD static void step2(const Shared &sh, int tid, Regs &r) {
        if constexpr (!THREE) {
            static_for<0, 16 / R2>([&](auto i) { fft_reg_dif<T, R2, decltype(i)::value * R2>(r.re, r.im); });
        } else {
            const int n3 = tau_of(tid) >> 4;
            fft_reg_dif<T, 16, 0>(r.re, r.im);
            static_for<1, 16>([&](auto p) {
                constexpr int K2 = bitrev_c(decltype(p)::value, 4);
                T wr, wi;
                twr_lookup(sh, 16u * (unsigned)n3 * K2, wr, wi);
                cmul(r.re[p], r.im[p], wr, wi);
            });
        }
    }

    // ---------------- step 3 (three-level only): 16/R3 radix-R3 butterflies, k2 = (tau >> 4) + R3*i ----------------
    PHAST_HD static void step3(Regs &r) {
        if constexpr (THREE)
            static_for<0, 16 / R3>([&](auto i) { fft_reg_dif<T, R3, decltype(i)::value * R3>(r.re, r.im); });
    }

    // frequency index (row of the tile FFT output) held by register P after the last step
    //   = krow_lane(tid) + krow_const<P>()
    PHAST_HD static unsigned krow_lane(int tid) { return (unsigned)tau_of(tid); }  // (tau&15) + 16*(tau>>4) = tau
    template <int P> PHAST_HD static constexpr unsigned krow_const() {
        if constexpr (!THREE) {
            constexpr int I = P / R2, PP = P % R2;
            return (unsigned)(R2 * I) + 16u * bitrev_c(PP, ilog2_c(R2));
        } else {
            constexpr int I = P / R3, PP = P % R3;
            return 16u * (unsigned)(R3 * I) + 256u * bitrev_c(PP, ilog2_c(R3));
        }
    }
    template <int P> PHAST_HD static unsigned krow(int tid) { return krow_lane(tid) + krow_const<P>(); }

    // ---------------- LDS exchange addresses; E = 1, 2 (three-level only), 3 (transpose only) ----------------
    template <int E, int P> PHAST_HD static int waddr(int tid) {
        const int col = col_of(tid), tau = tau_of(tid);
        if constexpr (E == 1) {  // row n'*E1S + k1 with n' = tau, k1 = bitrev4(P)
            constexpr int K1 = bitrev_c(P, 4);
            return ((tau * E1S) << LC) + col + (K1 << LC);
        } else if constexpr (E == 2) {  // [n3][k2][k1]
            constexpr int K2 = bitrev_c(P, 4);
            return ((((tau >> 4) * 256) + (tau & 15)) << LC) + col + ((K2 * 16) << LC);
        } else {  // [col][k]
            return col * CS + (int)krow_lane(tid) + (int)krow_const<P>();
        }
    }
    template <int E, int P> PHAST_HD static int raddr(int tid) {
        const int col = col_of(tid), tau = tau_of(tid);
        if constexpr (E == 1) {
            if constexpr (!THREE) {
                constexpr int I = P / R2, N2 = P % R2;  // n' = n2, k1 = tau + R2*I
                return (tau << LC) + col + ((N2 * E1S + R2 * I) << LC);
            } else {  // n' = n2*R3 + n3 with n2 = P, n3 = tau >> 4; k1 = tau & 15
                return ((((tau >> 4) * E1S) + (tau & 15)) << LC) + col + ((P * R3 * E1S) << LC);
            }
        } else if constexpr (E == 2) {
            constexpr int I = P / R3, N3 = P % R3;  // k2 = (tau >> 4) + R3*I
            return ((((tau >> 4) * 16) + (tau & 15)) << LC) + col + (((N3 * 16 + R3 * I) * 16) << LC);
        } else {
            const int f = P * NT + tid;
            return (f >> LR) * CS + (f & (ROWS - 1));
        }
    }
    // plane 0 = real parts through ex_re, plane 1 = imaginary parts through ex_im
    template <int E> PHAST_HD static void ex_write(const Shared &sh, int tid, const Regs &r, int plane) {
        T *dst = plane ? sh.ex_im : sh.ex_re;
        static_for<0, 16>([&](auto P) { dst[waddr<E, decltype(P)::value>(tid)] = plane ? r.im[P] : r.re[P]; });
    }
    template <int E> PHAST_HD static void ex_read(const Shared &sh, int tid, Regs &r, int plane) {
        const T *src = plane ? sh.ex_im : sh.ex_re;
        static_for<0, 16>([&](auto P) {
            const T v = src[raddr<E, decltype(P)::value>(tid)];
            if (plane)
                r.im[P] = v;
            else
                r.re[P] = v;
        });
    }

    // ---------------- store ----------------
    // out_col(g) for the tile's first column; tiles are COLS-aligned and COLS <= 2^out_lo_bits, so column
    // g0 + c is out_col(g0) + c*out_s1 (checked by make_passes)
    PHAST_HD static size_t out_base(const TileArgs &a, const Regs &r) {
        return (size_t)(r.g0 & ((1u << a.out_lo_bits) - 1u)) * a.out_s1 + (size_t)(r.g0 >> a.out_lo_bits) * a.out_s2 +
               (size_t)r.xform * a.out_dist;
    }
    PHAST_HD static void put(const TileArgs &a, size_t ubase, unsigned voff, T re, T im) {
        const T scale = (T)a.scale;
        if (!a.out_interleaved) {
            if constexpr (NT_HINT) {
                __builtin_nontemporal_store(re * scale, reinterpret_cast<T *>(a.out_re) + ubase + voff);
                __builtin_nontemporal_store(im * scale, reinterpret_cast<T *>(a.out_im) + ubase + voff);
            } else {
                (reinterpret_cast<T *>(a.out_re) + ubase)[voff] = re * scale;
                (reinterpret_cast<T *>(a.out_im) + ubase)[voff] = im * scale;
            }
        } else {
            cx v;
            v.x = (a.out_interleaved == 2 ? im : re) * scale;
            v.y = (a.out_interleaved == 2 ? re : im) * scale;
            (reinterpret_cast<cx *>(a.out_re) + ubase)[voff] = v;
        }
    }
    PHAST_HD static void store(const TileArgs &a, int tid, const Regs &r) {
        const size_t base = out_base(a, r);
        if constexpr (!TRANSPOSE) {  // register P holds row krow_lane + krow_const<P> of column g0 + col
            const unsigned voff = (unsigned)col_of(tid) * (unsigned)a.out_s1 + krow_lane(tid) * (unsigned)a.out_row_stride;
            static_for<0, 16>([&](auto P) {
                put(a, base + (size_t)krow_const<decltype(P)::value>() * a.out_row_stride, voff, r.re[P], r.im[P]);
            });
        } else {  // after exchange 3: register P holds flat element f = P*NT + tid of the [col][k] tile
            if constexpr (NT >= ROWS) {  // f -> column (P*NT >> LR) + (tid >> LR), row tid & (ROWS-1)
                const unsigned voff = (unsigned)(tid >> LR) * (unsigned)a.out_s1 +
                                      (unsigned)(tid & (ROWS - 1)) * (unsigned)a.out_row_stride;
                static_for<0, 16>([&](auto P) {
                    constexpr int C0 = (decltype(P)::value * NT) >> LR;
                    put(a, base + (size_t)C0 * a.out_s1, voff, r.re[P], r.im[P]);
                });
            } else {  // f -> column P*NT >> LR, row (P*NT & (ROWS-1)) + tid
                const unsigned voff = (unsigned)tid * (unsigned)a.out_row_stride;
                static_for<0, 16>([&](auto P) {
                    constexpr int C0 = (decltype(P)::value * NT) >> LR, K0 = (decltype(P)::value * NT) & (ROWS - 1);
                    put(a, base + (size_t)C0 * a.out_s1 + (size_t)K0 * a.out_row_stride, voff, r.re[P], r.im[P]);
                });
            }
        }
    }
};

#ifndef PHAST_MIN_WAVES
#define PHAST_MIN_WAVES(LR, LC) 1
#endif
template <typename T, int LR, int LC, bool PRE_TW, bool TRANSPOSE, bool SEQ>
__global__ void __launch_bounds__(1 << (LR + LC - 4), PHAST_MIN_WAVES(LR, LC)) tile_fft_kernel(const TileArgs a) {
    using Body = TileBody<T, LR, LC, PRE_TW, TRANSPOSE, SEQ>;
    using cx = cx_t<T>;
    constexpr int NT = Body::NT;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *ex_re = reinterpret_cast<T *>(smem);
    cx *l_tw3 = reinterpret_cast<cx *>(smem + (size_t)Body::EXCH * sizeof(T) * (Body::PLANE_SEQ ? 1 : 2));
    cx *l_twr = l_tw3 + (PRE_TW ? (3u << a.tw_bits) : 0u);
    const typename Body::Shared sh{ex_re, Body::PLANE_SEQ ? ex_re : ex_re + Body::EXCH, l_tw3, l_twr};

    const int tid = threadIdx.x;
    // Phase stamps exist only in the -DPHAST_TRACE build (tools/trace_tile.py): even behind a uniform branch the
    // drains below wreck register allocation (256 VGPRs + 300 spills), so the product kernels carry none.
#ifdef PHAST_TRACE
    int stamp_i = 0;
    auto stamp = [&]() {
        if (a.trace != nullptr) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (tid == 0 && stamp_i < 16) a.trace[(size_t)blockIdx.x * 16 + stamp_i] = now;
            ++stamp_i;
        }
    };
#else
    auto stamp = []() {};
#endif
    stamp();  // 0: kernel entry
    // the first tile's global loads are issued before the table loads so the two latencies overlap
    typename Body::Regs r;
    unsigned t = blockIdx.x;
    if (t < a.tiles_total) {
        Body::locate(a, t, r);
        Body::load_raw(a, tid, r);
    }
    for (int i = tid; i < 64; i += NT) l_twr[i] = reinterpret_cast<const cx *>(a.twr)[i];
    if constexpr (PRE_TW)
        for (unsigned i = tid; i < (3u << a.tw_bits); i += NT) l_tw3[i] = reinterpret_cast<const cx *>(a.tw3)[i];
    __syncthreads();
    stamp();  // 1: twiddle tables in LDS

    // one exchange: barrier (previous readers done), write, barrier, read -- per plane when PLANE_SEQ
    auto exchange = [&](auto e, typename Body::Regs &r) {
        constexpr int E = decltype(e)::value;
        if constexpr (!Body::PLANE_SEQ) {
            __syncthreads();
            Body::template ex_write<E>(sh, tid, r, 0);
            Body::template ex_write<E>(sh, tid, r, 1);
            __syncthreads();
            Body::template ex_read<E>(sh, tid, r, 0);
            Body::template ex_read<E>(sh, tid, r, 1);
        } else {
            for (int plane = 0; plane < 2; ++plane) {
                __syncthreads();
                Body::template ex_write<E>(sh, tid, r, plane);
                __syncthreads();
                Body::template ex_read<E>(sh, tid, r, plane);
            }
        }
    };

    while (t < a.tiles_total) {
        Body::pre_twiddle(a, sh, tid, r);
        stamp();  // 2: tile loaded (+ pre-twiddle)
        Body::step1(sh, tid, r);
        stamp();  // 3
        exchange(std::integral_constant<int, 1>{}, r);
        stamp();  // 4
        Body::step2(sh, tid, r);
        stamp();  // 5
        if constexpr (Body::THREE) {
            exchange(std::integral_constant<int, 2>{}, r);
            stamp();  // 6
            Body::step3(r);
            stamp();  // 7
        }
        if constexpr (TRANSPOSE) {
            exchange(std::integral_constant<int, 3>{}, r);
            stamp();  // 6 or 8
        }
        Body::store(a, tid, r);
        stamp();  // last: stores retired
        t += gridDim.x;
        if (t < a.tiles_total) {  // next tile's loads go out right behind the stores
            Body::locate(a, t, r);
            Body::load_raw(a, tid, r);
        }
    }
}

// host-side launcher for one (T, LR, LC, mode) instantiation
template <typename T, int LR, int LC, bool PRE_TW, bool TRANSPOSE, bool SEQ>
hipError_t launch_tile_inst(unsigned grid, hipStream_t stream, const TileArgs &a, bool query_only, int *blocks_per_cu,
                            size_t *lds_out, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    using Body = TileBody<T, LR, LC, PRE_TW, TRANSPOSE, SEQ>;
    auto kern = tile_fft_kernel<T, LR, LC, PRE_TW, TRANSPOSE, SEQ>;
    const size_t lds = Body::lds_bytes(a.tw_bits);
    if (lds_out) *lds_out = lds;
    // raise the dynamic-LDS limit only when it grows: the steady state issues no runtime call besides the
    // launch itself, so a launch sequence can be captured into a HIP graph
    static size_t lds_limit = 0;
    if (lds > lds_limit) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_limit = lds;
    }
    if (query_only) {
        // residency from the kernel's real register count and LDS request (MI355X_MICROARCH.md: 512 VGPRs per
        // lane per SIMD in granules of 8, 160 KiB LDS per CU, 32 waves per CU); the occupancy API is only a
        // cross-check (it under-reports for large dynamic-LDS requests)
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
        if (e != hipSuccess) return e;
        const int alloc = ((fa.numRegs + 7) / 8) * 8;
        int waves_per_simd = alloc > 0 ? 512 / alloc : 8;
        if (waves_per_simd > 8) waves_per_simd = 8;
        const int waves_per_wg = Body::NT / 64;
        int by_regs = waves_per_simd * 4 / waves_per_wg;
        int by_lds = (int)((160 * 1024) / lds);
        int by_waves = 32 / waves_per_wg;
        int b = by_regs < by_lds ? by_regs : by_lds;
        if (by_waves < b) b = by_waves;
        *blocks_per_cu = b < 1 ? 1 : b;
        return hipSuccess;
    }
    if (ev_start && ev_stop)  // events bound to the dispatch itself: their interval is the kernel's execution time
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), (uint32_t)lds, stream, ev_start, ev_stop, 0, a);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Body::NT), lds, stream, a);
    return hipGetLastError();
}

// Thread-by-thread host execution of one pass: the same TileBody phases, barriers replaced by
// "every thread finishes the phase".  Test infrastructure for the GPU-less build container.
template <typename T, int LR, int LC, bool PRE_TW, bool TRANSPOSE, bool SEQ> void emulate_tile_pass(const TileArgs &a) {
    using Body = TileBody<T, LR, LC, PRE_TW, TRANSPOSE, SEQ>;
    using Regs = typename Body::Regs;
    constexpr int NT = Body::NT;
    T *ex = new T[(size_t)Body::EXCH * 2];
    const typename Body::Shared sh{ex, Body::PLANE_SEQ ? ex : ex + Body::EXCH,
                                   reinterpret_cast<const cx_t<T> *>(a.tw3), reinterpret_cast<const cx_t<T> *>(a.twr)};
    Regs *regs = new Regs[NT];
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
    for (unsigned tile = 0; tile < a.tiles_total; ++tile) {
        for (int t = 0; t < NT; ++t) {
            Body::locate(a, tile, regs[t]);
            Body::load_raw(a, t, regs[t]);
            Body::pre_twiddle(a, sh, t, regs[t]);
            Body::step1(sh, t, regs[t]);
        }
        exchange(std::integral_constant<int, 1>{});
        for (int t = 0; t < NT; ++t) Body::step2(sh, t, regs[t]);
        if constexpr (Body::THREE) {
            exchange(std::integral_constant<int, 2>{});
            for (int t = 0; t < NT; ++t) Body::step3(regs[t]);
        }
        if constexpr (TRANSPOSE) exchange(std::integral_constant<int, 3>{});
        for (int t = 0; t < NT; ++t) Body::store(a, t, regs[t]);
    }
    delete[] regs;
    delete[] ex;
}

}  // namespace phast
