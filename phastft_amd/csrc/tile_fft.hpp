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
// Each thread holds P = 8, 16 or 32 complex points in registers.  Inside the tile the ROWS-point FFT is a
// decimation-in-frequency digit chain ROWS = R_1 * ... * R_S (R_i <= P); the radix-32/16/8/4/2 butterflies run
// in registers with literal twiddles, and data moves between the radix steps through LDS (layouts and the
// bank-conflict argument are with TileBody below; tests/test_emulator.py audits every shape).
// Pass A (TRANSPOSE) additionally transposes through LDS ([col][k], padded) so that each column's ROWS
// outputs leave as one contiguous run: this is where the digit reversal happens.  Later passes (PRE_TW)
// multiply by the inter-pass twiddle W_{ROWS*S}^{row*lo} on load, looked up as a product of three small LDS
// tables.  No MFMA: 6 FMA-class ops per 64 B moved per radix-2 stage -- HBM-bound.
//
// The per-thread work is written as __host__ __device__ phase functions (TileBody) so that the very
// same index arithmetic is executed thread-by-thread on the CPU by tests/emu/emu.hip (tests/test_emulator.py)
// -- the build container has no GPU.
#pragma once

#include <hip/hip_ext.h>

#include "common.hpp"

#ifndef PHAST_TW_PROG_MIN_LR  // shapes whose pre-twiddle is a progression instead of P look-ups (tuning: tools/cmp_throughput.py)
#define PHAST_TW_PROG_MIN_LR 9
#endif
#ifndef PHAST_TW_PROG_MIN_LP
#define PHAST_TW_PROG_MIN_LP 5
#endif

#ifndef PHAST_RUNNING_ROW_PTR  // measured, not adopted: see TileBody::load_raw
#define PHAST_RUNNING_ROW_PTR 0
#endif
#ifndef PHAST_FRESH_TID
#define PHAST_FRESH_TID 1
#endif
#ifndef PHAST_TW_PROG_G  // running values of the progression form of the pre-twiddle (common.hpp: tw_progression)
#define PHAST_TW_PROG_G(P) ((P) >= 32 ? 8 : 4)
#endif

namespace phast {

// ---- literal twiddles: (re, im) *= W_N^J = exp(-2*pi*i*J/N), N in {2,4,8,16,32}, 0 <= J < N/2 ----
// cos(a*pi/16), a = 0..8, correctly rounded
template <typename T> PHAST_HD constexpr T cos_pi16(int a) {
    constexpr long double c[9] = {1.0L,
                                  0.98078528040323044912618223613423903697L,
                                  0.92387953251128675612818318939678828682L,
                                  0.83146961230254523707878837761790575673L,
                                  0.70710678118654752440084436210484903928L,
                                  0.55557023301960222474283081394853287438L,
                                  0.38268343236508977172845998403039886676L,
                                  0.19509032201612826784828486847702224093L,
                                  0.0L};
    return a <= 8 ? (T)c[a] : (T)-c[16 - a];
}
template <typename T, int N, int J> PHAST_HD void mul_w(T &re, T &im) {  // T: a scalar or a packed pair (common.hpp: f32x2)
    using Sc = scalar_t<T>;
    constexpr Sc S = (Sc)0.70710678118654752440L;
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
        static_assert(N == 16 || N == 32, "general twiddle only for N = 16, 32");
        constexpr int A = J * (32 / N);                 // angle in units of pi/16, 0 < A < 16
        constexpr Sc c = cos_pi16<Sc>(A);
        constexpr Sc s = cos_pi16<Sc>(A <= 8 ? 8 - A : A - 8);  // sin(A pi/16) = cos((8 - A) pi/16) > 0
        T r = re * c + im * s;
        T i = im * c - re * s;
        re = r;
        im = i;
    }
}

// In-register radix-R decimation-in-frequency FFT on regs [OFF, OFF+R) of a P-register thread.
// Afterwards position OFF+p holds X[bitrev(p)].
template <typename T, int R, int OFF, int P> PHAST_HD void fft_reg_dif(T (&re)[P], T (&im)[P]) {
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

// One tile pass, generic in the points per thread P = 2^LP (32: the wide throughput tiles -- twice the row width
// at the same thread count, one exchange fewer; 16: throughput plans; 8: latency plans, twice the waves per tile --
// a one-tile-per-CU launch is issue-stall bound with one wave per SIMD; 2..32 = N in the small-transform kernel).
//
// The ROWS-point FFT is a DIF digit chain ROWS = R_1 * R_2 * ... * R_S with R_i <= P (R_1 = ... = P, the last
// takes the remainder).  K_i = R_1...R_i.  Step i consumes input digit n_i (most significant first) and
// produces output digit k_i (least significant first):
//   * butterfly id b in [0, ROWS/R_i): b = k_low + K_{i-1} * n_rest  (k_low: digits already produced,
//     n_rest: input digits still to go);  thread tau owns b = tau + M*I, I < P/R_i  (M = ROWS/P threads per column)
//   * register I*R_i + j holds digit value n_i = j before the butterfly and k_i = bitrev(j) after it
//   * then the twiddle W_{ROWS/K_{i-1}}^{n_rest * k_i} = W_ROWS^{n_rest * k_i * K_{i-1}}
//   * exchange i (LDS) stores row n_rest*K_i + k_i*K_{i-1} + k_low; step i+1 reads row (j*L + n_rest')*K_i + k_low'
//     with L = ROWS/(K_i R_{i+1}).  Lanes are (column fastest, tau): from exchange 2 on, a 32-lane access group
//     sees consecutive rows by construction; exchange 1 (whose writers stride by K_1 rows) gets one pad row per
//     n' = n_rest when the group spans G > 1 rows.
// After the last step register I*R_S + j of thread tau holds frequency row bitrev(j)*K_{S-1} + tau + M*I.
template <typename T, int LR, int LC, int LP, bool PRE_TW, bool TRANSPOSE, bool SEQ, bool ALLOW_DIRECT = true> struct TileBody {
    using cx = cx_t<T>;
    static constexpr int ROWS = 1 << LR;
    static constexpr int COLS = 1 << LC;
    static constexpr int P = 1 << LP;             // complex points per thread
    static constexpr int NT = ROWS * COLS / P;    // threads per workgroup
    static constexpr int M = ROWS / P;            // threads per column
    static constexpr int LM = LR - LP;
    static constexpr int G = COLS >= 32 ? 1 : 32 / COLS;  // rows seen by one 32-lane LDS access group
    static constexpr int S = (LR + LP - 1) / LP;  // number of radix steps
    static constexpr bool PLANE_SEQ = SEQ;  // exchange re and im one after the other (half the LDS, twice the barriers)
    // Column stride of the transposing exchange [col][k]: its WRITERS are lanes (col fastest, then tau = k), so
    // consecutive columns must land as many cells apart as a write group holds rows -- 32-lane groups over 32 cells
    // for 4-byte elements (G), but ds_write_b64 serves 16-lane groups over 16 eight-byte cells (MI355X_MICROARCH.md,
    // LDS table): 16/COLS rows per group.  The readers walk k contiguously and do not care.
    static constexpr int GW = sizeof(T) == 8 ? (COLS >= 16 ? 1 : 16 / COLS) : G;
    static constexpr int CS = ROWS + GW;
    static constexpr int E1S = P + (G > 1 ? 1 : 0);  // rows per n' in exchange 1 (P used + padding)
    static constexpr int TWR = ROWS > 1024 ? 32 + ROWS / 32 : 64;  // staged entries of the W_ROWS table (plan.hpp: host_twr)
    // DIRECT (first passes with at least one exchange and runs of >= 128 bytes per M threads): no transposing exchange.
    // The readers of the LAST exchange take the thread order (tau fastest, then column) instead of (column fastest, tau),
    // so that after the last radix step the lanes of a wave hold M consecutive output rows of a column: the stores go
    // out as runs of M elements straight from the registers, and one whole LDS round trip (two or four barriers) of the
    // pass is gone.  The last exchange then has a column pitch of COLS + 32/M cells, which keeps the row-fastest
    // readers (and the column-fastest writers) free of bank conflicts (tests/test_emulator.py audits every shape).
#ifndef PHAST_DIRECT_RUNS
#define PHAST_DIRECT_RUNS 1
#endif
    static constexpr bool DIRECT = PHAST_DIRECT_RUNS && ALLOW_DIRECT && TRANSPOSE && S >= 2 && M * (int)sizeof(T) >= 128 && M <= 32 &&
                                   COLS * (int)sizeof(T) >= 128;  // (narrower tiles: a write group spans rows, pitch rules differ)
    static constexpr int PITCH_L = DIRECT ? COLS + 32 / M : COLS;  // column pitch of exchange S - 1
    template <int E> static constexpr int pitch() { return (DIRECT && E == S - 1) ? PITCH_L : COLS; }
    static constexpr int EXCH_E1 = M * E1S * pitch<1>();
    static constexpr int EXCH_EL = S > 2 ? ROWS * pitch<S - 1>() : 0;
    static constexpr int EXCH_ET = (TRANSPOSE && !DIRECT) ? COLS * CS : 0;
    static constexpr int EXCH = EXCH_E1 > EXCH_ET ? (EXCH_E1 > EXCH_EL ? EXCH_E1 : EXCH_EL) : (EXCH_ET > EXCH_EL ? EXCH_ET : EXCH_EL);
    // Non-temporal global accesses when a tile row is a whole 128-byte line or more: every byte is touched once
    // per pass, and the strided-copy microbenchmark gains 5-10 % (profiles/r01_strided_copy_nt.log).  Narrower
    // rows share their line with the neighbouring tile and NEED the L2 (nt loads cost 25 % there).
    static constexpr bool NT_HINT = COLS * sizeof(T) >= 128;
#ifndef PHAST_TILE_NT_LOADS  // tools only: the two halves of the hint separately (profiles/r03_nt_halves.log)
#define PHAST_TILE_NT_LOADS 1
#endif
#ifndef PHAST_TILE_NT_STORES
#define PHAST_TILE_NT_STORES 1
#endif
    static constexpr bool NT_LD = NT_HINT && PHAST_TILE_NT_LOADS, NT_ST = NT_HINT && PHAST_TILE_NT_STORES;
    static_assert(LR >= 1 && LR <= 13, "tile FFT length 2..8192 (the multi-pass plans use 64..1024)");
    static_assert(LP >= 1 && LP <= 5 && LP <= LR, "2..32 points per thread");
    static_assert(NT <= 1024 && NT >= 64, "64..1024 threads per workgroup");
    static_assert(S >= 1, "at least one radix step");

    // log2 of radix R_i (1-based) and of K_i = R_1...R_i
    static constexpr int rbits(int i) { return i < S ? LP : LR - LP * (S - 1); }
    static constexpr int kbits(int i) { return i <= 0 ? 0 : (i < S ? LP * i : LR); }

    // the pre-twiddle as two look-ups + a progression (see pre_twiddle): six table entries per thread and tile
    // (round 4: in f32 every shape takes the progression -- profiles/r04_prog_f32_ab.log: one transform of 2^22 / 2^23 points
    // +3.7 %, 2^20 +1.5 %, 8 x 2^20 +2 %, 2^24 unchanged; the same switch in f64 is neutral to -3 %, so f64 keeps the shape
    // condition)
    static constexpr bool PROG = PRE_TW && (sizeof(T) == 4 || (LR >= PHAST_TW_PROG_MIN_LR && LP >= PHAST_TW_PROG_MIN_LP));
    // ... which may as well come straight from global memory when the three-level tables (48 KiB from N = 2^28 on)
    // no longer fit the LDS next to the tile: the 16384-point tiles stay available for the largest transforms
    // (2^28 f64: 1024 x 8 tiles with 64-byte rows were the fallback, 20 % slower)
    PHAST_HD static bool tw3_global(unsigned tw_bits) {
        const size_t exch = (size_t)EXCH * sizeof(T) * (PLANE_SEQ ? 1 : 2);
        return PROG && exch + (size_t)(3u << tw_bits) * sizeof(cx) + TWR * sizeof(cx) > (size_t)160 * 1024;
    }
    static size_t lds_bytes(unsigned tw_bits) {
        size_t exch = (size_t)EXCH * sizeof(T) * (PLANE_SEQ ? 1 : 2);
        size_t tw3 = (PRE_TW && !tw3_global(tw_bits)) ? (size_t)(3u << tw_bits) * sizeof(cx) : 0;
        return exch + tw3 + TWR * sizeof(cx);
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
        T re[P], im[P];
        unsigned xform, g0;
    };

    PHAST_HD static int col_of(int tid) { return tid & (COLS - 1); }
    PHAST_HD static int tau_of(int tid) { return tid >> LC; }

    // tile index -> (transform, first column).  XCD-aware order: workgroup b runs on XCD b%8 (observed);
    // each XCD gets one contiguous run of tiles so that neighbouring column groups -- which share DRAM
    // pages and L2 lines -- meet in one L2.
    PHAST_HD static void locate(const TileArgs &a, unsigned t, Regs &r) {
        // (starting every XCD at a different column-block phase of its run changes nothing -- the persistent workgroups
        // drift apart within a few tiles: profiles/r03_scratch_pad_ab.log, last section)
        const unsigned tile = ((a.tiles_total & 7u) == 0u) ? (t & 7u) * (a.tiles_total >> 3) + (t >> 3) : t;
        r.xform = tile >> (unsigned)__builtin_ctz(a.tiles_per_xform);  // a power of two (plan.hpp: geom_to_args)
        const unsigned ti = tile & (a.tiles_per_xform - 1u);
        r.g0 = a.cs_bits ? (((ti >> a.cb_bits) << a.cs_bits) | ((ti & ((1u << a.cb_bits) - 1u)) << LC)) : (ti << LC);
    }

    // ---------------- load rows n = j*M + tau, j = 0..P-1 ----------------
    // Addresses are (wave-uniform 64-bit base) + (32-bit per-lane element offset): the tile's columns share
    // the high part of in_col (tiles are COLS-aligned and COLS <= 2^log_s_in), the row j*M is uniform, and
    // only tau*2^log_s_in + col differs between lanes -- one VGPR for all P loads (saddr addressing).
    PHAST_HD static void load_raw(const TileArgs &a, int tid, Regs &r) {
        const int col = col_of(tid), tau = tau_of(tid);
        const size_t ubase = in_tile_base(a, r.xform, r.g0);
        const unsigned voff = (unsigned)tau * (unsigned)a.in_row_stride + (unsigned)col;
        // only a first pass (never PRE_TW) can see interleaved input, only a last pass (never TRANSPOSE) writes
        // interleaved output: the other kernels do not carry those paths (they cost registers: 4-dword loads
        // plus selects double the live state of a 32-point thread)
        if (PRE_TW || !a.in_interleaved) {
            const T *pr = reinterpret_cast<const T *>(a.in_re) + ubase;
            const T *pi = reinterpret_cast<const T *>(a.in_im) + ubase;
            // With P independent 64-bit row bases per plane the compiler materialises all of them up front -- 128 SGPRs for
            // 32 rows x 2 planes, 50-79 of them parked in VGPR lanes (v_writelane / v_readlane) in the 32-point kernels.
            // PHAST_RUNNING_ROW_PTR=1 makes them running pointers (two scalar adds per row, no SGPR spills) -- and the
            // loads then issue one address computation apart instead of back to back: the 256 x 64 pass of 2^24 x 4 ran
            // 4 % SLOWER, nothing else moved (profiles/r03_ablation_tid_rowptr.log).  Not adopted.
            const size_t ustep = (size_t)M * a.in_row_stride;
            (void)ustep;
            static_for<0, P>([&](auto j) {
#if PHAST_RUNNING_ROW_PTR
                const size_t urow = 0;
#else
                const size_t urow = (size_t)(decltype(j)::value * M) * a.in_row_stride;
#endif
                if constexpr (NT_LD) {
                    r.re[j] = __builtin_nontemporal_load(pr + urow + voff);
                    r.im[j] = __builtin_nontemporal_load(pi + urow + voff);
                } else {
                    r.re[j] = (pr + urow)[voff];
                    r.im[j] = (pi + urow)[voff];
                }
#if PHAST_RUNNING_ROW_PTR
                pr += ustep;
                pi += ustep;
#endif
            });
        } else {
            const cx *pz = reinterpret_cast<const cx *>(a.in_re) + ubase;
            static_for<0, P>([&](auto j) {
                const size_t urow = (size_t)(decltype(j)::value * M) * a.in_row_stride;
                cx v = (pz + urow)[voff];
                r.re[j] = a.in_interleaved == 2 ? v.y : v.x;
                r.im[j] = a.in_interleaved == 2 ? v.x : v.y;
            });
        }
    }
    // exponent of row j*M + tau of this thread's column: e0 + j*de (mod 2^32, a multiple of every table modulus)
    PHAST_HD static void pre_twiddle_exps(const TileArgs &a, int tid, const Regs &r, unsigned &e0, unsigned &de) {
        const int col = col_of(tid), tau = tau_of(tid);
        if (a.grid_mode) {  // input twiddle of a four-step split: (row << shift | glo) * (col0 + c)
            const unsigned g = r.g0 + (unsigned)col, k = a.grid_col0 + (g & a.grid_col_mask), glo = g >> a.tw_shift;
            const unsigned big = k << a.grid_row_shift;
            e0 = (unsigned)tau * big + glo * k;
            de = (unsigned)M * big;
        } else {
            const unsigned lo = ((r.g0 + (unsigned)col) >> a.tw_shift) & a.tw_mask;
            e0 = (unsigned)tau * lo;
            de = (unsigned)M * lo;
        }
    }
    // PROG: the progression W^e0 (W^de)^j from the six raw table entries (fetched by the caller from LDS or global memory)
    PHAST_HD static void pre_twiddle_progress(const Tw3Raw<T> &tb, const Tw3Raw<T> &td, Regs &r) {
        T br, bi, dr, di;
        tw3_combine<T>(tb, br, bi);
        tw3_combine<T>(td, dr, di);
        tw_progression<T, P, PHAST_TW_PROG_G(P)>(br, bi, dr, di, [&](auto j, T wr, T wi) {
            cmul(r.re[decltype(j)::value], r.im[decltype(j)::value], wr, wi);
        });
    }
    // inter-pass twiddle W_{ROWS*S_in}^{row*lo} on the freshly loaded rows (needs the LDS tables)
    PHAST_HD static void pre_twiddle(const TileArgs &a, const Shared &sh, int tid, Regs &r) {
        if constexpr (PRE_TW) {
            unsigned e0, de;
            pre_twiddle_exps(a, tid, r, e0, de);
            if constexpr (PROG) {
                // W^(e0 + j de) = W^e0 (W^de)^j: two look-ups and a geometric progression (tw_progression) instead of P look-ups
                // -- fewer complex products, 6 LDS reads per thread instead of 3 P, none of them conflicting
                // (the data-dependent table reads were 29 % of the LDS cycles of these passes, profiles/r01_sq_batch_lds.txt).
                // Measured (profiles/r02_tw_ladder.log): the 1024 x 16 pass of the batched 2^20 transforms 2.17 -> 2.07 ms
                // per 256 transforms; 512-row tiles gain 1-3 % (2^26: 49.3 -> 50.3 GSamples/s, 2^28: 39.1 -> 40.2); the 256 x 64
                // passes of 2^24 LOSE 3 % with it, hence the shape condition (rows >= 512, 32 points per thread).  The progression
                // adds <= 9 roundings: 1e-15 in f64 against the 1e-13 budget; in f32 the 1024 x 32 pass gains 9 % (2.12 -> 1.90 ms
                // per 512 transforms: this shape spills registers under 32 look-ups) and the batch's rel-L2 error stays at
                // 6e-7 against 1e-5.
                pre_twiddle_progress(tw3_fetch<T>(sh.tw3, a.tw_bits, e0), tw3_fetch<T>(sh.tw3, a.tw_bits, de), r);
            } else {
                static_for<0, P>([&](auto j) {
                    T wr, wi;
                    tw3_lookup<T>(sh.tw3, a.tw_bits, e0 + decltype(j)::value * de, wr, wi);
                    cmul(r.re[j], r.im[j], wr, wi);
                });
            }
        }
    }

    PHAST_HD static void twr_lookup(const Shared &sh, unsigned e, T &wr, T &wi) {  // W_ROWS^e, e < ROWS
        cx w0 = sh.twr[e & 31u], w1 = sh.twr[32u + (e >> 5)];
        wr = w0.x * w1.x - w0.y * w1.y;
        wi = w0.x * w1.y + w0.y * w1.x;
    }

    // ---------------- step I (1-based): P/R_I radix-R_I butterflies + the inter-digit twiddle ----------------
    template <int I> PHAST_HD static void step(const Shared &sh, int tid, Regs &r) {
        constexpr int RB = rbits(I), R = 1 << RB, NB = P / R, KB = kbits(I - 1);
        static_for<0, NB>([&](auto i) { fft_reg_dif<T, R, decltype(i)::value * R, P>(r.re, r.im); });
        if constexpr (I < S) {  // n_rest = (tau + M*i) >> KB; twiddle W_ROWS^(n_rest * k_I * K_{I-1})
            const int tau = tau_of(tid);
            static_for<0, NB>([&](auto i) {
                const unsigned n_rest = (unsigned)(tau + M * decltype(i)::value) >> KB;
                static_for<1, R>([&](auto p) {
                    constexpr int KI = bitrev_c(decltype(p)::value, RB);
                    T wr, wi;
                    twr_lookup(sh, (n_rest * KI) << KB, wr, wi);
                    cmul(r.re[decltype(i)::value * R + decltype(p)::value], r.im[decltype(i)::value * R + decltype(p)::value],
                         wr, wi);
                });
            });
        }
    }

    // frequency index (row of the tile FFT output) held by register Q after the last step
    //   = krow_lane(tid) + krow_const<Q>()
    // thread order of the last radix step: (column fastest, tau), or (tau fastest, column) for DIRECT passes
    PHAST_HD static int tau_last(int tid) { return DIRECT ? (tid & (M - 1)) : tau_of(tid); }
    PHAST_HD static int col_last(int tid) { return DIRECT ? (tid >> LM) : col_of(tid); }
    PHAST_HD static unsigned krow_lane(int tid) { return (unsigned)tau_last(tid); }
    template <int Q> PHAST_HD static constexpr unsigned krow_const() {
        constexpr int RB = rbits(S), R = 1 << RB;
        return (unsigned)(bitrev_c(Q % R, RB) << kbits(S - 1)) + (unsigned)(M * (Q / R));
    }

    // ---------------- LDS exchange addresses (in elements); E = 1..S-1, E = S is the transposing exchange ----
    // physical row of logical (n_rest, kk) with kk < K_E: exchange 1 pads one row per n_rest
    template <int E> PHAST_HD static int phys_row(int n_rest, int kk) {
        if constexpr (E == 1) return n_rest * E1S + kk;
        else return (n_rest << kbits(E)) + kk;
    }
    template <int E, int Q> PHAST_HD static int waddr(int tid) {
        const int col = col_of(tid), tau = tau_of(tid);
        if constexpr (E < S) {  // written by step E: register Q = I*R + j, k_E = bitrev(j)
            constexpr int RB = rbits(E), R = 1 << RB, KB = kbits(E - 1);
            constexpr int I = Q / R, KE = bitrev_c(Q % R, RB);
            const int b = tau + M * I, k_low = b & ((1 << KB) - 1), n_rest = b >> KB;
            return phys_row<E>(n_rest, (KE << KB) + k_low) * pitch<E>() + col;
        } else {  // [col][k]
            return col * CS + (int)krow_lane(tid) + (int)krow_const<Q>();
        }
    }
    template <int E, int Q> PHAST_HD static int raddr(int tid) {
        // the readers of the last exchange are the threads of the last step: their order may differ (DIRECT)
        const int col = (E == S - 1) ? col_last(tid) : col_of(tid), tau = (E == S - 1) ? tau_last(tid) : tau_of(tid);
        if constexpr (E < S) {  // read by step E+1: register Q = I*R + j holds input digit n_{E+1} = j
            constexpr int RB = rbits(E + 1), R = 1 << RB, KB = kbits(E);
            constexpr int I = Q / R, J = Q % R;
            constexpr int L = ROWS >> (KB + RB);  // values of the digits after n_{E+1}
            const int b = tau + M * I, k_low = b & ((1 << KB) - 1), n_rest = b >> KB;
            return phys_row<E>(J * L + n_rest, k_low) * pitch<E>() + col;
        } else {
            const int f = Q * NT + tid;
            return (f >> LR) * CS + (f & (ROWS - 1));
        }
    }
    // plane 0 = real parts through ex_re, plane 1 = imaginary parts through ex_im
    template <int E> PHAST_HD static void ex_write(const Shared &sh, int tid, const Regs &r, int plane) {
        T *dst = plane ? sh.ex_im : sh.ex_re;
        static_for<0, P>([&](auto Q) { dst[waddr<E, decltype(Q)::value>(tid)] = plane ? r.im[Q] : r.re[Q]; });
    }
    template <int E> PHAST_HD static void ex_read(const Shared &sh, int tid, Regs &r, int plane) {
        const T *src = plane ? sh.ex_im : sh.ex_re;
        static_for<0, P>([&](auto Q) {
            const T v = src[raddr<E, decltype(Q)::value>(tid)];
            if (plane)
                r.im[Q] = v;
            else
                r.re[Q] = v;
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
        if (TRANSPOSE || !a.out_interleaved) {
            if constexpr (NT_ST) {
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
        if constexpr (!TRANSPOSE || DIRECT) {  // register Q holds row krow_lane + krow_const<Q> of column g0 + col
            const unsigned voff = (unsigned)col_last(tid) * (unsigned)a.out_s1 + krow_lane(tid) * (unsigned)a.out_row_stride;
            static_for<0, P>([&](auto Q) {
                put(a, base + (size_t)krow_const<decltype(Q)::value>() * a.out_row_stride, voff, r.re[Q], r.im[Q]);
            });
        } else {  // after the transposing exchange: register Q holds flat element f = Q*NT + tid of the [col][k] tile
            if constexpr (NT >= ROWS) {  // f -> column (Q*NT >> LR) + (tid >> LR), row tid & (ROWS-1)
                const unsigned voff = (unsigned)(tid >> LR) * (unsigned)a.out_s1 +
                                      (unsigned)(tid & (ROWS - 1)) * (unsigned)a.out_row_stride;
                static_for<0, P>([&](auto Q) {
                    constexpr int C0 = (decltype(Q)::value * NT) >> LR;
                    put(a, base + (size_t)C0 * a.out_s1, voff, r.re[Q], r.im[Q]);
                });
            } else {  // f -> column Q*NT >> LR, row (Q*NT & (ROWS-1)) + tid
                const unsigned voff = (unsigned)tid * (unsigned)a.out_row_stride;
                static_for<0, P>([&](auto Q) {
                    constexpr int C0 = (decltype(Q)::value * NT) >> LR, K0 = (decltype(Q)::value * NT) & (ROWS - 1);
                    put(a, base + (size_t)C0 * a.out_s1 + (size_t)K0 * a.out_row_stride, voff, r.re[Q], r.im[Q]);
                });
            }
        }
    }

    // the whole per-tile chain between the load and the store, parameterised by how an exchange is performed
    // (barriers on the GPU, "every thread finishes the phase" in the emulator)
    template <typename StepFn, typename ExchFn> PHAST_HD static void chain(StepFn &&do_step, ExchFn &&do_exchange) {
        static_for<1, S + 1>([&](auto i) {
            do_step(i);
            if constexpr (decltype(i)::value < S) do_exchange(i);
        });
        if constexpr (TRANSPOSE && !DIRECT) do_exchange(std::integral_constant<int, S>{});
    }
};

// LDS exchange flavour per instantiation: f64 tiles with 16 points per thread and every 32-point tile exchange
// the re and im planes one after the other (half the LDS); the rest exchange both planes at once.
template <typename T, int LP> inline constexpr bool plane_seq_v = (sizeof(T) == 8 && LP == 4) || LP == 5;

#ifndef PHAST_MIN_WAVES
#define PHAST_MIN_WAVES(LR, LC) 1
#endif
template <typename T, int LR, int LC, int LP, bool PRE_TW, bool TRANSPOSE, bool SEQ>
__global__ void __launch_bounds__(1 << (LR + LC - LP), PHAST_MIN_WAVES(LR, LC)) tile_fft_kernel(const TileArgs a) {
    using Body = TileBody<T, LR, LC, LP, PRE_TW, TRANSPOSE, SEQ>;
    using cx = cx_t<T>;
    constexpr int NT = Body::NT;
    pin_tile_args(a);  // all scalar loads of the arguments at once (common.hpp)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *ex_re = reinterpret_cast<T *>(smem);
    cx *l_tw3 = reinterpret_cast<cx *>(smem + (size_t)Body::EXCH * sizeof(T) * (Body::PLANE_SEQ ? 1 : 2));
    const bool tw_global = Body::tw3_global(a.tw_bits);  // uniform: inter-pass tables read from global memory, not staged
    cx *l_twr = l_tw3 + ((PRE_TW && !tw_global) ? (3u << a.tw_bits) : 0u);
    const typename Body::Shared sh{ex_re, Body::PLANE_SEQ ? ex_re : ex_re + Body::EXCH, l_tw3, l_twr};

    int tid = threadIdx.x;
    // The thread id is not kept in a VGPR across the phases of a tile: the wave's first thread id sits in an SGPR and
    // the lane number is two mbcnt instructions away.  fresh_tid() launders the SGPR, so nothing derived from the
    // thread id (LDS exchange and twiddle-table addresses, ~100 values) can be hoisted out of the tile loop or kept
    // live -- or spilled -- across a phase; each phase recomputes what it needs with a few integer ops.  (Round 2
    // laundered a VGPR copy of the id once per tile: that VGPR and `col` were what the 32-point f32 kernel spilled.)
    unsigned wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    bool first_fresh = true;
    auto fresh_tid = [&]() {
#if PHAST_FRESH_TID
        asm volatile("" : "+s"(wave_base));
        return (int)(wave_base | __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
#else  // tools only: round 2's form, one laundered VGPR copy per tile
        (void)wave_base;
        if (first_fresh) asm volatile("" : "+v"(tid));
        return tid;
#endif
    };
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
#ifdef PHAST_TILE_STAGGER  // tools only: the workgroups' first loads PHAST_TILE_STAGGER x 64 cycles apart, 8 groups
    for (unsigned k = ((blockIdx.x >> 3) & 7u) * PHAST_TILE_STAGGER; k > 0; --k) __builtin_amdgcn_s_sleep(1);
#endif
    if (t < a.tiles_total) {
        Body::locate(a, t, r);
        Body::load_raw(a, tid, r);
    }
    for (int i = tid; i < Body::TWR; i += NT) l_twr[i] = reinterpret_cast<const cx *>(a.twr)[i];
    if constexpr (PRE_TW)
        if (!tw_global)
            for (unsigned i = tid; i < (3u << a.tw_bits); i += NT) l_tw3[i] = reinterpret_cast<const cx *>(a.tw3)[i];
    __syncthreads();
    stamp();  // 1: twiddle tables in LDS

    // one exchange: barrier (previous readers done), write, barrier, read -- per plane when PLANE_SEQ
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
            // `plane` must be a literal at every call: a run-time plane makes &r.re / &r.im a select and sends
            // the register arrays to scratch memory as soon as the compiler stops unrolling this loop (it does
            // at 32 points per thread)
            __syncthreads();
            Body::template ex_write<E>(sh, tid, r, 0);
            __syncthreads();
            Body::template ex_read<E>(sh, tid, r, 0);
            __syncthreads();
            Body::template ex_write<E>(sh, tid, r, 1);
            __syncthreads();
            Body::template ex_read<E>(sh, tid, r, 1);
        }
        stamp();
    };
    auto do_step = [&](auto i) {
        tid = fresh_tid();
        Body::template step<decltype(i)::value>(sh, tid, r);
        stamp();
    };

    while (t < a.tiles_total) {
        first_fresh = true;
        tid = fresh_tid();  // (see fresh_tid above: nothing derived from the thread id survives a phase)
        first_fresh = false;
        if constexpr (Body::PROG) {  // only the six table reads differ (ds_read / global_load); the arithmetic is shared
            unsigned e0, de;
            Body::pre_twiddle_exps(a, tid, r, e0, de);
            Tw3Raw<T> tb, td;
            if (tw_global) {
                tb = tw3_fetch<T>(reinterpret_cast<const cx *>(a.tw3), a.tw_bits, e0);
                td = tw3_fetch<T>(reinterpret_cast<const cx *>(a.tw3), a.tw_bits, de);
            } else {
                tb = tw3_fetch<T>(l_tw3, a.tw_bits, e0);
                td = tw3_fetch<T>(l_tw3, a.tw_bits, de);
            }
            Body::pre_twiddle_progress(tb, td, r);
        } else {
            Body::pre_twiddle(a, sh, tid, r);
        }
        stamp();  // 2: tile loaded (+ pre-twiddle)
        Body::chain(do_step, exchange);
        tid = fresh_tid();
        Body::store(a, tid, r);
        stamp();  // last: stores retired
        t += gridDim.x;
        if (t < a.tiles_total) {  // next tile's loads go out right behind the stores
            tid = fresh_tid();
            Body::locate(a, t, r);
            Body::load_raw(a, tid, r);
        }
    }
}

// host-side launcher for one (T, LR, LC, mode) instantiation
template <typename T, int LR, int LC, int LP, bool PRE_TW, bool TRANSPOSE, bool SEQ>
hipError_t launch_tile_inst(unsigned grid, hipStream_t stream, const TileArgs &a, bool query_only, int *blocks_per_cu,
                            size_t *lds_out, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    using Body = TileBody<T, LR, LC, LP, PRE_TW, TRANSPOSE, SEQ>;
    auto kern = tile_fft_kernel<T, LR, LC, LP, PRE_TW, TRANSPOSE, SEQ>;
    const size_t lds = Body::lds_bytes(a.tw_bits);
    if (lds_out) *lds_out = lds;
    // raise the dynamic-LDS limit only when it grows: the steady state issues no runtime call besides the
    // launch itself, so a launch sequence can be captured into a HIP graph
    if (lds > (size_t)160 * 1024) {  // does not fit one CU: the planner rejects the plan, nothing is asked of HIP
        if (query_only && blocks_per_cu) *blocks_per_cu = 0;
        return query_only ? hipSuccess : hipErrorInvalidValue;
    }
    static PerDeviceLimit lds_limit;
    if (hipError_t e = raise_lds_limit(lds_limit, reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
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
template <typename T, int LR, int LC, int LP, bool PRE_TW, bool TRANSPOSE, bool SEQ> void emulate_tile_pass(const TileArgs &a) {
    using Body = TileBody<T, LR, LC, LP, PRE_TW, TRANSPOSE, SEQ>;
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
    auto do_step = [&](auto i) {
        for (int t = 0; t < NT; ++t) Body::template step<decltype(i)::value>(sh, t, regs[t]);
    };
    for (unsigned tile = 0; tile < a.tiles_total; ++tile) {
        for (int t = 0; t < NT; ++t) {
            Body::locate(a, tile, regs[t]);
            Body::load_raw(a, t, regs[t]);
            Body::pre_twiddle(a, sh, t, regs[t]);
        }
        Body::chain(do_step, exchange);
        for (int t = 0; t < NT; ++t) Body::store(a, t, regs[t]);
    }
    delete[] regs;
    delete[] ex;
}

}  // namespace phast
