// plan.hpp -- host-only planning shared by the library (planner.hpp) and the CPU emulator (tests/emu/emu.hip):
// twiddle tables in long double, the factorisation of N = 2^L into tile passes, and the address
// geometry of every pass.  GPU counterpart of PlannerDit*::with_mode (planner.rs:65-100) -- but the
// tables here are O(N^(1/3)) small (three-level factored twiddles) instead of the reference's 2(N-64)
// scalars, because a streamed N-entry table would add ~50 % HBM traffic to a bandwidth-bound pass.
#pragma once

#include <algorithm>
#include <cmath>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace phast {

// ---- W_m^e = exp(-2*pi*i*e/m) in long double with exact octant symmetry ----
inline void twiddle_ld(unsigned long long e, unsigned long long m, long double &wr, long double &wi) {
    e %= m;
    const unsigned long long k = (8 * e) / m;   // octant
    const unsigned long long r = 8 * e - k * m;  // in [0, m): position inside the octant, units of (pi/4)/m
    const long double quarter_pi = 0.785398163397448309615660845819875721L;
    long double a, b;  // cos / sin of the reduced angle phi in [0, pi/4]
    const unsigned long long num = (k & 1) ? (m - r) : r;
    if (num == 0) {
        a = 1.0L;
        b = 0.0L;
    } else if (num == m) {
        a = b = 0.707106781186547524400844362104849039L;
    } else {
        const long double phi = quarter_pi * ((long double)num / (long double)m);
        a = cosl(phi);
        b = sinl(phi);
    }
    long double c, s;  // cos(theta), sin(theta), theta = 2*pi*e/m
    switch (k) {
    case 0: c = a; s = b; break;
    case 1: c = b; s = a; break;
    case 2: c = -b; s = a; break;
    case 3: c = -a; s = b; break;
    case 4: c = -a; s = -b; break;
    case 5: c = -b; s = -a; break;
    case 6: c = b; s = -a; break;
    default: c = a; s = -b; break;
    }
    wr = c;
    wi = -s;
    if (wr == 0.0L) wr = 0.0L;  // no negative zeros in the tables
    if (wi == 0.0L) wi = 0.0L;
}

template <typename T> inline cx_t<T> twiddle_t(unsigned long long e, unsigned long long m) {
    long double wr, wi;
    twiddle_ld(e, m, wr, wi);
    cx_t<T> v;
    v.x = (T)wr;
    v.y = (T)wi;
    return v;
}

inline unsigned tw3_bits_for(unsigned log_mod) {
    unsigned b = (log_mod + 2) / 3;
    return b ? b : 1;
}

// [3][1 << bits]: level l entry j = W_{2^log_mod}^{j << (l*bits)}   (tw3_lookup, common.hpp)
template <typename T> inline std::vector<cx_t<T>> host_tw3(unsigned log_mod, unsigned bits) {
    const unsigned long long m = 1ull << log_mod;
    std::vector<cx_t<T>> h((size_t)3 << bits);
    for (unsigned l = 0; l < 3; ++l)
        for (unsigned j = 0; j < (1u << bits); ++j) {
            const unsigned shift = l * bits;
            const unsigned long long e = shift >= 63 ? 0 : (((unsigned long long)j << shift) % m);
            h[((size_t)l << bits) + j] = twiddle_t<T>(e, m);
        }
    return h;
}

// [32 + max(64, rows/32)]: W_rows^j (j < 32) and W_rows^(32 j)   (TileBody::twr_lookup; the tile kernels stage
// the first 64 entries, enough for rows <= 1024, the small-transform kernel all of them)
inline unsigned twr_entries(unsigned rows) { return 32u + (rows / 32u > 64u ? rows / 32u : 64u); }
template <typename T> inline std::vector<cx_t<T>> host_twr(unsigned rows) {
    std::vector<cx_t<T>> h(twr_entries(rows));
    for (unsigned j = 0; j < 32; ++j) h[j] = twiddle_t<T>(j, rows);
    for (unsigned j = 0; j + 32 < h.size(); ++j) h[32 + j] = twiddle_t<T>(32ull * j, rows);
    return h;
}

// step-twiddle table of the four-wave 256-row kernel (quad_fft.hpp): W_256^j, j < 64, then W_64^j, j < 64
template <typename T> inline std::vector<cx_t<T>> host_twq() {
    std::vector<cx_t<T>> h(128);
    for (unsigned j = 0; j < 64; ++j) {
        h[j] = twiddle_t<T>(j, 256);
        h[64 + j] = twiddle_t<T>(j, 64);
    }
    return h;
}

// ---- tile shapes that exist as kernels: log2(rows), log2(cols) ----
// log2(rows), log2(cols), log2(points per thread).  16 points per thread: 4096-, 8192- and 16384-point tiles
// (256 / 512 / 1024 threads); 8 points per thread: 4096-point tiles with 512 threads (latency plans); 32 points
// per thread: 16384-point tiles with 512 threads (rows twice as wide at the same thread count, one LDS exchange
// fewer for 1024-point tile FFTs), plus the 8192- and 4096-point forms for comparison.
#define PHAST_TILE_SHAPES(X)                                                                                  \
    X(6, 6, 4) X(7, 5, 4) X(8, 4, 4) X(9, 3, 4) X(10, 2, 4) X(7, 6, 4) X(8, 5, 4) X(9, 4, 4) X(10, 3, 4)      \
    X(8, 6, 4) X(9, 5, 4) X(10, 4, 4) X(6, 6, 3) X(7, 5, 3) X(8, 4, 3) X(9, 3, 3) X(10, 2, 3)                 \
    X(10, 4, 5) X(9, 5, 5) X(8, 6, 5) X(10, 3, 5) X(9, 4, 5) X(8, 5, 5) X(10, 2, 5) X(11, 3, 5)                \
    X(6, 5, 3) X(7, 4, 3) X(8, 3, 3) X(6, 4, 3) X(7, 3, 3) X(6, 5, 4) X(7, 4, 4) X(8, 3, 4) X(6, 4, 4) X(7, 3, 4) \
    X(7, 4, 5)

// 4-byte elements only: 32768-point tiles (1024 threads x 32 points), rows twice as wide again; an f64 tile
// of that size does not fit the LDS.
#define PHAST_TILE_SHAPES_F32(X) X(10, 5, 5) X(9, 6, 5) X(8, 7, 5) X(11, 4, 5)

inline bool shape_exists(unsigned lr, unsigned lc, unsigned lp, size_t elem_bytes) {
#define PHAST_CHK(LR_, LC_, LP_) \
    if (lr == LR_ && lc == LC_ && lp == LP_) return true;
    PHAST_TILE_SHAPES(PHAST_CHK)
    if (elem_bytes == 4) {
        PHAST_TILE_SHAPES_F32(PHAST_CHK)
    }
#undef PHAST_CHK
    return false;
}

constexpr unsigned kWaveTiles = 0x10;  // flag in a plan's points-per-thread code, see make_passes
constexpr unsigned kFuseBelow = 0x20;  // real_plan only: ranked WITH the fused R2C last pass below the general threshold (planner.hpp: fuse_pays)

// Padding of the planner's scratch.  The intermediate arrays between the passes are the one part of the data whose layout
// is ours: S[r][q] (two passes) / S[u][r][q] (three).  With power-of-two pitches the rows a tile reads in the next pass are
// 2^a (or 2^(a+b)) elements apart and land on the same HBM channels; a few 128-byte lines of padding per row are worth 5-10 % of the
// pass that reads them (copy models of the three patterns, profiles/r03_pad_stride_probe.log: 1024 x 16 tiles at 8 KiB
// row distance 4.80 -> 5.29 TB/s, 512 x 32 in place at 4 KiB 5.26 -> 5.60, 256 x 64 at 2 MiB 4.74 -> 5.00).  The caller's
// arrays keep their natural layout: the first pass's reads and the last pass's writes cannot be helped this way.
// On the product kernels (A/B inside one gpurun call, five fresh processes each, profiles/r03_scratch_pad_ab.log):
// 1024 x 2^20 f64 78.4-79.0 -> 82.7-84.9 GSamples/s (pass A 6.76 -> 6.5 ms, pass B 6.85 -> 6.1-6.5), f32 140 -> 157;
// 2^21 x 8 59 -> 66; 2^14 / 2^16 batches +2-7 %; the three-pass plans +0-3 % in f64 and +3-5 % in f32; one transform of
// 2^20 (wave / quad tiles) unchanged.  128 ... 640 bytes of pad are equivalent within the noise except two points:
// 256 bytes is the best f32 pad (157 against 150 on the batch), 384 bytes the best f64 pad on the small batches.
#ifdef PHAST_SCRATCH_PAD_BYTES  // tools: one pad for both types (0 = round 2's power-of-two layout)
constexpr unsigned scratch_pad_bytes(size_t) { return PHAST_SCRATCH_PAD_BYTES; }
#else
constexpr unsigned scratch_pad_bytes(size_t elem_bytes) { return elem_bytes == 8 ? 384u : 256u; }
#endif

// ---- geometry of one pass (see TileArgs in common.hpp) ----
struct PassGeom {
    unsigned lr = 0, lc = 0;
    unsigned lp = 4;  // log2(points per thread): 4 = throughput tiles, 3 = latency tiles (twice the waves)
    bool wave = false;  // one wave per 64-row x 128-byte tile, exchange by cross-lane swaps (wave_fft.hpp)
    bool quad = false;  // four waves per 256-row x 128-byte tile, one LDS + one cross-lane exchange (quad_fft.hpp; later passes)
    bool pre_tw = false, transpose = false;
    unsigned log_s_in = 0, out_lo_bits = 0, tw_bits = 1;
    unsigned long long out_s1 = 0, out_s2 = 0, out_row_stride = 0;
    // input strides (TileArgs): 0 = the power-of-two defaults derived from log_s_in and lr (the caller's arrays, strided
    // batches); set by make_passes for passes that read the PADDED scratch
    unsigned in_lo_bits = 0;
    unsigned long long in_row_stride = 0, in_hi_stride = 0, in_mid_stride = 0;
    unsigned long long scratch_dist = 0;  // elements per transform and plane in the scratch (0 = n)
    // strided batches (make_strided_passes): see TileArgs; tw_log_mod = log2 of the twiddle modulus when it is not lr + log_s_in
    unsigned tw_shift = 0, tw_mask_bits = 0, cs_bits = 0, cb_bits = 0, tw_log_mod = 0;
    unsigned grid_log_n = 0, grid_row_shift = 0;  // first pass with the input twiddle of a four-step split (TileArgs::grid_*)
    bool strided = false;
    unsigned log_mod() const { return strided ? tw_log_mod : lr + log_s_in; }  // the inter-pass twiddle is W_{2^log_mod}^{row*lo}
};

// Default factorisations of L = log2 N (L > kSmallMaxLog), from the exhaustive MI355X sweeps in profiles/
// (r01_sweep_all_*.log; profiles/HISTORY.md section 5).  What the sweeps say: a pass runs at the copy rate of its access
// pattern, and that rate is set by the contiguous segment a tile row covers (64 B rows ~3.0-3.7 TB/s, 128 B ~4.4,
// >= 256 B ~4.6-5.3; 64-byte-row WRITES are the worst case), so
//   * `throughput` (>= kThroughputWork points in flight): as few passes as possible with the widest rows the
//     register file and LDS allow -- 16384-point tiles with 32 points per thread where a tile FFT is 256..1024
//     long, 8192-point tiles with 16 points per thread otherwise;
//   * `latency` (one small transform: launch- and latency-bound, not bandwidth-bound): fewest passes with
//     4096-point tiles and 8 points per thread, so that every CU gets a workgroup and every SIMD two waves.
// `lp` = log2(points per thread) of the plan.
template <typename T>
inline void heuristic_plan(unsigned L, bool latency, std::vector<unsigned> &lrs, std::vector<unsigned> &tls, unsigned &lp) {
    lrs.clear();
    tls.assign(1, 12);
    lp = 4;
    if (L < kTwinMinLog) return;  // (L = 12, 13: the multi-pass twins of the one-pass kernel's largest sizes, planner.hpp)
    auto split = [&](unsigned np) {
        lrs.clear();
        for (unsigned i = 0; i < np; ++i) lrs.push_back(L / np + (i < L % np ? 1 : 0));  // balanced, larger first
    };
    const bool f64 = sizeof(T) == 8;
    if (latency) {
        split(L <= 19 ? 2 : 3);
        // three passes: short outer FFTs (64 x 64 tiles: 512-byte rows) around a longer middle one; measured
        // 27.9 vs 31.6 us for one f64 2^20 against two 1024 x 4 passes (profiles/r01_sweep_all_f64_single.log)
        if (lrs.size() == 3 && L - 2 * (L / 3) <= 8) lrs = {L / 3, L - 2 * (L / 3), L / 3};  // middle rows stay >= 16 wide
        lp = 3;
        return;
    }
    if (L <= 13) {  // 64 x 64 / 128 x 32 tiles: rows >= 128 B in both types
        split(2);
    } else if (L <= (f64 ? 17u : 15u)) {
        split(2);
        tls.assign(1, 13);
    } else if (L <= 20) {  // tile FFTs of 256..1024 points: 32 x 8 .. 32 x 32 in registers, one LDS exchange
        split(2);
        tls.assign(1, (!f64 && L >= 17) ? 15 : 14);  // f32: 32768-point tiles keep the rows at 128 B
        lp = 5;
    } else if (L == 21 && f64) {  // 1024 x 16, then a 2048-point tile FFT (32 x 32 x 2) on 64-byte rows: two passes
        lrs = {10, 11};          // still beat three (59 vs 51 GSamples/s); not so for f32 or 2^22
        tls.assign(1, 14);
        lp = 5;
    } else if (L <= 23) {  // (three passes need tile FFTs of at least 64 points)
        split(3);
        if (!f64 && L % 3 == 1) std::swap(lrs[0], lrs[1]);  // f32: the odd one out in the middle (+2..8 %)
        tls.assign(1, 13);
    } else {  // (f64, L >= 28: the last pass's twiddle tables no longer fit the LDS next to a 16384-point tile -- those
        split(3);  //  passes read their six table entries per thread from global memory, TileBody::tw3_global)
        tls.assign(1, (!f64 && L >= 26) ? 15 : 14);
        lp = 5;
    }
}

// The plan for ONE transform (and two).  Round 2 ranked candidate plans by timing one buffer in place; from 2^21 to 2^24
// points that buffer sits in the 256 MiB Infinity Cache, which a caller's transform of fresh data never does.  Round 4
// ranked EVERY plan that exists as kernels -- factorisation x tile size PER PASS x points per thread, wave / quad tiles
// included -- by the time of one transform over a cold ring of distinct buffers (tools/sweep_single_cold.py,
// profiles/r04_sweep_single_cold_{f64,f32}.log) and kept what an interleaved re-measurement confirmed
// (tools/confirm_single.py with three scratch allocations per plan, profiles/r04_confirm_single.log; two runs of
// tools/size_ladder.py per library, profiles/r04_single_plans_ladder_ab.log).  What holds: a 256-row first pass on
// 8192-point tiles (32 columns = 256-byte row pieces in f64), a NARROW middle pass on 4096-point tiles (more workgroups in
// flight while it runs in place in the scratch) and a 128-row last pass beat the balanced splits on one tile size:
//   f64  2^14 13.7 -> 12.2 us, 2^15 14.3 -> 11.2, 2^21 45.6 -> 42.5, 2^22 89 -> 79, 2^23 175 -> 159, 2^24 351 -> 319,
//        2^25 700 -> 658, 2^27 2700 -> 2560, 2^28 5630 -> 5450       (2^18 .. 2^20 and 2^26: the old choice stands)
//   f32  2^19 19.6 -> 17.2, 2^25 362 -> 353                        (2^24, 2^26 .. 2^28: within the noise, not adopted)
// and, with 2048-point tiles in the sweep (SWEEP_TLS=10,11,12,13, profiles/r04_sweep_single_cold_tl11.log: a latency-bound
// transform wants workgroups, 2^16 points are 16 tiles of 4096): f64 2^16 14.6 -> 11.4, 2^17 16.1 -> 14.3; f32 2^14 9.5 -> 8.4,
// 2^15 9.75 -> 9.1, 2^19 17.2 -> 16.4
// From 2^25 points on a single sweep ranks the luck of each planner's scratch allocation more than the plan (+-5 %,
// profiles/r04_placement_probe.log) -- the sweep's own gains of 6 .. 11 % at f64 2^26 and f32 2^26 / 2^28 did not survive.
// Measured with one transform in flight -- with four or more of 2^19 / 2^20 points the older latency / mid plans win again
// (2^19 x 8: 48 GSamples/s on the wave plan against 69 on the two-pass latency plan, profiles/r02_sweep_batch_wave_quad.log),
// so Planner::plan_for uses it for ONE or two transforms -- and for small batches of small ones (up to four transforms, or
// 2^19 points in flight, below 2^21 points: profiles/r04_small_batch_plans.log).  Returns false where the latency plan
// already is the right one.
template <typename T>
inline bool single_plan(unsigned L, std::vector<unsigned> &lrs, std::vector<unsigned> &tls, unsigned &lp) {
    struct E {
        unsigned L, a, b, c, ta, tb, tc, lp;  // c = 0: two passes
    };
    constexpr unsigned W = kWaveTiles;  // 64 x 16 wave tiles (wave_fft.hpp) and the four-wave 256 x 16 pass (quad_fft.hpp)
    static const E f64[] = {{12, 6, 6, 0, 10, 10, 0, 3},      // (4096 / 8192 points: Planner::twin, planner.hpp)
                            {13, 6, 7, 0, 10, 11, 0, 3 | W},
                            {14, 6, 8, 0, 10, 12, 0, 4 | W},  {15, 7, 8, 0, 10, 12, 0, 3 | W},  {16, 8, 8, 0, 11, 11, 0, 3},      {17, 8, 9, 0, 11, 12, 0, 3},
                            {19, 6, 7, 6, 10, 11, 10, 3 | W}, {20, 6, 8, 6, 10, 12, 10, 3 | W},
                            {21, 6, 8, 7, 10, 12, 12, 3 | W}, {22, 8, 7, 7, 13, 12, 13, 4},     {23, 7, 9, 7, 13, 12, 13, 4},     {24, 8, 9, 7, 13, 12, 13, 4},
                            {25, 8, 9, 8, 12, 12, 14, 4},     {27, 8, 10, 9, 13, 14, 14, 5},    {28, 9, 9, 10, 14, 14, 14, 5}};
    // f32, round 6: the one-wave 64 x 32 tiles and the four-wave 256 x 32 pass on float2 column pairs (W) where they won the same-box
    // A/B against the plan that stood (profiles/r06_f32_wave_quad_ab.log): 2^19 19.0 -> 17.8 us, 2^20 22.8 -> 19.4, 2^22 43.2 -> 39.1
    static const E f32[] = {{12, 6, 6, 0, 10, 10, 0, 3},  {14, 7, 7, 0, 11, 11, 0, 3},  {15, 8, 7, 0, 12, 11, 0, 3},  {19, 6, 7, 6, 11, 12, 11, 3 | W},
                            {20, 6, 8, 6, 11, 13, 11, 3 | W}, {22, 6, 8, 8, 11, 13, 13, 3 | W}, {23, 8, 8, 7, 12, 12, 12, 4}, {25, 8, 9, 8, 14, 13, 14, 4}};
    const E *tab = sizeof(T) == 8 ? f64 : f32;
    const size_t cnt = (sizeof(T) == 8 ? sizeof f64 : sizeof f32) / sizeof(E);
    for (size_t i = 0; i < cnt; ++i)
        if (tab[i].L == L) {
            const E &e = tab[i];
            lrs = {e.a, e.b};
            tls = {e.ta, e.tb};
            if (e.c) {
                lrs.push_back(e.c);
                tls.push_back(e.tc);
            }
            lp = e.lp;
            return true;
        }
    return false;
}

// The inner N/2-point transform of ONE (or two) REAL transforms (`L` = log2 of the inner length).  The C2C plans above were
// ranked on planar input and output; R2C reads (re, im) PAIRS in its first pass and ends in the fused untangle pass (twice
// the work per tile, mirrored columns: r2c_fused.hpp), C2R starts with the fused preprocess pass (four streams per tile:
// c2r_fused.hpp) and ends writing pairs -- the best factorisations differ.  Ranked on the GPU over every 2- / 3-pass
// factorisation x tile size x points per thread (tools/sweep_real.py, profiles/r04_sweep_real_*.log), adopted where they
// beat the C2C choice by more than the run-to-run noise (3 %):
//   R2C  f32 2^25: 198 -> 177 us, 2^26: 429 -> 395, 2^27: 980 -> 820, 2^28: 1862 -> 1620 (the last two ran the untangle as a
//        sweep of its own until round 4: the throughput plan's 32-point last pass has no fused form);
//        f64 2^23: 122 -> 106, 2^24: 208 -> 174, 2^25: 401 -> 357, 2^26: 763 -> 695, 2^27: 1760 -> 1322, 2^28: 3385 -> 2785 (last pass + sweep 1699 -> fused 1034)
//   C2R  f32 2^23: 50.9 -> 48.1, 2^25: 201 -> 184, 2^26: 444 -> 389;  f64 2^23: 98 -> 82, 2^24: 188 -> 172, 2^25: 450 -> 356,
//        2^26: 822 -> 724, 2^27: 1510 -> 1365
// (real lengths; everything else keeps the C2C plan).  Returns false where there is no entry.
template <typename T>
inline bool real_plan(unsigned L, bool c2r, std::vector<unsigned> &lrs, std::vector<unsigned> &tls, unsigned &lp) {
    struct E {
        unsigned L, a, b, c, ta, tb, tc, lp;  // c = 0: two passes
    };
    constexpr unsigned W = kWaveTiles, F = kFuseBelow;
    // (r2c32 2^19 and c2r64 2^21: the C2C single_plan of that size has no fused form of the pass the real transform needs --
    //  a third pass in front of the untangle sweep, a wave tile as the first pass of C2R -- these keep the plans they had:
    //  r2c_fft_f32 2^20 20.6 us against 21.6, c2r_fft_f64 2^22 47.4 against 51.8, profiles/r04_single_plans_ladder_ab.log)
    // R2C 2^20 .. 2^22 (F): the fused last pass runs half as many workgroups, each with two tiles -- on 4096-point tiles that
    // leaves one workgroup per CU below 2^23 points and lost to the separate sweep (round 3).  On 2048-point last-pass tiles
    // it wins from 2^20 points on (tools/sweep_real.py with SWEEP_TLS=11,12,13, profiles/r04_r2c_fuse_small_tiles.log):
    //   r2c_fft_f32 2^21 31.5 -> 29.9 us, 2^22 42.9 -> 39.7, 2^23 65 -> 52;  r2c_fft_f64 2^22 68.6 -> 51.3, 2^23 119 -> 85
    // C2R 2^20 / 2^21 on 2048-point tiles (more workgroups for a latency-bound transform): f64 29.4 -> 24.0, 36.2 -> 33.2,
    //   f32 21.8 -> 20.3, 26.8 -> 23.9
    static const E r2c32[] = {{19, 10, 9, 0, 12, 12, 0, 3},     {20, 7, 7, 6, 12, 12, 11, 3 | F}, {21, 7, 8, 6, 12, 12, 11, 3 | F},
                              {22, 8, 8, 6, 13, 13, 12, 4 | F}, {24, 8, 9, 7, 12, 13, 12, 4},     {25, 8, 9, 8, 13, 13, 13, 4},
                              {26, 9, 9, 8, 13, 13, 13, 4},     {27, 10, 9, 8, 13, 13, 13, 4}};
    static const E r2c64[] = {{21, 9, 6, 6, 12, 11, 11, 3 | F}, {22, 8, 8, 6, 11, 11, 11, 3 | F}, {23, 9, 8, 6, 12, 12, 12, 3},
                              {24, 9, 9, 6, 12, 13, 12, 4},     {25, 9, 9, 7, 12, 13, 12, 4},     {26, 9, 9, 8, 13, 13, 13, 4},
                              {27, 9, 10, 8, 13, 13, 13, 4}};
    static const E c2r32[] = {{19, 6, 7, 6, 11, 11, 11, 3}, {20, 6, 7, 7, 11, 12, 12, 3}, {22, 8, 7, 7, 13, 12, 12, 4},
                              {24, 8, 8, 8, 13, 13, 13, 4}, {25, 8, 9, 8, 13, 13, 13, 4}};
    static const E c2r64[] = {{19, 7, 6, 6, 11, 11, 11, 3},     {20, 7, 6, 7, 11, 11, 11, 3}, {21, 8, 7, 6, 12, 12, 10, 3 | W}, {22, 8, 7, 7, 12, 13, 13, 4},
                              {23, 8, 8, 7, 12, 12, 13, 4},     {24, 8, 8, 8, 12, 12, 12, 4}, {25, 8, 9, 8, 13, 13, 13, 4},     {26, 8, 9, 9, 13, 13, 12, 4}};
    const E *tab = sizeof(T) == 4 ? (c2r ? c2r32 : r2c32) : (c2r ? c2r64 : r2c64);
    const size_t cnt = sizeof(T) == 4 ? (c2r ? sizeof c2r32 : sizeof r2c32) / sizeof(E) : (c2r ? sizeof c2r64 : sizeof r2c64) / sizeof(E);
    for (size_t i = 0; i < cnt; ++i)
        if (tab[i].L == L) {
            const E &e = tab[i];
            lrs = {e.a, e.b};
            tls = {e.ta, e.tb};
            if (e.c) {
                lrs.push_back(e.c);
                tls.push_back(e.tc);
            }
            lp = e.lp;
            return true;
        }
    return false;
}

// ... and for BATCHES of real transforms in the throughput regime (`L` = log2 of the inner length).  The C2C throughput plans
// end (and begin) in 32-point-per-thread passes on 16384- / 32768-point tiles, most of which have no fused untangle /
// preprocess form: batched R2C then ran three sweeps where two would do, and batched C2R kept a first pass cut for planar
// input.  Ranked with 2^27 real samples in flight (tools/sweep_real_batch.py, profiles/r04_sweep_real_batch.log):
// and kept where an alternating A/B of the table (PHAST_REAL_PLANS=0|1, profiles/r04_real_batch_ab.log) confirmed it:
//   r2c_fft_f32  2^15 749 -> 567 us, 2^17 613 -> 535, 2^18 711 -> 547, 2^19 736 -> 509, 2^20 924 -> 597 (145 -> 225
//                GSamples/s), 2^21 790 -> 686;   r2c_fft_f64  2^15 1401 -> 1088, 2^16 1168 -> 1001, 2^17 .. 2^20 + 8 .. 17 %
//   c2r_fft_f32  2^17 566 -> 477, 2^18 .. 2^20 + 3 .. 7 %;   c2r_fft_f64  2^15 .. 2^20 + 2 .. 11 %
//   (the three-pass f64 sizes 2^21 .. 2^23: 0 .. 12 % from run to run -- not adopted)
// What wins: R2C -- a long first pass on big tiles, then a SHORT fused last pass (64 .. 512 rows on 4096- / 8192-point
// tiles at 16 points per thread); C2R -- the mirror image, a short fused first pass.
template <typename T>
inline bool real_batch_plan(unsigned L, bool c2r, std::vector<unsigned> &lrs, std::vector<unsigned> &tls, unsigned &lp) {
    struct E {
        unsigned L, a, b, c, ta, tb, tc, lp;  // c = 0: two passes
    };
    static const E r2c32[] = {{13, 7, 6, 0, 12, 12, 0, 3},  {14, 8, 6, 0, 12, 12, 0, 3},  {15, 9, 6, 0, 14, 12, 0, 4},  {16, 10, 6, 0, 14, 12, 0, 4},
                              {17, 9, 8, 0, 14, 13, 0, 5},  {18, 10, 8, 0, 15, 13, 0, 5}, {19, 10, 9, 0, 14, 13, 0, 4}, {20, 10, 10, 0, 15, 14, 0, 5}};
    static const E r2c64[] = {{14, 8, 6, 0, 14, 12, 0, 4},  {15, 9, 6, 0, 14, 12, 0, 4},  {16, 9, 7, 0, 13, 12, 0, 4},  {17, 10, 7, 0, 14, 12, 0, 4},
                              {18, 10, 8, 0, 14, 13, 0, 4}, {19, 10, 9, 0, 14, 13, 0, 4}};
    static const E c2r32[] = {{13, 6, 7, 0, 12, 13, 0, 4},  {14, 6, 8, 0, 12, 13, 0, 4},  {15, 8, 7, 0, 13, 13, 0, 4},  {16, 8, 8, 0, 13, 14, 0, 4},
                              {17, 8, 9, 0, 13, 14, 0, 5},  {18, 8, 10, 0, 13, 15, 0, 5}, {19, 10, 9, 0, 15, 14, 0, 5}, {20, 10, 10, 0, 15, 15, 0, 5}};
    static const E c2r64[] = {{13, 6, 7, 0, 12, 13, 0, 4},  {14, 6, 8, 0, 12, 13, 0, 4},  {15, 7, 8, 0, 12, 14, 0, 4},  {16, 8, 8, 0, 12, 13, 0, 4},
                              {17, 8, 9, 0, 13, 13, 0, 4},  {18, 8, 10, 0, 13, 14, 0, 4}, {19, 9, 10, 0, 14, 14, 0, 4}};
    const E *tab = sizeof(T) == 4 ? (c2r ? c2r32 : r2c32) : (c2r ? c2r64 : r2c64);
    const size_t cnt = sizeof(T) == 4 ? (c2r ? sizeof c2r32 : sizeof r2c32) / sizeof(E) : (c2r ? sizeof c2r64 : sizeof r2c64) / sizeof(E);
    for (size_t i = 0; i < cnt; ++i)
        if (tab[i].L == L) {
            const E &e = tab[i];
            lrs = {e.a, e.b};
            tls = {e.ta, e.tb};
            if (e.c) {
                lrs.push_back(e.c);
                tls.push_back(e.tc);
            }
            lp = e.lp;
            return true;
        }
    return false;
}

// A third plan for "a few transforms in flight" where neither of the two above fits: N = 2^20, whose latency plan
// takes three passes (best for ONE transform) while 2..15 transforms are better served by two passes of
// 8192-point tiles (64 / 68 / 75 GSamples/s at 2 / 4 / 8 transforms against 47 / 44 / 49,
// profiles/r01_sweep_batch_f64.log).  Returns false where the latency plan already is the right one.
template <typename T>
inline bool mid_plan(unsigned L, std::vector<unsigned> &lrs, std::vector<unsigned> &tls, unsigned &lp) {
    if (L != 20) return false;
    lrs = {10, 10};
    tls.assign(1, 13);
    lp = 4;
    return true;
}

// points in flight (batch * n) from which the throughput plan is used; below it the chip is better filled by the
// latency plan's 4096-point tiles (measured crossovers, profiles/r01_sweep_batch_f64.log and the single-transform
// sweeps: 8192- and 16384-point tiles win from 2^24 points on, 32768-point tiles from 2^25)
inline size_t throughput_work(unsigned tile_log) {
    return (size_t)1 << (tile_log >= 15 ? 25 : tile_log >= 13 ? 24 : 22);
}

// N = 2^L as 2 passes (a, b) or 3 passes (a, b, c):
//   x[p][r][u] --A: FFT over p, runs out--> S[u][r][q] --B: FFT over r, in place--> S[u][kb][q]
//              --C: FFT over u--> x[kc][kb][q]            (2 passes: S[r][q] --B--> x[kb][q])
// `tile_logs` gives log2(points per tile) of every pass (one entry = the same for all passes).
inline bool make_passes(unsigned L, const std::vector<unsigned> &lrs, const std::vector<unsigned> &tile_logs,
                        std::vector<PassGeom> &ps, unsigned lp, size_t elem_bytes) {
    unsigned sum = 0;
    for (unsigned lr : lrs) sum += lr;
    if (lrs.size() < 2 || lrs.size() > 3 || sum != L) return false;
    if (tile_logs.size() != 1 && tile_logs.size() != lrs.size()) return false;
    ps.assign(lrs.size(), PassGeom());
    const unsigned a = lrs[0], b = lrs[1], c = lrs.size() == 3 ? lrs[2] : 0;
    // lp & 0x10 (kWaveTiles): every pass whose tile is 64 rows x 128 bytes (16 f64 / 32 f32 columns) runs as wave tiles,
    // every later f64 pass with 256 x 16 tiles as the four-wave kernel;
    // lp & 0xf = log2(points per thread) of the other passes
    const bool want_wave = (lp & kWaveTiles) != 0;
    lp &= 0xfu;
    for (size_t i = 0; i < lrs.size(); ++i) {
        const unsigned tl = tile_logs.size() == 1 ? tile_logs[0] : tile_logs[i];
        if (lrs[i] > tl) return false;
        ps[i].lr = lrs[i];
        ps[i].lc = tl - lrs[i];
        ps[i].wave = want_wave && lrs[i] == 6 && tl == (elem_bytes == 8 ? 10u : 11u);
        ps[i].quad = want_wave && lrs[i] == 8 && tl == (elem_bytes == 8 ? 12u : 13u) && i > 0;
        ps[i].lp = ps[i].wave ? (elem_bytes == 8 ? 4u : 5u) : ps[i].quad ? 4u : lp;
        if (!ps[i].wave && !ps[i].quad && !shape_exists(lrs[i], tl - lrs[i], lp, elem_bytes)) return false;
    }
    ps[0].transpose = true;  // FFT over the top `a` index bits; every column leaves as one contiguous run
    ps[0].log_s_in = L - a;
    ps[0].out_row_stride = 1;
    if (lrs.size() == 2) {
        ps[0].out_lo_bits = L - a;  // out_col(g) = g * 2^a
        ps[0].out_s1 = 1ull << a;
        ps[0].out_s2 = 0;
    } else {
        ps[0].out_lo_bits = c;  // g = r*2^c + u  ->  u*2^(a+b) + r*2^a
        ps[0].out_s1 = 1ull << (a + b);
        ps[0].out_s2 = 1ull << a;
    }
    // pass B: FFT over r (stride 2^a); columns (u, q); twiddle modulus 2^(a+b); same pattern in and out
    ps[1].pre_tw = true;
    ps[1].log_s_in = a;
    ps[1].out_lo_bits = a;
    ps[1].out_s1 = 1;
    ps[1].out_s2 = 1ull << (a + b);
    ps[1].out_row_stride = 1ull << a;
    ps[1].tw_bits = tw3_bits_for(a + b);
    if (lrs.size() == 3) {  // pass C: FFT over u (stride 2^(a+b)); columns (kb, q); modulus 2^L
        ps[2].pre_tw = true;
        ps[2].log_s_in = a + b;
        ps[2].out_lo_bits = a + b;
        ps[2].out_s1 = 1;
        ps[2].out_s2 = 1ull << L;
        ps[2].out_row_stride = 1ull << (a + b);
        ps[2].tw_bits = tw3_bits_for(L);
    }
    for (auto &p : ps) {
        if (p.lc > p.log_s_in) return false;    // a tile needs COLS adjacent columns sharing the row stride
        if (p.lc > p.out_lo_bits) return false;  // ... and the output column map must be linear inside a tile
    }
    // ---- padded scratch pitches: S[r][q] with pitch PR = 2^a + pad;  S[u][r][q] with PU = 2^b PR + pad ----
    const unsigned long long pad = elem_bytes ? scratch_pad_bytes(elem_bytes) / elem_bytes : 0;
    // Not for the wave / quad plans of ONE transform: their intermediate lives in the L2 / Infinity Cache, where the
    // padded layout only enlarges the footprint -- the last pass of the single 2^20 transform went from 6.6 to 7.1 us.
    bool pad_ok = pad != 0 && L + 1 <= 31 && !want_wave;  // (the per-lane offsets stay 32-bit with the padded strides)
    for (size_t i = 1; i < ps.size(); ++i) pad_ok = pad_ok && ps[i].lc <= a;  // tiles stay inside the contiguous q
    if (pad_ok) {
        const unsigned long long PR = (1ull << a) + pad;
        if (lrs.size() == 2) {
            const unsigned long long n_s = PR << b;
            ps[0].out_s1 = PR;  // out_col(g) = g * PR
            ps[0].scratch_dist = ps[1].scratch_dist = n_s;
            ps[1].in_row_stride = PR;
            ps[1].in_lo_bits = a;
            ps[1].in_hi_stride = n_s;  // never used: g < 2^a
        } else {
            const unsigned long long PU = (PR << b) + pad, n_s = PU << c;
            ps[0].out_s1 = PU;  // g = r 2^c + u  ->  u PU + r PR
            ps[0].out_s2 = PR;
            // pass B: columns g = (u, q), rows r; in place
            ps[1].in_row_stride = PR;
            ps[1].in_lo_bits = a;
            ps[1].in_hi_stride = PU;
            ps[1].out_s2 = PU;
            ps[1].out_row_stride = PR;
            // pass C: columns g = (kb, q) -> kb PR + q, rows u
            ps[2].in_row_stride = PU;
            ps[2].in_lo_bits = a;
            ps[2].in_mid_stride = PR;
            ps[2].in_hi_stride = n_s;  // never used: g < 2^(a+b)
            ps[0].scratch_dist = ps[1].scratch_dist = ps[2].scratch_dist = n_s;
        }
    }
    return L <= 31;  // per-lane offsets and twiddle exponents are 32-bit element indices
}

// elements per transform and plane the scratch of a pass list needs (>= n: the padded pitches of make_passes)
inline unsigned long long scratch_elems(const std::vector<PassGeom> &ps, unsigned log_n) {
    unsigned long long m = 1ull << log_n;
    for (const auto &p : ps)
        if (p.scratch_dist > m) m = p.scratch_dist;
    return m;
}

inline void geom_to_args(const PassGeom &p, unsigned log_n, size_t n_xforms, TileArgs &ta) {
    ta.in_row_stride = p.in_row_stride ? p.in_row_stride : 1ull << p.log_s_in;
    ta.in_hi_stride = p.in_hi_stride ? p.in_hi_stride : 1ull << (p.log_s_in + p.lr);
    ta.in_mid_stride = p.in_mid_stride;
    ta.in_lo_bits = p.in_row_stride ? p.in_lo_bits : p.log_s_in;
    ta.out_s1 = p.out_s1;
    ta.out_s2 = p.out_s2;
    ta.out_row_stride = p.out_row_stride;
    ta.log_s_in = p.log_s_in;
    ta.out_lo_bits = p.out_lo_bits;
    ta.tw_bits = p.tw_bits;
    if (p.strided) {  // ONE array of 2^log_n rows x 2^cs_bits columns, of which 2^cb_bits * COLS columns are transforms
        ta.tw_shift = p.tw_shift;
        ta.tw_mask = p.tw_mask_bits >= 32 ? 0xffffffffu : ((1u << p.tw_mask_bits) - 1u);
        ta.cs_bits = p.cs_bits;
        ta.cb_bits = p.cb_bits;
        ta.grid_mode = p.grid_log_n ? 1u : 0u;
        ta.grid_row_shift = p.grid_row_shift;
        ta.grid_col_mask = (1u << p.tw_shift) - 1u;
        ta.tiles_per_xform = 1u << (log_n - p.lr + p.cb_bits);
        ta.tiles_total = ta.tiles_per_xform;
        return;
    }
    ta.tw_shift = 0;
    ta.tw_mask = p.log_s_in >= 32 ? 0xffffffffu : ((1u << p.log_s_in) - 1u);
    ta.cs_bits = ta.cb_bits = 0;
    ta.tiles_per_xform = 1u << (log_n - p.lr - p.lc);
    ta.tiles_total = (unsigned)((size_t)ta.tiles_per_xform * n_xforms);
}

// ---- plan specifications and the candidate set of a tuning run (PlannerMode::Tune, planner.rs:18-32) ----
// A plan is written "a,b[,c]@ta,tb[,tc]:p<points per thread>[w]" -- log2(rows) of every pass, log2(points per tile) of every
// pass, points per thread of the generic tiles, `w` = wave / quad tiles where a pass has their shape (kWaveTiles) -- the
// notation of the sweep logs in profiles/ and of the wisdom text (wisdom.hpp).
struct PlanSpec {
    unsigned np = 0, lr[3] = {0, 0, 0}, tl[3] = {0, 0, 0};
    unsigned lp = 4;  // log2(points per thread) | kWaveTiles
    std::vector<unsigned> lrs() const { return std::vector<unsigned>(lr, lr + np); }
    std::vector<unsigned> tls() const { return std::vector<unsigned>(tl, tl + np); }
    bool operator==(const PlanSpec &o) const {
        if (np != o.np || lp != o.lp) return false;
        for (unsigned i = 0; i < np; ++i)
            if (lr[i] != o.lr[i] || tl[i] != o.tl[i]) return false;
        return true;
    }
};
inline std::string spec_to_string(const PlanSpec &s) {
    std::string out;
    for (unsigned i = 0; i < s.np; ++i) out += (i ? "," : "") + std::to_string(s.lr[i]);
    out += "@";
    for (unsigned i = 0; i < s.np; ++i) out += (i ? "," : "") + std::to_string(s.tl[i]);
    out += ":p" + std::to_string(1u << (s.lp & 0xfu));
    if (s.lp & kWaveTiles) out += "w";
    return out;
}
inline bool spec_from_string(const char *t, PlanSpec &s) {
    s = PlanSpec();
    auto list = [&](unsigned *dst, unsigned &cnt, char stop) {
        cnt = 0;
        for (;;) {
            if (*t < '0' || *t > '9' || cnt == 3) return false;
            unsigned v = 0;
            while (*t >= '0' && *t <= '9') v = v * 10 + (unsigned)(*t++ - '0');
            if (v > 31) return false;
            dst[cnt++] = v;
            if (*t == ',') {
                ++t;
                continue;
            }
            return *t++ == stop;
        }
    };
    unsigned na = 0, nb = 0;
    if (!t || !list(s.lr, na, '@') || !list(s.tl, nb, ':') || na != nb || na < 2) return false;
    s.np = na;
    if (*t++ != 'p') return false;
    unsigned pts = 0;
    while (*t >= '0' && *t <= '9') pts = pts * 10 + (unsigned)(*t++ - '0');
    if (pts != 8 && pts != 16 && pts != 32) return false;
    s.lp = pts == 8 ? 3u : pts == 16 ? 4u : 5u;
    if (*t == 'w') {
        s.lp |= kWaveTiles;
        ++t;
    }
    return *t == '\0' || *t == ' ' || *t == '\n';
}

// What a candidate is expected to cost before anything is timed -- only to ORDER the candidates, so that a tuning run cut
// short by its budget has seen the likely ones: a pass moves 4 * sizeof(T) bytes per point at the copy rate of its row width
// (profiles/HISTORY.md section 5: >= 256-byte rows 5.0 TB/s, 128 bytes 4.4, 64 bytes 3.3, 32 bytes 2.0) plus a kernel boundary.
inline double plan_model_us(const std::vector<PassGeom> &ps, unsigned L, size_t batch, size_t elem_bytes) {
    double us = 0;
    const double points = (double)batch * (double)(1ull << L);
    const double bytes = 4.0 * (double)elem_bytes * points;
    for (const PassGeom &p : ps) {
        const unsigned row = (unsigned)elem_bytes << p.lc;
        const double tbps = row >= 256 ? 5.0 : row >= 128 ? 4.4 : row >= 64 ? 3.3 : 2.0;
        // a launch with fewer threads than the chip holds (256 CUs x 512) is latency-bound: fewer points per thread help
        const double threads = points / (double)(1u << p.lp);
        const double fill = threads < 131072.0 ? std::sqrt(131072.0 / threads) : 1.0;
        us += 1.5 + fill * bytes / (tbps * 1e6);
    }
    return us;
}

// Every plan of a 2^L-point transform that exists as kernels: 2 or 3 passes, tile FFTs of 64 .. 2048 rows, a tile size PER
// PASS in [tl_lo, tl_hi], 8 / 16 / 32 points per thread, f64 also the one-wave 64 x 16 tiles and the four-wave 256 x 16 pass
// -- what tools/sweep_single_cold.py and tools/sweep_real*.py enumerated by hand until round 5.  Sorted by plan_model_us.
inline void enumerate_plans(unsigned L, size_t elem_bytes, size_t batch, unsigned tl_lo, unsigned tl_hi, std::vector<PlanSpec> &out) {
    out.clear();
    if (L < kTwinMinLog || L > 31) return;
    const unsigned lps[] = {3, 4, 5, 3 | kWaveTiles, 4 | kWaveTiles};
    const unsigned n_lps = 5;
    std::vector<std::pair<double, PlanSpec>> scored;
    std::vector<PassGeom> geo;
    for (unsigned np = 2; np <= 3; ++np) {
        unsigned lr[3] = {6, 6, 6};
        for (;;) {  // odometer over the row digits
            unsigned sum = 0;
            for (unsigned i = 0; i < np; ++i) sum += lr[i];
            if (sum == L) {
                unsigned tl[3] = {tl_lo, tl_lo, tl_lo};
                for (;;) {  // ... and over the tile sizes
                    bool cols_ok = true, wave_shaped = false;
                    for (unsigned i = 0; i < np; ++i) {
                        if (tl[i] < lr[i] + 2 || tl[i] > lr[i] + 7) cols_ok = false;
                        if ((lr[i] == 6 && tl[i] == (elem_bytes == 8 ? 10u : 11u)) || (lr[i] == 8 && tl[i] == (elem_bytes == 8 ? 12u : 13u) && i > 0))
                            wave_shaped = true;
                    }
                    for (unsigned k = 0; k < n_lps && cols_ok; ++k) {
                        if ((lps[k] & kWaveTiles) && !wave_shaped) continue;  // the same plan as without the flag
                        PlanSpec s;
                        s.np = np;
                        for (unsigned i = 0; i < np; ++i) {
                            s.lr[i] = lr[i];
                            s.tl[i] = tl[i];
                        }
                        s.lp = lps[k];
                        if (!make_passes(L, s.lrs(), s.tls(), geo, s.lp, elem_bytes)) continue;
                        scored.emplace_back(plan_model_us(geo, L, batch, elem_bytes), s);
                    }
                    unsigned i = 0;
                    while (i < np && ++tl[i] > tl_hi) tl[i++] = tl_lo;
                    if (i == np) break;
                }
            }
            unsigned i = 0;
            while (i < np && ++lr[i] > 11) lr[i++] = 6;
            if (i == np) break;
        }
    }
    std::stable_sort(scored.begin(), scored.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (auto &e : scored) out.push_back(e.second);
}

// Do two runs' per-transform digests {sum re, sum im, energy, one probed value} (fill.hip: digest_kernel) describe the same
// results?  What a tuning run asks before it adopts a plan (tune.hpp): far coarser than the parity gates -- two correct plans
// differ by rounding, 1e-15 / 1e-6 relative -- and enough to stop a plan whose geometry is wrong.
inline bool digests_agree(const double *a, const double *c, size_t batch, size_t n, size_t elem_bytes) {
    const double tol = elem_bytes == 8 ? 1e-9 : 1e-4;
    for (size_t b = 0; b < batch; ++b, a += 4, c += 4) {
        const double e = a[2], rms = std::sqrt(e > 0 ? e : 0), sum_tol = tol * rms * std::sqrt((double)n) + 1e-300;
        for (int i = 0; i < 4; ++i)
            if (!std::isfinite(a[i]) || !std::isfinite(c[i])) return false;
        // (the probed value against the rms BIN: a permutation of the outputs keeps the energy and both sums)
        const double bin_tol = (elem_bytes == 8 ? 1e-9 : 1e-3) * rms / std::sqrt((double)n) + 1e-300;
        if (std::fabs(c[2] - e) > tol * e || std::fabs(c[0] - a[0]) > sum_tol || std::fabs(c[1] - a[1]) > sum_tol ||
            std::fabs(c[3] - a[3]) > bin_tol)
            return false;
    }
    return true;
}

// the tile sizes worth trying for `points` in flight (batch * n): a latency-bound call wants many small tiles, a full chip
// the widest rows (section 5; the ranges the round-4 sweeps were run with)
inline void tune_tile_range(size_t points, size_t elem_bytes, unsigned &tl_lo, unsigned &tl_hi) {
    const unsigned lg = 63u - (unsigned)__builtin_clzll((unsigned long long)(points ? points : 1));
    tl_lo = lg <= 21 ? 10u : lg <= 24 ? 11u : 12u;
    tl_hi = lg <= 19 ? 13u : (elem_bytes == 4 && lg >= 22) ? 15u : 14u;
}

// ---- strided batches: 2^sb transforms of 2^L points, transform c at element c, its points 2^s elements apart ----
// (a matrix [2^L][2^s] whose first 2^sb columns are transformed along the rows' axis: "column FFTs").  The batch
// index c is the contiguous dimension everywhere, so NO pass needs the transposing store: with the digits of the
// transform index n = (p, r, u) high to low,
//     x[p][r][u][c] --0: FFT over p--> T1[r][u][kp][c] --1: x W_{2^(a+b)}^(r kp), FFT over r--> T2[u][kb][kp][c]
//                   --2: x W_N^(u (kp + 2^a kb)), FFT over u--> X[kc][kb][kp][c]      (natural order, k = kp + 2^a kb + ..)
// every pass reads rows with the full lower part as its (contiguous) column space and writes rows of >= COLS
// contiguous elements.  Pass 0 runs through the same pre-twiddle kernels with tw_mask = 0 (multiplier 0: W^0 = 1
// exactly).  Passes are NOT in place except the last: x -> scratch -> x (-> x).
inline unsigned pick_lp(unsigned lr, unsigned lc, size_t elem_bytes) {
    for (unsigned lp : {4u, 5u, 3u})
        if (lp <= lr && shape_exists(lr, lc, lp, elem_bytes)) return lp;
    return 0;
}
// grid_log_n != 0: the first pass multiplies x[j][c] by W_{2^grid_log_n}^(j (col0 + c)) on load (col0 per call).
inline bool make_strided_passes(unsigned L, unsigned s, unsigned sb, size_t elem_bytes, std::vector<PassGeom> &ps,
                                unsigned grid_log_n = 0) {
    if (L < 6 || sb > s || L + s > 31) return false;  // tile FFTs are 64..1024 points; 32-bit element offsets
    if (grid_log_n > 32 || (grid_log_n && grid_log_n < L)) return false;
    unsigned np = (L + 9) / 10;                        // rows <= 1024 per pass
    if (np > 3) return false;
    unsigned want_lc = elem_bytes == 8 ? 4u : 5u;      // 128-byte rows
    if (L == 11) {  // 6 + 5 has no 32-point tile FFT: one pass of 2048-point tile FFTs on 8 columns (the (11, 3) shape)
        np = 1;
        want_lc = 3;
    }
    std::vector<unsigned> lrs;
    for (unsigned i = 0; i < np; ++i) lrs.push_back(L / np + (i < L % np ? 1 : 0));
    for (unsigned lr : lrs)
        if (lr < 6) return false;
    const unsigned lc = sb < want_lc ? sb : want_lc;
    ps.assign(np, PassGeom());
    unsigned done = 0;  // log2 of the digits already transformed (they sit below the remaining ones, above c)
    for (unsigned i = 0; i < np; ++i) {
        PassGeom &g = ps[i];
        g.lr = lrs[i];
        g.lc = lc;
        g.lp = pick_lp(g.lr, g.lc, elem_bytes);
        if (!g.lp) return false;
        g.strided = true;
        g.pre_tw = true;  // pass 0 too: multiplier 0
        g.transpose = false;
        g.log_s_in = L - lrs[i] + s;  // rows of this pass are the TOP digit of what is left
        g.cs_bits = sb < s ? s : 0;   // all columns used: plain tile numbering
        g.cb_bits = sb - lc;
        const bool last = i + 1 == np;
        g.out_row_stride = 1ull << (done + s);  // the new digit k_i goes right above the digits already done
        if (last) {
            g.out_lo_bits = done + s;  // all columns are low digits already in place
            g.out_s1 = 1;
            g.out_s2 = 1ull << (L + s);  // never used: g < 2^out_lo_bits
        } else {
            g.out_lo_bits = done + s;          // (done digits, c) stay, the untransformed digits move up by lr bits
            g.out_s1 = 1;
            g.out_s2 = 1ull << (done + s + lrs[i]);
        }
        g.tw_shift = s;
        g.tw_mask_bits = i == 0 ? 0 : done;  // lo = the digits already done (kp, kb..), not the untransformed ones
        g.tw_log_mod = done + lrs[i];        // W_{2^(done + lr)}^(row * lo)
        g.tw_bits = tw3_bits_for(i == 0 ? 3 : g.tw_log_mod);
        if (i == 0 && grid_log_n) {
            g.grid_log_n = grid_log_n;
            g.grid_row_shift = L - lrs[0];  // weight of this pass's row digit in the transform index j
            g.tw_log_mod = grid_log_n;
            g.tw_bits = tw3_bits_for(grid_log_n);
        }
        done += lrs[i];
    }
    return true;
}

}  // namespace phast
