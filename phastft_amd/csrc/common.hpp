// common.hpp -- shared device helpers and launch-argument structs of libphastft_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "device_state.hpp"

#define PHAST_HD __host__ __device__ __forceinline__


namespace phast {

// ---- complex pair type: one 8/16-byte LDS or global access per twiddle ----
template <typename T> struct Cx;
template <> struct Cx<float> { using type = float2; };
template <> struct Cx<double> { using type = double2; };
template <typename T> using cx_t = typename Cx<T>::type;

// ---- the register type of a lane: one scalar, or (f32 wave / four-wave tiles, round 6) TWO ADJACENT COLUMNS packed in 8 bytes.
// A lane of an f32 wave tile holds float2 column pairs, so that its rows are 128 bytes per plane with 8-byte accesses and the
// f64 tiles' lane layout, exchanges and instruction count carry over unchanged (the arithmetic is elementwise on the pair;
// step twiddles are scalars, broadcast).  Arithmetic helpers below are written for either: constants and table entries are
// scalars (scalar_t<V>), data are V.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename V> struct ScalarOf { using type = V; static constexpr int W = 1; };
template <> struct ScalarOf<f32x2> { using type = float; static constexpr int W = 2; };
template <typename V> using scalar_t = typename ScalarOf<V>::type;
template <> struct Cx<f32x2> { using type = float2; };  // twiddle tables hold scalar complex numbers
template <typename T> struct LaneVec { using type = T; };
template <> struct LaneVec<float> { using type = f32x2; };
template <typename T> using lane_vec_t = typename LaneVec<T>::type;  // double -> double, float -> f32x2
template <typename V> PHAST_HD V splat(scalar_t<V> s) {
    if constexpr (ScalarOf<V>::W == 1) return s;
    else return V{s, s};
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, typename F> PHAST_HD void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

constexpr int bitrev_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int ilog2_c(int x) { return x <= 1 ? 0 : 1 + ilog2_c(x >> 1); }

// (re, im) *= (wr, wi)
template <typename T> PHAST_HD void cmul(T &re, T &im, T wr, T wi) {
    T r = re * wr - im * wi;
    T i = re * wi + im * wr;
    re = r;
    im = i;
}
// ... a packed pair of columns times ONE scalar twiddle
PHAST_HD void cmul(f32x2 &re, f32x2 &im, float wr, float wi) {
    f32x2 r = re * wr - im * wi;
    f32x2 i = re * wi + im * wr;
    re = r;
    im = i;
}

// Three-level twiddle lookup: W_{2^log_mod}^e = T0[e & m] * T1[(e >> B) & m] * T2[(e >> 2B) & m],
// tables laid out [3][1 << B] (built on the host in long double, plan.cpp).  `tab` may be LDS or global.
template <typename T> struct Tw3Raw {
    cx_t<T> a, b, c;
};
// the three table reads alone (a kernel issues them ahead of other loads and multiplies later) ...
template <typename T> PHAST_HD Tw3Raw<T> tw3_fetch(const cx_t<T> *tab, unsigned bits, unsigned e) {
    const unsigned m = (1u << bits) - 1u;
    Tw3Raw<T> t;
    t.a = tab[e & m];
    t.b = tab[(1u << bits) + ((e >> bits) & m)];
    t.c = tab[(2u << bits) + ((e >> (2 * bits)) & m)];
    return t;
}
// ... and their product
template <typename T> PHAST_HD void tw3_combine(const Tw3Raw<T> &t, T &wr, T &wi) {
    T r = t.a.x * t.b.x - t.a.y * t.b.y;
    T i = t.a.x * t.b.y + t.a.y * t.b.x;
    wr = r * t.c.x - i * t.c.y;
    wi = r * t.c.y + i * t.c.x;
}
template <typename T>
PHAST_HD void tw3_lookup(const cx_t<T> *tab, unsigned bits, unsigned e, T &wr, T &wi) {
    tw3_combine<T>(tw3_fetch<T>(tab, bits, e), wr, wi);
}

// W = B * D^j for j = 0..P-1, handed to apply(integral_constant<int, j>, wr, wi) in ascending j.  G running values
// B D^i (i < G, built by doubling: 1 + 2 + .. + G/2 products and log2 G squarings of D) stepped by D^G: G + P - G products
// in all where a table of the powers D^j costs 2 P, and only G values are live (a P-entry table is 4 P registers of
// f64 -- next to 4 P registers of data that spills).  Rounding: at most log2 G + P/G - 1 products on top of the
// log2 G squarings, ~1.1e-16 each.
template <typename T, int P, int G, typename F> PHAST_HD void tw_progression(T br, T bi, T dr, T di, F &&apply) {
    static_assert(G >= 1 && P % G == 0 && (G & (G - 1)) == 0, "group size: a power of two dividing P");
    T ar[G], ai[G];
    ar[0] = br;
    ai[0] = bi;
    T qr = dr, qi = di;  // D^(2^s)
    static_for<0, ilog2_c(G)>([&](auto s) {
        constexpr int S = 1 << decltype(s)::value;
        static_for<0, S>([&](auto i) {
            constexpr int I = decltype(i)::value;
            ar[S + I] = ar[I] * qr - ai[I] * qi;
            ai[S + I] = ar[I] * qi + ai[I] * qr;
        });
        const T t = qr * qr - qi * qi;
        qi = (qr * qi) * (scalar_t<T>)2;
        qr = t;
    });
    static_for<0, P / G>([&](auto h) {
        constexpr int H = decltype(h)::value;
        static_for<0, G>([&](auto i) {
            constexpr int I = decltype(i)::value;
            apply(std::integral_constant<int, H * G + I>{}, ar[I], ai[I]);
        });
        if constexpr (H + 1 < P / G) {
            static_for<0, G>([&](auto i) {
                constexpr int I = decltype(i)::value;
                const T t = ar[I] * qr - ai[I] * qi;
                ai[I] = ar[I] * qi + ai[I] * qr;
                ar[I] = t;
            });
        }
    });
}

// splitmix64-based synthetic input shared with oracle/pho_fill_* 
__host__ __device__ inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline double uniform_pm1(unsigned long long seed, unsigned long long id, unsigned long long idx) {
    unsigned long long u = splitmix64(seed ^ (id << 40) ^ idx);
    return (double)(u >> 11) * 0x1.0p-52 - 1.0;
}

// ---- launch arguments of one tiled FFT pass (tile_fft.hpp) ----
// A pass does ROWS-point FFTs along a strided axis of a 2^L array, COLS adjacent columns per tile.
//   input  element (row n, column g): in  + xform*in_dist  + in_col(g)  + n*in_row_stride
//   output element (row k, column g): out + xform*out_dist + out_col(g) + k*out_row_stride
//   in_col(g)  = (g >> log_s_in) * in_hi_stride + ((g >> in_lo_bits) & (2^(log_s_in - in_lo_bits) - 1)) * in_mid_stride
//                + (g & (2^in_lo_bits - 1))
//   out_col(g) = (g & (2^out_lo_bits - 1)) * out_s1 + (g >> out_lo_bits) * out_s2
// In the caller's arrays every stride is a power of two (in_row_stride = 2^log_s_in, in_hi_stride = 2^(log_s_in + LR), no
// middle part).  In the planner's SCRATCH the strides are padded (plan.hpp: scratch_pad_bytes): rows a power of two apart
// land on the same HBM channels, and a 128-byte pad per row is worth 5-10 % of a pass (profiles/r03_pad_stride_probe.log).
struct TileArgs {
    const void *in_re;
    const void *in_im;   // unused when in_interleaved
    void *out_re;
    void *out_im;        // unused when out_interleaved
    const void *tw3;     // [3][1 << tw_bits] complex: W_{ROWS * 2^log_s_in}^e  (pre-twiddle passes)
    const void *twr;     // [2][32] complex: W_ROWS^e two-level (e = e1*32 + e0)
    unsigned long long in_dist;
    unsigned long long out_dist;
    unsigned long long in_row_stride;  // elements between consecutive rows of the input
    unsigned long long in_hi_stride;   // weight of the column bits at and above log_s_in
    unsigned long long in_mid_stride;  // weight of the column bits [in_lo_bits, log_s_in) (0 bits wide unless padded)
    unsigned long long out_s1;
    unsigned long long out_s2;
    unsigned long long out_row_stride;
    unsigned tiles_per_xform;
    unsigned tiles_total;
    unsigned log_s_in;
    unsigned in_lo_bits;       // contiguous low column bits of the input (= log_s_in unless the layout is padded)
    unsigned out_lo_bits;
    unsigned tw_bits;
    unsigned in_interleaved;   // 1: input is one array of (re, im) pairs (R2C deinterleave fused into the load); 2: read as (im, re)
    unsigned out_interleaved;  // 1: output is one array of (re, im) pairs (C2R interleave fused into the store);
                               // 2: pairs stored as (im, re) -- the swap-trick inverse (algorithms/dit.rs:297-300)
    // pre-twiddle exponent multiplier of column g:  lo = (g >> tw_shift) & tw_mask  (contiguous transforms: shift 0,
    // mask 2^log_s_in - 1; strided batches carry the batch index in the low bits of g, which takes no part)
    unsigned tw_shift;
    unsigned tw_mask;
    // column of tile t (strided batches that fill only 2^cb_bits * COLS of the 2^cs_bits columns of a row; 0 = all):
    //   g0 = ((t >> cb_bits) << cs_bits) | ((t & (2^cb_bits - 1)) << LC)
    unsigned cs_bits;
    unsigned cb_bits;
    // first pass of a strided batch with an INPUT twiddle (four-step split: x[j][c] *= W_{2^grid_log_n}^(j (col0 + c)) on
    // load; j = row << grid_row_shift | the column's upper digits): exponent of row p = p K + glo k, K = k << grid_row_shift
    unsigned grid_mode;
    unsigned grid_col0;
    unsigned grid_row_shift;
    unsigned grid_col_mask;
    double scale;              // 1/N on the last pass of an inverse transform, else 1
    unsigned long long *trace; // tools/trace_tile.py only: [workgroup][16] s_memtime stamps of the first tile's phases
};

// input side of a tile whose first column is g0 (the tile's COLS columns are adjacent low bits): element offset of
// (row 0, column g0) of transform `xform`
PHAST_HD size_t in_tile_base(const TileArgs &a, unsigned xform, unsigned g0) {
    const unsigned lo = g0 & ((1u << a.in_lo_bits) - 1u);
    const unsigned mid = (g0 >> a.in_lo_bits) & ((1u << (a.log_s_in - a.in_lo_bits)) - 1u);
    return (size_t)xform * a.in_dist + (size_t)(g0 >> a.log_s_in) * a.in_hi_stride + (size_t)mid * a.in_mid_stride + lo;
}

// Kernel arguments are fetched with scalar loads where the compiler first needs them -- behind every branch of a kernel's
// prologue another round trip to the kernarg segment before the first global load can be issued (three of them in the
// wave-tile kernel: 1.4 us of a 7 us pass).  Naming every field as an input of an empty asm statement at kernel entry
// makes all the scalar loads go out together.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void pin_tile_args(const TileArgs &a) {
    asm volatile("" ::"s"(a.in_re), "s"(a.in_im), "s"(a.out_re), "s"(a.out_im), "s"(a.tw3), "s"(a.twr), "s"(a.in_dist), "s"(a.out_dist),
                 "s"(a.out_s1), "s"(a.out_s2), "s"(a.out_row_stride), "s"(a.in_row_stride), "s"(a.in_hi_stride), "s"(a.in_mid_stride),
                 "s"(a.in_lo_bits));
    asm volatile("" ::"s"(a.tiles_per_xform), "s"(a.tiles_total), "s"(a.log_s_in), "s"(a.out_lo_bits), "s"(a.tw_bits),
                 "s"(a.in_interleaved), "s"(a.out_interleaved), "s"(a.tw_shift), "s"(a.tw_mask), "s"(a.cs_bits), "s"(a.cb_bits),
                 "s"(a.grid_mode), "s"(a.grid_col0), "s"(a.grid_row_shift), "s"(a.grid_col_mask), "s"(a.scale));
}
__device__ __forceinline__ void pin_scalars(unsigned x, unsigned y) { asm volatile("" ::"s"(x), "s"(y)); }
#else
PHAST_HD inline void pin_tile_args(const TileArgs &) {}
PHAST_HD inline void pin_scalars(unsigned, unsigned) {}
#endif

// host: make sure `kern` may be launched with `lds` bytes of dynamic LDS ON THE CURRENT DEVICE.  The limit is an
// attribute of the function per device; `cache` (one per kernel instantiation) remembers what was raised where, so the
// steady state issues no runtime call besides the launch itself (a launch sequence can be captured into a HIP graph), and
// the first launches of two host threads do not race (device_state.hpp).
inline hipError_t raise_lds_limit(PerDeviceLimit &cache, const void *kern, size_t lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipError_t err = hipSuccess;
    const int rc = cache.ensure(dev, lds, [&](size_t want) {
        err = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        return err == hipSuccess ? 0 : 1;
    });
    if (rc < 0) return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  // ordinal beyond the cache
    return err;
}

}  // namespace phast
