// bitrev.hip -- in-place bit-reversal permutation on gfx950, bit-exact.
//
// GPU counterpart of algorithms/bravo.rs (bit_rev_bravo_f32/f64, bravo.rs:303-324).  Same index
// split as CO-BRAVO (bravo.rs:186-219): an index is [u | t | v] with |u| = |v| = beta bits; tile t is
// the B x B block {u*2^(L-beta) + t*B + v}; bit reversal sends element (u, t, v) to (rev v, rev t, rev u),
// i.e. tile t <-> tile rev(t) with the block transposed and both coordinates bit-reversed.  One
// workgroup stages tile t and tile rev(t) in LDS (the reference's stack buffer) and writes each into
// the other's place: every global access is a run of B contiguous elements, and the transpose happens in LDS
// with a +1 padded row so neither side has bank conflicts.  B = 64 for both element sizes (the reference's
// TILE_SIDE_F32; its TILE_SIDE_F64 = 32, bravo.rs:19-20, is what the small sizes below 2^12 still use).  Pure data
// movement: elements travel as integers, so NaN payloads and signed zeros survive.
#include <cstdlib>

#include "kernels.hpp"

namespace phast {

template <typename U> __global__ void __launch_bounds__(256) bitrev_simple_kernel(U *data, unsigned log_n, size_t dist) {
    U *x = data + (size_t)blockIdx.y * dist;
    const unsigned n = 1u << log_n;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned j = log_n ? (__brev(i) >> (32u - log_n)) : 0u;
        if (i < j) {
            const U a = x[i], b = x[j];
            x[i] = b;
            x[j] = a;
        }
    }
}

template <typename U, int BETA, int NTH>
__global__ void __launch_bounds__(NTH) bitrev_tiled_kernel(U *data, unsigned log_n, size_t dist, unsigned tiles) {
    constexpr int B = 1 << BETA;
    __shared__ U sa[B][B + 1];
    __shared__ U sb[B][B + 1];
    const unsigned tile_bits = log_n - 2 * BETA;
    const unsigned xf = blockIdx.x / tiles;
    const unsigned t = blockIdx.x - xf * tiles;
    const unsigned tr = tile_bits ? (__brev(t) >> (32u - tile_bits)) : 0u;
    if (t > tr) return;  // the pair is handled by the block of the smaller index
    U *x = data + (size_t)xf * dist;
    const unsigned ustride_log = log_n - BETA;

    for (int idx = threadIdx.x; idx < B * B; idx += NTH) {
        const unsigned u = idx >> BETA, v = idx & (B - 1);
        sa[u][v] = x[((size_t)u << ustride_log) + ((size_t)t << BETA) + v];
        if (t != tr) sb[u][v] = x[((size_t)u << ustride_log) + ((size_t)tr << BETA) + v];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < B * B; idx += NTH) {
        const unsigned u = idx >> BETA, v = idx & (B - 1);
        const unsigned ru = __brev(u) >> (32 - BETA), rv = __brev(v) >> (32 - BETA);
        if (t != tr) {
            x[((size_t)u << ustride_log) + ((size_t)t << BETA) + v] = sb[rv][ru];
            x[((size_t)u << ustride_log) + ((size_t)tr << BETA) + v] = sa[rv][ru];
        } else {
            x[((size_t)u << ustride_log) + ((size_t)t << BETA) + v] = sa[rv][ru];
        }
    }
}

// The tile pairs {t, rev t} of an array of 2^m tiles, ENUMERATED (round 4) -- every pair exactly once, by construction.
// Until now the persistent kernels walked all 2^m tiles grid-stride and skipped those with t > rev(t); which of a
// workgroup's items survive that test is decided by index bits the workgroup's items SHARE, so under the spread order of
// the second generation half of the workgroups kept all of their items and the other half none (in linear order a
// workgroup kept anything between none and all): half the chip idle or a long tail.
// With t = (a | mid | rev_h(b)), h = floor(m / 2) bits each for a and b, the pairs are the (a, b) with a <= b (times the
// middle bit when m is odd); the triangle is folded into an S/2 x (S + 1) rectangle (rows i and S - 1 - i together have
// S + 1 elements), so position p -> pair needs one division and every workgroup gets the same number of pairs.
struct PairEnum {
    unsigned m, h, mb, S;
    unsigned long long pairs;  // per array
    __host__ __device__ explicit PairEnum(unsigned tile_bits) : m(tile_bits), h(tile_bits / 2), mb(tile_bits - 2 * (tile_bits / 2)), S(1u << (tile_bits / 2)) {
        pairs = h == 0 ? (1ull << m) : ((unsigned long long)(S / 2) * (S + 1)) << mb;
    }
    __device__ unsigned tile(unsigned p) const {
        if (h == 0) return p;
        const unsigned mid = p & ((1u << mb) - 1u), q = p >> mb, i = q / (S + 1u), j = q - i * (S + 1u);
        unsigned a, b;
        if (j < S - i) {
            a = i;
            b = i + j;
        } else {
            a = S - 1u - i;
            b = a + (j - (S - i));
        }
        return (a << (m - h)) | (mid << h) | (__brev(b) >> (32u - h));
    }
};

// Persistent form: a workgroup walks the tile pairs (t <= rev t) grid-stride, with the NEXT pair's rows already in
// flight (registers) while the current pair goes through LDS -- the loads of pair i+1 overlap the transposing
// reads and the stores of pair i, and the half of the index space that would exit at once never becomes a
// workgroup.
template <typename U, int BETA, int NTH>
__global__ void __launch_bounds__(NTH) bitrev_persistent_kernel(U *data, unsigned log_n, size_t dist, unsigned tiles,
                                                                unsigned long long total) {
    constexpr int B = 1 << BETA, PER = B * B / NTH;
    __shared__ U sa[B][B + 1];
    __shared__ U sb[B][B + 1];
    const unsigned tile_bits = log_n - 2 * BETA;
    const unsigned ustride_log = log_n - BETA;
    auto rev_t = [&](unsigned t) { return tile_bits ? (__brev(t) >> (32u - tile_bits)) : 0u; };
    const PairEnum pe(tile_bits);  // `total` = arrays x pe.pairs: work item w = pair w % pe.pairs of array w / pe.pairs
    (void)tiles;
    auto advance = [&](unsigned long long w) { return w; };
    U ra[PER], rb[PER];
    auto load = [&](unsigned long long w) {
        const unsigned t = pe.tile((unsigned)(w % pe.pairs)), tr = rev_t(t);
        U *x = data + (size_t)(w / pe.pairs) * dist;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> BETA, v = idx & (B - 1);
            ra[i] = x[((size_t)u << ustride_log) + ((size_t)t << BETA) + v];
            rb[i] = x[((size_t)u << ustride_log) + ((size_t)tr << BETA) + v];
        }
    };
    unsigned long long w = advance(blockIdx.x);
    if (w < total) load(w);
    while (w < total) {
        const unsigned t = pe.tile((unsigned)(w % pe.pairs)), tr = rev_t(t);
        U *x = data + (size_t)(w / pe.pairs) * dist;
        __syncthreads();  // the previous pair's readers are done
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            sa[idx >> BETA][idx & (B - 1)] = ra[i];
            sb[idx >> BETA][idx & (B - 1)] = rb[i];
        }
        __syncthreads();
        U oa[PER], ob[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> BETA, v = idx & (B - 1);
            const unsigned ru = __brev(u) >> (32 - BETA), rv = __brev(v) >> (32 - BETA);
            oa[i] = sb[rv][ru];  // what tile t receives
            ob[i] = sa[rv][ru];  // what tile rev t receives
        }
        const unsigned long long wn = advance(w + gridDim.x);
        if (wn < total) load(wn);  // in flight during the stores below
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> BETA, v = idx & (B - 1);
            x[((size_t)u << ustride_log) + ((size_t)t << BETA) + v] = oa[i];
            if (t != tr) x[((size_t)u << ustride_log) + ((size_t)tr << BETA) + v] = ob[i];
        }
        w = wn;
    }
}

// Second generation of the persistent kernel (round 2):
//   * 16 bytes per lane on the global side (two adjacent elements: a 64-element f64 row is 32 lanes x 16 B), non-temporal
//     -- every byte is touched once;
//   * SPREAD tile order: work item w -> tile t with the bits of w dealt alternately to the LOW and the HIGH end of t.  In
//     linear order the tiles in flight at one moment are consecutive t, so their partners rev(t) differ only in their
//     HIGH bits: power-of-two strides of megabytes that pile onto few HBM channels / DRAM pages.  Dealing the bits to
//     both ends makes both the t side and the rev(t) side "32 neighbours x 32 far apart".
template <typename U> struct Vec2;  // clang vector types: what the nontemporal builtins accept
template <> struct Vec2<unsigned long long> { typedef unsigned long long type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<unsigned> { typedef unsigned type __attribute__((ext_vector_type(2))); };

template <typename U, int BETA, int NTH, bool SPREAD>
__global__ void __launch_bounds__(NTH) bitrev_persistent2_kernel(U *data, unsigned log_n, size_t dist, unsigned tiles,
                                                                 unsigned long long total) {
    using V2 = typename Vec2<U>::type;
    constexpr int B = 1 << BETA, PER = B * B / (2 * NTH);  // element PAIRS per thread per tile
    static_assert(PER >= 1, "tile too small for this workgroup");
    __shared__ U sa[B][B + 1];
    __shared__ U sb[B][B + 1];
    const unsigned tile_bits = log_n - 2 * BETA;
    const unsigned ustride_log = log_n - BETA;
    auto rev_t = [&](unsigned t) { return tile_bits ? (__brev(t) >> (32u - tile_bits)) : 0u; };
    // (round 4: the pairs are enumerated, PairEnum above; the SPREAD order of round 2 -- which "bought nothing" by its own
    // measurement and left half of the workgroups without a single pair -- is gone, the template flag is kept for the name)
    const PairEnum pe(tile_bits);
    (void)tiles;
    auto tile_of = [&](unsigned p) { return pe.tile(p); };
    auto advance = [&](unsigned long long w) { return w; };
    V2 ra[PER], rb[PER];
    auto load = [&](unsigned long long w) {
        const unsigned t = tile_of((unsigned)(w % pe.pairs)), tr = rev_t(t);
        const U *x = data + (size_t)(w / pe.pairs) * dist;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            ra[i] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(x + ((size_t)u << ustride_log) + ((size_t)t << BETA) + v));
            rb[i] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(x + ((size_t)u << ustride_log) + ((size_t)tr << BETA) + v));
        }
    };
    unsigned long long w = advance(blockIdx.x);
    if (w < total) load(w);
    while (w < total) {
        const unsigned t = tile_of((unsigned)(w % pe.pairs)), tr = rev_t(t);
        U *x = data + (size_t)(w / pe.pairs) * dist;
        __syncthreads();  // the previous pair's readers are done
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            sa[u][v] = ra[i].x;
            sa[u][v + 1] = ra[i].y;
            sb[u][v] = rb[i].x;
            sb[u][v + 1] = rb[i].y;
        }
        __syncthreads();
        V2 oa[PER], ob[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            const unsigned ru = __brev(u) >> (32 - BETA), rv = __brev(v) >> (32 - BETA);  // rev(v + 1) = rv + B/2
            oa[i].x = sb[rv][ru];
            oa[i].y = sb[rv + B / 2][ru];
            ob[i].x = sa[rv][ru];
            ob[i].y = sa[rv + B / 2][ru];
        }
        const unsigned long long wn = advance(w + gridDim.x);
        if (wn < total) load(wn);  // in flight during the stores below
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            __builtin_nontemporal_store(oa[i], reinterpret_cast<V2 *>(x + ((size_t)u << ustride_log) + ((size_t)t << BETA) + v));
            if (t != tr)
                __builtin_nontemporal_store(ob[i], reinterpret_cast<V2 *>(x + ((size_t)u << ustride_log) + ((size_t)tr << BETA) + v));
        }
        w = wn;
    }
}

// Third generation (round 2, late): tiles of 128 x 128 elements -- rows of 1 KiB in f64 where the 64 x 64 tiles have 512 B.
// A large array loses its bandwidth to the permutation's access pattern (profiles/r02_bitrev_stride_probe.log: the same
// kernel runs at 5.8 TB/s on contiguous tiles and 3.9 on rows 8 MiB apart -- every row of a tile on another page): twice
// the row halves the number of rows per byte.  Two 128 KiB tiles do not fit the LDS together, so the pair lives in
// REGISTERS (1024 threads x 2 x 8 x 16 B) and goes through ONE padded LDS buffer one tile at a time; the next pair's
// loads are issued as soon as the second tile has left the registers, under the last transposition and its stores.
template <typename U, int BETA, int NTH>
__global__ void __launch_bounds__(NTH) bitrev_persistent3_kernel(U *data, unsigned log_n, size_t dist, unsigned tiles,
                                                                 unsigned long long total) {
    using V2 = typename Vec2<U>::type;
    constexpr int B = 1 << BETA, PER = B * B / (2 * NTH);  // element PAIRS per thread per tile
    static_assert(PER >= 1, "tile too small for this workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    U(*s)[B + 1] = reinterpret_cast<U(*)[B + 1]>(smem_raw);
    const unsigned tile_bits = log_n - 2 * BETA;
    const unsigned ustride_log = log_n - BETA;
    auto rev_t = [&](unsigned t) { return tile_bits ? (__brev(t) >> (32u - tile_bits)) : 0u; };
    const PairEnum pe(tile_bits);  // the pairs enumerated: see PairEnum
    (void)tiles;
    auto advance = [&](unsigned long long w) { return w; };
    V2 ra[PER], rb[PER];
    auto load = [&](unsigned long long w) {
        const unsigned t = pe.tile((unsigned)(w % pe.pairs)), tr = rev_t(t);
        const U *x = data + (size_t)(w / pe.pairs) * dist;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            ra[i] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(x + ((size_t)u << ustride_log) + ((size_t)t << BETA) + v));
        }
        if (t != tr) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = i * NTH + threadIdx.x;
                const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
                rb[i] = __builtin_nontemporal_load(reinterpret_cast<const V2 *>(x + ((size_t)u << ustride_log) + ((size_t)tr << BETA) + v));
            }
        }
    };
    auto to_lds = [&](const V2(&r)[PER]) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            s[u][v] = r[i].x;
            s[u][v + 1] = r[i].y;
        }
    };
    auto transposed_to = [&](U *x, unsigned tile) {  // element (u, v) of `tile` <- s[rev v][rev u]
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * NTH + threadIdx.x;
            const unsigned u = idx >> (BETA - 1), v = (idx & (B / 2 - 1)) * 2;
            const unsigned ru = __brev(u) >> (32 - BETA), rv = __brev(v) >> (32 - BETA);  // rev(v + 1) = rv + B/2
            V2 o;
            o.x = s[rv][ru];
            o.y = s[rv + B / 2][ru];
            __builtin_nontemporal_store(o, reinterpret_cast<V2 *>(x + ((size_t)u << ustride_log) + ((size_t)tile << BETA) + v));
        }
    };
    unsigned long long w = advance(blockIdx.x);
    if (w < total) load(w);
    while (w < total) {
        const unsigned t = pe.tile((unsigned)(w % pe.pairs)), tr = rev_t(t);
        U *x = data + (size_t)(w / pe.pairs) * dist;
        const unsigned long long wn = advance(w + gridDim.x);
        __syncthreads();  // the previous pair's readers are done
        if (t != tr) {
            to_lds(rb);
            __syncthreads();
            transposed_to(x, t);  // tile t receives tile rev(t), transposed and reversed
            __syncthreads();
        }
        to_lds(ra);
        if (wn < total) load(wn);  // both tiles have left the registers: the next pair's loads fly under what follows
        __syncthreads();
        transposed_to(x, tr);
        w = wn;
    }
}

template <typename U, int BETA, int NTH>
static hipError_t launch_bitrev_persistent3(U *data, unsigned log_n, size_t batch, size_t dist, hipStream_t stream) {
    constexpr int B = 1 << BETA;
    const unsigned tiles = 1u << (log_n - 2 * BETA);
    const unsigned long long total = PairEnum(log_n - 2 * BETA).pairs * batch;  // work items: the enumerated tile pairs
    unsigned long long grid = 256ull;  // one workgroup per CU: the LDS buffer is 129 KiB
    if (grid > total) grid = total;
    const size_t lds = sizeof(U) * B * (B + 1);
    auto kern = bitrev_persistent3_kernel<U, BETA, NTH>;
    static PerDeviceLimit lds_limit;
    if (hipError_t e = raise_lds_limit(lds_limit, reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTH), lds, stream, data, log_n, dist, tiles, total);
    return hipGetLastError();
}

template <typename U, int BETA, int NTH, bool SPREAD>
static hipError_t launch_bitrev_persistent2(U *data, unsigned log_n, size_t batch, size_t dist, hipStream_t stream,
                                            unsigned wg_per_cu) {
    const unsigned tiles = 1u << (log_n - 2 * BETA);
    const unsigned long long total = PairEnum(log_n - 2 * BETA).pairs * batch;  // work items: the enumerated tile pairs
    unsigned long long grid = 256ull * wg_per_cu;
    if (grid > total) grid = total;
    hipLaunchKernelGGL((bitrev_persistent2_kernel<U, BETA, NTH, SPREAD>), dim3((unsigned)grid), dim3(NTH), 0, stream, data,
                       log_n, dist, tiles, total);
    return hipGetLastError();
}

template <typename U, int BETA, int NTH>
static hipError_t launch_bitrev_persistent(U *data, unsigned log_n, size_t batch, size_t dist, hipStream_t stream,
                                           unsigned wg_per_cu) {
    const unsigned tiles = 1u << (log_n - 2 * BETA);
    const unsigned long long total = PairEnum(log_n - 2 * BETA).pairs * batch;  // work items: the enumerated tile pairs
    unsigned long long grid = 256ull * wg_per_cu;
    if (grid > total) grid = total;
    hipLaunchKernelGGL((bitrev_persistent_kernel<U, BETA, NTH>), dim3((unsigned)grid), dim3(NTH), 0, stream, data, log_n,
                       dist, tiles, total);
    return hipGetLastError();
}

template <typename U, int BETA, int NTH>
static hipError_t launch_bitrev_u(U *data, unsigned log_n, size_t batch, size_t dist, hipStream_t stream) {
    if (log_n == 0 || batch == 0) return hipSuccess;
    if (log_n < 2 * BETA) {
        const unsigned n = 1u << log_n;
        const unsigned gx = (n + 255u) / 256u;
        for (size_t b0 = 0; b0 < batch; b0 += 65535) {
            const unsigned gy = (unsigned)((batch - b0) < 65535 ? (batch - b0) : 65535);
            hipLaunchKernelGGL((bitrev_simple_kernel<U>), dim3(gx, gy), dim3(256), 0, stream, data + b0 * dist, log_n,
                               dist);
        }
        return hipGetLastError();
    }
    const unsigned tiles = 1u << (log_n - 2 * BETA);
    // one grid per chunk of transforms so that gridDim.x stays below 2^31
    const size_t per_launch = (size_t)0x40000000u / tiles ? (size_t)0x40000000u / tiles : 1;
    for (size_t b0 = 0; b0 < batch; b0 += per_launch) {
        const size_t nb = (batch - b0) < per_launch ? (batch - b0) : per_launch;
        hipLaunchKernelGGL((bitrev_tiled_kernel<U, BETA, NTH>), dim3((unsigned)(nb * tiles)), dim3(NTH), 0, stream,
                           data + b0 * dist, log_n, dist, tiles);
    }
    return hipGetLastError();
}

// Which generation runs where (profiles/r01_sweep_bitrev.log, r02_sweep_bitrev.log; the other generations of those sweeps --
// 32 x 32 pairs per workgroup at every size, 1024-thread forms, the plain tile order of the 16-byte generation, a 128 x 128 u32
// form that needed scratch memory -- lost everywhere and were removed in round 4 together with the PHAST_BITREV_VARIANT hook that selected them):
//   N < 2^12             one pair of tiles per workgroup
//   default              persistent workgroups, 64 x 64 tiles (512-byte rows for f64, 256-byte for f32), 512 threads, 4 per CU:
//                        3.4-3.7 TB/s at 2^26..2^30 f64, 3.4-4.6 f32 -- for 4-byte elements it stays ahead at every size
//   f64, >= 2^25 points  past the 256 MiB Infinity Cache the 16-byte-per-lane, non-temporal generation is 4-10 % faster
//                        (2^26: 3.9 vs 3.6 TB/s)
//   f64, N >= 2^27       128 x 128 tiles held in registers (1 KiB rows): 2^27 3.4 -> 4.1 TB/s, 2^28 3.9 -> 4.0, 2^30 3.8 -> 4.0;
//                        below that there are too few tiles per CU for its one workgroup per CU (2^26: 3.7 vs 3.9)
#ifndef PHAST_BITREV_P3_MIN_LOG   // tuning (tools/ab.sh with build(extra=...)): thresholds of the generations
#define PHAST_BITREV_P3_MIN_LOG 27
#endif
#ifndef PHAST_BITREV_P2_MIN_LOG
#define PHAST_BITREV_P2_MIN_LOG 25
#endif
#ifndef PHAST_BITREV_F32_P2_MIN_LOG
#define PHAST_BITREV_F32_P2_MIN_LOG 99
#endif
template <> hipError_t launch_bitrev<double>(double *data, unsigned log_n, size_t batch, size_t dist, hipStream_t s) {
    auto *p = reinterpret_cast<unsigned long long *>(data);
    if (log_n < 12) return launch_bitrev_u<unsigned long long, 5, 256>(p, log_n, batch, dist, s);
    const bool even = (dist & 1) == 0 && (reinterpret_cast<size_t>(data) & 15) == 0;  // 16-byte accesses need it
    if (even && log_n >= PHAST_BITREV_P3_MIN_LOG) return launch_bitrev_persistent3<unsigned long long, 7, 1024>(p, log_n, batch, dist, s);
    if (even && (((size_t)batch << log_n) >= ((size_t)1 << PHAST_BITREV_P2_MIN_LOG)))
        return launch_bitrev_persistent2<unsigned long long, 6, 256, true>(p, log_n, batch, dist, s, 4);
    return launch_bitrev_persistent<unsigned long long, 6, 512>(p, log_n, batch, dist, s, 4);
}
template <> hipError_t launch_bitrev<float>(float *data, unsigned log_n, size_t batch, size_t dist, hipStream_t s) {
    auto *p = reinterpret_cast<unsigned *>(data);
    if (log_n < 12) return launch_bitrev_u<unsigned, 6, 512>(p, log_n, batch, dist, s);
    const bool even = (dist & 3) == 0 && (reinterpret_cast<size_t>(data) & 15) == 0;
    // (the default of 99 means "never": the comparison is guarded -- a 64-bit shift by 99 is undefined)
    if (even && PHAST_BITREV_F32_P2_MIN_LOG < 64 && (((size_t)batch << log_n) >= ((size_t)1 << (PHAST_BITREV_F32_P2_MIN_LOG & 63))))
        return launch_bitrev_persistent2<unsigned, 6, 256, true>(p, log_n, batch, dist, s, 4);
    return launch_bitrev_persistent<unsigned, 6, 512>(p, log_n, batch, dist, s, 4);
}

}  // namespace phast
