// bitrev.hip -- in-place bit-reversal permutation on gfx950, bit-exact.
//
// GPU counterpart of algorithms/bravo.rs (bit_rev_bravo_f32/f64, bravo.rs:303-324).  Same index
// split as CO-BRAVO (bravo.rs:186-219): an index is [u | t | v] with |u| = |v| = beta bits; tile t is
// the B x B block {u*2^(L-beta) + t*B + v}; bit reversal sends element (u, t, v) to (rev v, rev t, rev u),
// i.e. tile t <-> tile rev(t) with the block transposed and both coordinates bit-reversed.  One
// workgroup stages tile t and tile rev(t) in LDS (the reference's stack buffer) and writes each into
// the other's place: every global access is a run of B contiguous elements (256 B), and the transpose
// happens in LDS with a +1 padded row so neither side has bank conflicts.  B = 32 for 8-byte and 64 for
// 4-byte elements -- the reference's TILE_SIDE_F64 / TILE_SIDE_F32 (bravo.rs:19-20).  Pure data
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

static int bitrev_variant() {  // tuning hook: PHAST_BITREV_VARIANT=0 (default) | 1 | 2 | 3 (tools/sweep_bitrev.py)
    const char *e = getenv("PHAST_BITREV_VARIANT");
    return e ? atoi(e) : 0;
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

template <> hipError_t launch_bitrev<double>(double *data, unsigned log_n, size_t batch, size_t dist, hipStream_t s) {
    auto *p = reinterpret_cast<unsigned long long *>(data);
    const int v = bitrev_variant();
    if (v == 1 && log_n >= 12) return launch_bitrev_u<unsigned long long, 6, 512>(p, log_n, batch, dist, s);
    if (v == 2 && log_n >= 12) return launch_bitrev_u<unsigned long long, 6, 1024>(p, log_n, batch, dist, s);
    if (v == 3) return launch_bitrev_u<unsigned long long, 5, 512>(p, log_n, batch, dist, s);
    return launch_bitrev_u<unsigned long long, 5, 256>(p, log_n, batch, dist, s);
}
template <> hipError_t launch_bitrev<float>(float *data, unsigned log_n, size_t batch, size_t dist, hipStream_t s) {
    auto *p = reinterpret_cast<unsigned *>(data);
    const int v = bitrev_variant();
    if (v == 1) return launch_bitrev_u<unsigned, 6, 256>(p, log_n, batch, dist, s);
    if (v == 2) return launch_bitrev_u<unsigned, 6, 1024>(p, log_n, batch, dist, s);
    if (v == 3 && log_n >= 14) return launch_bitrev_u<unsigned, 7, 1024>(p, log_n, batch, dist, s);
    return launch_bitrev_u<unsigned, 6, 512>(p, log_n, batch, dist, s);  // measured best (profiles/r01_sweep_bitrev.log)
}

}  // namespace phast
