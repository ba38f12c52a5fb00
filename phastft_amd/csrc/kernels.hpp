// kernels.hpp -- host-callable launchers of every HIP kernel in libphastft_hip.so.
#pragma once

#include "common.hpp"

namespace phast {

// ---- small_fft.hip: batches of whole small transforms on chip, one pass (N <= 8192; row_fft.hpp) ----
struct SmallArgs {
    const void *in_re;
    const void *in_im;
    void *out_re;
    void *out_im;
    const void *tw;  // plan.hpp host_twr(N)  (unused for N = 1)
    unsigned long long in_dist;
    unsigned long long out_dist;
    unsigned log_n;
    unsigned batch;
    unsigned in_interleaved;
    unsigned out_interleaved;  // 1 = (re, im), 2 = (im, re)
    double scale;
    // real transforms of 2N points around this N-point core (row_fft.hpp: RowArgs::real_mode): 0 none,
    // 1 R2C untangle as the epilogue, 2 C2R preprocess as the prologue; rtw3 = W_{2N} tables
    unsigned real_mode;
    unsigned rtw_bits;
    const void *rtw3;
};
constexpr unsigned kSmallMaxLog = 13;
constexpr unsigned kTwinMinLog = 12;  // the shortest length that also exists as a multi-pass plan (64 x 64: Planner::twin)
template <typename T>
hipError_t launch_small_fft(const SmallArgs &a, hipStream_t stream, hipEvent_t ev_start = nullptr,
                            hipEvent_t ev_stop = nullptr);

// ---- bitrev.hip: in-place bit-reversal permutation of `batch` arrays of 2^log_n elements ----
template <typename T> hipError_t launch_bitrev(T *data, unsigned log_n, size_t batch, size_t dist, hipStream_t stream);

// ---- r2c.hip ----
struct UntangleArgs {
    void *re;  // [half + 1], in place
    void *im;
    const void *tw3;  // [3][1 << tw_bits]: W_N^e, N = 2*half
    unsigned long long dist;
    unsigned half;
    unsigned tw_bits;
    unsigned batch;
};
template <typename T>
hipError_t launch_untangle(const UntangleArgs &a, hipStream_t stream, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

struct C2rPreArgs {
    const void *in_re;  // [half + 1]
    const void *in_im;
    void *z_re;  // [half]
    void *z_im;
    const void *tw3;
    unsigned long long in_dist;
    unsigned long long z_dist;
    unsigned half;
    unsigned tw_bits;
    unsigned batch;
};
template <typename T>
hipError_t launch_c2r_preprocess(const C2rPreArgs &a, hipStream_t stream, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// ---- twiddle.hip: block (r, c) *= W_N^((row0 + r) * (col0 + c)), N = 2^log_n (four-step inter-factor twiddle) ----
struct TwiddleGridArgs {
    void *re;  // [rows][row_pitch], in place
    void *im;
    const void *tw3;  // [3][1 << tw_bits]: W_N^e
    unsigned long long rows, cols, row_pitch, row0, col0;
    unsigned log_n;
    unsigned tw_bits;
};
template <typename T> hipError_t launch_twiddle_grid(const TwiddleGridArgs &a, hipStream_t stream);

// ---- fill.hip ----
template <typename T>
hipError_t launch_fill(T *re, T *im, size_t n, size_t batch, size_t dist, unsigned long long seed,
                       unsigned long long first_id, hipStream_t stream);
template <typename T>
hipError_t launch_digest(const T *re, const T *im, size_t n, size_t batch, size_t dist, size_t probe, double *digest,
                         hipStream_t stream);
// ---- complex_nums.hip: Complex<T> <-> planes (complex_nums.rs:11-56) ----
template <typename T> hipError_t launch_deinterleave(const T *in, T *a, T *b, size_t pairs, hipStream_t stream);
template <typename T> hipError_t launch_combine(const T *re, const T *im, T *out, size_t n, hipStream_t stream);
// harness: this box's streaming ceilings (probe.hip) -- out_gbps = {read, write, copy}
hipError_t stream_probe(const void *d_a, void *d_b, size_t bytes, int reps, int cus, double *out_gbps, hipStream_t stream);

}  // namespace phast
