/*
 * phastft_oracle.h -- CPU restatement of QuState/PhastFT 0.3.0's planar DIT FFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product path (phastft_amd/ + libphastft_hip.so) never
 * links, imports or calls it.
 *
 * Why a restatement: the reference is Rust and there is no cargo/rustc in the build image, so the
 * reference itself cannot be compiled or imported.  Every function below
 * cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Pinning status: pinned against the reference's own in-module known-answer tests (bit reversal
 * exact for n=2..23, bravo.rs:373-407; analytic R2C answers, r2c.rs:1235-1386; C2C ramp vs an
 * independent FFT at abs 0.01, lib.rs:298-338; codelet == staged kernels, codelets.rs:522-698;
 * round trips, lib.rs:381-425) -- see tests/test_oracle_pin.py.  Bit-level parity with a real
 * PhastFT binary is UNPINNED (no Rust toolchain; fearless_simd's mul_add is assumed fused, as it
 * is on every AVX2+FMA / NEON dispatch level).
 */
#ifndef PHASTFT_ORACLE_H
#define PHASTFT_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* planner.rs:10-16 */
#define PHO_FORWARD 1
#define PHO_REVERSE (-1)

/* status codes: one per reference assert (the Rust code panics; C returns) */
#define PHO_OK 0
#define PHO_ERR_NOT_POW2 1       /* planner.rs:66, dit.rs:285 */
#define PHO_ERR_LEN_MISMATCH 2   /* dit.rs:284 */
#define PHO_ERR_PLANNER_SIZE 3   /* dit.rs:289 */
#define PHO_ERR_R2C_N 4          /* planner.rs:195  "n must be a power of 2 >= 4" */
#define PHO_ERR_R2C_INPUT_LEN 5  /* r2c.rs:543     "input length must match planner size" */
#define PHO_ERR_R2C_OUT_RE_LEN 6 /* r2c.rs:544-548 "output_re must have length N/2 + 1" */
#define PHO_ERR_R2C_OUT_IM_LEN 7 /* r2c.rs:549-553 "output_im must have length N/2 + 1" */
#define PHO_ERR_C2R_OUTPUT_LEN 8 /* r2c.rs:750     "output length must match planner size" */
#define PHO_ERR_C2R_IN_RE_LEN 9  /* r2c.rs:751-755 "input_re must have length N/2 + 1" */
#define PHO_ERR_C2R_IN_IM_LEN 10 /* r2c.rs:756-760 "input_im must have length N/2 + 1" */
#define PHO_ERR_C2R_SCRATCH_RE 11 /* r2c.rs:761    "scratch_re must have length N/2" */
#define PHO_ERR_C2R_SCRATCH_IM 12 /* r2c.rs:762    "scratch_im must have length N/2" */
#define PHO_ERR_ALLOC 13

/* options.rs:10-24 */
typedef struct pho_options {
    int multithreaded_bit_reversal;       /* ignored: default features are empty (Cargo.toml:32) */
    size_t smallest_parallel_chunk_size;  /* ignored likewise */
} pho_options;

/* options.rs:38-43; returns PHO_ERR_NOT_POW2 for input_size == 0 (ilog2 panics) */
int pho_options_guess(size_t input_size, pho_options *out);

typedef struct pho_planner_dit64 pho_planner_dit64;
typedef struct pho_planner_dit32 pho_planner_dit32;
typedef struct pho_planner_r2c64 pho_planner_r2c64;
typedef struct pho_planner_r2c32 pho_planner_r2c32;

/* planner.rs:55-100 (PlannerMode is ignored by the reference: planner.rs:65 `_mode`) */
int pho_planner_dit64_new(size_t n, pho_planner_dit64 **out);
int pho_planner_dit32_new(size_t n, pho_planner_dit32 **out);
void pho_planner_dit64_free(pho_planner_dit64 *p);
void pho_planner_dit32_free(pho_planner_dit32 *p);
/* test access to the per-stage tables: stage >= 6; returns dist (= 1<<stage) or 0 */
size_t pho_planner_dit64_stage(const pho_planner_dit64 *p, size_t stage, const double **re, const double **im);
size_t pho_planner_dit32_stage(const pho_planner_dit32 *p, size_t stage, const float **re, const float **im);

/* planner.rs:120-212 */
int pho_planner_r2c64_new(size_t n, pho_planner_r2c64 **out);
int pho_planner_r2c32_new(size_t n, pho_planner_r2c32 **out);
void pho_planner_r2c64_free(pho_planner_r2c64 *p);
void pho_planner_r2c32_free(pho_planner_r2c32 *p);
void pho_planner_r2c64_twiddles(const pho_planner_r2c64 *p, const double **w_re, const double **w_im);
void pho_planner_r2c32_twiddles(const pho_planner_r2c32 *p, const float **w_re, const float **w_im);

/* lib.rs:180-226 / algorithms/dit.rs:263-401 */
int pho_fft_64_dit(double *reals, size_t re_len, double *imags, size_t im_len, int direction);
int pho_fft_32_dit(float *reals, size_t re_len, float *imags, size_t im_len, int direction);
int pho_fft_64_dit_with_planner(double *reals, size_t re_len, double *imags, size_t im_len, int direction,
                                const pho_planner_dit64 *planner);
int pho_fft_32_dit_with_planner(float *reals, size_t re_len, float *imags, size_t im_len, int direction,
                                const pho_planner_dit32 *planner);

/* the same with the crate's optional `parallel` feature emulated (Cargo.toml:34, parallel.rs:6-25): two-way
 * join of the bit reversals when log2 N >= 16, recursive two-way join while size > 16384, the stages spanning
 * both halves serial.  Results are bit-identical to the single-threaded entry points. */
int pho_fft_64_dit_with_planner_parallel(double *reals, size_t re_len, double *imags, size_t im_len, int direction,
                                         const pho_planner_dit64 *planner);
int pho_fft_32_dit_with_planner_parallel(float *reals, size_t re_len, float *imags, size_t im_len, int direction,
                                         const pho_planner_dit32 *planner);

/* complex_nums.rs:11-56 (bench-internals surface): deinterleave = `input.chunks_exact(2).map(|c| (c[0], c[1])).unzip()` -- an odd
 * last element is dropped; out_a / out_b hold len / 2 elements.  combine_re_im: out[k] = Complex::new(reals[k], imags[k]). */
void pho_deinterleave_f64(const double *input, size_t len, double *out_a, double *out_b);
void pho_deinterleave_f32(const float *input, size_t len, float *out_a, float *out_b);
void pho_combine_re_im_f64(const double *reals, const double *imags, size_t n, double *out);
void pho_combine_re_im_f32(const float *reals, const float *imags, size_t n, float *out);

/* algorithms/bravo.rs:303-345 (bench-internals surface) */
void pho_bit_rev_f64(double *data, unsigned log_n);
void pho_bit_rev_f32(float *data, unsigned log_n);
/* the three regimes individually, for the pin tests (bravo.rs:77-251) */
void pho_bit_rev_scalar_f64(double *data, unsigned log_n);
void pho_bit_rev_bravo_f64(double *data, unsigned log_n);   /* needs 2^log_n >= 64 */
void pho_bit_rev_cobravo_f64(double *data, unsigned log_n); /* needs log_n >= 10 */
void pho_bit_rev_scalar_f32(float *data, unsigned log_n);
void pho_bit_rev_bravo_f32(float *data, unsigned log_n);    /* needs 2^log_n >= 64 */
void pho_bit_rev_cobravo_f32(float *data, unsigned log_n);  /* needs log_n >= 12 */

/* kernels exposed for the codelet==staged pin test (codelets.rs:522-698) */
void pho_codelet_16_f64(double *re, double *im, size_t len);
void pho_codelet_32_f32(float *re, float *im, size_t len);
/* one staged radix-2 DIT stage over the whole slice (kernels/dit.rs); stage < 6 only */
void pho_stage_const_f64(double *re, double *im, size_t len, unsigned stage);
void pho_stage_const_f32(float *re, float *im, size_t len, unsigned stage);

/* algorithms/r2c.rs:521-662 */
int pho_r2c_fft_f64(const double *input, size_t n, double *out_re, size_t out_re_len, double *out_im,
                    size_t out_im_len);
int pho_r2c_fft_f32(const float *input, size_t n, float *out_re, size_t out_re_len, float *out_im,
                    size_t out_im_len);
int pho_r2c_fft_f64_with_planner(const double *input, size_t n, double *out_re, size_t out_re_len,
                                 double *out_im, size_t out_im_len, const pho_planner_r2c64 *planner);
int pho_r2c_fft_f32_with_planner(const float *input, size_t n, float *out_re, size_t out_re_len,
                                 float *out_im, size_t out_im_len, const pho_planner_r2c32 *planner);

/* algorithms/r2c.rs:695-895 */
int pho_c2r_fft_f64(const double *in_re, size_t in_re_len, const double *in_im, size_t in_im_len,
                    double *output, size_t n);
int pho_c2r_fft_f32(const float *in_re, size_t in_re_len, const float *in_im, size_t in_im_len, float *output,
                    size_t n);
int pho_c2r_fft_f64_with_planner_and_scratch(const double *in_re, size_t in_re_len, const double *in_im,
                                             size_t in_im_len, double *output, size_t n,
                                             const pho_planner_r2c64 *planner, double *scratch_re,
                                             size_t scratch_re_len, double *scratch_im, size_t scratch_im_len);
int pho_c2r_fft_f32_with_planner_and_scratch(const float *in_re, size_t in_re_len, const float *in_im,
                                             size_t in_im_len, float *output, size_t n,
                                             const pho_planner_r2c32 *planner, float *scratch_re,
                                             size_t scratch_re_len, float *scratch_im, size_t scratch_im_len);

/* the exact panic strings of the reference (r2c.rs:1392-1540) */
const char *pho_strerror(int code);

/*
 * Timing helper for bench.py's cpu_baseline leg: runs `iters` forward transforms with the planner
 * built outside the timer and the input regenerated before every timed call, as
 * examples/benchmark.rs:19-63 does; returns the SUM of the timed seconds.
 */
double pho_time_fft_64_dit(size_t n, int iters, unsigned long long seed);
double pho_time_fft_32_dit(size_t n, int iters, unsigned long long seed);
double pho_time_fft_64_dit_parallel(size_t n, int iters, unsigned long long seed, int threads); /* feature `parallel` emulated; threads <= 0: OpenMP default */
int pho_parallel_threads(void);
double pho_time_fft_64_roundtrip(size_t n, int iters, unsigned long long seed); /* forward + inverse per iteration */
double pho_time_r2c_fft_f32(size_t n, int iters, unsigned long long seed);
double pho_time_c2r_fft_f32(size_t n, int iters, unsigned long long seed);

/* counter-based synthetic input shared with the HIP fill kernel:
 * u = splitmix64(seed ^ (transform_id << 40) ^ (2*i + is_imag)); value = (u >> 11) * 2^-52 - 1 in [-1, 1) */
void pho_fill_f64(double *re, double *im, size_t n, unsigned long long seed, unsigned long long transform_id);
void pho_fill_f32(float *re, float *im, size_t n, unsigned long long seed, unsigned long long transform_id);

#ifdef __cplusplus
}
#endif
#endif
