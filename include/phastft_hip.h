/*
 * phastft_hip.h -- C ABI of libphastft_hip.so, the MI355X (gfx950) drop-in for PhastFT's planar
 * power-of-two FFT path.
 *
 * The reference (QuState/PhastFT 0.3.0, /root/reference) has no FFI layer: its boundary is the
 * crate's public Rust API.  Every entry point below is named after, and
 * cites, the Rust item it replaces; a Rust shim crate binds them 1:1.
 *
 *   - plain pointers + explicit lengths (so the library re-validates what the Rust `assert!`s check)
 *   - `int direction`: +1 = Direction::Forward, -1 = Direction::Reverse  (planner.rs:10-16)
 *   - return value: PHAST_OK or one status per reference assert; phast_strerror() returns the
 *     reference's panic text so a shim can `panic!` with it (r2c.rs:1392-1540 tests the strings)
 *   - host-slice calls (no suffix) take HOST pointers, stage H2D/D2H internally and return with the
 *     result visible in the caller's slices -- the drop-in semantics of the Rust API
 *   - `_dev` calls take DEVICE pointers + a hipStream_t (as void*) and are asynchronous on that
 *     stream; they are what bench.py measures.  `batch` independent transforms, transform b at
 *     pointer + b*dist elements.
 *
 * Planners may be shared by concurrent host threads AND streams (planner.rs:38-39: the reference's planner is an
 * immutable value borrowed by `&`).  What a call mutates -- the inter-pass scratch, the staging buffer and pinned mirror of
 * the host-slice calls -- lives in a WORKSPACE, and a planner keeps a small pool of them (up to PHAST_MAX_WORKSPACES = 8,
 * made on demand): a call checks one out while it enqueues (a blocking host-slice call: for the whole call, on the
 * workspace's own non-blocking stream, never the NULL stream), calls on one stream come back to the same workspace, calls
 * on other streams get another one, and only when the pool is exhausted does a stream wait -- on the device, behind an
 * event -- for another stream's work.  N threads x N streams on one planner run side by side.
 * Graphs: a workspace used under stream capture belongs to the captured graph(s) until the planner is freed -- eager
 * calls never touch it again and none of its buffers is ever released early, so replays stay valid whatever the planner
 * is used for afterwards.  Replays of several graphs captured from ONE planner on ONE capture stream share that workspace:
 * order them among themselves (or capture them on different streams).
 *
 * Devices: a planner belongs to the HIP device that is current when it is created (its tables and scratch live
 * there).  One process may hold planners on several devices (one host thread per GPU, or one thread switching):
 * every call on a planner runs on the planner's device whatever the calling thread's current device is, and the
 * caller's current device is restored before the call returns.  Data pointers and the stream of a _dev call must
 * belong to (or be accessible from) the planner's device.
 *
 * Memory: scratch and staging buffers grow on demand (geometrically); an outgrown buffer is released as soon as the
 * work that used it has completed, never under stream capture.  When the device is out of memory a batched call
 * falls back to smaller chunks before it reports PHAST_ERR_HIP.
 */
#ifndef PHASTFT_HIP_H
#define PHASTFT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- planner.rs:10-32 ---- */
#define PHAST_FORWARD 1  /* Direction::Forward */
#define PHAST_REVERSE (-1) /* Direction::Reverse */
#define PHAST_MODE_HEURISTIC 0 /* PlannerMode::Heuristic */
#define PHAST_MODE_TUNE 1      /* PlannerMode::Tune: the planner measures its plans on the device at plan time (below) */

/* ---- status codes: one per reference assert ---- */
#define PHAST_OK 0
#define PHAST_ERR_NOT_POW2 1        /* planner.rs:66, algorithms/dit.rs:285,359 */
#define PHAST_ERR_LEN_MISMATCH 2    /* algorithms/dit.rs:284,358 */
#define PHAST_ERR_PLANNER_SIZE 3    /* algorithms/dit.rs:289,363 */
#define PHAST_ERR_R2C_N 4           /* planner.rs:195 */
#define PHAST_ERR_R2C_INPUT_LEN 5   /* algorithms/r2c.rs:543,615 */
#define PHAST_ERR_R2C_OUT_RE_LEN 6  /* algorithms/r2c.rs:544-548,616-620 */
#define PHAST_ERR_R2C_OUT_IM_LEN 7  /* algorithms/r2c.rs:549-553,621-625 */
#define PHAST_ERR_C2R_OUTPUT_LEN 8  /* algorithms/r2c.rs:750,846 */
#define PHAST_ERR_C2R_IN_RE_LEN 9   /* algorithms/r2c.rs:751-755,847-851 */
#define PHAST_ERR_C2R_IN_IM_LEN 10  /* algorithms/r2c.rs:756-760,852-856 */
#define PHAST_ERR_C2R_SCRATCH_RE 11 /* algorithms/r2c.rs:761,857 */
#define PHAST_ERR_C2R_SCRATCH_IM 12 /* algorithms/r2c.rs:762,858 */
#define PHAST_ERR_ALLOC 13          /* host allocation failed (incl. std::bad_alloc inside the library) */
#define PHAST_ERR_HIP 14            /* a HIP runtime call failed; see phast_last_hip_error() */
#define PHAST_ERR_NO_DEVICE 15      /* no gfx950 device visible: the library never falls back to CPU */
#define PHAST_ERR_INVALID_ARG 16    /* null pointer / bad direction / bit-reversal length != 2^n (bravo.rs:228) */

/* exact panic text of the reference for the code (or a description for the HIP-side codes) */
const char *phast_strerror(int code);
/* hipGetErrorString of the last failing HIP call on this thread ("" if none) */
const char *phast_last_hip_error(void);
/* library / device identification: returns PHAST_OK and fills what it can */
int phast_device_info(char *name, size_t name_len, int *compute_units, size_t *lds_per_block,
                      size_t *global_mem_bytes);

/* ---- options.rs:8-43 ---- */
typedef struct phast_options {
    int multithreaded_bit_reversal;      /* options.rs:17; CPU threading hint -- ignored on the GPU */
    size_t smallest_parallel_chunk_size; /* options.rs:24; ignored on the GPU */
} phast_options;
void phast_options_default(phast_options *out);               /* options.rs:26-33 */
int phast_options_guess(size_t input_size, phast_options *out); /* options.rs:38-43 */

/* ---- planner.rs:34-114 ----
 * num_points: a power of two (else PHAST_ERR_NOT_POW2, the reference's assert) up to 2^30 for f64 and 2^31 for
 * f32; larger sizes return PHAST_ERR_INVALID_ARG (the reference is bounded by host memory only). */
typedef struct phast_planner_dit64 phast_planner_dit64; /* PlannerDit64 */
typedef struct phast_planner_dit32 phast_planner_dit32; /* PlannerDit32 */
int phast_planner_dit64_new(size_t num_points, phast_planner_dit64 **out);                 /* planner.rs:55 */
int phast_planner_dit64_with_mode(size_t num_points, int mode, phast_planner_dit64 **out); /* planner.rs:65 */
void phast_planner_dit64_free(phast_planner_dit64 *p);
int phast_planner_dit32_new(size_t num_points, phast_planner_dit32 **out);
int phast_planner_dit32_with_mode(size_t num_points, int mode, phast_planner_dit32 **out);
void phast_planner_dit32_free(phast_planner_dit32 *p);
/* device footprint of a planner (twiddle tables + scratch), and its pass plan as text */
size_t phast_planner_dit64_device_bytes(const phast_planner_dit64 *p);
size_t phast_planner_dit32_device_bytes(const phast_planner_dit32 *p);
int phast_planner_dit64_describe(const phast_planner_dit64 *p, char *buf, size_t buf_len);
int phast_planner_dit32_describe(const phast_planner_dit32 *p, char *buf, size_t buf_len);
/* the plan a call of `kind` (PHAST_TUNE_*) with `batch` transforms runs, as text: "<which> [rows x cols ...]..." */
int phast_planner_dit64_describe_call(const phast_planner_dit64 *p, size_t batch, int kind, char *buf, size_t buf_len);
int phast_planner_dit32_describe_call(const phast_planner_dit32 *p, size_t batch, int kind, char *buf, size_t buf_len);
/* optional: size the scratch for `max_batch` transforms in flight (default 1); realloc on demand otherwise */
int phast_planner_dit64_reserve_batch(phast_planner_dit64 *p, size_t max_batch);
int phast_planner_dit32_reserve_batch(phast_planner_dit32 *p, size_t max_batch);
/* HIP graphs: the workspaces that captured calls work in are kept until the planner is freed (the library cannot know when a
 * graph dies).  A long-lived planner that is captured again and again may hand them back -- the caller promises that every
 * graph captured on this planner so far is gone.  Returns the device bytes released. */
size_t phast_planner_dit64_release_graph_workspaces(phast_planner_dit64 *p);
size_t phast_planner_dit32_release_graph_workspaces(phast_planner_dit32 *p);

/* ---- PlannerMode::Tune (planner.rs:18-32: "benchmarks both paths at plan time and picks whichever is faster, at the cost of
 * additional planning time") ----
 * The reference chooses between two codelet paths; here the choice is the PLAN -- how N = 2^L is cut into 2 or 3 passes, the
 * tile size of every pass, points per thread, wave / quad tiles, and for r2c whether the untangle rides in the last pass.
 * PHAST_MODE_HEURISTIC: static rules ranked on an MI355X (plus built-in / imported wisdom, below), zero planning overhead.
 * PHAST_MODE_TUNE (`_with_mode`): as the reference's -- the planner times every plan that exists for its length on the
 * current device, for ONE transform per call (the reference's only case), and keeps the fastest if it beats the static rule
 * by more than 3 %.  Costs 0.1 .. 3 s (under 1 s at N = 2^20).  Nothing to measure for N <= 4096 (one kernel).
 * `_tune` does the same for another batch size or call kind on an existing planner: a result covers batches in
 * (2^(b-1), 2^b] around `batch_hint`.  Tuning is synchronous, allocates a ring of input sets (up to 1.25 GiB, or three sets)
 * and must not run under stream capture.  Results are bit-identical for a given plan; which plan runs changes the last bits
 * (same tolerance: every plan is tested against the oracle). */
#define PHAST_TUNE_C2C 0             /* fft_*_dit* on planar arrays */
#define PHAST_TUNE_C2C_INTERLEAVED 1 /* the Complex<T> forms (lib.rs:41-140) */
#define PHAST_TUNE_R2C 2             /* r2c_fft_*   (PlannerR2c* only) */
#define PHAST_TUNE_C2R 3             /* c2r_fft_*   (PlannerR2c* only) */
typedef struct phast_tune_report {
    int adopted;             /* 1: a measured plan replaced the static rule's for this (kind, batch bucket) */
    unsigned candidates;     /* plans timed */
    float us_heuristic;      /* per call, static rule's plan (median of the interleaved rounds) */
    float us_best;           /* per call, the plan now in force */
    double seconds;          /* what the tuning run took */
    char plan[96];           /* "a,b[,c]@ta,tb[,tc]:p<points>[w][ fused]" or "heuristic" / "one pass" */
} phast_tune_report;
int phast_planner_dit64_tune(phast_planner_dit64 *p, size_t batch_hint, int kind, phast_tune_report *report /* or NULL */);
int phast_planner_dit32_tune(phast_planner_dit32 *p, size_t batch_hint, int kind, phast_tune_report *report);
/* Wisdom: what tuning runs found, as text (one line per type / kind / log2 length / batch bucket, csrc/wisdom.hpp).  Planners
 * created after an import start with the plans it names (entries measured on a device with another CU count or gfx architecture, or by another generation of the library's kernels -- the header line's cus= / arch= / lib= -- are kept but not applied).
 * PHAST_WISDOM=<path>: read at first use, rewritten after every tuning run.  The library also carries built-in wisdom measured
 * on an MI355X (PHAST_BUILTIN_WISDOM=0 turns it off).  No device needed for these three calls. */
int phast_wisdom_export(char *buf, size_t buf_len, size_t *needed /* bytes incl. NUL, or NULL */); /* without the built-in layer */
int phast_wisdom_import(const char *text); /* PHAST_ERR_INVALID_ARG: not a wisdom text (nothing of it is kept); lines that do not parse are skipped */
void phast_wisdom_forget(void);            /* everything but the built-in layer */
int phast_wisdom_builtin(int enable);      /* the built-in layer off / on again at run time (planners made afterwards); returns what it was (1 on, 0 off) so a caller can put it back */
size_t phast_wisdom_count(int layer);      /* entries of a layer: 0 built-in, 1 PHAST_WISDOM file, 2 imported, 3 measured here; -1 all */

/* ---- planner.rs:164-212 ---- */
typedef struct phast_planner_r2c64 phast_planner_r2c64; /* PlannerR2c64 */
typedef struct phast_planner_r2c32 phast_planner_r2c32; /* PlannerR2c32 */
int phast_planner_r2c64_new(size_t n, phast_planner_r2c64 **out); /* planner.rs:194 */
void phast_planner_r2c64_free(phast_planner_r2c64 *p);
int phast_planner_r2c32_new(size_t n, phast_planner_r2c32 **out);
void phast_planner_r2c32_free(phast_planner_r2c32 *p);
/* (no reference counterpart: PlannerR2c*::new has no mode -- the same switch as PlannerDit*::with_mode, for r2c_fft and c2r_fft
 * of one transform per call; `_tune`: kind = PHAST_TUNE_R2C or PHAST_TUNE_C2R) */
int phast_planner_r2c64_with_mode(size_t n, int mode, phast_planner_r2c64 **out);
int phast_planner_r2c32_with_mode(size_t n, int mode, phast_planner_r2c32 **out);
int phast_planner_r2c64_describe_call(const phast_planner_r2c64 *p, size_t batch, int kind, char *buf, size_t buf_len);
int phast_planner_r2c32_describe_call(const phast_planner_r2c32 *p, size_t batch, int kind, char *buf, size_t buf_len);
int phast_planner_r2c64_tune(phast_planner_r2c64 *p, size_t batch_hint, int kind, phast_tune_report *report);
int phast_planner_r2c32_tune(phast_planner_r2c32 *p, size_t batch_hint, int kind, phast_tune_report *report);

/* ---- C2C, host slices: lib.rs:143-226, algorithms/dit.rs:263,338 ----
 * The forms without a planner argument make one per call in the reference (lib.rs:181,224).  Here a planner owns device
 * memory, so the library keeps the few most recently used ones (per type, size, device; planes up to 64 MiB) and the
 * second call of a size costs what the _with_planner form costs.  Same results, same errors; PHAST_PLANNER_CACHE=0 turns
 * it off.  The same holds for the real-transform and interleaved forms below. */
int phast_fft_64_dit(double *reals, size_t reals_len, double *imags, size_t imags_len, int direction); /* lib.rs:180 */
int phast_fft_32_dit(float *reals, size_t reals_len, float *imags, size_t imags_len, int direction);   /* lib.rs:223 */
int phast_fft_64_dit_with_planner(double *reals, size_t reals_len, double *imags, size_t imags_len,
                                  int direction, const phast_planner_dit64 *planner); /* lib.rs:143 */
int phast_fft_32_dit_with_planner(float *reals, size_t reals_len, float *imags, size_t imags_len, int direction,
                                  const phast_planner_dit32 *planner); /* lib.rs:186 */
int phast_fft_64_dit_with_planner_and_opts(double *reals, size_t reals_len, double *imags, size_t imags_len,
                                           int direction, const phast_planner_dit64 *planner,
                                           const phast_options *opts); /* algorithms/dit.rs:263 */
int phast_fft_32_dit_with_planner_and_opts(float *reals, size_t reals_len, float *imags, size_t imags_len,
                                           int direction, const phast_planner_dit32 *planner,
                                           const phast_options *opts); /* algorithms/dit.rs:338 */

/* ---- C2C, device-resident, batched, asynchronous on `stream` (hipStream_t) ---- */
int phast_fft_64_dit_dev(double *d_reals, double *d_imags, size_t n, size_t batch, size_t dist, int direction,
                         const phast_planner_dit64 *planner, void *stream);
int phast_fft_32_dit_dev(float *d_reals, float *d_imags, size_t n, size_t batch, size_t dist, int direction,
                         const phast_planner_dit32 *planner, void *stream);
/* `count` independent transforms of n points at arbitrary device addresses (HOST arrays of device pointers), each run
 * exactly as a single-transform call, enqueued back to back by ONE host call -- a caller with many separate signals
 * (the Rust API takes one pair of slices per call, lib.rs:143) pays its FFI / interpreter overhead once. */
int phast_fft_64_dit_many_dev(double *const *d_reals, double *const *d_imags, size_t count, size_t n, int direction,
                              const phast_planner_dit64 *planner, void *stream);
int phast_fft_32_dit_many_dev(float *const *d_reals, float *const *d_imags, size_t count, size_t n, int direction,
                              const phast_planner_dit32 *planner, void *stream);
/* Strided batches (no reference counterpart): transform b occupies elements b*dist + j*stride, j < n.
 * stride == 1 is the call above.  dist == 1 with stride, batch powers of two, batch <= stride, n >= 64 are the
 * "column FFTs" of a row-major [n][stride] array (first `batch` columns), in place, natural order in and out -- what
 * a four-step split (phastft_amd/distributed.py) and multi-dimensional transforms need.  The planner's scratch grows
 * to n*stride elements per plane.  The kernels work on tiles of adjacent columns: batch >= 16 is always served,
 * batch == 8 for every n except 2^6, 2^12 and 2^13, batch == 4 only for n = 2^10 and 2^20, narrower batches never --
 * those (and anything else outside the description above) return PHAST_ERR_INVALID_ARG and nothing has run: transpose
 * and use the contiguous batch (phastft_amd/distributed.py does, and falls back on THAT code only). */
int phast_fft_64_dit_strided_dev(double *d_reals, double *d_imags, size_t n, size_t batch, size_t dist, size_t stride,
                                 int direction, const phast_planner_dit64 *planner, void *stream);
int phast_fft_32_dit_strided_dev(float *d_reals, float *d_imags, size_t n, size_t batch, size_t dist, size_t stride,
                                 int direction, const phast_planner_dit32 *planner, void *stream);
/* The same column FFTs with an INPUT twiddle fused into the first pass's load: element j of transform b is multiplied by
 * W_{tw_n}^(j * (tw_col0 + b)) before it is transformed -- the inter-factor twiddle of a four-step split (N = N1 N2 = tw_n,
 * this call = the second factor's transforms on rank-local columns tw_col0 ...), which otherwise is a sweep of its own
 * (phast_twiddle_grid*_apply_dev).  tw_n a power of two, n <= tw_n <= 2^32; dist == 1 only. */
int phast_fft_64_dit_strided_tw_dev(double *d_reals, double *d_imags, size_t n, size_t batch, size_t dist, size_t stride,
                                    int direction, const phast_planner_dit64 *planner, size_t tw_n, size_t tw_col0,
                                    void *stream);
int phast_fft_32_dit_strided_tw_dev(float *d_reals, float *d_imags, size_t n, size_t batch, size_t dist, size_t stride,
                                    int direction, const phast_planner_dit32 *planner, size_t tw_n, size_t tw_col0,
                                    void *stream);

/* ---- C2C on interleaved Complex<T> signals: lib.rs:41-140 (feature `complex-nums`) ----
 * `signal` holds n complex numbers as (re, im) pairs, transformed in place.  The reference copies into two planar
 * Vecs, runs the planar path and copies back (lib.rs:56-58); here the (de)interleave is fused into the first
 * pass's load and the last pass's store.  `dist` of the _dev form counts complex elements. */
int phast_fft_64_interleaved(double *signal, size_t n, int direction);                               /* lib.rs:120 */
int phast_fft_32_interleaved(float *signal, size_t n, int direction);
int phast_fft_64_interleaved_with_planner(double *signal, size_t n, int direction,
                                          const phast_planner_dit64 *planner);                       /* lib.rs:87 */
int phast_fft_32_interleaved_with_planner(float *signal, size_t n, int direction, const phast_planner_dit32 *planner);
int phast_fft_64_interleaved_with_planner_and_opts(double *signal, size_t n, int direction,
                                                   const phast_planner_dit64 *planner,
                                                   const phast_options *opts);                        /* lib.rs:50 */
int phast_fft_32_interleaved_with_planner_and_opts(float *signal, size_t n, int direction,
                                                   const phast_planner_dit32 *planner, const phast_options *opts);
int phast_fft_64_interleaved_dev(double *d_signal, size_t n, size_t batch, size_t dist, int direction,
                                 const phast_planner_dit64 *planner, void *stream);
int phast_fft_32_interleaved_dev(float *d_signal, size_t n, size_t batch, size_t dist, int direction,
                                 const phast_planner_dit32 *planner, void *stream);

/* ---- bit reversal: algorithms/bravo.rs:303,317 (public with feature bench-internals, lib.rs:20-23) ---- */
int phast_bit_rev_f64(double *data, size_t len, unsigned log_n); /* host slice */
int phast_bit_rev_f32(float *data, size_t len, unsigned log_n);
int phast_bit_rev_f64_dev(double *d_data, unsigned log_n, size_t batch, size_t dist, void *stream);
int phast_bit_rev_f32_dev(float *d_data, unsigned log_n, size_t batch, size_t dist, void *stream);

/* ---- Complex<T> <-> planes: complex_nums.rs (public with feature bench-internals, like the bit reversal) ----
 * deinterleave: [1, 2, 3, 4] -> ([1, 3], [2, 4]) for any length; `chunks_exact(2)` drops an odd last element
 * (complex_nums.rs:11-17).  deinterleave_complex64 / _complex32 (:25-39) are this on the cast slice: pass the n Complex<T>
 * as 2 n scalars.  combine_re_im (:47-56): `assert_eq!(reals.len(), imags.len())` -> PHAST_ERR_LEN_MISMATCH.  The reference
 * returns new Vecs; a C caller brings the outputs, and the host-slice forms check their lengths (len / 2 each; 2 n). */
int phast_deinterleave_f64(const double *input, size_t len, double *out_a, size_t a_len, double *out_b, size_t b_len);
int phast_deinterleave_f32(const float *input, size_t len, float *out_a, size_t a_len, float *out_b, size_t b_len);
int phast_deinterleave_f64_dev(const double *d_input, size_t len, double *d_out_a, double *d_out_b, void *stream);
int phast_deinterleave_f32_dev(const float *d_input, size_t len, float *d_out_a, float *d_out_b, void *stream);
int phast_combine_re_im_f64(const double *reals, size_t reals_len, const double *imags, size_t imags_len, double *out, size_t out_len);
int phast_combine_re_im_f32(const float *reals, size_t reals_len, const float *imags, size_t imags_len, float *out, size_t out_len);
int phast_combine_re_im_f64_dev(const double *d_reals, const double *d_imags, size_t n, double *d_out, void *stream);
int phast_combine_re_im_f32_dev(const float *d_reals, const float *d_imags, size_t n, float *d_out, void *stream);

/* ---- R2C: algorithms/r2c.rs:521-662 ---- */
int phast_r2c_fft_f64(const double *input_re, size_t input_len, double *output_re, size_t output_re_len,
                      double *output_im, size_t output_im_len); /* r2c.rs:521 */
int phast_r2c_fft_f32(const float *input_re, size_t input_len, float *output_re, size_t output_re_len,
                      float *output_im, size_t output_im_len); /* r2c.rs:598 */
int phast_r2c_fft_f64_with_planner(const double *input_re, size_t input_len, double *output_re,
                                   size_t output_re_len, double *output_im, size_t output_im_len,
                                   const phast_planner_r2c64 *planner); /* r2c.rs:535 */
int phast_r2c_fft_f32_with_planner(const float *input_re, size_t input_len, float *output_re, size_t output_re_len,
                                   float *output_im, size_t output_im_len,
                                   const phast_planner_r2c32 *planner); /* r2c.rs:607 */
/* device-resident: input N reals, outputs N/2+1 each; batch b at +b*in_dist / +b*out_dist elements */
int phast_r2c_fft_f64_dev(const double *d_input, double *d_output_re, double *d_output_im, size_t batch,
                          size_t in_dist, size_t out_dist, const phast_planner_r2c64 *planner, void *stream);
int phast_r2c_fft_f32_dev(const float *d_input, float *d_output_re, float *d_output_im, size_t batch,
                          size_t in_dist, size_t out_dist, const phast_planner_r2c32 *planner, void *stream);

/* ---- C2R: algorithms/r2c.rs:695-895 ---- */
int phast_c2r_fft_f64(const double *input_re, size_t input_re_len, const double *input_im, size_t input_im_len,
                      double *output, size_t output_len); /* r2c.rs:695 */
int phast_c2r_fft_f32(const float *input_re, size_t input_re_len, const float *input_im, size_t input_im_len,
                      float *output, size_t output_len); /* r2c.rs:804 */
int phast_c2r_fft_f64_with_planner(const double *input_re, size_t input_re_len, const double *input_im,
                                   size_t input_im_len, double *output, size_t output_len,
                                   const phast_planner_r2c64 *planner); /* r2c.rs:710 */
int phast_c2r_fft_f32_with_planner(const float *input_re, size_t input_re_len, const float *input_im,
                                   size_t input_im_len, float *output, size_t output_len,
                                   const phast_planner_r2c32 *planner); /* r2c.rs:813 */
/* the scratch slices are validated for length exactly as the reference does and otherwise unused:
 * the device path keeps its workspace in the planner (r2c.rs:740,836) */
int phast_c2r_fft_f64_with_planner_and_scratch(const double *input_re, size_t input_re_len, const double *input_im,
                                               size_t input_im_len, double *output, size_t output_len,
                                               const phast_planner_r2c64 *planner, double *scratch_re,
                                               size_t scratch_re_len, double *scratch_im, size_t scratch_im_len);
int phast_c2r_fft_f32_with_planner_and_scratch(const float *input_re, size_t input_re_len, const float *input_im,
                                               size_t input_im_len, float *output, size_t output_len,
                                               const phast_planner_r2c32 *planner, float *scratch_re,
                                               size_t scratch_re_len, float *scratch_im, size_t scratch_im_len);
int phast_c2r_fft_f64_dev(const double *d_input_re, const double *d_input_im, double *d_output, size_t batch,
                          size_t in_dist, size_t out_dist, const phast_planner_r2c64 *planner, void *stream);
int phast_c2r_fft_f32_dev(const float *d_input_re, const float *d_input_im, float *d_output, size_t batch,
                          size_t in_dist, size_t out_dist, const phast_planner_r2c32 *planner, void *stream);

/* ---- harness support: deterministic on-device inputs and digests ---- */
/* value(i) = uniform [-1,1) from splitmix64(seed ^ (transform_id << 40) ^ (2*i + is_imag)); transform b of the
 * batch uses transform_id = first_id + b.  Same generator as oracle/pho_fill_*. */
int phast_fill_f64_dev(double *d_reals, double *d_imags, size_t n, size_t batch, size_t dist,
                       unsigned long long seed, unsigned long long first_id, void *stream);
int phast_fill_f32_dev(float *d_reals, float *d_imags, size_t n, size_t batch, size_t dist,
                       unsigned long long seed, unsigned long long first_id, void *stream);
/* per-transform digest {sum re, sum im, sum re^2+im^2, re[probe]} in f64, 4 doubles per transform */
int phast_digest_f64_dev(const double *d_reals, const double *d_imags, size_t n, size_t batch, size_t dist,
                         size_t probe, double *d_digest, void *stream);
int phast_digest_f32_dev(const float *d_reals, const float *d_imags, size_t n, size_t batch, size_t dist,
                         size_t probe, double *d_digest, void *stream);

/* harness: the streaming ceilings of the current device, measured with hand-written grid-stride kernels (probe.hip):
 * d_a (read) and d_b (written) are device buffers of `bytes` each (a multiple of 16, >= 1 MiB; use >= 1 GiB so that
 * the 256 MiB Infinity Cache does not help); out_gbps[3] = {read-only, write-only, 1:1 copy with read + write counted}
 * in GB/s, each the best of several access widths / grid sizes over `reps` back-to-back launches.  Blocks until done.
 * bench.py reports them as roofline.stream_probe: the copy figure is what a pass that reads and writes every byte once
 * can reach on this box. */
int phast_stream_probe_dev(const void *d_a, void *d_b, size_t bytes, int reps, double *out_gbps, void *stream);

/* The _dev entry points are stream-capture safe (no allocation, no synchronisation in the steady state), so a
 * launch-bound sequence of transforms is captured into a HIP graph with the plain HIP API around them.  This helper is
 * hipGraphUpload for hosts that hold a hipGraphExec_t but cannot call HIP themselves (bench.py: the exec handle of a
 * torch CUDAGraph): the first launch of an instantiated graph otherwise pays the upload inside the timed region. */
int phast_hip_graph_upload(void *graph_exec, void *stream);

/* ---- one transform spread over several GPUs (no reference counterpart: the reference's
 * recursion, algorithms/dit.rs:33-164, never leaves one address space).  A four-step split N = N1*N2 needs, between
 * its two local FFT stages, every element (r, c) of a rank's row-major slab multiplied by W_N^((row0 + r)*(col0 + c));
 * a twiddle grid owns the device tables of W_N (three-level, as the planners').  The exchanges themselves are the
 * host side's (phastft_amd/distributed.py: RCCL all-to-all through torch.distributed). ---- */
typedef struct phast_twiddle_grid64 phast_twiddle_grid64;
typedef struct phast_twiddle_grid32 phast_twiddle_grid32;
int phast_twiddle_grid64_new(size_t n, phast_twiddle_grid64 **out);   /* n = N, a power of two <= 2^32 */
int phast_twiddle_grid32_new(size_t n, phast_twiddle_grid32 **out);
void phast_twiddle_grid64_free(phast_twiddle_grid64 *g);
void phast_twiddle_grid32_free(phast_twiddle_grid32 *g);
int phast_twiddle_grid64_apply_dev(const phast_twiddle_grid64 *g, double *d_re, double *d_im, size_t rows, size_t cols,
                                   size_t row_pitch, size_t row0, size_t col0, void *stream);
int phast_twiddle_grid32_apply_dev(const phast_twiddle_grid32 *g, float *d_re, float *d_im, size_t rows, size_t cols,
                                   size_t row_pitch, size_t row0, size_t col0, void *stream);

/* ---- tuning hook used by tools/ and tests to force a pass plan (n_passes = 0 restores the heuristic) ----
 * log_rows[i] = log2 of pass i's tile FFT length, tile_logs[i] = log2 of the points per tile of pass i
 * (12, 13 or 14), points_log = log2 of the complex points each thread holds (4, or 3 for the 4096-point
 * latency tiles).  The forced plan serves every batch size.  Returns PHAST_ERR_INVALID_ARG when the
 * factorisation is not realisable with the compiled tile shapes. */
int phast_planner_dit64_set_plan(phast_planner_dit64 *p, const unsigned *log_rows, const unsigned *tile_logs,
                                 size_t n_passes, unsigned points_log);
int phast_planner_dit32_set_plan(phast_planner_dit32 *p, const unsigned *log_rows, const unsigned *tile_logs,
                                 size_t n_passes, unsigned points_log);

/* tuning hook: force the number of resident workgroups per CU the tile passes are launched with (0 = planner's own
 * residency estimate).  Process-wide; for sweeps in tools/ only. */
void phast_debug_set_wg_per_cu(int wg_per_cu);
/* debug hooks for the sanitizer pass (the boxes run with xnack off, so device-side ASan is not
 * available): scratch buffers allocated AFTER phast_debug_set_guard_bytes(b) carry a b-byte guard band filled with 0xA5
 * on either side; ..._debug_check_guards blocks until the device is idle and counts the band bytes a kernel overwrote */
void phast_debug_set_guard_bytes(size_t bytes);
int phast_planner_dit64_debug_check_guards(const phast_planner_dit64 *p, size_t *bad_bytes);
int phast_planner_dit32_debug_check_guards(const phast_planner_dit32 *p, size_t *bad_bytes);

/* No C++ exception leaves the library (every entry point is a function-try-block, csrc/c_abi.hip): host memory exhaustion
 * comes back as PHAST_ERR_ALLOC, any other exception as PHAST_ERR_HIP with its text in phast_last_hip_error().  Test hook:
 * throws std::bad_alloc (1), std::runtime_error (2) or an int (3) INSIDE the library and returns what the caller would see. */
int phast_debug_throw(int what);

/* debug hook: device buffer of [3 passes][4096 workgroups][16] s_memtime stamps written by the tile kernels while
 * set (NULL = off; adds drains, so never leave it on while measuring).  tools/trace_tile.py decodes it. */
void phast_debug_set_trace(unsigned long long *d_trace);

/* ---- measurement hook (bench.py "roofline"): runs `reps` batched forward transforms in place on the given
 * buffers with hipEvents recorded on `stream` around every pass kernel; pass_ms[i] = average duration of pass i
 * in milliseconds (sum over batch chunks), *n_passes = number of passes (<= 3).  Blocks until done. ---- */
int phast_planner_dit64_time_passes(const phast_planner_dit64 *p, double *d_reals, double *d_imags, size_t batch,
                                    size_t dist, int reps, float *pass_ms, int *n_passes, void *stream);
int phast_planner_dit32_time_passes(const phast_planner_dit32 *p, float *d_reals, float *d_imags, size_t batch,
                                    size_t dist, int reps, float *pass_ms, int *n_passes, void *stream);

/* the same for a real transform: slots 0..n-2 = the passes of the inner N/2-point complex transform, slot n-1 = the
 * untangle sweep (r2c.rs:150-242); a transform whose inner N/2 <= 8192 runs as ONE kernel (*n_passes = 1, slot 0 not
 * timed per kernel -- use stream events around the call).  pass_ms must hold 4 floats. */
int phast_planner_r2c64_time_passes(const phast_planner_r2c64 *p, const double *d_input, double *d_output_re,
                                    double *d_output_im, size_t batch, size_t in_dist, size_t out_dist, int reps,
                                    float *pass_ms, int *n_passes, void *stream);
int phast_planner_r2c32_time_passes(const phast_planner_r2c32 *p, const float *d_input, float *d_output_re,
                                    float *d_output_im, size_t batch, size_t in_dist, size_t out_dist, int reps,
                                    float *pass_ms, int *n_passes, void *stream);
/* ... and for the inverse real transform (r2c.rs:740-895): slots 0..np-1 = the passes of the inner transform -- the first
 * of them forms z from the half-spectrum on load wherever its fused form exists (every multi-pass plan whose first pass
 * is a generic tile; csrc/c2r_fused.hpp), slot np = the preprocess sweep (r2c.rs:263-433) otherwise. */
int phast_planner_r2c64_time_c2r_passes(const phast_planner_r2c64 *p, const double *d_input_re, const double *d_input_im,
                                        double *d_output, size_t batch, size_t in_dist, size_t out_dist, int reps,
                                        float *pass_ms, int *n_passes, void *stream);
int phast_planner_r2c32_time_c2r_passes(const phast_planner_r2c32 *p, const float *d_input_re, const float *d_input_im,
                                        float *d_output, size_t batch, size_t in_dist, size_t out_dist, int reps,
                                        float *pass_ms, int *n_passes, void *stream);
/* tuning hook, as phast_planner_dit*_set_plan but for the inner N/2-point transform of a real-transform planner (R2C and
 * C2R then run that plan for every batch size; np = 0 restores the library's own plans).  tools/sweep_real.py */
int phast_planner_r2c64_set_inner_plan(phast_planner_r2c64 *p, const unsigned *log_rows, const unsigned *tile_logs,
                                       size_t n_passes, unsigned points_log);
int phast_planner_r2c32_set_inner_plan(phast_planner_r2c32 *p, const unsigned *log_rows, const unsigned *tile_logs,
                                       size_t n_passes, unsigned points_log);

/* plan of the inner N/2-point complex transform, as phast_planner_dit*_describe */
int phast_planner_r2c64_describe(const phast_planner_r2c64 *p, char *buf, size_t buf_len);
int phast_planner_r2c32_describe(const phast_planner_r2c32 *p, char *buf, size_t buf_len);

#ifdef __cplusplus
}
#endif
#endif
