/*
 * phastft_oracle.c -- CPU restatement of PhastFT 0.3.0's planar DIT FFT / bit reversal / R2C path.
 * TEST INFRASTRUCTURE ONLY -- see phastft_oracle.h for the rules and the pinning status.
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off).
 */
#define _GNU_SOURCE
#include "phastft_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static inline unsigned pho_ilog2(size_t x) { return 63u - (unsigned)__builtin_clzll((unsigned long long)x); }

/* bravo.rs:338-345 */
static inline size_t pho_reverse_bits(size_t x, unsigned bits) {
    if (bits == 0) return 0;
    uint64_t v = (uint64_t)x;
    v = ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
    v = __builtin_bswap64(v);
    return (size_t)(v >> (64 - bits));
}

static inline uint64_t pho_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline double pho_uniform_pm1(uint64_t seed, uint64_t transform_id, uint64_t idx) {
    uint64_t u = pho_splitmix64(seed ^ (transform_id << 40) ^ idx);
    return (double)(u >> 11) * 0x1.0p-52 - 1.0;
}

/* options.rs:38-43 */
int pho_options_guess(size_t input_size, pho_options *out) {
    if (input_size == 0) return PHO_ERR_NOT_POW2;
    out->multithreaded_bit_reversal = pho_ilog2(input_size) >= 16;
    out->smallest_parallel_chunk_size = 16384;
    return PHO_OK;
}

const char *pho_strerror(int code) {
    switch (code) {
    case PHO_OK: return "ok";
    case PHO_ERR_NOT_POW2: return "assertion failed: num_points > 0 && num_points.is_power_of_two()";
    case PHO_ERR_LEN_MISMATCH: return "assertion `left == right` failed: reals.len() == imags.len()";
    case PHO_ERR_PLANNER_SIZE: return "assertion `left == right` failed: log_n == planner.log_n";
    case PHO_ERR_R2C_N: return "n must be a power of 2 >= 4";
    case PHO_ERR_R2C_INPUT_LEN: return "input length must match planner size";
    case PHO_ERR_R2C_OUT_RE_LEN: return "output_re must have length N/2 + 1";
    case PHO_ERR_R2C_OUT_IM_LEN: return "output_im must have length N/2 + 1";
    case PHO_ERR_C2R_OUTPUT_LEN: return "output length must match planner size";
    case PHO_ERR_C2R_IN_RE_LEN: return "input_re must have length N/2 + 1";
    case PHO_ERR_C2R_IN_IM_LEN: return "input_im must have length N/2 + 1";
    case PHO_ERR_C2R_SCRATCH_RE: return "scratch_re must have length N/2";
    case PHO_ERR_C2R_SCRATCH_IM: return "scratch_im must have length N/2";
    case PHO_ERR_ALLOC: return "allocation failed";
    default: return "unknown error";
    }
}

/*
 * Literal twiddles of kernels/dit.rs:141-967 and kernels/codelets.rs:118-194,400-488.
 * All are cos(2*pi*j/64), j = 0..16, as the reference WRITES them: the f32 literals are the
 * correctly rounded values; five f64 literals are 1 ulp off the correctly rounded cosine
 * (j = 3, 5, 13, 14, and j = 12 in the 16-point kernels only) and are kept as written.
 */
static const double PHO_C64_F64[17] = {
    1.0,
    0.9951847266721969,
    0.9807852804032304,
    0.9569403357322089, /* kernels/dit.rs chunk_64: correctly rounded is ...088 */
    0.9238795325112867,
    0.8819212643483549, /* correctly rounded is ...355 */
    0.8314696123025452,
    0.773010453362737,
    0.70710678118654752440, /* std::f64::consts::FRAC_1_SQRT_2 */
    0.6343932841636455,
    0.5555702330196022,
    0.47139673682599764,
    0.3826834323650898,  /* chunk_32 / chunk_64 spelling (correctly rounded) */
    0.29028467725446233, /* correctly rounded is ...624 */
    0.19509032201612825, /* correctly rounded is ...828 */
    0.0980171403295606,
    0.0,
};
static const double PHO_C64_12_ALT_F64 = 0.38268343236508984; /* chunk_16 + codelet_16 spelling */

static const float PHO_C64_F32[17] = {
    1.0f,        0.9951847f,  0.98078525f, 0.95694035f, 0.9238795f,  0.8819213f,
    0.8314696f,  0.77301043f, 0.70710677f, 0.6343933f,  0.55557024f, 0.47139674f,
    0.38268343f, 0.29028466f, 0.19509032f, 0.09801714f, 0.0f,
};

/* W_M^k = (re, im), M in {8,16,32,64}, k < M/2; j = k*64/M is the angle index on the 64-gon */
static inline double pho_c64_f64(unsigned m, unsigned j) {
    if (m == 16 && j == 12) return PHO_C64_12_ALT_F64;
    return PHO_C64_F64[j];
}
static inline double pho_const_re_f64(unsigned m, unsigned k) {
    unsigned j = k * (64 / m);
    return j <= 16 ? pho_c64_f64(m, j) : -pho_c64_f64(m, 32 - j);
}
static inline double pho_const_im_f64(unsigned m, unsigned k) {
    unsigned j = k * (64 / m);
    if (j == 0) return 0.0; /* the reference writes +0.0 for W^0.im */
    return -pho_c64_f64(m, j <= 16 ? 16 - j : j - 16);
}
static inline float pho_const_re_f32(unsigned m, unsigned k) {
    unsigned j = k * (64 / m);
    return j <= 16 ? PHO_C64_F32[j] : -PHO_C64_F32[32 - j];
}
static inline float pho_const_im_f32(unsigned m, unsigned k) {
    unsigned j = k * (64 / m);
    if (j == 0) return 0.0f;
    return -PHO_C64_F32[j <= 16 ? 16 - j : j - 16];
}

/* ---- f64 instantiation ---- */
#define T double
#define SFX 64
#define FSFX f64
#define FMA fma
#define COS cos
#define SIN sin
#define PI_T 3.14159265358979323846
#define CODELET_STAGES 4
#define TILE_SIDE 32 /* bravo.rs:20 */
#define CONST_RE(M, k) pho_const_re_f64((unsigned)(M), (unsigned)(k))
#define CONST_IM(M, k) pho_const_im_f64((unsigned)(M), (unsigned)(k))
#define FRAC_1_SQRT_2_T 0.70710678118654752440
#include "dit_impl.inc"
#undef T
#undef SFX
#undef FSFX
#undef FMA
#undef COS
#undef SIN
#undef PI_T
#undef CODELET_STAGES
#undef TILE_SIDE
#undef CONST_RE
#undef CONST_IM
#undef FRAC_1_SQRT_2_T

/* ---- f32 instantiation ---- */
#define T float
#define SFX 32
#define FSFX f32
#define FMA fmaf
#define COS cosf
#define SIN sinf
#define PI_T ((float)3.14159265358979323846) /* std::f32::consts::PI */
#define CODELET_STAGES 5
#define TILE_SIDE 64 /* bravo.rs:19 */
#define CONST_RE(M, k) pho_const_re_f32((unsigned)(M), (unsigned)(k))
#define CONST_IM(M, k) pho_const_im_f32((unsigned)(M), (unsigned)(k))
#define FRAC_1_SQRT_2_T 0.70710677f
#include "dit_impl.inc"

/* ---- timing helpers for bench.py's cpu_baseline leg (examples/benchmark.rs:19-63) ---- */
static double pho_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double time_fft_64(size_t n, int iters, unsigned long long seed, int par);
double pho_time_fft_64_dit(size_t n, int iters, unsigned long long seed) { return time_fft_64(n, iters, seed, 0); }
double pho_time_fft_64_dit_parallel(size_t n, int iters, unsigned long long seed, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    return time_fft_64(n, iters, seed, 1);
}
int pho_parallel_threads(void) { return omp_get_max_threads(); }

static double time_fft_64(size_t n, int iters, unsigned long long seed, int par) {
    pho_planner_dit64 *planner;
    if (pho_planner_dit64_new(n, &planner)) return -1.0;
    double *re = malloc(n * sizeof(double)), *im = malloc(n * sizeof(double));
    double total = 0.0;
    for (int it = 0; it < iters && re && im; ++it) {
        pho_fill_f64(re, im, n, seed, (unsigned long long)it);
        double t0 = pho_now();
        if (par) pho_fft_64_dit_with_planner_parallel(re, n, im, n, PHO_FORWARD, planner);
        else pho_fft_64_dit_with_planner(re, n, im, n, PHO_FORWARD, planner);
        total += pho_now() - t0;
    }
    free(re);
    free(im);
    pho_planner_dit64_free(planner);
    return total;
}

/* f32 twin of pho_time_fft_64_dit (bench.py configs f32_2p20 / f32_2p26): planner outside the timer, input regenerated
 * before every timed call */
double pho_time_fft_32_dit(size_t n, int iters, unsigned long long seed) {
    pho_planner_dit32 *planner;
    if (pho_planner_dit32_new(n, &planner)) return -1.0;
    float *re = malloc(n * sizeof(float)), *im = malloc(n * sizeof(float));
    double total = 0.0;
    for (int it = 0; it < iters && re && im; ++it) {
        pho_fill_f32(re, im, n, seed, (unsigned long long)it);
        double t0 = pho_now();
        pho_fft_32_dit_with_planner(re, n, im, n, PHO_FORWARD, planner);
        total += pho_now() - t0;
    }
    free(re);
    free(im);
    pho_planner_dit32_free(planner);
    return total;
}

/* BASELINE configs[2]: forward then inverse on the same buffers, timed together */
double pho_time_fft_64_roundtrip(size_t n, int iters, unsigned long long seed) {
    pho_planner_dit64 *planner;
    if (pho_planner_dit64_new(n, &planner)) return -1.0;
    double *re = malloc(n * sizeof(double)), *im = malloc(n * sizeof(double));
    double total = 0.0;
    for (int it = 0; it < iters && re && im; ++it) {
        pho_fill_f64(re, im, n, seed, (unsigned long long)it);
        double t0 = pho_now();
        pho_fft_64_dit_with_planner(re, n, im, n, PHO_FORWARD, planner);
        pho_fft_64_dit_with_planner(re, n, im, n, PHO_REVERSE, planner);
        total += pho_now() - t0;
    }
    free(re);
    free(im);
    pho_planner_dit64_free(planner);
    return total;
}

double pho_time_r2c_fft_f32(size_t n, int iters, unsigned long long seed) {
    pho_planner_r2c32 *planner;
    if (pho_planner_r2c32_new(n, &planner)) return -1.0;
    size_t half = n / 2;
    float *in = malloc(n * sizeof(float)), *dummy = malloc(n * sizeof(float));
    float *ore = malloc((half + 1) * sizeof(float)), *oim = malloc((half + 1) * sizeof(float));
    double total = 0.0;
    for (int it = 0; it < iters && in && dummy && ore && oim; ++it) {
        pho_fill_f32(in, dummy, n, seed, (unsigned long long)it);
        double t0 = pho_now();
        pho_r2c_fft_f32_with_planner(in, n, ore, half + 1, oim, half + 1, planner);
        total += pho_now() - t0;
    }
    free(in);
    free(dummy);
    free(ore);
    free(oim);
    pho_planner_r2c32_free(planner);
    return total;
}

/* the inverse leg (bench.py configs.c2r_f32_2p24): c2r_fft_f32_with_planner_and_scratch (r2c.rs:836-895) on a half-spectrum regenerated
 * before every timed call, planner outside the timer */
double pho_time_c2r_fft_f32(size_t n, int iters, unsigned long long seed) {
    pho_planner_r2c32 *planner;
    if (pho_planner_r2c32_new(n, &planner)) return -1.0;
    size_t half = n / 2;
    float *out = malloc(n * sizeof(float));
    float *ire = malloc((half + 1) * sizeof(float)), *iim = malloc((half + 1) * sizeof(float));
    float *sre = malloc(half * sizeof(float)), *sim = malloc(half * sizeof(float)); /* the caller's scratch (r2c.rs:836) */
    double total = 0.0;
    for (int it = 0; it < iters && out && ire && iim && sre && sim; ++it) {
        pho_fill_f32(ire, iim, half + 1, seed, (unsigned long long)it);
        double t0 = pho_now();
        pho_c2r_fft_f32_with_planner_and_scratch(ire, half + 1, iim, half + 1, out, n, planner, sre, half, sim, half);
        total += pho_now() - t0;
    }
    free(sre);
    free(sim);
    free(out);
    free(ire);
    free(iim);
    pho_planner_r2c32_free(planner);
    return total;
}

/* ---- complex_nums.rs:11-56: Complex<T> <-> planes ---- */
#define PHO_COMPLEX_NUMS(T, FS)                                                                         \
    void pho_deinterleave_##FS(const T *input, size_t len, T *out_a, T *out_b) { /* complex_nums.rs:16 */ \
        for (size_t k = 0; k + 1 < len; k += 2) {                                                       \
            out_a[k / 2] = input[k];                                                                    \
            out_b[k / 2] = input[k + 1];                                                                \
        }                                                                                               \
    }                                                                                                   \
    void pho_combine_re_im_##FS(const T *reals, const T *imags, size_t n, T *out) { /* complex_nums.rs:50-55 */ \
        for (size_t k = 0; k < n; ++k) {                                                                \
            out[2 * k] = reals[k];                                                                      \
            out[2 * k + 1] = imags[k];                                                                  \
        }                                                                                               \
    }
PHO_COMPLEX_NUMS(double, f64)
PHO_COMPLEX_NUMS(float, f32)
