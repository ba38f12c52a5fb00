// c_abi.hip -- the C ABI of include/phastft_hip.h over the host side of libphastft_hip.so (planner.hpp, exec.hpp, host_api.hpp).
// Mirrors PhastFT's public surface (lib.rs:143-226, planner.rs, options.rs,
// algorithms/r2c.rs:521-895, algorithms/bravo.rs:303-324); see DESIGN.md for the mapping.
//
// There is NO CPU fallback in this library: without a gfx950 device every compute entry point returns
// PHAST_ERR_NO_DEVICE / PHAST_ERR_HIP.
#include "host_util.hpp"
#include "workspace.hpp"
#include "planner.hpp"
#include "planner_pool.hpp"
#include "planner_plans.hpp"
#include "exec.hpp"
#include "planner_r2c.hpp"
#include "entry.hpp"
#include "host_api.hpp"
#include "tune.hpp"

// ================================================================================================
// C ABI
// ================================================================================================
using namespace phast;

struct phast_planner_dit64 : Planner<double> {};
struct phast_planner_dit32 : Planner<float> {};
struct phast_planner_r2c64 : PlannerR2c<double> {};
struct phast_planner_r2c32 : PlannerR2c<float> {};

// W_N^(r*c) tables of a four-step split (twiddle.hip)
template <typename T> struct TwiddleGrid {
    unsigned log_n = 0, tw_bits = 1;
    int device = -1;
    void *d_tw3 = nullptr;
    ~TwiddleGrid() {
        DeviceGuard on(device);
        if (d_tw3) hipFree(d_tw3);
    }
    int init(size_t n) {
        int rc = ensure_device(&device);
        if (rc) return rc;
        log_n = ilog2(n);
        if (log_n > 32) return PHAST_ERR_INVALID_ARG;  // exponents are reduced to 32 bits
        tw_bits = tw3_bits_for(log_n);
        if (((size_t)3 << tw_bits) * sizeof(cx_t<T>) > (size_t)160 * 1024) return PHAST_ERR_INVALID_ARG;  // tables must fit one CU's LDS
        return upload<T>(host_tw3<T>(log_n, tw_bits), &d_tw3);
    }
    int apply(T *d_re, T *d_im, size_t rows, size_t cols, size_t row_pitch, size_t row0, size_t col0, hipStream_t s) const {
        if ((!d_re || !d_im) && rows * cols) return PHAST_ERR_INVALID_ARG;
        if (row_pitch < cols) return PHAST_ERR_INVALID_ARG;
        PHAST_ON_DEVICE(device);
        TwiddleGridArgs a{};
        a.re = d_re;
        a.im = d_im;
        a.tw3 = d_tw3;
        a.rows = rows;
        a.cols = cols;
        a.row_pitch = row_pitch;
        a.row0 = row0;
        a.col0 = col0;
        a.log_n = log_n;
        a.tw_bits = tw_bits;
        PHAST_HIP(launch_twiddle_grid<T>(a, s));
        return PHAST_OK;
    }
};
struct phast_twiddle_grid64 : TwiddleGrid<double> {};
struct phast_twiddle_grid32 : TwiddleGrid<float> {};

// PlannerMode::Tune at construction: one transform per call (the reference's only case) -- unless wisdom already holds a
// measurement for this type, length, kind and device
template <typename P> static int tune_new(P *p, int kind) {
    const size_t eb = sizeof(typename P::value_type);
    if (WisdomStore::instance().lookup(eb, kind, p->wisdom_log_n(), 0, cus_of(p->device_of()), arch_of(p->device_of()), nullptr)) return PHAST_OK;
    return p->tune(kind, 1, nullptr);
}
template <typename R> static void report_to_c(const R &r, phast_tune_report *rep) {
    if (!rep) return;
    rep->adopted = r.adopted;
    rep->candidates = r.candidates;
    rep->us_heuristic = r.us_heuristic;
    rep->us_best = r.us_best;
    rep->seconds = r.seconds;
    std::snprintf(rep->plan, sizeof rep->plan, "%s", r.plan.c_str());
}

// No C++ exception leaves the library: the callers are C, Rust (unwinding across `extern "C"` is undefined there), ctypes.
// Host memory exhaustion (std::bad_alloc, std::length_error) becomes PHAST_ERR_ALLOC; anything else PHAST_ERR_HIP with the
// exception's text in phast_last_hip_error().
static int cxx_exception_rc() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return PHAST_ERR_ALLOC;
    } catch (const std::length_error &) {
        return PHAST_ERR_ALLOC;
    } catch (const std::exception &e) {
        std::snprintf(g_hip_err, sizeof g_hip_err, "C++ exception in the host library: %s", e.what());
        return PHAST_ERR_HIP;
    } catch (...) {
        std::snprintf(g_hip_err, sizeof g_hip_err, "unknown C++ exception in the host library");
        return PHAST_ERR_HIP;
    }
}
#define PHAST_CATCH_RC catch (...) { return cxx_exception_rc(); }
#define PHAST_CATCH_ZERO catch (...) { return 0; }
#define PHAST_CATCH_VOID catch (...) {}

extern "C" {

int phast_wisdom_export(char *buf, size_t buf_len, size_t *needed) try {
    const std::string text = WisdomStore::instance().export_text();
    if (needed) *needed = text.size() + 1;
    if (!buf || buf_len == 0) return needed ? PHAST_OK : PHAST_ERR_INVALID_ARG;
    if (buf_len < text.size() + 1) return PHAST_ERR_INVALID_ARG;
    std::memcpy(buf, text.c_str(), text.size() + 1);
    return PHAST_OK;
} PHAST_CATCH_RC
int phast_wisdom_import(const char *text) try {
    if (!text) return PHAST_ERR_INVALID_ARG;
    return WisdomStore::instance().import_text(text, 2) == 0 ? PHAST_OK : PHAST_ERR_INVALID_ARG;
} PHAST_CATCH_RC
void phast_wisdom_forget(void) try { WisdomStore::instance().forget(); } PHAST_CATCH_VOID
int phast_wisdom_builtin(int enable) try { return WisdomStore::instance().set_builtin(enable != 0) ? 1 : 0; } PHAST_CATCH_ZERO
size_t phast_wisdom_count(int layer) try { return WisdomStore::instance().count(layer); } PHAST_CATCH_ZERO

const char *phast_strerror(int code) {
    switch (code) {
    case PHAST_OK: return "ok";
    case PHAST_ERR_NOT_POW2: return "assertion failed: num_points > 0 && num_points.is_power_of_two()";
    case PHAST_ERR_LEN_MISMATCH: return "assertion `left == right` failed: reals.len() == imags.len()";
    case PHAST_ERR_PLANNER_SIZE: return "assertion `left == right` failed: log_n == planner.log_n";
    case PHAST_ERR_R2C_N: return "n must be a power of 2 >= 4";
    case PHAST_ERR_R2C_INPUT_LEN: return "input length must match planner size";
    case PHAST_ERR_R2C_OUT_RE_LEN: return "output_re must have length N/2 + 1";
    case PHAST_ERR_R2C_OUT_IM_LEN: return "output_im must have length N/2 + 1";
    case PHAST_ERR_C2R_OUTPUT_LEN: return "output length must match planner size";
    case PHAST_ERR_C2R_IN_RE_LEN: return "input_re must have length N/2 + 1";
    case PHAST_ERR_C2R_IN_IM_LEN: return "input_im must have length N/2 + 1";
    case PHAST_ERR_C2R_SCRATCH_RE: return "scratch_re must have length N/2";
    case PHAST_ERR_C2R_SCRATCH_IM: return "scratch_im must have length N/2";
    case PHAST_ERR_ALLOC: return "host allocation failed";
    case PHAST_ERR_HIP: return "HIP runtime error (see phast_last_hip_error)";
    case PHAST_ERR_NO_DEVICE: return "no HIP device visible: libphastft_hip has no CPU fallback";
    case PHAST_ERR_INVALID_ARG: return "invalid argument";
    default: return "unknown error";
    }
}

const char *phast_last_hip_error(void) { return g_hip_err; }

int phast_hip_graph_upload(void *graph_exec, void *stream) try {
    if (!graph_exec) return PHAST_ERR_INVALID_ARG;
    PHAST_HIP(hipGraphUpload(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
    return PHAST_OK;
} PHAST_CATCH_RC

int phast_stream_probe_dev(const void *d_a, void *d_b, size_t bytes, int reps, double *out_gbps, void *stream) try {
    if (!d_a || !d_b || !out_gbps || reps < 1 || bytes < ((size_t)1 << 20) || (bytes & 15)) return PHAST_ERR_INVALID_ARG;
    int dev = 0;
    int rc = ensure_device(&dev);
    if (rc) return rc;
    PHAST_HIP(stream_probe(d_a, d_b, bytes, reps, cus_of(dev), out_gbps, static_cast<hipStream_t>(stream)));
    return PHAST_OK;
} PHAST_CATCH_RC

int phast_debug_throw(int what) try {
    if (what == 1) throw std::bad_alloc();
    if (what == 2) throw std::runtime_error("phast_debug_throw");
    if (what == 3) throw 3;
    return PHAST_OK;
} PHAST_CATCH_RC

void phast_debug_set_guard_bytes(size_t bytes) try { g_guard_bytes = (bytes + 255) & ~(size_t)255; } PHAST_CATCH_VOID

void phast_debug_set_wg_per_cu(int wg_per_cu) try { g_wg_per_cu_override = wg_per_cu; } PHAST_CATCH_VOID
void phast_debug_set_trace(unsigned long long *d_trace) try { g_trace = d_trace; } PHAST_CATCH_VOID

int phast_device_info(char *name, size_t name_len, int *compute_units, size_t *lds_per_block,
                      size_t *global_mem_bytes) try {
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    PHAST_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PHAST_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (lds_per_block) *lds_per_block = prop.sharedMemPerBlock;
    if (global_mem_bytes) *global_mem_bytes = prop.totalGlobalMem;
    return PHAST_OK;
} PHAST_CATCH_RC

void phast_options_default(phast_options *out) try {
    if (!out) return;
    out->multithreaded_bit_reversal = 0;
    out->smallest_parallel_chunk_size = 16384;
} PHAST_CATCH_VOID

int phast_options_guess(size_t input_size, phast_options *out) try {
    if (!out) return PHAST_ERR_INVALID_ARG;
    if (input_size == 0) return PHAST_ERR_NOT_POW2;  // usize::ilog2(0) panics (options.rs:40)
    phast_options_default(out);
    out->multithreaded_bit_reversal = ilog2(input_size) >= 16;
    return PHAST_OK;
} PHAST_CATCH_RC

#define PHAST_PLANNER_API(SFX, T)                                                                                       \
    int phast_planner_dit##SFX##_new(size_t n, phast_planner_dit##SFX **out) try {                                      \
        return planner_new(n, out);                                                                                     \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_dit##SFX##_with_mode(size_t n, int mode, phast_planner_dit##SFX **out) try {                      \
        if (mode != PHAST_MODE_HEURISTIC && mode != PHAST_MODE_TUNE) return PHAST_ERR_INVALID_ARG;                      \
        int rc = planner_new(n, out);                                                                                   \
        if (rc == PHAST_OK && mode == PHAST_MODE_TUNE) rc = tune_new(*out, kC2C);                                       \
        if (rc != PHAST_OK && out && *out) {                                                                            \
            delete *out;                                                                                                \
            *out = nullptr;                                                                                             \
        }                                                                                                               \
        return rc;                                                                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_dit##SFX##_tune(phast_planner_dit##SFX *p, size_t batch, int kind, phast_tune_report *rep) try {  \
        if (!p) return PHAST_ERR_INVALID_ARG;                                                                           \
        Planner<T>::TuneReport r;                                                                                       \
        int rc = p->tune(kind, batch, &r);                                                                              \
        if (rc == PHAST_OK) report_to_c(r, rep);                                                                        \
        return rc;                                                                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    void phast_planner_dit##SFX##_free(phast_planner_dit##SFX *p) try { delete p; } PHAST_CATCH_VOID                    \
    size_t phast_planner_dit##SFX##_device_bytes(const phast_planner_dit##SFX *p) try {                                 \
        return p ? p->device_bytes() : 0;                                                                               \
    } PHAST_CATCH_ZERO                                                                                                  \
    int phast_planner_dit##SFX##_debug_check_guards(const phast_planner_dit##SFX *p, size_t *bad_bytes) try {           \
        if (!p || !bad_bytes) return PHAST_ERR_INVALID_ARG;                                                             \
        return p->check_guards(bad_bytes);                                                                              \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_dit##SFX##_describe(const phast_planner_dit##SFX *p, char *buf, size_t len) try {                 \
        return describe_to<T>(p, buf, len);                                                                             \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_dit##SFX##_describe_call(const phast_planner_dit##SFX *p, size_t batch, int kind, char *buf,      \
                                               size_t len) try {                                                        \
        if (!p || !buf || !len || (kind != kC2C && kind != kC2CI)) return PHAST_ERR_INVALID_ARG;                        \
        std::snprintf(buf, len, "%s", p->describe_call(kind, batch).c_str());                                           \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_describe_call(const phast_planner_r2c##SFX *p, size_t batch, int kind, char *buf,      \
                                               size_t len) try {                                                        \
        if (!p || !buf || !len || (kind != kR2C && kind != kC2R)) return PHAST_ERR_INVALID_ARG;                         \
        const PlannerR2c<T> *q = p->route_small(kind == kC2R, batch);                                                   \
        std::snprintf(buf, len, "%s", q->dit.passes.empty() ? "one-pass" : q->dit.describe_call(kind, batch).c_str());  \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_dit##SFX##_reserve_batch(phast_planner_dit##SFX *p, size_t max_batch) try {                       \
        if (!p || max_batch == 0) return PHAST_ERR_INVALID_ARG;                                                         \
        return p->reserve_batch(max_batch);                                                                             \
    } PHAST_CATCH_RC                                                                                                    \
    size_t phast_planner_dit##SFX##_release_graph_workspaces(phast_planner_dit##SFX *p) try {                           \
        return p ? p->release_graph_workspaces() : 0;                                                                   \
    } PHAST_CATCH_ZERO                                                                                                  \
    int phast_planner_dit##SFX##_set_plan(phast_planner_dit##SFX *p, const unsigned *lr, const unsigned *tl,            \
                                          size_t np, unsigned points_log) try {                                         \
        return set_plan_c<T>(p, lr, tl, np, points_log);                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_dit##SFX##_time_passes(const phast_planner_dit##SFX *p, T *d_re, T *d_im, size_t batch,           \
                                             size_t dist, int reps, float *pass_ms, int *n_passes, void *stream) try {  \
        return time_passes<T>(p, d_re, d_im, batch, dist, reps, pass_ms, n_passes,                                      \
                              static_cast<hipStream_t>(stream));                                                        \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_new(size_t n, phast_planner_r2c##SFX **out) try {                                      \
        return r2c_planner_new(n, out);                                                                                 \
    } PHAST_CATCH_RC                                                                                                    \
    void phast_planner_r2c##SFX##_free(phast_planner_r2c##SFX *p) try { delete p; } PHAST_CATCH_VOID                    \
    int phast_planner_r2c##SFX##_with_mode(size_t n, int mode, phast_planner_r2c##SFX **out) try {                      \
        if (mode != PHAST_MODE_HEURISTIC && mode != PHAST_MODE_TUNE) return PHAST_ERR_INVALID_ARG;                      \
        int rc = r2c_planner_new(n, out);                                                                               \
        if (rc == PHAST_OK && mode == PHAST_MODE_TUNE) rc = tune_new(*out, kR2C);                                       \
        if (rc == PHAST_OK && mode == PHAST_MODE_TUNE) rc = tune_new(*out, kC2R);                                       \
        if (rc != PHAST_OK && out && *out) {                                                                            \
            delete *out;                                                                                                \
            *out = nullptr;                                                                                             \
        }                                                                                                               \
        return rc;                                                                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_tune(phast_planner_r2c##SFX *p, size_t batch, int kind, phast_tune_report *rep) try {  \
        if (!p) return PHAST_ERR_INVALID_ARG;                                                                           \
        Planner<T>::TuneReport r;                                                                                       \
        int rc = p->tune(kind, batch, &r);                                                                              \
        if (rc == PHAST_OK) report_to_c(r, rep);                                                                        \
        return rc;                                                                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_time_passes(const phast_planner_r2c##SFX *p, const T *d_in, T *d_ore, T *d_oim,        \
                                             size_t batch, size_t in_dist, size_t out_dist, int reps,                   \
                                             float *pass_ms, int *n_passes, void *stream) try {                         \
        return time_passes_r2c<T>(p, d_in, d_ore, d_oim, batch, in_dist, out_dist, reps, pass_ms, n_passes,             \
                                  static_cast<hipStream_t>(stream));                                                    \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_time_c2r_passes(const phast_planner_r2c##SFX *p, const T *d_ire, const T *d_iim,       \
                                                 T *d_out, size_t batch, size_t in_dist, size_t out_dist, int reps,     \
                                                 float *pass_ms, int *n_passes, void *stream) try {                     \
        return time_passes_c2r<T>(p, d_ire, d_iim, d_out, batch, in_dist, out_dist, reps, pass_ms, n_passes,            \
                                  static_cast<hipStream_t>(stream));                                                    \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_set_inner_plan(phast_planner_r2c##SFX *p, const unsigned *lr, const unsigned *tl,      \
                                                size_t np, unsigned points_log) try {                                   \
        if (!p) return PHAST_ERR_INVALID_ARG;                                                                           \
        PlannerR2c<T> *q = (p->dit.passes.empty() && p->twin) ? p->twin.get() : p;                                      \
        int rc = set_plan_c<T>(&q->dit, lr, tl, np, points_log);                                                        \
        if (rc == PHAST_OK && np == 0 && !q->dit.passes.empty()) rc = q->dit.make_c2r_plans();                          \
        return rc;                                                                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_planner_r2c##SFX##_describe(const phast_planner_r2c##SFX *p, char *buf, size_t len) try {                 \
        if (!p || !buf || !len) return PHAST_ERR_INVALID_ARG;                                                           \
        std::string s = p->dit.describe();                                                                              \
        if (p->twin)                                                                                                    \
            s += std::string(p->route_small(false) != p ? " | one transform: " : " | one c2r transform: ") +            \
                 p->twin->dit.describe();                                                                               \
        std::snprintf(buf, len, "%s", s.c_str());                                                                       \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC

PHAST_PLANNER_API(64, double)
PHAST_PLANNER_API(32, float)

#define PHAST_FFT_API(SFX, FS, T)                                                                                       \
    int phast_fft_##SFX##_dit(T *re, size_t re_len, T *im, size_t im_len, int direction) try {                          \
        return fft_host_noplanner<T>(re, re_len, im, im_len, direction);                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_dit_with_planner(T *re, size_t re_len, T *im, size_t im_len, int direction,                   \
                                           const phast_planner_dit##SFX *pl) try {                                      \
        return fft_host<T>(re, re_len, im, im_len, direction, pl);                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_dit_with_planner_and_opts(T *re, size_t re_len, T *im, size_t im_len, int direction,          \
                                                    const phast_planner_dit##SFX *pl, const phast_options *opts) try {  \
        if (!opts) return PHAST_ERR_INVALID_ARG;                                                                        \
        return fft_host<T>(re, re_len, im, im_len, direction, pl);                                                      \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_dit_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, int direction,                 \
                                  const phast_planner_dit##SFX *pl, void *stream) try {                                 \
        return fft_dev<T>(d_re, d_im, n, batch, dist, direction, pl, static_cast<hipStream_t>(stream));                 \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_dit_many_dev(T *const *d_re, T *const *d_im, size_t count, size_t n, int direction,           \
                                       const phast_planner_dit##SFX *pl, void *stream) try {                            \
        return fft_dev_many<T>(d_re, d_im, count, n, direction, pl, static_cast<hipStream_t>(stream));                  \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_dit_strided_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, size_t stride,         \
                                          int direction, const phast_planner_dit##SFX *pl, void *stream) try {          \
        return fft_strided_dev<T>(d_re, d_im, n, batch, dist, stride, direction, pl,                                    \
                                  static_cast<hipStream_t>(stream));                                                    \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_dit_strided_tw_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, size_t stride,      \
                                             int direction, const phast_planner_dit##SFX *pl, size_t tw_n,              \
                                             size_t tw_col0, void *stream) try {                                        \
        if (tw_n == 0) return PHAST_ERR_INVALID_ARG;                                                                    \
        return fft_strided_dev<T>(d_re, d_im, n, batch, dist, stride, direction, pl,                                    \
                                  static_cast<hipStream_t>(stream), tw_n, tw_col0);                                     \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_interleaved(T *signal, size_t n, int direction) try {                                         \
        std::shared_ptr<Planner<T>> pl; /* lib.rs:121: a planner per call -- kept, see PlannerCache */                  \
        int rc = PlannerCache<Planner<T>>::instance().get(                                                              \
            n, sizeof(T), [](size_t m, Planner<T> **o) { return planner_new(m, o); }, &pl);                             \
        if (rc) return rc;                                                                                              \
        return fft_interleaved_host<T>(signal, n, direction, pl.get());                                                 \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_interleaved_with_planner(T *signal, size_t n, int direction,                                  \
                                                   const phast_planner_dit##SFX *pl) try {                              \
        return fft_interleaved_host<T>(signal, n, direction, pl);                                                       \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_interleaved_with_planner_and_opts(T *signal, size_t n, int direction,                         \
                                                            const phast_planner_dit##SFX *pl,                           \
                                                            const phast_options *opts) try {                            \
        if (!opts) return PHAST_ERR_INVALID_ARG;                                                                        \
        return fft_interleaved_host<T>(signal, n, direction, pl);                                                       \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fft_##SFX##_interleaved_dev(T *d_signal, size_t n, size_t batch, size_t dist, int direction,              \
                                          const phast_planner_dit##SFX *pl, void *stream) try {                         \
        return fft_interleaved_dev<T>(d_signal, n, batch, dist, direction, pl, static_cast<hipStream_t>(stream));       \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_bit_rev_##FS(T *data, size_t len, unsigned log_n) try { return bitrev_host<T>(data, len, log_n); } PHAST_CATCH_RC \
    int phast_bit_rev_##FS##_dev(T *d, unsigned log_n, size_t batch, size_t dist, void *stream) try {                   \
        if (!d || log_n > 31 || (batch > 1 && dist < ((size_t)1 << log_n))) return PHAST_ERR_INVALID_ARG;               \
        int rc = ensure_device();                                                                                       \
        if (rc) return rc;                                                                                              \
        PHAST_HIP(launch_bitrev<T>(d, log_n, batch, dist, static_cast<hipStream_t>(stream)));                           \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_deinterleave_##FS(const T *in, size_t len, T *a, size_t a_len, T *b, size_t b_len) try {                  \
        return deinterleave_host<T>(in, len, a, a_len, b, b_len);                                                       \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_deinterleave_##FS##_dev(const T *d_in, size_t len, T *d_a, T *d_b, void *stream) try {                    \
        if (len < 2) return PHAST_OK; /* chunks_exact(2) of fewer than two elements: nothing */                         \
        if (!d_in || !d_a || !d_b) return PHAST_ERR_INVALID_ARG;                                                        \
        int rc = ensure_device();                                                                                       \
        if (rc) return rc;                                                                                              \
        PHAST_HIP(launch_deinterleave<T>(d_in, d_a, d_b, len / 2, static_cast<hipStream_t>(stream)));                   \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_combine_re_im_##FS(const T *re, size_t re_len, const T *im, size_t im_len, T *out, size_t out_len) try {  \
        return combine_host<T>(re, re_len, im, im_len, out, out_len);                                                   \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_combine_re_im_##FS##_dev(const T *d_re, const T *d_im, size_t n, T *d_out, void *stream) try {            \
        if (n == 0) return PHAST_OK;                                                                                    \
        if (!d_re || !d_im || !d_out) return PHAST_ERR_INVALID_ARG;                                                     \
        int rc = ensure_device();                                                                                       \
        if (rc) return rc;                                                                                              \
        PHAST_HIP(launch_combine<T>(d_re, d_im, d_out, n, static_cast<hipStream_t>(stream)));                           \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_r2c_fft_##FS(const T *in, size_t in_len, T *ore, size_t ore_len, T *oim, size_t oim_len) try {            \
        std::shared_ptr<PlannerR2c<T>> pl; /* r2c.rs:522: planner from input_re.len() */                                \
        int rc = PlannerCache<PlannerR2c<T>>::instance().get(                                                           \
            in_len, sizeof(T), [](size_t m, PlannerR2c<T> **o) { return r2c_planner_new(m, o); }, &pl);                 \
        if (rc) return rc;                                                                                              \
        return r2c_host<T>(in, in_len, ore, ore_len, oim, oim_len, pl.get());                                           \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_r2c_fft_##FS##_with_planner(const T *in, size_t in_len, T *ore, size_t ore_len, T *oim,                   \
                                          size_t oim_len, const phast_planner_r2c##SFX *pl) try {                       \
        return r2c_host<T>(in, in_len, ore, ore_len, oim, oim_len, pl);                                                 \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_r2c_fft_##FS##_dev(const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist, size_t out_dist,      \
                                 const phast_planner_r2c##SFX *pl, void *stream) try {                                  \
        if (!pl || !d_in || !d_ore || !d_oim) return PHAST_ERR_INVALID_ARG;                                             \
        if (batch > 1 && (in_dist < pl->n || out_dist < pl->n / 2 + 1)) return PHAST_ERR_INVALID_ARG;                   \
        return pl->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, static_cast<hipStream_t>(stream));                 \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_c2r_fft_##FS(const T *ire, size_t ire_len, const T *iim, size_t iim_len, T *out, size_t out_len) try {    \
        std::shared_ptr<PlannerR2c<T>> pl; /* r2c.rs:696: planner from output.len() */                                  \
        int rc = PlannerCache<PlannerR2c<T>>::instance().get(                                                           \
            out_len, sizeof(T), [](size_t m, PlannerR2c<T> **o) { return r2c_planner_new(m, o); }, &pl);                \
        if (rc) return rc;                                                                                              \
        return c2r_host<T>(ire, ire_len, iim, iim_len, out, out_len, pl.get(), false, 0, 0);                            \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_c2r_fft_##FS##_with_planner(const T *ire, size_t ire_len, const T *iim, size_t iim_len, T *out,           \
                                          size_t out_len, const phast_planner_r2c##SFX *pl) try {                       \
        return c2r_host<T>(ire, ire_len, iim, iim_len, out, out_len, pl, false, 0, 0);                                  \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_c2r_fft_##FS##_with_planner_and_scratch(const T *ire, size_t ire_len, const T *iim, size_t iim_len,       \
                                                      T *out, size_t out_len, const phast_planner_r2c##SFX *pl,         \
                                                      T *sre, size_t sre_len, T *sim, size_t sim_len) try {             \
        (void)sre;                                                                                                      \
        (void)sim;                                                                                                      \
        return c2r_host<T>(ire, ire_len, iim, iim_len, out, out_len, pl, true, sre_len, sim_len);                       \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_c2r_fft_##FS##_dev(const T *d_ire, const T *d_iim, T *d_out, size_t batch, size_t in_dist,                \
                                 size_t out_dist, const phast_planner_r2c##SFX *pl, void *stream) try {                 \
        if (!pl || !d_ire || !d_iim || !d_out) return PHAST_ERR_INVALID_ARG;                                            \
        if (batch > 1 && (in_dist < pl->n / 2 + 1 || out_dist < pl->n)) return PHAST_ERR_INVALID_ARG;                   \
        return pl->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, static_cast<hipStream_t>(stream));                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_fill_##FS##_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, unsigned long long seed,           \
                              unsigned long long first_id, void *stream) try {                                          \
        if (!d_re) return PHAST_ERR_INVALID_ARG;                                                                        \
        int rc = ensure_device();                                                                                       \
        if (rc) return rc;                                                                                              \
        PHAST_HIP(launch_fill<T>(d_re, d_im, n, batch, dist, seed, first_id, static_cast<hipStream_t>(stream)));        \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC                                                                                                    \
    int phast_digest_##FS##_dev(const T *d_re, const T *d_im, size_t n, size_t batch, size_t dist, size_t probe,        \
                                double *d_digest, void *stream) try {                                                   \
        if (!d_re || !d_im || !d_digest) return PHAST_ERR_INVALID_ARG;                                                  \
        int rc = ensure_device();                                                                                       \
        if (rc) return rc;                                                                                              \
        PHAST_HIP(launch_digest<T>(d_re, d_im, n, batch, dist, probe, d_digest, static_cast<hipStream_t>(stream)));     \
        return PHAST_OK;                                                                                                \
    } PHAST_CATCH_RC

PHAST_FFT_API(64, f64, double)
PHAST_FFT_API(32, f32, float)

#define PHAST_TWIDDLE_API(SFX, T)                                                                                       \
    int phast_twiddle_grid##SFX##_new(size_t n, phast_twiddle_grid##SFX **out) try {                                    \
        return planner_new(n, out);                                                                                     \
    } PHAST_CATCH_RC                                                                                                    \
    void phast_twiddle_grid##SFX##_free(phast_twiddle_grid##SFX *g) try { delete g; } PHAST_CATCH_VOID                  \
    int phast_twiddle_grid##SFX##_apply_dev(const phast_twiddle_grid##SFX *g, T *d_re, T *d_im, size_t rows,            \
                                            size_t cols, size_t row_pitch, size_t row0, size_t col0, void *stream) try { \
        if (!g) return PHAST_ERR_INVALID_ARG;                                                                           \
        return g->apply(d_re, d_im, rows, cols, row_pitch, row0, col0, static_cast<hipStream_t>(stream));               \
    } PHAST_CATCH_RC
PHAST_TWIDDLE_API(64, double)
PHAST_TWIDDLE_API(32, float)

}  // extern "C"
