// phastft.hpp -- header-only C++17 host side above the C ABI (phastft_hip.h), mirroring PhastFT 0.3.0's
// public Rust API for the planar FFT path: same names, argument meaning and error behaviour.
//
//   Rust (reference file:line)                               C++ (namespace phastft)
//   ---------------------------------------------------      -------------------------------------------
//   planner::Direction {Forward = 1, Reverse = -1}  planner.rs:10    enum class Direction
//   planner::PlannerMode {Heuristic, Tune}          planner.rs:24    enum class PlannerMode
//   options::Options, Options::guess_options        options.rs:10    struct Options, Options::guess_options
//   PlannerDit64/32::{new, with_mode}               planner.rs:55    class PlannerDit64/32 (ctor, with_mode; + tune, wisdom_*)
//   PlannerR2c64/32::new                            planner.rs:194   class PlannerR2c64/32
//   fft_64_dit, fft_32_dit                          lib.rs:180,223   fft_64_dit, fft_32_dit
//   fft_*_dit_with_planner[_and_opts]               lib.rs:143,186; dit.rs:263,338
//   r2c_fft_f64/f32[_with_planner]                  r2c.rs:521,535,598,607
//   c2r_fft_f64/f32[_with_planner[_and_scratch]]    r2c.rs:695,710,740,804,813,836
//   fft_*_interleaved[_with_planner[_and_opts]]     lib.rs:50,87,120 (feature complex-nums)
//   bit_rev_bravo_f64/f32                           bravo.rs:303,317 (feature bench-internals)
//   deinterleave[_complex64/32], combine_re_im      complex_nums.rs:11,25,37,47 (feature bench-internals)
//
// A Rust `&mut [T]` is a (pointer, length) pair here -- `Slice<T>` converts from std::vector / std::array /
// raw pointer + length.  Where the reference panics (`assert!`), these functions throw `phastft::Panic` whose
// what() is the reference's panic message; HIP failures throw `phastft::HipError`.  There is no CPU fallback.
#ifndef PHASTFT_HPP
#define PHASTFT_HPP

#include <complex>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "phastft_hip.h"

namespace phastft {

struct Panic : std::logic_error {
    int code;
    Panic(int c, const std::string &m) : std::logic_error(m), code(c) {}
};
struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc == PHAST_OK) return;
    if (rc == PHAST_ERR_ALLOC || rc == PHAST_ERR_HIP || rc == PHAST_ERR_NO_DEVICE)
        throw HipError(rc, std::string(phast_strerror(rc)) + ": " + phast_last_hip_error());
    throw Panic(rc, phast_strerror(rc));
}

enum class Direction : int { Forward = PHAST_FORWARD, Reverse = PHAST_REVERSE };   // planner.rs:10-16
enum class PlannerMode : int { Heuristic = PHAST_MODE_HEURISTIC, Tune = PHAST_MODE_TUNE };  // planner.rs:24-32

// options.rs:8-43 -- CPU threading knobs, carried for source compatibility, ignored on the GPU
struct Options {
    bool multithreaded_bit_reversal = false;
    std::size_t smallest_parallel_chunk_size = 16384;
    static Options guess_options(std::size_t input_size) {
        phast_options o;
        check(phast_options_guess(input_size, &o));
        return Options{o.multithreaded_bit_reversal != 0, o.smallest_parallel_chunk_size};
    }
    phast_options to_c() const { return phast_options{multithreaded_bit_reversal ? 1 : 0, smallest_parallel_chunk_size}; }
};

// the Rust slice: pointer + length
template <typename T> struct Slice {
    T *ptr;
    std::size_t len;
    Slice(T *p, std::size_t n) : ptr(p), len(n) {}
    template <typename A> Slice(std::vector<T, A> &v) : ptr(v.data()), len(v.size()) {}
    template <typename U, typename A, typename = std::enable_if_t<std::is_same<const U, T>::value>>
    Slice(const std::vector<U, A> &v) : ptr(v.data()), len(v.size()) {}
};

#define PHASTFT_PLANNER(NAME, CT, NEW_EXPR, FREE)                                   \
    class NAME {                                                                    \
      public:                                                                       \
        NAME(const NAME &) = delete;                                                \
        NAME &operator=(const NAME &) = delete;                                     \
        NAME(NAME &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }                      \
        ~NAME() {                                                                   \
            if (h_) FREE(h_);                                                       \
        }                                                                           \
        const CT *get() const { return h_; }                                        \
        CT *get() { return h_; }                                                    \
        NEW_EXPR                                                                    \
      private:                                                                      \
        CT *h_ = nullptr;                                                           \
    };

// planner.rs:34-114
PHASTFT_PLANNER(PlannerDit64, phast_planner_dit64,
                explicit PlannerDit64(std::size_t num_points, PlannerMode mode = PlannerMode::Heuristic) {
                    check(phast_planner_dit64_with_mode(num_points, static_cast<int>(mode), &h_));
                } static PlannerDit64 with_mode(std::size_t n, PlannerMode mode) { return PlannerDit64(n, mode); }
                /* PlannerMode::Tune for another batch size / call kind (PHAST_TUNE_C2C, PHAST_TUNE_C2C_INTERLEAVED) */
                phast_tune_report tune(std::size_t batch = 1, int kind = PHAST_TUNE_C2C) {
                    phast_tune_report r{};
                    check(phast_planner_dit64_tune(h_, batch, kind, &r));
                    return r;
                },
                phast_planner_dit64_free)
PHASTFT_PLANNER(PlannerDit32, phast_planner_dit32,
                explicit PlannerDit32(std::size_t num_points, PlannerMode mode = PlannerMode::Heuristic) {
                    check(phast_planner_dit32_with_mode(num_points, static_cast<int>(mode), &h_));
                } static PlannerDit32 with_mode(std::size_t n, PlannerMode mode) { return PlannerDit32(n, mode); }
                /* PlannerMode::Tune for another batch size / call kind (PHAST_TUNE_C2C, PHAST_TUNE_C2C_INTERLEAVED) */
                phast_tune_report tune(std::size_t batch = 1, int kind = PHAST_TUNE_C2C) {
                    phast_tune_report r{};
                    check(phast_planner_dit32_tune(h_, batch, kind, &r));
                    return r;
                },
                phast_planner_dit32_free)
// planner.rs:164-212
PHASTFT_PLANNER(PlannerR2c64, phast_planner_r2c64,
                explicit PlannerR2c64(std::size_t n, PlannerMode mode = PlannerMode::Heuristic) {
                    check(phast_planner_r2c64_with_mode(n, static_cast<int>(mode), &h_));
                }
                phast_tune_report tune(std::size_t batch = 1, int kind = PHAST_TUNE_R2C) {
                    phast_tune_report r{};
                    check(phast_planner_r2c64_tune(h_, batch, kind, &r));
                    return r;
                },
                phast_planner_r2c64_free)
PHASTFT_PLANNER(PlannerR2c32, phast_planner_r2c32,
                explicit PlannerR2c32(std::size_t n, PlannerMode mode = PlannerMode::Heuristic) {
                    check(phast_planner_r2c32_with_mode(n, static_cast<int>(mode), &h_));
                }
                phast_tune_report tune(std::size_t batch = 1, int kind = PHAST_TUNE_R2C) {
                    phast_tune_report r{};
                    check(phast_planner_r2c32_tune(h_, batch, kind, &r));
                    return r;
                },
                phast_planner_r2c32_free)
#undef PHASTFT_PLANNER

// ---- wisdom: what PlannerMode::Tune measured, as text (csrc/wisdom.hpp; no reference counterpart) ----
inline std::string wisdom_export() {
    std::size_t need = 0;
    check(phast_wisdom_export(nullptr, 0, &need));
    std::string text(need, '\0');
    check(phast_wisdom_export(&text[0], need, nullptr));
    text.resize(need ? need - 1 : 0);
    return text;
}
inline void wisdom_import(const std::string &text) { check(phast_wisdom_import(text.c_str())); }
inline void wisdom_forget() { phast_wisdom_forget(); }

// ---- C2C, planar (lib.rs:143-226, algorithms/dit.rs:263,338) ----
inline void fft_64_dit_with_planner_and_opts(Slice<double> reals, Slice<double> imags, Direction direction,
                                             const PlannerDit64 &planner, const Options &opts) {
    const phast_options o = opts.to_c();
    check(phast_fft_64_dit_with_planner_and_opts(reals.ptr, reals.len, imags.ptr, imags.len, static_cast<int>(direction),
                                                 planner.get(), &o));
}
inline void fft_32_dit_with_planner_and_opts(Slice<float> reals, Slice<float> imags, Direction direction,
                                             const PlannerDit32 &planner, const Options &opts) {
    const phast_options o = opts.to_c();
    check(phast_fft_32_dit_with_planner_and_opts(reals.ptr, reals.len, imags.ptr, imags.len, static_cast<int>(direction),
                                                 planner.get(), &o));
}
inline void fft_64_dit_with_planner(Slice<double> reals, Slice<double> imags, Direction direction, const PlannerDit64 &planner) {
    check(phast_fft_64_dit_with_planner(reals.ptr, reals.len, imags.ptr, imags.len, static_cast<int>(direction), planner.get()));
}
inline void fft_32_dit_with_planner(Slice<float> reals, Slice<float> imags, Direction direction, const PlannerDit32 &planner) {
    check(phast_fft_32_dit_with_planner(reals.ptr, reals.len, imags.ptr, imags.len, static_cast<int>(direction), planner.get()));
}
inline void fft_64_dit(Slice<double> reals, Slice<double> imags, Direction direction) {
    check(phast_fft_64_dit(reals.ptr, reals.len, imags.ptr, imags.len, static_cast<int>(direction)));
}
inline void fft_32_dit(Slice<float> reals, Slice<float> imags, Direction direction) {
    check(phast_fft_32_dit(reals.ptr, reals.len, imags.ptr, imags.len, static_cast<int>(direction)));
}

// ---- C2C, interleaved Complex<T> (lib.rs:41-140) ----
inline void fft_64_interleaved(Slice<std::complex<double>> signal, Direction direction) {
    check(phast_fft_64_interleaved(reinterpret_cast<double *>(signal.ptr), signal.len, static_cast<int>(direction)));
}
inline void fft_32_interleaved(Slice<std::complex<float>> signal, Direction direction) {
    check(phast_fft_32_interleaved(reinterpret_cast<float *>(signal.ptr), signal.len, static_cast<int>(direction)));
}
inline void fft_64_interleaved_with_planner(Slice<std::complex<double>> signal, Direction direction, const PlannerDit64 &planner) {
    check(phast_fft_64_interleaved_with_planner(reinterpret_cast<double *>(signal.ptr), signal.len, static_cast<int>(direction),
                                                planner.get()));
}
inline void fft_32_interleaved_with_planner(Slice<std::complex<float>> signal, Direction direction, const PlannerDit32 &planner) {
    check(phast_fft_32_interleaved_with_planner(reinterpret_cast<float *>(signal.ptr), signal.len, static_cast<int>(direction),
                                                planner.get()));
}

inline void fft_64_interleaved_with_planner_and_opts(Slice<std::complex<double>> signal, Direction direction,
                                                     const PlannerDit64 &planner, const Options &opts) {  // lib.rs:50
    const phast_options o = opts.to_c();
    check(phast_fft_64_interleaved_with_planner_and_opts(reinterpret_cast<double *>(signal.ptr), signal.len,
                                                         static_cast<int>(direction), planner.get(), &o));
}
inline void fft_32_interleaved_with_planner_and_opts(Slice<std::complex<float>> signal, Direction direction,
                                                     const PlannerDit32 &planner, const Options &opts) {
    const phast_options o = opts.to_c();
    check(phast_fft_32_interleaved_with_planner_and_opts(reinterpret_cast<float *>(signal.ptr), signal.len,
                                                         static_cast<int>(direction), planner.get(), &o));
}

// ---- bit reversal (algorithms/bravo.rs:303,317) ----
// `n` is validated BEFORE it is used as a shift count (a shift by >= 64 is undefined; the reference asserts and panics)
inline void bit_rev_bravo_f64(Slice<double> data, unsigned n) {
    if (n >= 8 * sizeof(std::size_t) || data.len != (std::size_t(1) << n))
        throw Panic(PHAST_ERR_INVALID_ARG, "Data length must be 2^n");  // bravo.rs:228
    check(phast_bit_rev_f64(data.ptr, data.len, n));
}
inline void bit_rev_bravo_f32(Slice<float> data, unsigned n) {
    if (n >= 8 * sizeof(std::size_t) || data.len != (std::size_t(1) << n))
        throw Panic(PHAST_ERR_INVALID_ARG, "Data length must be 2^n");
    check(phast_bit_rev_f32(data.ptr, data.len, n));
}

// ---- Complex<T> <-> planes (complex_nums.rs:11-56; feature bench-internals) ----
template <typename T> struct ComplexNumsAbi;
template <> struct ComplexNumsAbi<double> {
    static int deinterleave(const double *in, std::size_t len, double *a, double *b) { return phast_deinterleave_f64(in, len, a, len / 2, b, len / 2); }
    static int combine(const double *re, std::size_t n, const double *im, std::size_t m, double *out) { return phast_combine_re_im_f64(re, n, im, m, out, 2 * n); }
};
template <> struct ComplexNumsAbi<float> {
    static int deinterleave(const float *in, std::size_t len, float *a, float *b) { return phast_deinterleave_f32(in, len, a, len / 2, b, len / 2); }
    static int combine(const float *re, std::size_t n, const float *im, std::size_t m, float *out) { return phast_combine_re_im_f32(re, n, im, m, out, 2 * n); }
};
// complex_nums.rs:11 -- [1, 2, 3, 4] -> ([1, 3], [2, 4]); an odd last element is dropped (chunks_exact(2))
template <typename T> inline std::pair<std::vector<T>, std::vector<T>> deinterleave(Slice<const T> input) {
    std::pair<std::vector<T>, std::vector<T>> out{std::vector<T>(input.len / 2), std::vector<T>(input.len / 2)};
    check(ComplexNumsAbi<T>::deinterleave(input.ptr, input.len, out.first.data(), out.second.data()));
    return out;
}
// complex_nums.rs:25,37 -- std::complex<T> is layout-compatible with T[2] (re, im), as Complex<T> is repr(C)
inline std::pair<std::vector<double>, std::vector<double>> deinterleave_complex64(Slice<const std::complex<double>> signal) {
    return deinterleave<double>(Slice<const double>(reinterpret_cast<const double *>(signal.ptr), 2 * signal.len));
}
inline std::pair<std::vector<float>, std::vector<float>> deinterleave_complex32(Slice<const std::complex<float>> signal) {
    return deinterleave<float>(Slice<const float>(reinterpret_cast<const float *>(signal.ptr), 2 * signal.len));
}
// complex_nums.rs:47 -- panics unless reals.len() == imags.len()
template <typename T> inline std::vector<std::complex<T>> combine_re_im(Slice<const T> reals, Slice<const T> imags) {
    if (reals.len != imags.len) throw Panic(PHAST_ERR_LEN_MISMATCH, "assertion `left == right` failed");  // complex_nums.rs:48
    std::vector<std::complex<T>> out(reals.len);
    check(ComplexNumsAbi<T>::combine(reals.ptr, reals.len, imags.ptr, imags.len, reinterpret_cast<T *>(out.data())));
    return out;
}

// ---- R2C / C2R (algorithms/r2c.rs:521-895) ----
inline void r2c_fft_f64_with_planner(Slice<const double> input_re, Slice<double> output_re, Slice<double> output_im,
                                     const PlannerR2c64 &planner) {
    check(phast_r2c_fft_f64_with_planner(input_re.ptr, input_re.len, output_re.ptr, output_re.len, output_im.ptr, output_im.len,
                                         planner.get()));
}
inline void r2c_fft_f32_with_planner(Slice<const float> input_re, Slice<float> output_re, Slice<float> output_im,
                                     const PlannerR2c32 &planner) {
    check(phast_r2c_fft_f32_with_planner(input_re.ptr, input_re.len, output_re.ptr, output_re.len, output_im.ptr, output_im.len,
                                         planner.get()));
}
inline void r2c_fft_f64(Slice<const double> input_re, Slice<double> output_re, Slice<double> output_im) {
    check(phast_r2c_fft_f64(input_re.ptr, input_re.len, output_re.ptr, output_re.len, output_im.ptr, output_im.len));
}
inline void r2c_fft_f32(Slice<const float> input_re, Slice<float> output_re, Slice<float> output_im) {
    check(phast_r2c_fft_f32(input_re.ptr, input_re.len, output_re.ptr, output_re.len, output_im.ptr, output_im.len));
}
inline void c2r_fft_f64_with_planner_and_scratch(Slice<const double> input_re, Slice<const double> input_im, Slice<double> output,
                                                 const PlannerR2c64 &planner, Slice<double> scratch_re, Slice<double> scratch_im) {
    check(phast_c2r_fft_f64_with_planner_and_scratch(input_re.ptr, input_re.len, input_im.ptr, input_im.len, output.ptr, output.len,
                                                     planner.get(), scratch_re.ptr, scratch_re.len, scratch_im.ptr, scratch_im.len));
}
inline void c2r_fft_f32_with_planner_and_scratch(Slice<const float> input_re, Slice<const float> input_im, Slice<float> output,
                                                 const PlannerR2c32 &planner, Slice<float> scratch_re, Slice<float> scratch_im) {
    check(phast_c2r_fft_f32_with_planner_and_scratch(input_re.ptr, input_re.len, input_im.ptr, input_im.len, output.ptr, output.len,
                                                     planner.get(), scratch_re.ptr, scratch_re.len, scratch_im.ptr, scratch_im.len));
}
inline void c2r_fft_f64_with_planner(Slice<const double> input_re, Slice<const double> input_im, Slice<double> output,
                                     const PlannerR2c64 &planner) {
    check(phast_c2r_fft_f64_with_planner(input_re.ptr, input_re.len, input_im.ptr, input_im.len, output.ptr, output.len, planner.get()));
}
inline void c2r_fft_f32_with_planner(Slice<const float> input_re, Slice<const float> input_im, Slice<float> output,
                                     const PlannerR2c32 &planner) {
    check(phast_c2r_fft_f32_with_planner(input_re.ptr, input_re.len, input_im.ptr, input_im.len, output.ptr, output.len, planner.get()));
}
inline void c2r_fft_f64(Slice<const double> input_re, Slice<const double> input_im, Slice<double> output) {
    check(phast_c2r_fft_f64(input_re.ptr, input_re.len, input_im.ptr, input_im.len, output.ptr, output.len));
}
inline void c2r_fft_f32(Slice<const float> input_re, Slice<const float> input_im, Slice<float> output) {
    check(phast_c2r_fft_f32(input_re.ptr, input_re.len, input_im.ptr, input_im.len, output.ptr, output.len));
}

}  // namespace phastft
#endif
