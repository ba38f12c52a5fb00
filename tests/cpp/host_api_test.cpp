// host_api_test.cpp -- the reference's own tests (lib.rs:238-461, r2c.rs:914-1540), re-read through the C++ host
// side (include/phastft.hpp) over libphastft_hip.so.  Built by tests/test_cpp_host.py; with a GPU it runs the
// numerical checks, without one it checks that argument asserts still panic and compute calls fail loudly.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <complex>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "phastft.hpp"
#include "sanitizer_exit.hpp"

using namespace phastft;

static int failures = 0;
#define EXPECT(cond)                                                            \
    do {                                                                        \
        if (!(cond)) {                                                          \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);         \
            ++failures;                                                         \
        }                                                                       \
    } while (0)

template <typename F> static std::string panic_message(F &&f) {
    try {
        f();
    } catch (const Panic &p) {
        return p.what();
    } catch (const HipError &e) {
        return std::string("HipError: ") + e.what();
    }
    return "<no panic>";
}

// O(N^2) long-double DFT: the independent oracle for small N (RustFFT's role in lib.rs:298-338)
static void naive_dft(const std::vector<double> &re, const std::vector<double> &im, std::vector<double> &ore,
                      std::vector<double> &oim) {
    const size_t n = re.size();
    const long double tau = 6.283185307179586476925286766559005768L;
    ore.assign(n, 0);
    oim.assign(n, 0);
    for (size_t k = 0; k < n; ++k) {
        long double sr = 0, si = 0;
        for (size_t j = 0; j < n; ++j) {
            const long double a = -tau * (long double)((k * j) % n) / (long double)n;
            const long double c = cosl(a), s = sinl(a);
            sr += re[j] * c - im[j] * s;
            si += re[j] * s + im[j] * c;
        }
        ore[k] = (double)sr;
        oim[k] = (double)si;
    }
}

int main(int argc, char **argv) {
    const bool have_gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;

    // ---- panics that need no device (planner.rs:66,195; lib.rs:238-257; bravo.rs:228) ----
    EXPECT(panic_message([] { PlannerDit64 p(5); }) == "assertion failed: num_points > 0 && num_points.is_power_of_two()");
    EXPECT(panic_message([] { PlannerDit32 p(5); }) == "assertion failed: num_points > 0 && num_points.is_power_of_two()");
    EXPECT(panic_message([] { PlannerR2c64 p(6); }) == "n must be a power of 2 >= 4");
    EXPECT(panic_message([] { PlannerR2c32 p(2); }) == "n must be a power of 2 >= 4");
    {
        std::vector<double> d(10);
        EXPECT(panic_message([&] { bit_rev_bravo_f64(d, 3); }) == "Data length must be 2^n");
        // n is validated before it is used as a shift count (a shift by >= 64 is undefined behaviour)
        EXPECT(panic_message([&] { bit_rev_bravo_f64(d, 64); }) == "Data length must be 2^n");
        EXPECT(panic_message([&] { bit_rev_bravo_f64(d, 4000000000u); }) == "Data length must be 2^n");
        std::vector<float> f(10);
        EXPECT(panic_message([&] { bit_rev_bravo_f32(f, 77); }) == "Data length must be 2^n");
    }
    const Options o = Options::guess_options(size_t(1) << 20);  // options.rs:38-43
    EXPECT(o.multithreaded_bit_reversal && o.smallest_parallel_chunk_size == 16384);
    EXPECT(!Options::guess_options(size_t(1) << 15).multithreaded_bit_reversal);
    EXPECT(static_cast<int>(Direction::Forward) == 1 && static_cast<int>(Direction::Reverse) == -1);

    if (!have_gpu) {  // no CPU fallback: compute must fail loudly and leave the data alone
        std::vector<double> re(16, 1.0), im(16, 0.0);
        const std::string m = panic_message([&] { fft_64_dit(re, im, Direction::Forward); });
        EXPECT(m.rfind("HipError:", 0) == 0);
        EXPECT(re[3] == 1.0);
        std::printf("host_api_test (no GPU): %d failure(s)\n", failures);
        return failures ? 1 : 0;
    }

    // ---- fft_correctness (lib.rs:298-338): ramp vs an independent DFT, abs 0.01 ----
    for (int k = 4; k <= 10; ++k) {
        const size_t n = size_t(1) << k;
        std::vector<double> re(n), im(n), ore, oim;
        for (size_t i = 0; i < n; ++i) re[i] = im[i] = double(i + 1);
        naive_dft(re, im, ore, oim);
        fft_64_dit(re, im, Direction::Forward);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(re[i] - ore[i]) < 0.01 && std::fabs(im[i] - oim[i]) < 0.01);
    }
    // ---- round trip with a planner, both precisions (lib.rs:381-461) ----
    for (int k = 4; k <= 14; ++k) {
        const size_t n = size_t(1) << k;
        std::vector<double> re(n), im(n);
        double norm = 0;
        for (size_t i = 0; i < n; ++i) {
            re[i] = std::sin(0.37 * double(i) + 1.0);
            im[i] = std::cos(1.91 * double(i));
            norm += re[i] * re[i] + im[i] * im[i];
        }
        for (size_t i = 0; i < n; ++i) {
            re[i] /= std::sqrt(norm);
            im[i] /= std::sqrt(norm);
        }
        const std::vector<double> re0 = re, im0 = im;
        PlannerDit64 planner = PlannerDit64::with_mode(n, PlannerMode::Tune);
        fft_64_dit_with_planner(re, im, Direction::Forward, planner);
        fft_64_dit_with_planner_and_opts(re, im, Direction::Reverse, planner, Options::guess_options(n));
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(re[i] - re0[i]) < 1e-10 && std::fabs(im[i] - im0[i]) < 1e-10);
        std::vector<float> fr(re0.begin(), re0.end()), fi(im0.begin(), im0.end());
        const std::vector<float> fr0 = fr, fi0 = fi;
        fft_32_dit(fr, fi, Direction::Forward);
        fft_32_dit(fr, fi, Direction::Reverse);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(fr[i] - fr0[i]) < 1e-6f && std::fabs(fi[i] - fi0[i]) < 1e-6f);
    }
    // ---- wrong_num_points_in_planner (lib.rs:259-296) and mismatched slices ----
    {
        PlannerDit64 planner(16);
        std::vector<double> re(1 << 16), im(1 << 16);
        EXPECT(panic_message([&] { fft_64_dit_with_planner(re, im, Direction::Forward, planner); }) ==
               "assertion `left == right` failed: log_n == planner.log_n");
        std::vector<double> a(8), b(16);
        EXPECT(panic_message([&] { fft_64_dit(a, b, Direction::Forward); }) ==
               "assertion `left == right` failed: reals.len() == imags.len()");
    }
    // ---- interleaved == planar (lib.rs:340-378) ----
    {
        const size_t n = 1024;
        std::vector<std::complex<double>> sig(n);
        std::vector<double> re(n), im(n, 0.0);
        for (size_t i = 0; i < n; ++i) {
            sig[i] = {double(i + 1), 0.0};
            re[i] = double(i + 1);
        }
        fft_64_interleaved(sig, Direction::Forward);
        fft_64_dit(re, im, Direction::Forward);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(sig[i].real() - re[i]) < 1e-10 * 1e6 && std::fabs(sig[i].imag() - im[i]) < 1e-10 * 1e6);
        // the _with_planner_and_opts forms (lib.rs:50): forward then reverse gives the signal back
        PlannerDit64 planner(n);
        const Options opts = Options::guess_options(n);
        std::vector<std::complex<double>> sig2(n);
        for (size_t i = 0; i < n; ++i) sig2[i] = {double(i + 1), -0.5 * double(i)};
        std::vector<std::complex<double>> orig = sig2;
        fft_64_interleaved_with_planner_and_opts(sig2, Direction::Forward, planner, opts);
        fft_64_interleaved_with_planner_and_opts(sig2, Direction::Reverse, planner, opts);
        for (size_t i = 0; i < n; ++i) EXPECT(std::abs(sig2[i] - orig[i]) < 1e-9);
        PlannerDit32 planner32(n);
        std::vector<std::complex<float>> sig3(n), orig3;
        for (size_t i = 0; i < n; ++i) sig3[i] = {float(i % 7) - 3.0f, float(i % 5)};
        orig3 = sig3;
        fft_32_interleaved_with_planner_and_opts(sig3, Direction::Forward, planner32, opts);
        fft_32_interleaved_with_planner_and_opts(sig3, Direction::Reverse, planner32, opts);
        for (size_t i = 0; i < n; ++i) EXPECT(std::abs(sig3[i] - orig3[i]) < 1e-3f);
    }
    // ---- R2C known answers and panics (r2c.rs:1235-1540), R2C -> C2R round trip (r2c.rs:958-976) ----
    {
        const size_t n = 16, half = 8;
        std::vector<double> in(n, 1.0), ore(half + 1, 7.0), oim(half + 1, 7.0);
        r2c_fft_f64(in, ore, oim);  // dc_only
        EXPECT(std::fabs(ore[0] - double(n)) < 1e-10);
        for (size_t k = 1; k <= half; ++k) EXPECT(std::fabs(ore[k]) < 1e-10 && std::fabs(oim[k]) < 1e-10);
        for (size_t i = 0; i < n; ++i) in[i] = (i % 2 == 0) ? 1.0 : -1.0;  // nyquist_only
        r2c_fft_f64(in, ore, oim);
        for (size_t k = 0; k <= half; ++k) EXPECT(std::fabs(ore[k] - (k == half ? double(n) : 0.0)) < 1e-10 && std::fabs(oim[k]) < 1e-10);
        std::vector<double> bad(half);
        EXPECT(panic_message([&] { r2c_fft_f64(in, bad, oim); }) == "output_re must have length N/2 + 1");
        EXPECT(panic_message([&] { r2c_fft_f64(in, ore, bad); }) == "output_im must have length N/2 + 1");
        PlannerR2c64 planner(n);
        std::vector<double> shortin(8);
        EXPECT(panic_message([&] { r2c_fft_f64_with_planner(shortin, ore, oim, planner); }) == "input length must match planner size");
        std::vector<double> out(n), s7(7), s8(8);
        EXPECT(panic_message([&] { c2r_fft_f64_with_planner_and_scratch(ore, oim, out, planner, s7, s8); }) == "scratch_re must have length N/2");
        EXPECT(panic_message([&] { c2r_fft_f64_with_planner_and_scratch(ore, oim, out, planner, s8, s7); }) == "scratch_im must have length N/2");
        EXPECT(panic_message([&] { c2r_fft_f64(bad, oim, out); }) == "input_re must have length N/2 + 1");
        EXPECT(panic_message([&] { c2r_fft_f64(ore, bad, out); }) == "input_im must have length N/2 + 1");
        EXPECT(panic_message([&] { c2r_fft_f64_with_planner(ore, oim, s8, planner); }) == "output length must match planner size");
    }
    for (int k = 2; k <= 16; ++k) {
        const size_t n = size_t(1) << k;
        std::vector<double> x(n), ore(n / 2 + 1), oim(n / 2 + 1), back(n);
        for (size_t i = 0; i < n; ++i) x[i] = double(i + 1);
        r2c_fft_f64(x, ore, oim);
        c2r_fft_f64(ore, oim, back);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(back[i] - x[i]) < 1e-6);
        std::vector<float> xf(x.begin(), x.end()), fre(n / 2 + 1), fim(n / 2 + 1), fback(n);
        r2c_fft_f32(xf, fre, fim);
        c2r_fft_f32(fre, fim, fback);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(fback[i] - xf[i]) < 1e-2f * (1.0f + std::fabs(xf[i])));
    }
    // ---- the fused passes of the large real transforms (round 3: untangle in the last pass from 2^23 points, preprocess in
    // the first pass's load): host slices in, host slices out, bounded pseudo-random signal ----
    for (int k : {20, 24}) {
        const size_t n = size_t(1) << k;
        std::vector<float> xf(n), fre(n / 2 + 1), fim(n / 2 + 1), fback(n);
        for (size_t i = 0; i < n; ++i) xf[i] = float((uint32_t(i) * 2654435761u >> 8) & 0xffffu) / 65536.0f - 0.5f;
        PlannerR2c32 planner(n);
        r2c_fft_f32_with_planner(xf, fre, fim, planner);
        EXPECT(fim[0] == 0.0f && fim[n / 2] == 0.0f);
        double dc = 0;
        for (size_t i = 0; i < n; ++i) dc += xf[i];
        EXPECT(std::fabs(fre[0] - dc) < 1e-3 * std::sqrt(double(n)));
        c2r_fft_f32_with_planner(fre, fim, fback, planner);
        float worst = 0;
        for (size_t i = 0; i < n; ++i) worst = std::max(worst, std::fabs(fback[i] - xf[i]));
        EXPECT(worst < 1e-4f);
    }
    // ---- PlannerMode::Tune beyond one transform per call, the real planners, wisdom (round 5; planner.rs:18-32) ----
    {
        const size_t n = size_t(1) << 15;
        std::vector<double> x(n), ore(n / 2 + 1), oim(n / 2 + 1), back(n);
        for (size_t i = 0; i < n; ++i) x[i] = std::sin(0.001 * double(i)) + 0.25 * double(i % 11);
        PlannerR2c64 tuned(n, PlannerMode::Tune);            // r2c and c2r of one transform measured at plan time
        PlannerR2c64 plain(n);
        r2c_fft_f64_with_planner(x, ore, oim, tuned);
        std::vector<double> pre(n / 2 + 1), pim(n / 2 + 1);
        r2c_fft_f64_with_planner(x, pre, pim, plain);
        double worst = 0, peak = 0;
        for (size_t k = 0; k <= n / 2; ++k) {
            worst = std::max(worst, std::max(std::fabs(ore[k] - pre[k]), std::fabs(oim[k] - pim[k])));
            peak = std::max(peak, std::fabs(pre[k]));
        }
        EXPECT(worst <= 1e-12 * peak);                       // another plan: other last bits, the same transform
        c2r_fft_f64_with_planner(ore, oim, back, tuned);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(back[i] - x[i]) < 1e-9);
        const phast_tune_report rep = tuned.tune(8, PHAST_TUNE_C2R);   // a batch bucket, the other call kind
        EXPECT(rep.candidates >= 8 && rep.us_best <= rep.us_heuristic * 1.0001f && rep.seconds < 5.0);
        PlannerDit32 c32(size_t(1) << 16);
        const phast_tune_report r32 = c32.tune(4, PHAST_TUNE_C2C_INTERLEAVED);
        EXPECT(r32.candidates >= 8 && r32.plan[0] != '\0');
        EXPECT(panic_message([&] { (void)c32.tune(4, PHAST_TUNE_R2C); }) == "invalid argument");
        // what was measured travels as text
        const std::string text = wisdom_export();
        EXPECT(text.find("f64 c2r 15 3 ") != std::string::npos && text.find("f32 c2ci 16 2 ") != std::string::npos);
        wisdom_forget();
        wisdom_import(text);
        EXPECT(wisdom_export() == text);
        EXPECT(panic_message([] { wisdom_import("not wisdom"); }) == "invalid argument");
        PlannerR2c64 again(n);                               // starts with the imported plans
        c2r_fft_f64_with_planner(ore, oim, back, again);
        for (size_t i = 0; i < n; ++i) EXPECT(std::fabs(back[i] - x[i]) < 1e-9);
        wisdom_forget();
    }
    // ---- bit reversal exact (bravo.rs:373-407) ----
    for (unsigned nb = 2; nb <= 18; ++nb) {
        const size_t n = size_t(1) << nb;
        std::vector<double> d(n);
        for (size_t i = 0; i < n; ++i) d[i] = double(i);
        bit_rev_bravo_f64(d, nb);
        bool ok = true;
        for (size_t i = 0; i < n && ok; ++i) {
            size_t r = 0;
            for (unsigned b = 0; b < nb; ++b) r |= ((i >> b) & 1) << (nb - 1 - b);
            ok = d[i] == double(r);
        }
        EXPECT(ok);
    }
    // ---- Complex<T> <-> planes (complex_nums.rs:73-118): the reference's list of lengths, then separate -> combine ----
    for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(3), size_t(15), size_t(16), size_t(17), size_t(127), size_t(128), size_t(129),
                     size_t(130), size_t(135), size_t(100500)}) {
        std::vector<double> in(n);
        for (size_t i = 0; i < n; ++i) in[i] = double(i);
        const auto ab = deinterleave<double>(Slice<const double>(in.data(), in.size()));
        bool ok = ab.first.size() == n / 2 && ab.second.size() == n / 2;
        for (size_t i = 0; i < n / 2 && ok; ++i) ok = ab.first[i] == in[2 * i] && ab.second[i] == in[2 * i + 1];
        EXPECT(ok);
    }
    {
        std::vector<std::complex<float>> z(1000);
        for (size_t i = 0; i < z.size(); ++i) z[i] = {float(i) * 0.5f, -float(i)};
        const auto ri = deinterleave_complex32(Slice<const std::complex<float>>(z.data(), z.size()));
        const auto back = combine_re_im<float>(Slice<const float>(ri.first.data(), ri.first.size()), Slice<const float>(ri.second.data(), ri.second.size()));
        EXPECT(back == z);
        std::vector<float> shorter(999);
        EXPECT(panic_message([&] { combine_re_im<float>(Slice<const float>(ri.first.data(), ri.first.size()), Slice<const float>(shorter.data(), shorter.size())); }) ==
               "assertion `left == right` failed");
    }
    std::printf("host_api_test (GPU): %d failure(s)\n", failures);
    phast_test_exit(failures ? 1 : 0);
}
