// planner_stress_test.cpp -- the workspace pool under abuse (round 4).  Eight host threads share THREE planners (C2C f64 2^15
// and 2^17, R2C f32 2^16) and mix, at random: blocking host-slice calls, _dev calls on a private stream (batches of 1..6,
// results copied back and compared), R2C / C2R, and -- every few iterations -- destroy their stream and make a new one, as
// any caller may as soon as its work is done.  Every result is compared bit for bit with the one a single thread produced
// before the threads started.  Run with the default pool and with PHAST_MAX_WORKSPACES=2 (streams queue behind each other on
// the device), plain and under AddressSanitizer (tools/sanitize_host.sh).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "phastft_hip.h"
#include "sanitizer_exit.hpp"

#define CHECK(x)                                                                                                  \
    do {                                                                                                          \
        int rc_ = (int)(x);                                                                                       \
        if (rc_ != 0) {                                                                                           \
            std::printf("FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, phast_last_hip_error());      \
            std::fflush(stdout);                                                                                  \
            std::_Exit(1);                                                                                        \
        }                                                                                                         \
    } while (0)

struct Rng {
    unsigned long long s;
    unsigned next() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (unsigned)(s >> 33);
    }
    double unit() { return (double)(next() & 0xffffff) / 8388608.0 - 1.0; }
};

int main() {
    const int T = 8, ITERS = 160, K = 4;  // K distinct inputs per planner
    // (ONE transform -- or a few -- runs a plan of its own at most sizes, plan.hpp: single_plan: another factorisation, other
    //  last bits than the same transform inside a larger batch.  The references below are therefore kept PER BATCH SIZE.)
    const size_t n_a = 1 << 16, n_b = 1 << 17, n_r = 1 << 16, h1 = n_r / 2 + 1;
    const size_t MAXB = 6;
    phast_planner_dit64 *pa = nullptr, *pb = nullptr;
    phast_planner_r2c32 *pr = nullptr;
    CHECK(phast_planner_dit64_new(n_a, &pa));
    CHECK(phast_planner_dit64_new(n_b, &pb));
    CHECK(phast_planner_r2c32_new(n_r, &pr));
    Rng rng{12345};
    auto fill = [&](std::vector<double> &v) { for (auto &x : v) x = rng.unit(); };
    // inputs and single-threaded references
    std::vector<std::vector<double>> a_re(K, std::vector<double>(n_a)), a_im(K, std::vector<double>(n_a)), A_re(K), A_im(K);
    std::vector<std::vector<double>> b_re(K, std::vector<double>(n_b)), b_im(K, std::vector<double>(n_b)), B_re(K), B_im(K);
    std::vector<std::vector<float>> r_in(K, std::vector<float>(n_r)), R_re(K, std::vector<float>(h1)), R_im(K, std::vector<float>(h1)), R_back(K, std::vector<float>(n_r));
    for (int k = 0; k < K; ++k) {
        fill(a_re[k]); fill(a_im[k]); fill(b_re[k]); fill(b_im[k]);
        for (auto &x : r_in[k]) x = (float)rng.unit();
        A_re[k] = a_re[k]; A_im[k] = a_im[k]; B_re[k] = b_re[k]; B_im[k] = b_im[k];
        CHECK(phast_fft_64_dit_with_planner(A_re[k].data(), n_a, A_im[k].data(), n_a, PHAST_FORWARD, pa));
        CHECK(phast_fft_64_dit_with_planner(B_re[k].data(), n_b, B_im[k].data(), n_b, PHAST_FORWARD, pb));
        CHECK(phast_r2c_fft_f32_with_planner(r_in[k].data(), n_r, R_re[k].data(), h1, R_im[k].data(), h1, pr));
        CHECK(phast_c2r_fft_f32_with_planner(R_re[k].data(), h1, R_im[k].data(), h1, R_back[k].data(), n_r, pr));
    }
    // _dev references: [planner][input][batch - 1] = the transform of input k as every row of a batch of `batch` copies gives it
    std::vector<std::vector<std::vector<std::vector<double>>>> D_re(2), D_im(2);
    {
        double *t_re, *t_im;
        CHECK(hipMalloc((void **)&t_re, MAXB * n_b * 8));
        CHECK(hipMalloc((void **)&t_im, MAXB * n_b * 8));
        for (int big = 0; big < 2; ++big) {
            const size_t n = big ? n_b : n_a;
            D_re[big].assign(K, std::vector<std::vector<double>>(MAXB, std::vector<double>(n)));
            D_im[big].assign(K, std::vector<std::vector<double>>(MAXB, std::vector<double>(n)));
            for (int k = 0; k < K; ++k)
                for (size_t batch = 1; batch <= MAXB; ++batch) {
                    const auto &ir = big ? b_re[k] : a_re[k], &ii = big ? b_im[k] : a_im[k];
                    for (size_t b = 0; b < batch; ++b) {
                        CHECK(hipMemcpy(t_re + b * n, ir.data(), n * 8, hipMemcpyHostToDevice));
                        CHECK(hipMemcpy(t_im + b * n, ii.data(), n * 8, hipMemcpyHostToDevice));
                    }
                    CHECK(phast_fft_64_dit_dev(t_re, t_im, n, batch, n, PHAST_FORWARD, big ? pb : pa, nullptr));
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipMemcpy(D_re[big][k][batch - 1].data(), t_re + (batch - 1) * n, n * 8, hipMemcpyDeviceToHost));
                    CHECK(hipMemcpy(D_im[big][k][batch - 1].data(), t_im + (batch - 1) * n, n * 8, hipMemcpyDeviceToHost));
                }
        }
        (void)hipFree(t_re);
        (void)hipFree(t_im);
    }
    std::atomic<int> bad{0};
    std::atomic<long> ops{0};
    auto worker = [&](int t) {
        Rng r{(unsigned long long)(777 + 31 * t)};
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const size_t maxb = MAXB;
        double *d_re, *d_im;
        float *d_x, *d_ore, *d_oim, *d_y;
        CHECK(hipMalloc((void **)&d_re, maxb * n_b * 8));
        CHECK(hipMalloc((void **)&d_im, maxb * n_b * 8));
        CHECK(hipMalloc((void **)&d_x, n_r * 4));
        CHECK(hipMalloc((void **)&d_ore, h1 * 4));
        CHECK(hipMalloc((void **)&d_oim, h1 * 4));
        CHECK(hipMalloc((void **)&d_y, n_r * 4));
        std::vector<double> h_re(maxb * n_b), h_im(maxb * n_b);
        std::vector<float> f_re(h1), f_im(h1), f_y(n_r);
        for (int it = 0; it < ITERS; ++it) {
            const int k = (int)(r.next() % K), op = (int)(r.next() % 6);
            if (op == 0) {  // blocking host-slice call, planner a
                std::vector<double> x = a_re[k], y = a_im[k];
                CHECK(phast_fft_64_dit_with_planner(x.data(), n_a, y.data(), n_a, PHAST_FORWARD, pa));
                if (std::memcmp(x.data(), A_re[k].data(), n_a * 8) || std::memcmp(y.data(), A_im[k].data(), n_a * 8)) bad++;
            } else if (op == 1 || op == 2) {  // _dev call, batch of 1..6 copies of input k, planner a or b
                const bool big = op == 2;
                const size_t n = big ? n_b : n_a, batch = 1 + r.next() % maxb;
                const auto &ir = big ? b_re[k] : a_re[k], &ii = big ? b_im[k] : a_im[k];
                for (size_t b = 0; b < batch; ++b) {
                    CHECK(hipMemcpyAsync(d_re + b * n, ir.data(), n * 8, hipMemcpyHostToDevice, s));
                    CHECK(hipMemcpyAsync(d_im + b * n, ii.data(), n * 8, hipMemcpyHostToDevice, s));
                }
                CHECK(phast_fft_64_dit_dev(d_re, d_im, n, batch, n, PHAST_FORWARD, big ? pb : pa, s));
                CHECK(hipMemcpyAsync(h_re.data(), d_re, batch * n * 8, hipMemcpyDeviceToHost, s));
                CHECK(hipMemcpyAsync(h_im.data(), d_im, batch * n * 8, hipMemcpyDeviceToHost, s));
                CHECK(hipStreamSynchronize(s));
                const auto &wr = D_re[big][k][batch - 1], &wi = D_im[big][k][batch - 1];
                for (size_t b = 0; b < batch; ++b)
                    if (std::memcmp(h_re.data() + b * n, wr.data(), n * 8) || std::memcmp(h_im.data() + b * n, wi.data(), n * 8)) bad++;
            } else if (op == 3) {  // R2C on host slices
                CHECK(phast_r2c_fft_f32_with_planner(r_in[k].data(), n_r, f_re.data(), h1, f_im.data(), h1, pr));
                if (std::memcmp(f_re.data(), R_re[k].data(), h1 * 4) || std::memcmp(f_im.data(), R_im[k].data(), h1 * 4)) bad++;
            } else if (op == 4) {  // R2C then C2R on the stream
                CHECK(hipMemcpyAsync(d_x, r_in[k].data(), n_r * 4, hipMemcpyHostToDevice, s));
                CHECK(phast_r2c_fft_f32_dev(d_x, d_ore, d_oim, 1, n_r, h1, pr, s));
                CHECK(phast_c2r_fft_f32_dev(d_ore, d_oim, d_y, 1, h1, n_r, pr, s));
                CHECK(hipMemcpyAsync(f_y.data(), d_y, n_r * 4, hipMemcpyDeviceToHost, s));
                CHECK(hipStreamSynchronize(s));
                if (std::memcmp(f_y.data(), R_back[k].data(), n_r * 4)) bad++;
            } else {  // the caller is done with its stream: destroy it, make another
                CHECK(hipStreamSynchronize(s));
                CHECK(hipStreamDestroy(s));
                CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            }
            ops++;
        }
        CHECK(hipStreamSynchronize(s));
        (void)hipStreamDestroy(s);
        (void)hipFree(d_re); (void)hipFree(d_im); (void)hipFree(d_x); (void)hipFree(d_ore); (void)hipFree(d_oim); (void)hipFree(d_y);
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < T; ++t) ts.emplace_back(worker, t);
    for (auto &t : ts) t.join();
    std::printf("{\"threads\": %d, \"ops\": %ld, \"mismatches\": %d, \"device_bytes\": [%zu, %zu]}\n", T, ops.load(), bad.load(),
                phast_planner_dit64_device_bytes(pa), phast_planner_dit64_device_bytes(pb));
    phast_planner_dit64_free(pa);
    phast_planner_dit64_free(pb);
    phast_planner_r2c32_free(pr);
    phast_test_exit(bad.load() ? 1 : 0);
}
