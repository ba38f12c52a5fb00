// concurrent_planner_test.cpp -- ONE planner under concurrent callers (planner.rs:38-39: the reference's planner is an
// immutable value borrowed by `&`; algorithms/dit.rs:263 takes `&PlannerDit64`, so N threads transform N buffers at once).
// Plain C++ threads over the C ABI (a Python harness would measure the GIL): (a) blocking host-slice calls from 1 and 4
// threads, (b) _dev calls on 1 and 4 streams, each call followed by a stream synchronisation (a consumer loop), for several
// batch sizes.  Every result is compared bit for bit with the single-threaded one.  Prints one JSON line.
// Built and run by tests/test_gpu_parity_r4.py.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "phastft_hip.h"
#include "sanitizer_exit.hpp"

#define CHECK(x)                                                                        \
    do {                                                                                \
        int rc_ = (int)(x);                                                             \
        if (rc_ != 0) {                                                                 \
            std::printf("FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, phast_last_hip_error()); \
            std::exit(1);                                                               \
        }                                                                               \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double fill(std::vector<double> &v, unsigned seed) {
    unsigned long long s = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (auto &x : v) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        x = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
    return v[0];
}

template <typename F> static double run_threads(int threads, F &&body) {  // returns seconds
    std::vector<std::thread> ts;
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    for (int t = 0; t < threads; ++t)
        ts.emplace_back([&, t] {
            ready++;
            while (!go.load()) std::this_thread::yield();
            body(t);
        });
    while (ready.load() < threads) std::this_thread::yield();
    const double t0 = now();
    go = true;
    for (auto &t : ts) t.join();
    return now() - t0;
}

int main() {
    const size_t n = 1 << 16;
    const int T = 4;
    phast_planner_dit64 *pl = nullptr;
    CHECK(phast_planner_dit64_new(n, &pl));
    int ok = 1;

    // ---------------- (a) host-slice calls ----------------
    std::vector<std::vector<double>> in_re(T, std::vector<double>(n)), in_im(T, std::vector<double>(n)), ref_re(T), ref_im(T);
    for (int t = 0; t < T; ++t) {
        fill(in_re[t], 2 * t);
        fill(in_im[t], 2 * t + 1);
        ref_re[t] = in_re[t];
        ref_im[t] = in_im[t];
        CHECK(phast_fft_64_dit_with_planner(ref_re[t].data(), n, ref_im[t].data(), n, PHAST_FORWARD, pl));
    }
    auto host_loop = [&](int iters) {
        return [&, iters](int t) {
            std::vector<double> r(n), m(n);
            for (int i = 0; i < iters; ++i) {
                std::memcpy(r.data(), in_re[t].data(), n * 8);
                std::memcpy(m.data(), in_im[t].data(), n * 8);
                CHECK(phast_fft_64_dit_with_planner(r.data(), n, m.data(), n, PHAST_FORWARD, pl));
                if ((i & 7) == 0 && (std::memcmp(r.data(), ref_re[t].data(), n * 8) || std::memcmp(m.data(), ref_im[t].data(), n * 8))) ok = 0;
            }
        };
    };
    run_threads(T, host_loop(20));
    const int hi = 400;
    const double h1 = 1 * hi / run_threads(1, host_loop(hi));
    const double h4 = T * hi / run_threads(T, host_loop(hi));

    // ---------------- (b) _dev calls: forward + inverse per iteration, synchronised per call ----------------
    double d1[3], d4[3];
    const size_t batches[3] = {1, 8, 64};
    for (int bi = 0; bi < 3; ++bi) {
        const size_t batch = batches[bi], elems = n * batch;
        std::vector<hipStream_t> st(T);
        std::vector<double *> d_re(T), d_im(T);
        std::vector<std::vector<double>> h_re(T, std::vector<double>(elems)), h_im(T, std::vector<double>(elems));
        for (int t = 0; t < T; ++t) {
            CHECK(hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking));
            CHECK(hipMalloc((void **)&d_re[t], elems * 8));
            CHECK(hipMalloc((void **)&d_im[t], elems * 8));
            fill(h_re[t], 100 + 2 * t);
            fill(h_im[t], 101 + 2 * t);
        }
        auto reset = [&] {
            for (int t = 0; t < T; ++t) {
                CHECK(hipMemcpy(d_re[t], h_re[t].data(), elems * 8, hipMemcpyHostToDevice));
                CHECK(hipMemcpy(d_im[t], h_im[t].data(), elems * 8, hipMemcpyHostToDevice));
            }
        };
        auto dev_loop = [&](int iters) {
            return [&, iters](int t) {
                for (int i = 0; i < iters; ++i) {
                    CHECK(phast_fft_64_dit_dev(d_re[t], d_im[t], n, batch, n, (i & 1) ? PHAST_REVERSE : PHAST_FORWARD, pl, st[t]));
                    CHECK(hipStreamSynchronize(st[t]));
                }
            };
        };
        const int di = 301;  // odd: ends on a forward transform
        reset();
        run_threads(T, dev_loop(11));
        reset();
        d1[bi] = 1 * di / run_threads(1, dev_loop(di));
        std::vector<double> want_re(elems), want_im(elems), got(elems);
        CHECK(hipMemcpy(want_re.data(), d_re[0], elems * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(want_im.data(), d_im[0], elems * 8, hipMemcpyDeviceToHost));
        reset();
        d4[bi] = T * di / run_threads(T, dev_loop(di));
        CHECK(hipMemcpy(got.data(), d_re[0], elems * 8, hipMemcpyDeviceToHost));
        if (std::memcmp(got.data(), want_re.data(), elems * 8)) ok = 0;
        CHECK(hipMemcpy(got.data(), d_im[0], elems * 8, hipMemcpyDeviceToHost));
        if (std::memcmp(got.data(), want_im.data(), elems * 8)) ok = 0;
        for (int t = 0; t < T; ++t) {
            (void)hipFree(d_re[t]);
            (void)hipFree(d_im[t]);
            (void)hipStreamDestroy(st[t]);
        }
    }
    std::printf("{\"n\": %zu, \"threads\": %d, \"bit_identical\": %s, \"host_calls_per_s\": [%.1f, %.1f], "
                "\"dev_calls_per_s_batch1\": [%.1f, %.1f], \"dev_calls_per_s_batch8\": [%.1f, %.1f], "
                "\"dev_calls_per_s_batch64\": [%.1f, %.1f], \"device_bytes\": %zu}\n",
                n, T, ok ? "true" : "false", h1, h4, d1[0], d4[0], d1[1], d4[1], d1[2], d4[2], phast_planner_dit64_device_bytes(pl));
    phast_planner_dit64_free(pl);
    phast_test_exit(ok ? 0 : 1);
}
