// tune_beside_callers_test.cpp -- a tuning run on a planner that other threads are calling (round 5).  Three threads keep
// batches of 16 x 2^18 f64 transforms going through ONE planner, each on its own stream and buffers; the main thread tunes that
// very call (built-in wisdom off: the static rule's plan loses there, so a plan is installed under them), then another bucket,
// then makes a second planner with PHAST_MODE_TUNE while the first is still in use.  Every call must succeed and every
// transform's energy must be N x its input's (Parseval, through phast_digest_f64_dev) whichever plan ran it.  The Python
// counterpart (tests/test_gpu_parity_r5.py) checks every bin against a float64 reference; this program exists to run the same
// interleaving under ThreadSanitizer / AddressSanitizer (tools/sanitize_host.sh).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "phastft_hip.h"
#include "sanitizer_exit.hpp"

#define CHECK(x)                                                                                             \
    do {                                                                                                     \
        int rc_ = (int)(x);                                                                                  \
        if (rc_ != 0) {                                                                                      \
            std::printf("FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, phast_last_hip_error()); \
            std::fflush(stdout);                                                                             \
            std::_Exit(1);                                                                                   \
        }                                                                                                    \
    } while (0)

int main() {
    if (std::getenv("PHAST_TSAN_CONTROL")) {  // positive control for the sanitizer pass: a real race, no GPU needed -- TSan must report it
        static int racy = 0;
        std::thread a([] { for (int i = 0; i < 100000; ++i) ++racy; }), b([] { for (int i = 0; i < 100000; ++i) ++racy; });
        a.join();
        b.join();
        std::printf("control: racy = %d\n", racy);
        return 0;
    }
    const size_t n = (size_t)1 << 18, batch = 16;
    const int T = 3;
    phast_wisdom_builtin(0);  // the static rules alone: the tuning run below has something to install
    phast_planner_dit64 *pl = nullptr;
    CHECK(phast_planner_dit64_new(n, &pl));
    std::atomic<bool> stop{false};
    std::atomic<long> calls[T];
    std::atomic<int> bad{0};
    for (auto &c : calls) c = 0;
    auto caller = [&](int k) {
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        double *re, *im, *dig;
        CHECK(hipMalloc((void **)&re, batch * n * sizeof(double)));
        CHECK(hipMalloc((void **)&im, batch * n * sizeof(double)));
        CHECK(hipMalloc((void **)&dig, 2 * batch * 4 * sizeof(double)));
        std::vector<double> h(2 * batch * 4);
        while (!stop.load()) {
            CHECK(phast_fill_f64_dev(re, im, n, batch, n, 0xCAFEull + (unsigned)k, (unsigned long long)calls[k].load() * batch, s));
            CHECK(phast_digest_f64_dev(re, im, n, batch, n, 1, dig, s));
            CHECK(phast_fft_64_dit_dev(re, im, n, batch, n, PHAST_FORWARD, pl, s));
            CHECK(phast_digest_f64_dev(re, im, n, batch, n, 1, dig + batch * 4, s));
            CHECK(hipMemcpyAsync(h.data(), dig, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            CHECK(hipStreamSynchronize(s));
            for (size_t b = 0; b < batch; ++b) {
                const double e_in = h[b * 4 + 2], e_out = h[(batch + b) * 4 + 2];
                if (!(e_in > 0) || !(std::fabs(e_out / ((double)n * e_in) - 1.0) < 1e-12)) ++bad;
            }
            ++calls[k];
        }
        CHECK(hipFree(re));
        CHECK(hipFree(im));
        CHECK(hipFree(dig));
        CHECK(hipStreamDestroy(s));
    };
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k) th.emplace_back(caller, k);
    auto least = [&] {
        long m = calls[0].load();
        for (auto &c : calls) m = c.load() < m ? c.load() : m;
        return m;
    };
    while (least() < 3) std::this_thread::yield();
    phast_tune_report rep{}, rep2{};
    CHECK(phast_planner_dit64_tune(pl, batch, PHAST_TUNE_C2C, &rep));
    const long during = least();
    CHECK(phast_planner_dit64_tune(pl, 2, PHAST_TUNE_C2C, &rep2));
    phast_planner_dit64 *pl2 = nullptr;  // PlannerMode::Tune at construction, the first planner still in use
    CHECK(phast_planner_dit64_with_mode(n, PHAST_MODE_TUNE, &pl2));
    char what[512];
    CHECK(phast_planner_dit64_describe_call(pl, batch, PHAST_TUNE_C2C, what, sizeof what));
    const long mid = least();
    while (least() < mid + 10) std::this_thread::yield();
    stop = true;
    for (auto &t : th) t.join();
    phast_planner_dit64_free(pl2);
    phast_planner_dit64_free(pl);
    phast_wisdom_forget();
    std::printf("{\"threads\": %d, \"calls\": [%ld, %ld, %ld], \"calls_during_first_tune_at_least\": %ld, \"bad_transforms\": %d, "
                "\"tune\": {\"adopted\": %d, \"plan\": \"%s\", \"us_heuristic\": %.1f, \"us_best\": %.1f, \"candidates\": %u}, "
                "\"tune_x2\": \"%s\", \"call_now\": \"%s\"}\n",
                T, calls[0].load(), calls[1].load(), calls[2].load(), during, bad.load(), rep.adopted, rep.plan, rep.us_heuristic, rep.us_best,
                rep.candidates, rep2.plan, what);
    phast_test_exit(bad.load() == 0 ? 0 : 1);
}
