// shard_host.cpp -- a TORCH-FREE host for BASELINE configs[4] ("Batch of 8192 independent f64 FFTs at N=2^20 sharded across
// 8 x MI355X, 1024 per GPU"), written against the C ABI alone: what north_star's "Rust host calls hand-written HIP kernels
// through a thin C-ABI FFI layer ... one shard per GPU, RCCL over xGMI only for the trivial gather" looks like in the language
// this image can compile (rust/phastft-hip/examples/shard.rs is the same program as Rust source).
//
//   one host thread per visible device (the reference's only parallel construct is rayon::join, parallel.rs:13-24; its planners
//   are plain values usable from any thread, planner.rs:38-39) -- each thread: hipSetDevice, its own planner, its shard of
//   `shard` transforms generated on the device (phast_fill_f64_dev, transform ids [g * shard, (g + 1) * shard)),
//   K timed steps of phast_fft_64_dit_dev between thread barriers, then a fresh step + phast_digest_f64_dev and ONE
//   ncclAllGather of the 32-byte digests (librccl.so, communicators from ncclCommInitAll) -- no data-path collective.
//
// Prints bench.py's JSON line (value = samples of ALL devices / MAX-over-devices time) and writes every gathered digest to
// --digests-out for the checker: tests/test_gpu_parity_r5.py compares them with digests of the CPU oracle's output.
//
//   hipcc -O2 -std=c++17 tests/cpp/shard_host.cpp -I include -L phastft_amd/lib -lphastft_hip -L /opt/rocm/lib -lrccl -pthread
//   ./shard_host [--gpus G] [--shard 1024] [--steps 10] [--warmup 2] [--log-n 20] [--digests-out file]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "phastft_hip.h"

namespace {

struct Barrier {  // (std::barrier is C++20)
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long generation = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
};

// [first, first + count) of `total` units for rank r of `world`: contiguous, sizes differ by at most one
// (phastft_amd/sharding.py: shard_bounds)
void shard_bounds(size_t total, int r, int world, size_t *first, size_t *count) {
    const size_t base = total / (size_t)world, extra = total % (size_t)world;
    *first = (size_t)r * base + ((size_t)r < extra ? (size_t)r : extra);
    *count = base + ((size_t)r < extra ? 1 : 0);
}

std::atomic<int> g_failed{0};
#define CHECK_HIP(x)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            std::fprintf(stderr, "shard_host: %s: %s\n", #x, hipGetErrorString(e_));          \
            g_failed = 1;                                                                     \
            return;                                                                           \
        }                                                                                     \
    } while (0)
#define CHECK_PHAST(x)                                                                                        \
    do {                                                                                                      \
        int rc_ = (x);                                                                                        \
        if (rc_ != PHAST_OK) {                                                                                \
            std::fprintf(stderr, "shard_host: %s: %s (%s)\n", #x, phast_strerror(rc_), phast_last_hip_error()); \
            g_failed = 1;                                                                                     \
            return;                                                                                           \
        }                                                                                                     \
    } while (0)
#define CHECK_NCCL(x)                                                                         \
    do {                                                                                      \
        ncclResult_t r_ = (x);                                                                \
        if (r_ != ncclSuccess) {                                                              \
            std::fprintf(stderr, "shard_host: %s: %s\n", #x, ncclGetErrorString(r_));         \
            g_failed = 1;                                                                     \
            return;                                                                           \
        }                                                                                     \
    } while (0)

}  // namespace

int main(int argc, char **argv) {
    int want_gpus = 0, steps = 10, warmup = 2, log_n = 20;
    size_t shard = 1024;
    std::string digests_out;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char *flag) -> const char * { return (!std::strcmp(argv[i], flag) && i + 1 < argc) ? argv[++i] : nullptr; };
        if (const char *v = val("--gpus")) want_gpus = std::atoi(v);
        else if (const char *v = val("--shard")) shard = (size_t)std::atoll(v);
        else if (const char *v = val("--steps")) steps = std::atoi(v);
        else if (const char *v = val("--warmup")) warmup = std::atoi(v);
        else if (const char *v = val("--log-n")) log_n = std::atoi(v);
        else if (const char *v = val("--digests-out")) digests_out = v;
        else {
            std::fprintf(stderr, "usage: shard_host [--gpus G] [--shard S] [--steps K] [--warmup W] [--log-n L] [--digests-out file]\n");
            return 2;
        }
    }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) {
        std::fprintf(stderr, "shard_host: no HIP device visible (the library has no CPU path)\n");
        return 2;
    }
    const int G = want_gpus > 0 ? want_gpus : have;
    if (G > have) {  // never report a G-GPU number measured on fewer devices (bench.py: fail())
        std::fprintf(stderr, "shard_host: --gpus %d but %d device(s) visible; nothing measured\n", G, have);
        return 2;
    }
    const size_t n = (size_t)1 << log_n, total = shard * (size_t)G;
    std::vector<int> devs(G);
    for (int g = 0; g < G; ++g) devs[g] = g;
    std::vector<ncclComm_t> comms(G);
    if (ncclCommInitAll(comms.data(), G, devs.data()) != ncclSuccess) {
        std::fprintf(stderr, "shard_host: ncclCommInitAll failed\n");
        return 1;
    }
    Barrier bar(G);
    std::vector<double> seconds(G, 0.0);
    std::vector<double> all_digests;  // device 0's copy of the gather: [total][4]
    std::vector<std::string> plans(G);

    auto worker = [&](int g) {
        size_t first = 0, count = 0;
        shard_bounds(total, g, G, &first, &count);
        CHECK_HIP(hipSetDevice(g));
        hipStream_t st;
        CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        phast_planner_dit64 *pl = nullptr;
        CHECK_PHAST(phast_planner_dit64_new(n, &pl));  // one planner per device: tables and scratch live where they are used
        double *re = nullptr, *im = nullptr, *dig = nullptr, *dig_all = nullptr;
        CHECK_HIP(hipMalloc((void **)&re, count * n * sizeof(double)));
        CHECK_HIP(hipMalloc((void **)&im, count * n * sizeof(double)));
        CHECK_HIP(hipMalloc((void **)&dig, count * 4 * sizeof(double)));
        CHECK_HIP(hipMalloc((void **)&dig_all, total * 4 * sizeof(double)));
        auto refill = [&] { return phast_fill_f64_dev(re, im, n, count, n, 0xCAFEull, first, st); };
        auto step = [&] { return phast_fft_64_dit_dev(re, im, n, count, n, PHAST_FORWARD, pl, st); };
        CHECK_PHAST(refill());
        for (int i = 0; i < warmup; ++i) CHECK_PHAST(step());
        CHECK_PHAST(refill());  // values grow by sqrt(N) per in-place step: K timed steps from fresh inputs stay far from overflow
        CHECK_HIP(hipStreamSynchronize(st));
        bar.wait();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) CHECK_PHAST(step());
        CHECK_HIP(hipStreamSynchronize(st));
        seconds[g] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bar.wait();
        // after the timed region: one step on fresh inputs, its digests, and the only collective of the job
        CHECK_PHAST(refill());
        CHECK_PHAST(step());
        CHECK_PHAST(phast_digest_f64_dev(re, im, n, count, n, 1, dig, st));
        CHECK_NCCL(ncclAllGather(dig, dig_all, count * 4, ncclDouble, comms[g], st));  // (equal shards: total = shard * G)
        CHECK_HIP(hipStreamSynchronize(st));
        if (g == 0) {
            all_digests.resize(total * 4);
            CHECK_HIP(hipMemcpy(all_digests.data(), dig_all, total * 4 * sizeof(double), hipMemcpyDeviceToHost));
        }
        char buf[1024];
        if (phast_planner_dit64_describe_call(pl, count, PHAST_TUNE_C2C, buf, sizeof buf) == PHAST_OK) plans[g] = buf;
        bar.wait();
        phast_planner_dit64_free(pl);
        for (void *p : {(void *)re, (void *)im, (void *)dig, (void *)dig_all}) (void)hipFree(p);
        (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> threads;
    for (int g = 0; g < G; ++g) threads.emplace_back([&, g] {
        worker(g);
        if (g_failed) std::_Exit(1);  // a failed device thread would leave the others at a barrier
    });
    for (auto &t : threads) t.join();
    int comm_count = 0, nccl_version = 0;
    (void)ncclCommCount(comms[0], &comm_count);
    (void)ncclGetVersion(&nccl_version);
    for (int g = 0; g < G; ++g) ncclCommDestroy(comms[g]);
    if (g_failed) return 1;

    double worst = 0;
    for (double s : seconds) worst = s > worst ? s : worst;
    bool finite = all_digests.size() == total * 4;
    for (double d : all_digests) finite = finite && d == d && d - d == 0.0;
    if (!digests_out.empty()) {
        FILE *f = std::fopen(digests_out.c_str(), "wb");
        if (!f || std::fwrite(all_digests.data(), sizeof(double), all_digests.size(), f) != all_digests.size()) {
            std::fprintf(stderr, "shard_host: cannot write %s\n", digests_out.c_str());
            return 1;
        }
        std::fclose(f);
    }
    const double ms = 1e3 * worst / steps, value = (double)total * (double)n * steps / worst / 1e9;
    double fastest = worst;
    for (double s : seconds) fastest = s < fastest ? s : fastest;
    // the self-diagnosing keys of the N > 1 line (tests/golden/multi_gpu_line.schema.json; bench.py --gpus N prints the same
    // ones): every device thread's own time, the ranks the communicator really has, the collective library's version
    const int ranks_seen = comm_count;
    char name[128] = "";
    int cus = 0;
    phast_device_info(name, sizeof name, &cus, nullptr, nullptr);
    std::printf("{\"metric\": \"GSamples/s f64 forward FFT N=2^%d\", \"value\": %.4f, \"unit\": \"GSamples/s\", \"n_gpus\": %d, "
                "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"weak\", "
                "\"vs_baseline\": null, \"dtype\": \"f64\", \"data\": \"synthetic (counter-based uniform [-1,1), seed 0xCAFE, generated on device)\", "
                "\"config\": {\"workload\": \"%zu independent f64 forward FFTs N=2^%d, %zu per GPU, in place (BASELINE configs[4])\", "
                "\"host\": \"tests/cpp/shard_host.cpp: one C++ thread per device over the C ABI, no torch\", "
                "\"digest_gather\": \"ncclAllGather of %zu x 32 B digests (librccl, ncclCommInitAll over %d device(s))\", "
                "\"digest_ok\": %s, \"plan_used\": \"%s\", \"device\": \"%s\", "
                "\"rank_ms_min\": %.6f, \"rank_ms_max\": %.6f, \"ranks_seen\": %d, \"backend\": \"rccl\", \"shard\": %zu, "
                "\"rccl_version\": \"%d.%d.%d\"}}\n",
                log_n, value, G, steps, warmup, ms, total, log_n, shard, total, G, finite ? "true" : "false", plans[0].c_str(), name,
                1e3 * fastest / steps, ms, ranks_seen, shard, nccl_version / 10000, (nccl_version / 100) % 100, nccl_version % 100);
    return finite ? 0 : 1;
}
