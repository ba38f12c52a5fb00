// device_state_test.cpp -- the per-device caches of libphastft_hip.so (phastft_amd/csrc/device_state.hpp) driven with
// FAKE device ordinals on the CPU: what lets one process hold planners on several GPUs (planner.rs:38-39: a planner is
// a plain value usable from any thread) is that "was this kernel's dynamic-LDS limit raised" and "how many CUs" are
// keyed by the device, and that two host threads launching the same kernel for the first time do not race.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "device_state.hpp"

#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
            std::exit(1);                                                 \
        }                                                                 \
    } while (0)

int main() {
    using namespace phast;
    {  // keyed by device: raising on device 0 says nothing about device 3
        PerDeviceLimit lim;
        int raises[kMaxDevices] = {0};
        auto raise_on = [&](int dev) { return [&raises, dev](size_t) { ++raises[dev]; return 0; }; };
        CHECK(lim.ensure(0, 1000, raise_on(0)) == 0 && raises[0] == 1);
        CHECK(lim.ensure(0, 1000, raise_on(0)) == 0 && raises[0] == 1);  // steady state: no call
        CHECK(lim.ensure(0, 500, raise_on(0)) == 0 && raises[0] == 1);   // monotone
        CHECK(lim.ensure(3, 1000, raise_on(3)) == 0 && raises[3] == 1);  // another device: its own raise
        CHECK(lim.ensure(0, 2000, raise_on(0)) == 0 && raises[0] == 2);  // growth raises again
        CHECK(lim.get(0) == 2000 && lim.get(3) == 1000 && lim.get(5) == 0);
        // a failing raise is not remembered
        CHECK(lim.ensure(5, 10, [](size_t) { return 7; }) == 7 && lim.get(5) == 0);
        CHECK(lim.ensure(5, 10, raise_on(5)) == 0 && raises[5] == 1);
        CHECK(lim.ensure(-1, 10, raise_on(0)) == -1 && lim.ensure(kMaxDevices, 10, raise_on(0)) == -1);
    }
    {  // first launches from many threads: exactly one raise per (device, growth), nobody proceeds before it is done
        PerDeviceLimit lim;
        std::atomic<int> raises[4];
        std::atomic<bool> up[4];
        for (int d = 0; d < 4; ++d) raises[d] = 0, up[d] = false;
        std::atomic<int> early{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 16; ++t)
            th.emplace_back([&, t] {
                const int dev = t & 3;
                for (int rep = 0; rep < 1000; ++rep) {
                    lim.ensure(dev, 4096, [&](size_t) {
                        std::this_thread::yield();
                        ++raises[dev];
                        up[dev] = true;
                        return 0;
                    });
                    if (!up[dev]) ++early;  // "launched" before the limit of ITS device was raised
                }
            });
        for (auto &x : th) x.join();
        for (int d = 0; d < 4; ++d) CHECK(raises[d] == 1);
        CHECK(early == 0);
    }
    {  // CU count: computed once per device
        PerDeviceInt cus;
        int calls = 0;
        CHECK(cus.get(0, [&] { ++calls; return 256; }) == 256);
        CHECK(cus.get(0, [&] { ++calls; return 1; }) == 256 && calls == 1);
        CHECK(cus.get(1, [&] { ++calls; return 304; }) == 304 && calls == 2);
    }
    std::printf("device_state: ok\n");
    return 0;
}
