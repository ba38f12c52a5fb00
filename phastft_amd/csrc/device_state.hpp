// device_state.hpp -- per-DEVICE host-side state of libphastft_hip.so.
//
// PhastFT's planners are plain values usable from any thread (planner.rs:38-39); a host that drives several GPUs from
// one process (one thread per device) must be able to hold planners on all of them.  What the
// library caches per device: the CU count (grid sizes) and, per kernel instantiation, the dynamic-LDS limit already
// raised with hipFuncSetAttribute (an attribute of the function ON THE CURRENT DEVICE).  Both are keyed by the device
// ordinal here, and the raise is serialised: two host threads launching the same instantiation for the first time must
// not race on "was it raised yet" (one of them would launch before the limit is up and fail).
//
// No HIP types in this header: tests/cpp/device_state_test.cpp compiles it with g++ and drives it with fake device ids.
#pragma once

#include <atomic>
#include <cstddef>
#include <mutex>

namespace phast {

constexpr int kMaxDevices = 64;

// A monotone per-device limit.  ensure(dev, want, raise): make the limit of `dev` at least `want`, calling
// raise(want) -> int (0 = ok) at most once per growth; the steady state is one relaxed atomic load.
class PerDeviceLimit {
  public:
    PerDeviceLimit() {
        for (auto &c : cur_) c.store(0, std::memory_order_relaxed);
    }
    template <typename Raise> int ensure(int dev, size_t want, Raise &&raise) {
        if (dev < 0 || dev >= kMaxDevices) return -1;
        if (cur_[dev].load(std::memory_order_acquire) >= want) return 0;
        std::lock_guard<std::mutex> lk(mu_);
        if (cur_[dev].load(std::memory_order_relaxed) >= want) return 0;
        const int rc = raise(want);
        if (rc == 0) cur_[dev].store(want, std::memory_order_release);
        return rc;
    }
    size_t get(int dev) const { return (dev < 0 || dev >= kMaxDevices) ? 0 : cur_[dev].load(std::memory_order_acquire); }

  private:
    std::mutex mu_;
    std::atomic<size_t> cur_[kMaxDevices];
};

// A per-device value computed once per device (the CU count).  get(dev, compute): compute() runs once per device.
class PerDeviceInt {
  public:
    PerDeviceInt() {
        for (auto &c : val_) c.store(0, std::memory_order_relaxed);
    }
    template <typename Compute> int get(int dev, Compute &&compute) {
        if (dev < 0 || dev >= kMaxDevices) return compute();
        int v = val_[dev].load(std::memory_order_acquire);
        if (v) return v;
        std::lock_guard<std::mutex> lk(mu_);
        v = val_[dev].load(std::memory_order_relaxed);
        if (v) return v;
        v = compute();
        val_[dev].store(v, std::memory_order_release);
        return v;
    }

  private:
    std::mutex mu_;
    std::atomic<int> val_[kMaxDevices];
};

}  // namespace phast
