// host_util.hpp -- error plumbing, the device guard and small helpers of the host side of libphastft_hip.so (one translation
// unit: c_abi.hip).  There is NO CPU fallback in this library: without a gfx950 device every compute entry point returns
// PHAST_ERR_NO_DEVICE / PHAST_ERR_HIP.
#pragma once

#include "../../include/phastft_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"
#include "tile_dispatch.hpp"
#include "r2c_fused.hpp"
#include "c2r_fused.hpp"
#include "wisdom.hpp"

namespace phast {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_hip_err[256] = "";

static int hip_fail(hipError_t e, const char *what) {
    std::snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
    return e == hipErrorNoDevice ? PHAST_ERR_NO_DEVICE : PHAST_ERR_HIP;
}
#define PHAST_HIP(call)                                     \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(e_, #call);   \
    } while (0)

static PerDeviceInt g_cus_of;  // CU count per device ordinal (device_state.hpp)
static int g_wg_per_cu_override = 0;           // tuning hook (phast_debug_set_wg_per_cu)
static unsigned long long *g_trace = nullptr;  // tuning hook (phast_debug_set_trace)

// Debug hook (phast_debug_set_guard_bytes): every scratch / workspace the planners allocate afterwards gets a guard band
// of this many bytes on either side, filled with 0xA5; phast_planner_*_debug_check_guards counts the bytes a kernel has
// overwritten.  The device-side half of the sanitizer pass: the boxes run gfx950 with xnack off,
// so ASan's device instrumentation is not available -- out-of-bounds WRITES of the pass kernels are caught by the bands.
static size_t g_guard_bytes = 0;
static constexpr unsigned char kGuardFill = 0xA5;

static int cus_of(int dev) {
    return g_cus_of.get(dev, [&] {
        hipDeviceProp_t prop;
        int c = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
        return c > 0 ? c : 256;
    });
}
// gfx number of the device as hex digits (gfx950 -> 0x950; wisdom.hpp: arch_from_name), -1 cached for "unknown" -> 0
static PerDeviceInt g_arch_of;
static int arch_of(int dev) {
    const int v = g_arch_of.get(dev, [&] {
        hipDeviceProp_t prop;
        int a = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) a = arch_from_name(prop.gcnArchName);
        return a > 0 ? a : -1;
    });
    return v > 0 ? v : 0;
}

// Is there a device at all, and which one is current?  The library holds no process-wide device: a planner belongs to
// the device that is current when it is created (its tables and scratch live there), several planners on several
// devices may coexist in one process (planner.rs:38-39: a planner is a plain value usable from any thread), and every
// call on a planner runs on the planner's device whatever the calling thread's current device is (DeviceGuard).
static int ensure_device(int *dev_out = nullptr) {
    static std::once_flag once;
    static int status = PHAST_OK;
    static char why[160] = "";
    std::call_once(once, [] {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count == 0) {
            std::snprintf(why, sizeof why, "hipGetDeviceCount: %s", hipGetErrorString(e));
            status = PHAST_ERR_NO_DEVICE;
        }
    });
    if (status != PHAST_OK) {
        std::snprintf(g_hip_err, sizeof g_hip_err, "%s", why);
        return status;
    }
    int dev = 0;
    PHAST_HIP(hipGetDevice(&dev));
    if (dev_out) *dev_out = dev;
    return PHAST_OK;
}

// Run a call on the device `want` owns its memory on; the caller's current device is restored on exit.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int want) {
        if (want < 0) return;
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) {
            err = hipSetDevice(want);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define PHAST_ON_DEVICE(dev)                                              \
    DeviceGuard device_guard_(dev);                                       \
    if (device_guard_.err != hipSuccess) return hip_fail(device_guard_.err, "hipSetDevice(planner's device)")

static inline bool is_pow2(size_t n) { return n != 0 && (n & (n - 1)) == 0; }
static inline unsigned ilog2(size_t n) { return 63u - (unsigned)__builtin_clzll((unsigned long long)n); }

template <typename T> static int upload(const std::vector<cx_t<T>> &h, void **d_out) {
    void *d = nullptr;
    PHAST_HIP(hipMalloc(&d, h.size() * sizeof(cx_t<T>)));
    hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(cx_t<T>), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(d);
        return hip_fail(e, "hipMemcpy(twiddles)");
    }
    *d_out = d;
    return PHAST_OK;
}

// a device allocation that goes with its scope
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) hipFree(p);
    }
    int alloc(size_t bytes) {
        PHAST_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return PHAST_OK;
    }
};

}  // namespace phast
