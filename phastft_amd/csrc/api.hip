// api.hip -- host side of libphastft_hip.so: planners, pass plans, launches and the C ABI of
// include/phastft_hip.h.  Mirrors PhastFT's public surface (lib.rs:143-226, planner.rs, options.rs,
// algorithms/r2c.rs:521-895, algorithms/bravo.rs:303-324); see DESIGN.md for the mapping.
//
// There is NO CPU fallback in this library: without a gfx950 device every compute entry point returns
// PHAST_ERR_NO_DEVICE / PHAST_ERR_HIP.
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
#include <string>
#include <vector>

#include "kernels.hpp"
#include "plan.hpp"
#include "tile_dispatch.hpp"
#include "r2c_fused.hpp"
#include "c2r_fused.hpp"

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

// ------------------------------------------------------------------------------------------------
// planner
// ------------------------------------------------------------------------------------------------
struct PassDesc : PassGeom {
    void *d_tw3 = nullptr;
    void *d_twr = nullptr;
    int blocks_per_cu = 1;
    size_t lds = 0;
    // last pass only: the fused R2C form of this pass exists (r2c_fused.hpp); its W_{2 rows}^k table and residency
    void *d_twu = nullptr;
    int r2c_blocks = 0;
    // first pass only: the fused C2R form of this pass exists (c2r_fused.hpp; d_twu is its W_{2 rows}^n table)
    int c2r_blocks = 0;
};

// what the R2C planner hands to Planner::exec so that the last pass can take the untangle with it
struct R2cFuse {
    const void *tw3n;  // W_N three-level table, N = 2 * (inner transform length)
    unsigned twn_bits;
};
static bool c2r_fuse_enabled() {  // PHAST_C2R_FUSE=0: keep the C2R preprocess as a sweep of its own (tools, A/B)
    static const bool v = [] {
        const char *e = std::getenv("PHAST_C2R_FUSE");
        return !(e && *e == '0');
    }();
    return v;
}
static bool r2c_fuse_enabled() {  // PHAST_R2C_FUSE=0: keep the untangle as a sweep of its own (tools, A/B)
    static const bool v = [] {
        const char *e = std::getenv("PHAST_R2C_FUSE");
        return !(e && *e == '0');
    }();
    return v;
}

template <typename T> struct Types;
template <> struct Types<double> {
    static hipError_t launch_a(int lr, int lc, int lp, unsigned g, hipStream_t s, const TileArgs &a, bool q, int *b,
                               size_t *l, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
        return launch_tile_f64_a(lr, lc, lp, g, s, a, q, b, l, e0, e1);
    }
    static hipError_t launch_bc(int lr, int lc, int lp, unsigned g, hipStream_t s, const TileArgs &a, bool q, int *b,
                               size_t *l, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
        return launch_tile_f64_bc(lr, lc, lp, g, s, a, q, b, l, e0, e1);
    }
};
template <typename T> static hipError_t launch_wave(bool transpose, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                                                    hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if constexpr (sizeof(T) == 8) return launch_wave_f64(transpose, s, a, q, b, l, e0, e1);
#ifdef PHAST_EXPERIMENTAL_WAVE_F32
    else return launch_wave_f32(transpose, s, a, q, b, l, e0, e1);
#else
    else return hipErrorInvalidValue;  // the f32 wave tiles are not in the product library (build.py --experimental)
#endif
}
template <typename T> static hipError_t launch_quad(unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                                                    hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if constexpr (sizeof(T) == 8) return launch_quad_f64(grid, s, a, q, b, l, e0, e1);
    else return hipErrorInvalidValue;
}
template <> struct Types<float> {
    static hipError_t launch_a(int lr, int lc, int lp, unsigned g, hipStream_t s, const TileArgs &a, bool q, int *b,
                               size_t *l, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
        return launch_tile_f32_a(lr, lc, lp, g, s, a, q, b, l, e0, e1);
    }
    static hipError_t launch_bc(int lr, int lc, int lp, unsigned g, hipStream_t s, const TileArgs &a, bool q, int *b,
                               size_t *l, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
        return launch_tile_f32_bc(lr, lc, lp, g, s, a, q, b, l, e0, e1);
    }
};

static size_t scratch_target_bytes() {
    const char *env = std::getenv("PHAST_SCRATCH_MB");
    if (env && *env) {
        long v = std::atol(env);
        if (v > 0) return (size_t)v << 20;
    }
    // measured (profiles/r01_sweep_scratch_chunk.log): chunks sized to the 256 MiB Infinity Cache buy nothing, while
    // launches of >= 256 transforms run the 1024 x 8 passes ~25 % faster than 16-transform launches; round 2: the whole
    // 1024-transform shard of BASELINE configs[4] in one chunk is another 1 % (76.4 vs 75.6 GSamples/s; 2 GiB chunks:
    // 71.6) -- 16 GiB of a 288 GB device, allocated only when a batch that large arrives
    return (size_t)16384 << 20;
}

// measurement hook: hipEvents recorded on the launch stream around every pass kernel (bench.py "roofline")
struct PassTimer {
    std::vector<hipEvent_t> ev;       // start/stop pairs in launch order
    std::vector<int> pass_of;         // pass index of every pair
    ~PassTimer() {
        for (auto e : ev) hipEventDestroy(e);
    }
    // a fresh (start, stop) pair; the launcher binds it to the dispatch (hipExtLaunchKernelGGL), so the
    // interval is the kernel's own execution time -- what rocprofv3 --kernel-trace reports
    hipError_t pair(int pass, hipEvent_t *e0, hipEvent_t *e1) {
        hipError_t rc = hipEventCreate(e0);
        if (rc == hipSuccess) rc = hipEventCreate(e1);
        if (rc != hipSuccess) return rc;
        ev.push_back(*e0);
        ev.push_back(*e1);
        pass_of.push_back(pass);
        return hipSuccess;
    }
};

// ------------------------------------------------------------------------------------------------
// Workspaces: what ONE call sequence works in.  The reference's planner is an immutable value shared by `&`
// (planner.rs:38-39; algorithms/dit.rs:263 takes `&PlannerDit64`): N host threads transform N buffers at once with one
// planner.  Here the immutable part is the tables and plans; everything a call mutates -- the inter-pass scratch, the
// staging buffer and pinned mirror of the host-slice calls, the C2R workspace -- lives in a Workspace, and a planner
// keeps a small pool of them.  A call checks one out for the time it ENQUEUES (a blocking host-slice call: for the whole
// call), so concurrent callers of one planner run side by side instead of one behind the other.
//   * a workspace is bound to the stream its last work went to: calls on that stream come back to it (stream order makes
//     the reuse of its scratch safe without any synchronisation);
//   * a call on another stream takes a workspace whose work has drained (its own `idle` event, recorded behind every _dev
//     call, has completed), or makes a new one (up to PHAST_MAX_WORKSPACES, default 8), or -- pool exhausted -- takes one
//     whose work is still in flight BEHIND that event: the new stream waits on the device, the host never blocks;
//   * host-slice calls run on the workspace's own non-blocking stream, never on the NULL stream;
//   * a workspace that was used under stream capture belongs to the captured graph(s) from then on: replays may run at any
//     time on streams this library never sees, so eager calls never take it and none of its buffers is ever freed before
//     the planner is (a graph replayed after an outgrown scratch had been released read freed memory).
// ------------------------------------------------------------------------------------------------
static bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return true;  // cannot tell: behave as if it were
    }
    return st != hipStreamCaptureStatusNone;
}

static size_t max_workspaces() {
    static const size_t v = [] {
        const char *e = std::getenv("PHAST_MAX_WORKSPACES");
        long n = e && *e ? std::atol(e) : 8;
        return (size_t)(n < 1 ? 1 : n > 64 ? 64 : n);
    }();
    return v;
}

struct Workspace {
    void *d_scratch = nullptr;  // [cap][2][stride]: re plane then im plane per transform (typed by the planner)
    size_t cap = 0;
    size_t guard = 0;           // bytes of guard band before and after the scratch (debug hook, normally 0)
    size_t per = 0;             // bytes per transform the scratch was cut for (2 * stride * sizeof(T))
    void *d_stage = nullptr;    // device staging of the host-slice entry points (grow-only)
    size_t stage_bytes = 0;
    void *h_pin = nullptr;      // pinned host mirror of the staging buffer for SMALL host-slice calls
    size_t pin_bytes = 0;
    void *d_z = nullptr;        // unfused C2R: the preprocess workspace [z_cap][2][n/2] (PlannerR2c)
    size_t z_cap = 0, z_bytes = 0;
    hipStream_t stream = nullptr;  // the stream the last work of this workspace went to (valid while `pending`)
    bool pending = false;          // work may still be running on `stream`
    hipStream_t own = nullptr;     // the non-blocking stream of host-slice calls (created on first use)
    hipEvent_t idle = nullptr;     // recorded behind the last _dev call's work (Planner::check_in); owned by the workspace
    bool busy = false;             // checked out by a host thread
    bool captured = false;         // used under stream capture: pinned to the captured graphs (see above)
    // A buffer that has to grow is replaced, never freed inside the call that outgrew it: kernels already enqueued may
    // still use the old one.  The predecessor is RETIRED with an event recorded on the workspace's stream behind them; a
    // later call frees it once that event has completed (never under capture, never for a captured workspace).
    struct Retired {
        void *p;
        size_t bytes;
        hipEvent_t done;  // nullptr: released with the planner
        bool pinned;
    };
    std::vector<Retired> retired;
    size_t retired_dev_bytes = 0;

    void retire(void *p, size_t bytes, bool pinned, hipStream_t on) {
        if (!p) return;
        Retired r{p, bytes, nullptr, pinned};
        if (!captured && !stream_capturing(on)) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
                if (hipEventRecord(ev, on) == hipSuccess) r.done = ev;
                else hipEventDestroy(ev);
            }
            (void)hipGetLastError();
        }
        if (!pinned) retired_dev_bytes += bytes;
        retired.push_back(r);
    }
    // free what is provably idle; `wait`: block for it (out-of-memory recovery).  Not under capture of `on`.
    void reap(hipStream_t on, bool wait = false) {
        if (retired.empty() || stream_capturing(on)) return;
        size_t keep = 0;
        for (size_t i = 0; i < retired.size(); ++i) {
            Retired &r = retired[i];
            bool idle = false;
            if (r.done) idle = (wait ? hipEventSynchronize(r.done) : hipEventQuery(r.done)) == hipSuccess;
            if (idle) {
                if (r.pinned) hipHostFree(r.p);
                else {
                    hipFree(r.p);
                    retired_dev_bytes -= r.bytes;
                }
                hipEventDestroy(r.done);
            } else {
                retired[keep++] = r;
            }
        }
        retired.resize(keep);
        (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error of the call being made
    }
    size_t device_bytes() const { return cap * per + stage_bytes + z_bytes + retired_dev_bytes; }
    void release() {  // with the planner (hipFree waits for the device: whatever still used the buffers is done afterwards)
        if (d_scratch) hipFree(reinterpret_cast<char *>(d_scratch) - guard);
        if (d_stage) hipFree(d_stage);
        if (d_z) hipFree(d_z);
        if (h_pin) hipHostFree(h_pin);
        for (const Retired &r : retired) {
            if (r.pinned) hipHostFree(r.p);
            else hipFree(r.p);
            if (r.done) hipEventDestroy(r.done);
        }
        retired.clear();
        if (own) hipStreamDestroy(own);
        if (idle) hipEventDestroy(idle);
        d_scratch = d_stage = d_z = h_pin = nullptr;
        own = nullptr;
        idle = nullptr;
        cap = stage_bytes = z_cap = z_bytes = pin_bytes = retired_dev_bytes = 0;
    }
};

template <typename T> struct Planner {
    size_t n = 0;
    unsigned log_n = 0;
    std::vector<PassDesc> passes;      // throughput plan; empty => small path
    std::vector<PassDesc> passes_lat;  // latency plan (one small transform); may equal `passes`
    std::vector<PassDesc> passes_mid;  // a few transforms in flight, where that wants a plan of its own (plan.hpp)
    std::vector<PassDesc> passes_one;  // ONE (or two) transforms: wave / quad tiles etc. (plan.hpp: single_plan)
    // C2R only (PlannerR2c::init): passes_one / passes_lat with the pass ORDER reversed, where that gives the first pass --
    // which reads the caller's planar half-spectrum -- the wide rows the C2C order gives the last (see make_c2r_plans)
    std::vector<PassDesc> passes_c2r_one, passes_c2r_lat;
    std::vector<PassDesc> passes_r2c_tp, passes_c2r_tp;  // batches of real transforms in the throughput regime (plan.hpp: real_batch_plan)
    // R2C only: the plan of ONE (or two) real transforms where plan.hpp (real_plan) has a better one than the C2C choice;
    // passes_c2r_one is C2R's (from the same table, else the reversal above)
    std::vector<PassDesc> passes_r2c;
    // passes_r2c was ranked with its fused last pass below the general threshold (plan.hpp: kFuseBelow); written by
    // make_c2r_plans after the plan swap, read by calls in flight on other threads
    std::atomic<bool> r2c_table_fuses{false};
    void *d_small_tw = nullptr;
    // elements per transform and plane in the scratch: n plus the padding of the intermediate layouts (plan.hpp:
    // scratch_pad_bytes); the largest over this planner's plans, fixed before the first allocation grows past it
    size_t scratch_stride = 0;
    size_t sstride() const { return scratch_stride ? scratch_stride : n; }
    mutable size_t reserve = 1;
    mutable size_t table_bytes = 0;
    int device = -1;  // the device this planner's tables and scratch live on (current at creation); calls run there
    // the workspace pool (see Workspace); `mu` guards the pool's bookkeeping, never a launch
    mutable std::mutex mu;
    mutable std::condition_variable cv;
    mutable std::vector<std::unique_ptr<Workspace>> pool;
    // plans are read by every call and replaced by set_plan (a tuning hook): shared for the enqueue, exclusive to swap
    mutable std::shared_mutex plan_mu;
    // tables a set_plan replaced: kernels already enqueued (or captured) may still read them -- released with the planner
    mutable std::vector<void *> old_tables;
    mutable size_t old_table_bytes = 0;
    // plans of strided batches (column FFTs), built on first use per (log2 stride, log2 batch): see make_strided_passes;
    // guarded by `mu`, entries never move
    struct StridedPlan {
        unsigned s = 0, sb = 0, grid_log_n = 0;
        std::vector<PassDesc> passes;
    };
    mutable std::vector<std::unique_ptr<StridedPlan>> strided_plans;

    static bool capturing(hipStream_t s) { return stream_capturing(s); }

    // One checked-out workspace + the shared hold on the plans, for the duration of a call's enqueue (host-slice calls: of
    // the whole blocking call).  `stream` is where the call's work goes: the caller's for _dev calls, the workspace's own
    // for host-slice calls.
    struct Lease {
        const Planner *pl = nullptr;
        Workspace *ws = nullptr;
        hipStream_t stream = nullptr;
        bool host = false;
        std::shared_lock<std::shared_mutex> plans;
        Lease() = default;
        Lease(const Lease &) = delete;
        Lease &operator=(const Lease &) = delete;
        ~Lease() {
            if (pl && ws) pl->check_in(ws, stream, host);
        }
    };
    // which = 0: a _dev call on `stream`; 1: a host-slice call (runs on the workspace's own stream); 2: bookkeeping only
    // (reserve_batch: any free eager workspace, nothing is enqueued).
    // The library never touches a caller's stream handle after the call that was given it has returned -- the caller may
    // destroy the stream the moment its work is done (round 4's first version asked hipStreamQuery about the stream a
    // workspace had last served: a use-after-free inside the HIP runtime once that stream was gone, found by the ASan pass of
    // tests/cpp/concurrent_planner_test.cpp).  What outlives a call is the workspace's OWN event, recorded behind the call's
    // work while the stream is certainly alive (check_in): "has this workspace drained?" is hipEventQuery(idle), "order that
    // stream behind it" is hipStreamWaitEvent(stream, idle).
    int check_out(Lease &L, hipStream_t stream, int which = 0) const {
        L.plans = std::shared_lock<std::shared_mutex>(plan_mu);
        const bool cap = which == 0 && capturing(stream);
        std::unique_lock<std::mutex> lk(mu);
        Workspace *pick = nullptr;
        auto drained = [](Workspace &w) {  // nothing of this workspace's work can still be running
            if (!w.pending) return true;
            if (w.idle && hipEventQuery(w.idle) == hipSuccess) {
                w.pending = false;
                return true;
            }
            (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure
            return false;
        };
        // under capture nothing may be allocated: a workspace "fits" if its scratch exists and was cut for the pitches of the
        // plans as they are now (set_plan may have widened them since it was made; ensure_scratch would have to re-cut it)
        const size_t per_now = 2 * sstride() * sizeof(T);
        auto fits = [&](const Workspace &w) { return w.cap > 0 && w.per == per_now; };
        for (;;) {
            if (which == 0 && !cap)  // 1. the workspace this stream used last: stream order protects its buffers
                for (auto &w : pool)
                    if (!w->busy && w->pending && w->stream == stream && !w->captured) {
                        pick = w.get();
                        break;
                    }
            if (cap) {
                // 1b. under capture nothing executes now and nothing may be allocated or queried: the stream's own workspace
                // or any eager one, whichever FITS and is largest (a larger batch runs in fewer chunks; the stream's own on
                // a tie) -- it belongs to the graph from here on, so no eager call can meet a replay in it.  What was
                // enqueued in it BEFORE the capture is ordered before the replays by the caller (a capture stream is always
                // forked from the stream that did the warm-up).
                for (auto &w : pool) {
                    const bool own_ws = w->pending && w->stream == stream;
                    if (w->busy || !fits(*w) || (w->captured && !own_ws)) continue;
                    if (!pick || w->cap > pick->cap || (w->cap == pick->cap && own_ws)) pick = w.get();
                }
                // 1c. none fits (no eager call since the plan changed, or none at all): the stream's own, then any eager one --
                // ensure_scratch will have to allocate and the capture fails with the runtime's message, as documented
                // ("HIP graphs": run the call once eagerly before capturing it)
                if (!pick)
                    for (auto &w : pool)
                        if (!w->busy && w->pending && w->stream == stream) {
                            pick = w.get();
                            break;
                        }
                if (!pick)
                    for (auto &w : pool)
                        if (!w->busy && !w->captured && (!pick || w->cap > pick->cap)) pick = w.get();
            }
            if (!pick && !cap)  // 2. one with nothing in flight (never one that belongs to a captured graph)
                for (auto &w : pool)
                    if (!w->busy && !w->captured && drained(*w)) {
                        pick = w.get();
                        break;
                    }
            if (pick) break;
            size_t eager = 0;
            for (auto &w : pool) eager += !w->captured;
            if (cap || eager < max_workspaces()) {  // 3. a new one
                pool.emplace_back(new (std::nothrow) Workspace());
                if (!pool.back()) {
                    pool.pop_back();
                    return PHAST_ERR_ALLOC;
                }
                pick = pool.back().get();
                break;
            }
            // 4. pool exhausted: queue behind another stream's work ON THE DEVICE (below: the new stream waits for `idle`)
            bool any_busy = false;
            for (auto &w : pool) {
                if (w->busy) any_busy = true;
                else if (!w->captured && w->idle) {
                    pick = w.get();
                    break;
                }
            }
            if (pick) break;
            if (!any_busy) {  // nothing to wait for: one workspace beyond the limit
                pool.emplace_back(new (std::nothrow) Workspace());
                if (!pool.back()) {
                    pool.pop_back();
                    return PHAST_ERR_ALLOC;
                }
                pick = pool.back().get();
                break;
            }
            cv.wait(lk);  // 5. every candidate is checked out by a thread that is enqueueing: wait for one
        }
        pick->busy = true;
        lk.unlock();
        auto give_back = [&](int rc) {
            std::lock_guard<std::mutex> g(mu);
            pick->busy = false;
            cv.notify_one();
            return rc;
        };
        hipStream_t work = stream;
        if (which == 1) {
            if (!pick->own) {
                hipError_t e = hipStreamCreateWithFlags(&pick->own, hipStreamNonBlocking);
                if (e != hipSuccess) return give_back(hip_fail(e, "hipStreamCreateWithFlags(workspace stream)"));
            }
            work = pick->own;
        }
        if (!cap && pick->pending && pick->idle) {
            // Work of this workspace may still be in flight.  Same stream handle as last time: stream order already covers
            // it -- unless the handle belongs to a NEW stream that reuses a destroyed one's address, so the (cheap) query
            // decides; another stream (case 4, or a host-slice call after a _dev call): order this call's stream behind the
            // workspace's last work on the device.  Bookkeeping calls (which == 2) enqueue nothing: they wait here.
            if (hipEventQuery(pick->idle) != hipSuccess) {
                (void)hipGetLastError();
                hipError_t e = which == 2 ? hipEventSynchronize(pick->idle) : hipStreamWaitEvent(work, pick->idle, 0);
                if (e != hipSuccess) return give_back(hip_fail(e, "hipStreamWaitEvent(workspace idle)"));
            }
            if (which == 2) pick->pending = false;
        }
        L.pl = this;
        L.ws = pick;
        L.stream = work;
        L.host = which == 1;
        if (which != 2) {
            if (cap) pick->captured = true;
            pick->stream = work;
            pick->pending = true;
            if (!cap) pick->reap(work);
        }
        return PHAST_OK;
    }
    // the call has enqueued everything (a host-slice call: and waited for it)
    void check_in(Workspace *ws, hipStream_t stream, bool host_synchronised) const {
        if (host_synchronised) {
            (void)hipStreamSynchronize(stream);      // (already drained on the success path; an early error return may not be)
            if (!ws->captured) ws->pending = false;  // returns with its stream drained
        } else if (ws->pending && !ws->captured && !capturing(stream)) {
            // behind this call's work, while the caller's stream is certainly alive: the only thing later calls look at
            hipError_t e = ws->idle ? hipSuccess : hipEventCreateWithFlags(&ws->idle, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ws->idle, stream);
            if (e != hipSuccess) {  // cannot mark it: make sure nothing is in flight instead
                (void)hipGetLastError();
                (void)hipStreamSynchronize(stream);
                ws->pending = false;
            }
        }
        std::lock_guard<std::mutex> g(mu);
        ws->busy = false;
        cv.notify_one();
    }

    ~Planner() { release(); }
    // Host-slice calls up to this many staged bytes go through the pinned mirror (one memcpy each way on the host,
    // one asynchronous copy each way over PCIe): hipMemcpy from pageable memory costs 50-200 us per call
    // whatever the size, which is all a small transform's time.  Larger calls copy straight from the slices.
    static size_t pinned_max_bytes() {
        static const size_t v = [] {
            const char *e = getenv("PHAST_PINNED_MAX_KB");
            return (size_t)(e ? atol(e) : 1024) << 10;
        }();
        return v;
    }
    int pinned(const Lease &L, size_t bytes, void **out) const {
        Workspace &w = *L.ws;
        if (w.pin_bytes < bytes) {
            w.retire(w.h_pin, w.pin_bytes, true, L.stream);
            w.h_pin = nullptr;
            w.pin_bytes = 0;
            PHAST_HIP(hipHostMalloc(&w.h_pin, bytes ? bytes : 1, hipHostMallocDefault));
            w.pin_bytes = bytes;
        }
        *out = w.h_pin;
        return PHAST_OK;
    }
    // device staging buffer of at least `bytes`
    int stage(const Lease &L, size_t bytes, void **out) const {
        Workspace &w = *L.ws;
        if (w.stage_bytes < bytes) {
            w.retire(w.d_stage, w.stage_bytes, false, L.stream);
            w.d_stage = nullptr;
            w.stage_bytes = 0;
            PHAST_HIP(hipMalloc(&w.d_stage, bytes ? bytes : 1));
            w.stage_bytes = bytes;
        }
        *out = w.d_stage;
        return PHAST_OK;
    }
    static void free_passes(std::vector<PassDesc> &v) {
        for (auto &p : v) {
            if (p.d_tw3) hipFree(p.d_tw3);
            if (p.d_twr) hipFree(p.d_twr);
            if (p.d_twu) hipFree(p.d_twu);
        }
        v.clear();
    }
    void retire_passes(std::vector<PassDesc> &v) {  // plan_mu held exclusively
        for (auto &p : v) {
            for (void *t : {p.d_tw3, p.d_twr, p.d_twu})
                if (t) old_tables.push_back(t);
            old_table_bytes += (p.pre_tw ? ((size_t)3 << p.tw_bits) : 0) * sizeof(cx_t<T>) + 64 * sizeof(cx_t<T>) +
                               (p.d_twu ? ((size_t)1 << p.lr) * sizeof(cx_t<T>) : 0);
        }
        v.clear();
    }
    void release_passes() {
        for (auto &sp : strided_plans) free_passes(sp->passes);
        strided_plans.clear();
        free_passes(passes);
        free_passes(passes_lat);
        free_passes(passes_mid);
        free_passes(passes_one);
        free_passes(passes_c2r_one);
        free_passes(passes_c2r_lat);
        free_passes(passes_r2c_tp);
        free_passes(passes_c2r_tp);
        free_passes(passes_r2c);
        for (void *t : old_tables) hipFree(t);
        old_tables.clear();
        old_table_bytes = 0;
    }
    void release() {
        DeviceGuard on(device);
        release_passes();
        if (d_small_tw) hipFree(d_small_tw);
        d_small_tw = nullptr;
        for (auto &w : pool) w->release();
        pool.clear();
    }

    // the plan for `batch` transforms in flight: throughput when its tiles fill the chip, else the mid plan for
    // more than one transform (where there is one), else the latency plan
    const std::vector<PassDesc> &plan_for(size_t batch) const {
        if (passes_lat.empty() || passes.empty()) return passes;
        // ranked for ONE transform, whatever its size (plan.hpp: single_plan) -- and still ahead for a few small ones: up to
        // 2^19 points in flight, or four transforms, below 2^21 points (tools/small_batch_plans.py,
        // profiles/r04_small_batch_plans.log: 2^15 x 8 f64 14.6 -> 12.4 us, 2^16 x 4 15.0 -> 13.5, 2^19 x 3 42.7 -> 36.0;
        // from 2^21 points per transform on, four in flight already prefer the throughput tiles)
        if (!passes_one.empty() && (batch <= 2 || (log_n <= 20 && (batch <= 4 || batch * n <= ((size_t)1 << 19))))) return passes_one;
        unsigned tl = 0;
        for (const PassDesc &p : passes) tl = std::max(tl, p.lr + p.lc);
        // 4-byte elements: the same tile holds half the bytes, and the measured crossover sits one octave higher (one f32
        // transform of 2^24 points: 149.7 us on the latency tiles, 180.0 on the throughput tiles)
        if (batch * n >= throughput_work(tl) * (sizeof(T) == 4 && tl < 15 ? 2 : 1)) return passes;
        if (batch > 1 && !passes_mid.empty()) return passes_mid;
        return passes_lat;
    }

    // R2C (with the untangle fused into the last pass): where the plan for `batch` ends in a pass that has no fused form
    // (wave / quad tiles of the single-transform plans) but the latency plan's generic tiles do, the latency plan runs --
    // its passes are a few per cent slower, the sweep it saves is a quarter of the transform (R2C of 2^24..2^26 f64 points:
    // +11..15 %; beyond 2^25 inner points the latency plan's passes lose more than the sweep gives: measured, tools/r2c_f64_probe.py)
    const std::vector<PassDesc> &plan_for_r2c(size_t batch, bool fusing = true) const {
        if (batch <= 2 && !passes_r2c.empty()) return passes_r2c;  // ranked for R2C itself (plan.hpp: real_plan)
        const std::vector<PassDesc> &ps = plan_for(batch);
        if (&ps == &passes && !passes_r2c_tp.empty()) return passes_r2c_tp;  // ... and for batches of them (real_batch_plan)
        if (fusing && !ps.empty() && ps.back().r2c_blocks == 0 && log_n <= 25 && !passes_lat.empty() && passes_lat.back().r2c_blocks > 0 &&
            r2c_lat_ok())
            return passes_lat;
        return ps;
    }
    // C2R, the same on the other side: a plan whose FIRST pass is a wave tile has no fused form of it (c2r_fused.hpp)
    const std::vector<PassDesc> &plan_for_c2r(size_t batch) const {
        const std::vector<PassDesc> &ps0 = plan_for(batch);
        if (&ps0 == &passes && batch > 2 && !passes_c2r_tp.empty() && passes_c2r_tp.front().c2r_blocks > 0) return passes_c2r_tp;
        const std::vector<PassDesc> &ps = (batch <= 2 && !passes_c2r_one.empty())                 ? passes_c2r_one
                                          : (&ps0 == &passes_lat && !passes_c2r_lat.empty()) ? passes_c2r_lat
                                                                                             : ps0;
        if (!ps.empty() && ps.front().c2r_blocks == 0 && !passes_lat.empty() && passes_lat.front().c2r_blocks > 0 && c2r_lat_ok())
            return passes_lat;
        return ps;
    }
    static bool c2r_lat_ok() {  // PHAST_C2R_LAT=0: tools (A/B)
        static const bool v = [] {
            const char *e = std::getenv("PHAST_C2R_LAT");
            return !(e && *e == '0');
        }();
        return v;
    }
    static bool r2c_lat_ok() {  // PHAST_R2C_LAT=0: tools (A/B)
        static const bool v = [] {
            const char *e = std::getenv("PHAST_R2C_LAT");
            return !(e && *e == '0');
        }();
        return v;
    }

    // which: 0 = one plan for every batch size, 1 = throughput plan only, 2 = latency plan only, 3 = mid plan only,
    // 4 = the plan for one transform;
    // lp = log2(points per thread)
    int set_plan(const std::vector<unsigned> &lrs, const std::vector<unsigned> &tls, int which = 0, unsigned lp = 4) {
        std::vector<PassGeom> geo;
        if (!make_passes(log_n, lrs, tls, geo, lp, sizeof(T))) return PHAST_ERR_INVALID_ARG;
#ifndef PHAST_EXPERIMENTAL_WAVE_F32
        if (sizeof(T) == 4)  // f32 wave tiles: measured, slower than the generic tiles everywhere, built with --experimental only
            for (const PassGeom &g : geo)
                if (g.wave) return PHAST_ERR_INVALID_ARG;
#endif
        PHAST_ON_DEVICE(device);
        std::vector<PassDesc> ps(geo.size());
        for (size_t i = 0; i < geo.size(); ++i) static_cast<PassGeom &>(ps[i]) = geo[i];
        size_t tb = 0;
        {
            int rc = prepare_passes(ps, &tb);
            if (rc) return rc;
        }
        const size_t need = (size_t)scratch_elems(geo, log_n);
        // every call reads the pass vectors under a shared hold of plan_mu for as long as it enqueues: a plan is never
        // swapped under a launch sequence; kernels already enqueued (or captured) keep reading the old tables, which
        // are therefore kept until the planner goes, not freed
        std::unique_lock<std::shared_mutex> plans(plan_mu);
        if (which == 2) {
            retire_passes(passes_lat);
            retire_passes(passes_c2r_lat);  // derived from the plan that goes
            passes_lat = std::move(ps);
        } else if (which == 3) {
            retire_passes(passes_mid);
            passes_mid = std::move(ps);
        } else if (which == 4) {
            retire_passes(passes_one);
            retire_passes(passes_c2r_one);
            passes_one = std::move(ps);
        } else if (which >= 5 && which <= 9) {  // the real transforms' own plans: additional, table_bytes and pitch below
            std::vector<PassDesc> &dst = which == 5   ? passes_c2r_one
                                         : which == 6 ? passes_c2r_lat
                                         : which == 7 ? passes_r2c
                                         : which == 8 ? passes_c2r_tp
                                                      : passes_r2c_tp;
            retire_passes(dst);
            dst = std::move(ps);
            table_bytes += tb;
            if (need > sstride()) scratch_stride = need;
            return PHAST_OK;
        } else {
            retire_passes(passes);
            passes = std::move(ps);
            if (which == 0) {  // one plan for every batch size
                retire_passes(passes_lat);
                retire_passes(passes_mid);
                retire_passes(passes_one);
                retire_passes(passes_c2r_one);
                retire_passes(passes_c2r_lat);
                retire_passes(passes_r2c);
                retire_passes(passes_r2c_tp);
                retire_passes(passes_c2r_tp);
            }
        }
        table_bytes = tb;
        // a plan with wider pitches than the scratches were cut for: every workspace re-cuts its scratch on next use
        // (ensure_scratch compares Workspace::per)
        if (need > sstride()) scratch_stride = need;
        return PHAST_OK;
    }

    // N = 2^kSmallMaxLog (8192 points) only: the same length as a MULTI-pass planner, for up to 128 transforms.  The one-pass
    // kernel keeps a whole transform in one workgroup -- right for batches (one sweep over the data), but a single 8192-point
    // transform is then ONE workgroup's chain of six LDS round trips: 16.2 us (f64) where the two-pass wave / quad plan of
    // 2^14 points takes 12.1 (profiles/r04_size_ladder.log).  PHAST_SMALL_TWIN=0: tools (A/B).
    std::unique_ptr<Planner<T>> twin;
    static bool twin_enabled() {
        static const bool v = [] {
            const char *e = std::getenv("PHAST_SMALL_TWIN");
            return !(e && *e == '0');
        }();
        return v;
    }
    // ... and a batch of them is one workgroup EACH: below half the chip's CUs the twin still wins (2^13 x 128 f64: 17.8 us
    // against 21.8, x 32: 11.3 against 17.3; profiles/r04_small_twin_batch.log).  PHAST_SMALL_TWIN_MAX_BATCH: tools.
    static size_t twin_max_batch() {
        static const size_t v = [] {
            const char *e = std::getenv("PHAST_SMALL_TWIN_MAX_BATCH");
            return (e && *e) ? (size_t)std::atoll(e) : (size_t)128;
        }();
        return v;
    }

    int init(size_t num_points, bool force_multi = false, bool with_twin = true) {
        n = num_points;
        log_n = ilog2(n);
        int rc = ensure_device(&device);
        if (rc) return rc;
        if (log_n <= kSmallMaxLog && !(force_multi && log_n == kSmallMaxLog)) {
            std::vector<cx_t<T>> h = host_twr<T>((unsigned)n);  // W_N^j two-level table of the one-pass kernel
            table_bytes = h.size() * sizeof(cx_t<T>);
            rc = upload<T>(h, &d_small_tw);
            if (rc == PHAST_OK && with_twin && log_n == kSmallMaxLog && twin_enabled()) {
                twin.reset(new (std::nothrow) Planner<T>());
                if (twin && twin->init(n, true) != PHAST_OK) twin.reset();  // an optimisation: without it the one-pass kernel serves
            }
            return rc;
        }
        return default_plans();
    }

    // the library's own two plans (plan.hpp: heuristic_plan)
    int default_plans() {
        std::vector<unsigned> lrs, tls;
        unsigned lp = 4;
        heuristic_plan<T>(log_n, false, lrs, tls, lp);
        int rc = set_plan(lrs, tls, 1, lp);
        // the largest sizes: the last pass's twiddle tables (3 * 2^ceil(L/3) entries) may not leave room for the
        // heuristic's tile -- step the tile size down until the plan fits one CU's LDS
        while (rc == PHAST_ERR_INVALID_ARG && tls[0] > 12) {
            tls.assign(1, tls[0] - 1);
            if (tls[0] < 14 && lp == 5) lp = 4;
            rc = set_plan(lrs, tls, 1, lp);
        }
        if (rc) return rc;
        // the other two plans are optimisations: where their tiles do not exist (N >= 2^31) the throughput plan serves
        heuristic_plan<T>(log_n, true, lrs, tls, lp);
        rc = set_plan(lrs, tls, 2, lp);
        if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
        if (mid_plan<T>(log_n, lrs, tls, lp)) {
            rc = set_plan(lrs, tls, 3, lp);
            if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
        }
        if (single_plan<T>(log_n, lrs, tls, lp)) {
            rc = set_plan(lrs, tls, 4, lp);
            if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
        }
        return PHAST_OK;
    }

    // C2R reads the caller's PLANAR half-spectrum in its first pass and writes (im, re) PAIRS in its last -- the mirror image
    // of what the C2C plans were cut for (and of R2C: pairs in, planes out).  Where a plan gives its first pass rows of less
    // than a 128-byte line of planar elements and its last pass wider ones (f32: [256x16A][256x16][128x32] -- 64-byte rows
    // exactly where C2R has FOUR streams of them per tile: re / im of the element and of its mirror partner), the same passes
    // in reverse order serve C2R better: [128x32A][256x16][256x16].  PHAST_C2R_REV=0: tools (A/B).
    int make_c2r_plans() {
        static const bool rev = [] {
            const char *e = std::getenv("PHAST_C2R_REV");
            return !(e && *e == '0');
        }();
        static const bool table = [] {  // PHAST_REAL_PLANS=0: R2C / C2R keep the C2C plans (tools: A/B, tools/sweep_real.py)
            const char *e = std::getenv("PHAST_REAL_PLANS");
            return !(e && *e == '0');
        }();
        // 1. the ranked plans of ONE real transform (plan.hpp: real_plan)
        for (int c2r = 0; c2r < 2 && table; ++c2r) {
            std::vector<unsigned> lrs, tls;
            unsigned lp = 4;
            if (!real_plan<T>(log_n, c2r != 0, lrs, tls, lp)) continue;
            const bool fuse_below = (lp & kFuseBelow) != 0;
            int rc = set_plan(lrs, tls, c2r ? 5 : 7, lp & ~kFuseBelow);
            if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
            if (!c2r) r2c_table_fuses = rc == PHAST_OK && fuse_below;
        }
        // 1b. ... and of batches of them in the throughput regime (plan.hpp: real_batch_plan)
        for (int c2r = 0; c2r < 2 && table; ++c2r) {
            std::vector<unsigned> lrs, tls;
            unsigned lp = 4;
            if (!real_batch_plan<T>(log_n, c2r != 0, lrs, tls, lp)) continue;
            int rc = set_plan(lrs, tls, c2r ? 8 : 9, lp);
            if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
        }
        // 2. C2R, two-pass plans: the reversed order
        for (int k = 0; k < 2 && rev; ++k) {
            if (k == 0 && !passes_c2r_one.empty()) continue;
            const std::vector<PassDesc> &src = k == 0 ? passes_one : passes_lat;
            // (three-pass plans: measured and NOT reversed -- f32 2^24 first pass 38 -> 35 us but the middle pass, now behind
            // a 128-row first pass, 25 -> 31: profiles/r04_c2r_rev_ab.log; two-pass plans: 2^20 20.2 -> 17.9 us)
            if (src.size() != 2 || src.front().wave || src.front().quad || src.back().wave || src.back().quad) continue;
            if ((sizeof(T) << src.front().lc) >= 128 || src.back().lc <= src.front().lc) continue;
            std::vector<unsigned> lrs, tls;
            for (size_t i = src.size(); i-- > 0;) {
                lrs.push_back(src[i].lr);
                tls.push_back(src[i].lr + src[i].lc);
            }
            int rc = set_plan(lrs, tls, 5 + k, src.front().lp);
            if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
            std::vector<PassDesc> &dst = k == 0 ? passes_c2r_one : passes_c2r_lat;
            if (rc == PHAST_OK && (dst.empty() || dst.front().c2r_blocks <= 0)) {  // no fused first pass: not worth having
                std::unique_lock<std::shared_mutex> plans(plan_mu);
                retire_passes(dst);
            }
        }
        return PHAST_OK;
    }

    // scratch for `want` transforms in flight (capped by the target footprint, at least 1) in the leased workspace.  Grows
    // geometrically: a sequence of slowly growing batches retires O(log) buffers whose sizes sum to less than the live one
    // (and reap() frees them as soon as the work that used them is done).  Out of device memory: idle retired buffers are
    // released, then the request is halved (exec() loops over chunks) down to the reserved batch.
    int ensure_scratch(const Lease &L, size_t batch, size_t *cap_out, bool exact = false) const {
        if (passes.empty() && !exact) {  // whole transforms on chip: no scratch (strided batches of such sizes need one)
            *cap_out = batch;
            return PHAST_OK;
        }
        Workspace &w = *L.ws;
        hipStream_t stream = L.stream;
        const size_t per = 2 * sstride() * sizeof(T);
        if (w.cap && w.per != per) {  // cut for another plan's pitches (set_plan): start over
            w.retire(reinterpret_cast<char *>(w.d_scratch) - w.guard, w.cap * w.per + 2 * w.guard, false, stream);
            w.d_scratch = nullptr;
            w.cap = w.guard = 0;
        }
        w.per = per;
        size_t target = scratch_target_bytes() / (2 * n * sizeof(T));  // counted in transforms of the caller's size: the
        if (target < 1) target = 1;                                   // row padding of the scratch rides on top (1.5-3 %)
        if (target < reserve) target = reserve;
        size_t want = target;
        if (want > batch && batch >= reserve) want = batch;
        if (exact && want < batch) want = batch;  // work that cannot be cut into chunks (strided batches)
        if (w.cap < want && !exact && w.cap >= std::max<size_t>(reserve, 1) && capturing(stream)) {
            *cap_out = w.cap;  // under capture nothing may be allocated: the batch runs in the chunks this scratch allows
            return PHAST_OK;
        }
        if (w.cap < want) {
            if (!exact && w.cap && want < 2 * w.cap) want = std::min(2 * w.cap, std::max(target, want));
            const size_t floor_cap = exact ? want : std::max<size_t>(std::max<size_t>(reserve, 1), w.cap + 1);
            const size_t guard = g_guard_bytes;
            void *d = nullptr;
            hipError_t e = hipMalloc(&d, want * per + 2 * guard);
            if (e == hipErrorOutOfMemory && !capturing(stream)) {
                (void)hipGetLastError();
                w.reap(stream, true);  // whatever retired buffers are idle (or become so) go first
                e = hipMalloc(&d, want * per + 2 * guard);
                while (e == hipErrorOutOfMemory && want > floor_cap) {
                    (void)hipGetLastError();
                    want = std::max(floor_cap, want / 2);
                    e = hipMalloc(&d, want * per + 2 * guard);
                }
                if (e == hipErrorOutOfMemory && !exact && w.cap >= std::max<size_t>(reserve, 1)) {
                    (void)hipGetLastError();  // cannot grow: keep working in the chunks the present scratch allows
                    *cap_out = w.cap;
                    return PHAST_OK;
                }
            }
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(scratch)");
            if (guard) {
                PHAST_HIP(hipMemset(d, kGuardFill, guard));
                PHAST_HIP(hipMemset(reinterpret_cast<char *>(d) + guard + want * per, kGuardFill, guard));
            }
            w.retire(w.d_scratch ? reinterpret_cast<char *>(w.d_scratch) - w.guard : nullptr, w.cap * per + 2 * w.guard, false, stream);
            w.d_scratch = reinterpret_cast<char *>(d) + guard;
            w.guard = guard;
            w.cap = want;
        }
        *cap_out = w.cap;
        return PHAST_OK;
    }

    // debug: bytes of the scratches' guard bands that no longer hold the fill value (blocks until the device is idle)
    int check_guards(size_t *bad_out) const {
        PHAST_ON_DEVICE(device);
        std::lock_guard<std::mutex> lk(mu);
        size_t bad = 0;
        PHAST_HIP(hipDeviceSynchronize());
        for (auto &wp : pool) {
            const Workspace &w = *wp;
            if (!w.d_scratch || !w.guard) continue;
            std::vector<unsigned char> h(w.guard);
            const char *lo = reinterpret_cast<const char *>(w.d_scratch) - w.guard;
            const char *hi = reinterpret_cast<const char *>(w.d_scratch) + w.cap * w.per;
            for (const char *band : {lo, hi}) {
                PHAST_HIP(hipMemcpy(h.data(), band, w.guard, hipMemcpyDeviceToHost));
                for (unsigned char c : h) bad += c != kGuardFill;
            }
        }
        *bad_out = bad;
        return PHAST_OK;
    }
    // live tables and scratch plus what is retired but not yet released
    size_t device_bytes() const {
        std::lock_guard<std::mutex> lk(mu);
        size_t b = table_bytes + old_table_bytes;
        for (auto &w : pool) b += w->device_bytes();
        if (twin) b += twin->device_bytes();
        return b;
    }

    // tables + launch parameters of a pass list (shared by set_plan and the strided plans)
    int prepare_passes(std::vector<PassDesc> &ps, size_t *table_bytes_out) const {
        size_t tb = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            int rc = ps[i].quad ? upload<T>(host_twq<T>(), &ps[i].d_twr) : upload<T>(host_twr<T>(1u << ps[i].lr), &ps[i].d_twr);
            if (rc == PHAST_OK && ps[i].pre_tw) {
                rc = upload<T>(host_tw3<T>(ps[i].log_mod(), ps[i].tw_bits), &ps[i].d_tw3);
                tb += ((size_t)3 << ps[i].tw_bits) * sizeof(cx_t<T>);
            }
            tb += 64 * sizeof(cx_t<T>);
            if (rc == PHAST_OK) {
                TileArgs ta{};
                ta.tw_bits = ps[i].tw_bits;
                hipError_t e = ps[i].wave ? launch_wave<T>(ps[i].transpose, nullptr, ta, true, &ps[i].blocks_per_cu, &ps[i].lds)
                               : ps[i].quad ? launch_quad<T>(0, nullptr, ta, true, &ps[i].blocks_per_cu, &ps[i].lds)
                               : ps[i].transpose
                                   ? Types<T>::launch_a(ps[i].lr, ps[i].lc, (int)ps[i].lp, 0, nullptr, ta, true, &ps[i].blocks_per_cu, &ps[i].lds)
                                   : Types<T>::launch_bc(ps[i].lr, ps[i].lc, (int)ps[i].lp, 0, nullptr, ta, true, &ps[i].blocks_per_cu, &ps[i].lds);
                if (e != hipSuccess) rc = hip_fail(e, "occupancy query");
                if (ps[i].blocks_per_cu < 1) ps[i].blocks_per_cu = 1;
                if (rc == PHAST_OK && ps[i].lds > 160 * 1024) rc = PHAST_ERR_INVALID_ARG;  // tile does not fit one CU's LDS
                // the fused R2C form of a LAST pass: generic tiles with <= 16 points per thread whose columns span at least
                // two tiles (r2c_fused.hpp); anything else keeps the separate untangle sweep
                const PassDesc &q = ps[i];
                if (rc == PHAST_OK && i + 1 == ps.size() && i > 0 && !q.wave && !q.quad && !q.strided && q.pre_tw &&
                    q.log_s_in >= q.lc + 1 && r2c_shape_ok(q.lr, q.lc, q.lp, sizeof(T))) {
                    std::vector<cx_t<T>> h((size_t)1 << q.lr);
                    for (size_t k = 0; k < h.size(); ++k) h[k] = twiddle_t<T>(k, 2ull << q.lr);
                    rc = upload<T>(h, &ps[i].d_twu);
                    if (rc == PHAST_OK) {
                        R2cFuseArgs fa{};
                        int b = 0;
                        hipError_t e2 = launch_r2c_last<T>((int)q.lr, (int)q.lc, (int)q.lp, 0, nullptr, ta, fa, true, &b);
                        ps[i].r2c_blocks = e2 == hipSuccess ? b : 0;
                        (void)hipGetLastError();
                    }
                }
            }
            // the fused C2R form of a FIRST pass of a contiguous transform: generic tiles, at least two of them per transform
            if (rc == PHAST_OK) {
                const PassDesc &q = ps[i];
                if (i == 0 && ps.size() > 1 && !q.wave && !q.quad && !q.strided && q.transpose && !q.pre_tw &&
                    q.log_s_in >= q.lc + 1 && c2r_shape_ok(q.lr, q.lc, q.lp, sizeof(T))) {
                    std::vector<cx_t<T>> h((size_t)1 << q.lr);
                    for (size_t k = 0; k < h.size(); ++k) h[k] = twiddle_t<T>(k, 2ull << q.lr);
                    rc = upload<T>(h, &ps[i].d_twu);
                    if (rc == PHAST_OK) {
                        TileArgs ta{};
                        ta.tw_bits = q.tw_bits;
                        C2rFuseArgs fa{};
                        int b = 0;
                        hipError_t e2 = launch_c2r_first<T>((int)q.lr, (int)q.lc, (int)q.lp, 0, nullptr, ta, fa, true, &b);
                        ps[i].c2r_blocks = e2 == hipSuccess ? b : 0;
                        (void)hipGetLastError();
                    }
                }
            }
            if (rc != PHAST_OK) {
                for (auto &p : ps) {
                    if (p.d_tw3) hipFree(p.d_tw3);
                    if (p.d_twr) hipFree(p.d_twr);
                    if (p.d_twu) hipFree(p.d_twu);
                    p.d_tw3 = p.d_twr = p.d_twu = nullptr;
                }
                return rc;
            }
        }
        if (table_bytes_out) *table_bytes_out = tb;
        return PHAST_OK;
    }

    hipError_t launch_pass(const PassDesc &p, const TileArgs &ta, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) const {
        unsigned grid = (unsigned)(g_wg_per_cu_override > 0 ? g_wg_per_cu_override : p.blocks_per_cu) * (unsigned)cus_of(device);
        if (grid > ta.tiles_total) grid = ta.tiles_total;
        // keep tile%8 == workgroup%8 (XCD affinity of the tile order) -- where that order is in force (TileBody::locate: tile
        // counts that are a multiple of 8); rounding 12 tiles down to 8 workgroups made four of them run two tiles in a row:
        // 3 x 2^14 f64 23.2 us where 4 x 2^14 takes 14.1 (profiles/r04_small_batch_plans.log)
        if (grid >= 8 && (ta.tiles_total & 7u) == 0u) grid &= ~7u;
        return p.wave      ? launch_wave<T>(p.transpose, stream, ta, false, nullptr, nullptr, e0, e1)
               : p.quad    ? launch_quad<T>(grid, stream, ta, false, nullptr, nullptr, e0, e1)
               : p.transpose ? Types<T>::launch_a(p.lr, p.lc, (int)p.lp, grid, stream, ta, false, nullptr, nullptr, e0, e1)
                             : Types<T>::launch_bc(p.lr, p.lc, (int)p.lp, grid, stream, ta, false, nullptr, nullptr, e0, e1);
    }

    // Strided batch ("column FFTs"): 2^sb transforms, transform c at element c, points 2^s elements apart, in place in
    // the caller's planes (forward arithmetic; `scale` on the last store).  make_strided_passes has the layouts.
    int exec_strided(T *re, T *im, unsigned s_bits, unsigned sb_bits, double scale, hipStream_t stream,
                     unsigned grid_log_n = 0, unsigned grid_col0 = 0) const {
        PHAST_ON_DEVICE(device);
        Lease L;
        int rc = check_out(L, stream);
        if (rc) return rc;
        const StridedPlan *plan = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (const auto &sp : strided_plans)
                if (sp->s == s_bits && sp->sb == sb_bits && sp->grid_log_n == grid_log_n) plan = sp.get();
        }
        if (!plan) {
            std::vector<PassGeom> geo;
            if (!make_strided_passes(log_n, s_bits, sb_bits, sizeof(T), geo, grid_log_n)) return PHAST_ERR_INVALID_ARG;
            std::unique_ptr<StridedPlan> sp(new StridedPlan());
            sp->s = s_bits;
            sp->sb = sb_bits;
            sp->grid_log_n = grid_log_n;
            sp->passes.resize(geo.size());
            for (size_t i = 0; i < geo.size(); ++i) static_cast<PassGeom &>(sp->passes[i]) = geo[i];
            rc = prepare_passes(sp->passes, nullptr);
            if (rc) return rc;
            std::lock_guard<std::mutex> lk(mu);
            for (const auto &q : strided_plans)  // another thread may have built the same plan meanwhile
                if (q->s == s_bits && q->sb == sb_bits && q->grid_log_n == grid_log_n) plan = q.get();
            if (plan) {
                free_passes(sp->passes);
            } else {
                strided_plans.push_back(std::move(sp));
                plan = strided_plans.back().get();
            }
        }
        // the whole [2^log_n][2^s] array is one unit of work: scratch for all of it (2^s "transforms" of n points)
        const size_t cols = (size_t)1 << s_bits;
        size_t cap = 0;
        rc = ensure_scratch(L, cols, &cap, true);
        if (rc) return rc;
        T *s_re = reinterpret_cast<T *>(L.ws->d_scratch), *s_im = s_re + cap * n;
        const size_t np = plan->passes.size();
        for (size_t i = 0; i < np; ++i) {
            const PassDesc &p = plan->passes[i];
            TileArgs ta{};
            // x -> scratch -> x (-> x): every pass but the last moves the data (digits change places)
            const bool from_x = (i % 2) == 0 || i + 1 == np && np == 3;
            const bool to_x = (i % 2) == 1 || i + 1 == np;
            ta.in_re = from_x ? re : s_re;
            ta.in_im = from_x ? im : s_im;
            ta.out_re = to_x ? re : s_re;
            ta.out_im = to_x ? im : s_im;
            ta.in_dist = ta.out_dist = 0;
            ta.scale = i + 1 == np ? scale : 1.0;
            ta.tw3 = p.d_tw3;
            ta.twr = p.d_twr;
            geom_to_args(p, log_n, 1, ta);
            ta.grid_col0 = grid_col0;
            hipError_t e = launch_pass(p, ta, stream, nullptr, nullptr);
            if (e != hipSuccess) return hip_fail(e, "tile_fft launch (strided)");
        }
        return PHAST_OK;
    }

    std::string describe() const {
        char buf[512];
        std::string s = "n=2^" + std::to_string(log_n);
        auto add = [&](const char *tag, const std::vector<PassDesc> &v) {
            s += std::string(" ") + tag + "=" + std::to_string(v.size()) + "p";
            for (auto &p : v) {
                std::snprintf(buf, sizeof buf, "[%ux%u%s %s%u lds=%zu wg/cu=%d]", 1u << p.lr, 1u << p.lc,
                              p.transpose ? "A" : "", p.wave ? "w" : p.quad ? "q" : "p", 1u << p.lp, p.lds, p.blocks_per_cu);
                s += buf;
            }
        };
        if (passes.empty()) {
            s += " one pass (whole transforms on chip)";
            if (twin) add("single", twin->plan_for(1));
            return s;
        }
        add("throughput", passes);
        if (!passes_mid.empty()) add("mid", passes_mid);
        if (!passes_lat.empty()) add("latency", passes_lat);
        if (!passes_one.empty()) add("single", passes_one);
        if (!passes_c2r_one.empty()) add("c2r-single", passes_c2r_one);
        if (!passes_c2r_lat.empty()) add("c2r-latency", passes_c2r_lat);
        if (!passes_r2c.empty()) add("r2c-single", passes_r2c);
        if (!passes_r2c_tp.empty()) add("r2c-batch", passes_r2c_tp);
        if (!passes_c2r_tp.empty()) add("c2r-batch", passes_c2r_tp);
        return s;
    }

    // Small real transforms (the N-point core runs in the one-pass kernel): R2C untangle / C2R preprocess fused
    // into that kernel (row_fft.hpp, RowArgs::real_mode).  rtw3 = W_{2N} tables of the R2C planner.
    int exec_small_real(unsigned mode, const void *in_a, const void *in_b, size_t in_dist, void *out_a, void *out_b,
                        size_t out_dist, size_t batch, double scale, const void *rtw3, unsigned rtw_bits,
                        hipStream_t stream) const {
        if (batch == 0) return PHAST_OK;
        PHAST_ON_DEVICE(device);  // no workspace: the one-pass kernel keeps whole transforms on chip
        const size_t chunk = (size_t)1 << 30;
        const size_t in_el = mode == 1 ? 2 * sizeof(T) : sizeof(T), out_el = mode == 1 ? sizeof(T) : 2 * sizeof(T);
        for (size_t b0 = 0; b0 < batch; b0 += chunk) {
            SmallArgs sa{};
            const size_t nb = batch - b0 < chunk ? batch - b0 : chunk;
            sa.in_re = (const char *)in_a + b0 * in_dist * in_el;
            sa.in_im = in_b ? (const char *)in_b + b0 * in_dist * in_el : nullptr;
            sa.out_re = (char *)out_a + b0 * out_dist * out_el;
            sa.out_im = out_b ? (char *)out_b + b0 * out_dist * out_el : nullptr;
            sa.tw = d_small_tw;
            sa.in_dist = in_dist;
            sa.out_dist = out_dist;
            sa.log_n = log_n;
            sa.batch = (unsigned)nb;
            sa.in_interleaved = mode == 1 ? 1 : 0;
            sa.out_interleaved = mode == 1 ? 0 : 2;
            sa.scale = scale;
            sa.real_mode = mode;
            sa.rtw_bits = rtw_bits;
            sa.rtw3 = rtw3;
            PHAST_HIP(launch_small_fft<T>(sa, stream, nullptr, nullptr));
        }
        return PHAST_OK;
    }

    // One batched transform: in -> out (may alias for the planar in-place case), forward arithmetic,
    // output scaled by `scale`.  in_mode/out_mode: 0 planar, 1 interleaved (re,im), 2 interleaved (im,re).
    // The fused R2C last pass runs HALF as many tiles, each twice as long: it pays once the tiles fill the chip -- from
    // 2^23 complex points in flight (profiles/r03_r2c_fused_ab.log: f32 N = 2^24 108.6 -> 89.9 us; below, one transform is
    // latency-bound and loses: N = 2^20 19.5 -> 31.6 us, 2^22 39.4 -> 42.9).
    static unsigned fuse_min_log() {  // PHAST_R2C_FUSE_MIN_LOG: tools (A/B of the threshold)
        static const unsigned v = [] {
            const char *e = std::getenv("PHAST_R2C_FUSE_MIN_LOG");
            return (e && *e) ? (unsigned)std::atoi(e) : 0u;
        }();
        return v;
    }
    bool fuse_pays(size_t batch) const {
        const unsigned min_log = fuse_min_log() ? fuse_min_log() : 23u;
        if (!r2c_fuse_enabled()) return false;
        if (batch <= 2 && r2c_table_fuses && !passes_r2c.empty()) return true;  // a plan cut for it: 2048-point last-pass tiles
        return batch * n >= ((size_t)1 << min_log);
    }
    // a _dev call: checks a workspace out for the enqueue
    int exec(const void *in_re, const void *in_im, size_t in_dist, unsigned in_mode, void *out_re, void *out_im,
             size_t out_dist, unsigned out_mode, size_t batch, double scale, hipStream_t stream,
             PassTimer *timer = nullptr) const {
        if (batch == 0) return PHAST_OK;
        if (twin && batch <= twin_max_batch())  // one 8192-point transform: two passes over the whole chip instead of one workgroup
            return twin->exec(in_re, in_im, in_dist, in_mode, out_re, out_im, out_dist, out_mode, batch, scale, stream, timer);
        PHAST_ON_DEVICE(device);
        Lease L;
        if (!passes.empty()) {  // (the one-pass kernel keeps whole transforms on chip: nothing to check out)
            int rc = check_out(L, stream);
            if (rc) return rc;
        } else {
            L.stream = stream;
        }
        return exec_in(L, in_re, in_im, in_dist, in_mode, out_re, out_im, out_dist, out_mode, batch, scale, timer);
    }
    // The launches of one batched transform in the leased workspace, on L.stream.
    // `fuse` (R2C): the last pass takes the untangle with it where its fused form exists; *fused_out says whether it did;
    // *np_out = the number of passes of the plan that ran (the timer slots 0 .. np - 1 belong to them)
    int exec_in(const Lease &L, const void *in_re, const void *in_im, size_t in_dist, unsigned in_mode, void *out_re,
                void *out_im, size_t out_dist, unsigned out_mode, size_t batch, double scale, PassTimer *timer = nullptr,
                const R2cFuse *fuse = nullptr, bool *fused_out = nullptr, size_t *np_out = nullptr) const {
        if (fused_out) *fused_out = false;
        if (np_out) *np_out = 1;
        if (batch == 0) return PHAST_OK;
        hipStream_t stream = L.stream;
        if (passes.empty()) {
            const size_t chunk = (size_t)1 << 30;  // transforms per launch: the tile count stays below 2^32
            for (size_t b0 = 0; b0 < batch; b0 += chunk) {
                SmallArgs sa{};
                const size_t nb = batch - b0 < chunk ? batch - b0 : chunk;
                const size_t isz = in_mode ? 2 * sizeof(T) : sizeof(T), osz = out_mode ? 2 * sizeof(T) : sizeof(T);
                sa.in_re = (const char *)in_re + b0 * in_dist * isz;
                sa.in_im = in_im ? (const char *)in_im + b0 * in_dist * isz : nullptr;
                sa.out_re = (char *)out_re + b0 * out_dist * osz;
                sa.out_im = out_im ? (char *)out_im + b0 * out_dist * osz : nullptr;
                sa.tw = d_small_tw;
                sa.in_dist = in_dist;
                sa.out_dist = out_dist;
                sa.log_n = log_n;
                sa.batch = (unsigned)nb;
                sa.in_interleaved = in_mode;
                sa.out_interleaved = out_mode;
                sa.scale = scale;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (timer) PHAST_HIP(timer->pair(0, &e0, &e1));
                PHAST_HIP(launch_small_fft<T>(sa, stream, e0, e1));
            }
            return PHAST_OK;
        }
        size_t cap = 0;
        int rc = ensure_scratch(L, batch, &cap, false);
        if (rc) return rc;
        const size_t sd = sstride();       // elements per transform and plane in the (padded) scratch
        T *s_re = reinterpret_cast<T *>(L.ws->d_scratch);  // plane layout: all re planes, then all im planes
        T *s_im = s_re + cap * sd;
        // R2C: fused or not is decided ONCE per call, from the size of a full chunk -- a smaller tail chunk follows the others
        // (the caller runs the untangle sweep over the whole batch or not at all)
        const bool r2c_fuse = fuse && in_mode != 3 && fuse_pays(batch < cap ? batch : cap);
        const std::vector<PassDesc> &passes = in_mode == 3 ? plan_for_c2r(batch) : fuse ? plan_for_r2c(batch, r2c_fuse) : plan_for(batch);
        const size_t np = passes.size();
        if (np_out) *np_out = np;
        for (size_t b0 = 0; b0 < batch; b0 += cap) {
            const size_t nb = batch - b0 < cap ? batch - b0 : cap;
            for (size_t i = 0; i < np; ++i) {
                const PassDesc &p = passes[i];
                TileArgs ta{};
                const bool first = i == 0, last = i + 1 == np;
                if (first) {
                    const size_t isz = (in_mode == 1 || in_mode == 2) ? 2 * sizeof(T) : sizeof(T);
                    ta.in_re = (const char *)in_re + b0 * in_dist * isz;
                    ta.in_im = in_im ? (const char *)in_im + b0 * in_dist * isz : nullptr;
                    ta.in_dist = in_dist;
                    ta.in_interleaved = in_mode;
                } else {
                    ta.in_re = s_re;
                    ta.in_im = s_im;
                    ta.in_dist = sd;
                }
                if (last) {
                    const size_t osz = out_mode ? 2 * sizeof(T) : sizeof(T);
                    ta.out_re = (char *)out_re + b0 * out_dist * osz;
                    ta.out_im = out_im ? (char *)out_im + b0 * out_dist * osz : nullptr;
                    ta.out_dist = out_dist;
                    ta.out_interleaved = out_mode;
                    ta.scale = scale;
                } else {
                    ta.out_re = s_re;
                    ta.out_im = s_im;
                    ta.out_dist = sd;
                    ta.scale = 1.0;
                }
                ta.tw3 = p.d_tw3;
                ta.twr = p.d_twr;
                ta.trace = g_trace ? g_trace + (size_t)i * 16 * 4096 : nullptr;
                if (((unsigned long long)nb << (log_n - p.lr - p.lc)) > 0xffffffffull) return PHAST_ERR_INVALID_ARG;
                geom_to_args(p, log_n, nb, ta);
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (timer) PHAST_HIP(timer->pair((int)i, &e0, &e1));
                if (first && in_mode == 3) {  // C2R: the half-spectrum planes, z formed on load (c2r_fused.hpp)
                    if (!fuse || p.c2r_blocks <= 0) return PHAST_ERR_INVALID_ARG;
                    ta.in_interleaved = 0;
                    C2rFuseArgs fa{};
                    fa.tw3n = fuse->tw3n;
                    fa.twn_bits = fuse->twn_bits;
                    fa.twu = p.d_twu;
                    unsigned grid = (unsigned)p.c2r_blocks * (unsigned)cus_of(device);
                    if (grid > ta.tiles_total) grid = ta.tiles_total;
                    if (grid >= 8 && (ta.tiles_total & 7u) == 0u) grid &= ~7u;
                    hipError_t e = launch_c2r_first<T>((int)p.lr, (int)p.lc, (int)p.lp, grid, stream, ta, fa, false, nullptr, e0, e1);
                    if (e != hipSuccess) return hip_fail(e, "c2r_first_pass launch");
                    continue;
                }
                if (last && r2c_fuse && p.r2c_blocks > 0 && out_mode == 0 && scale == 1.0) {
                    R2cFuseArgs fa{};
                    fa.tw3n = fuse->tw3n;
                    fa.twn_bits = fuse->twn_bits;
                    fa.twu = p.d_twu;
                    fa.tiles_per_xform = (1u << (p.log_s_in - p.lc - 1)) + 1u;
                    if ((unsigned long long)nb * fa.tiles_per_xform > 0xffffffffull) return PHAST_ERR_INVALID_ARG;
                    fa.tiles_total = (unsigned)(nb * fa.tiles_per_xform);
                    fa.pair_tiles = (unsigned)(nb * (fa.tiles_per_xform - 1u));
                    unsigned grid = (unsigned)p.r2c_blocks * (unsigned)cus_of(device);
                    if (grid > fa.tiles_total) grid = fa.tiles_total;
                    if (grid >= 8) grid &= ~7u;
                    hipError_t e = launch_r2c_last<T>((int)p.lr, (int)p.lc, (int)p.lp, grid, stream, ta, fa, false, nullptr, e0, e1);
                    if (e != hipSuccess) return hip_fail(e, "r2c_last_pass launch");
                    if (fused_out) *fused_out = true;
                    continue;
                }
                hipError_t e = launch_pass(p, ta, stream, e0, e1);
                if (e != hipSuccess) return hip_fail(e, "tile_fft launch");
            }
        }
        return PHAST_OK;
    }
};

// ------------------------------------------------------------------------------------------------
// R2C planner (planner.rs:164-212)
// ------------------------------------------------------------------------------------------------
template <typename T> struct PlannerR2c {
    using Lease = typename Planner<T>::Lease;
    size_t n = 0;
    Planner<T> dit;         // the inner N/2-point transform; its workspace pool serves the real transforms too
    void *d_tw3 = nullptr;  // W_N^e three-level table for the untangle / c2r-preprocess passes
    unsigned tw_bits = 1;
    ~PlannerR2c() {
        DeviceGuard on(dit.device);
        if (d_tw3) hipFree(d_tw3);
    }
    std::unique_ptr<PlannerR2c<T>> twin;  // N/2 = 8192 only: the multi-pass form for ONE real transform (Planner::twin)
    int init(size_t n_, bool force_multi = false) {
        n = n_;
        int rc = dit.init(n / 2, force_multi, false);
        if (rc) return rc;
        if (!dit.passes.empty()) {
            rc = dit.make_c2r_plans();
            if (rc) return rc;
        }
        tw_bits = tw3_bits_for(ilog2(n));
        rc = upload<T>(host_tw3<T>(ilog2(n), tw_bits), &d_tw3);
        if (rc == PHAST_OK && !force_multi && dit.log_n == kSmallMaxLog && Planner<T>::twin_enabled()) {
            twin.reset(new (std::nothrow) PlannerR2c<T>());
            if (twin && twin->init(n, true) != PHAST_OK) twin.reset();
        }
        return rc;
    }
    // unfused C2R: the preprocess workspace for `batch` transforms in the leased workspace; an outgrown one is retired
    // (freed once idle, Workspace::reap), growth is geometric
    int ensure_z(const Lease &L, size_t batch, size_t *cap_out) const {
        Workspace &w = *L.ws;
        hipStream_t stream = L.stream;
        const size_t per = n * sizeof(T);  // 2 planes of n/2
        size_t target = scratch_target_bytes() / per;
        if (target < 1) target = 1;
        size_t want = target < batch ? target : batch;
        if (w.z_cap < want) {
            if (w.z_cap && want < 2 * w.z_cap) want = std::min(2 * w.z_cap, target);
            void *d = nullptr;
            hipError_t e = hipMalloc(&d, want * per);
            while (e == hipErrorOutOfMemory && want > w.z_cap + 1 && !Planner<T>::capturing(stream)) {
                (void)hipGetLastError();
                w.reap(stream, true);
                want = std::max(w.z_cap + 1, want / 2);
                e = hipMalloc(&d, want * per);
            }
            if (e == hipErrorOutOfMemory && w.z_cap) {
                (void)hipGetLastError();
                *cap_out = w.z_cap;
                return PHAST_OK;
            }
            if (e != hipSuccess) return hip_fail(e, "hipMalloc(c2r workspace)");
            w.retire(w.d_z, w.z_bytes, false, stream);
            w.d_z = d;
            w.z_cap = want;
            w.z_bytes = want * per;
        }
        *cap_out = w.z_cap;
        return PHAST_OK;
    }

    bool fuses(size_t batch) const {
        if (dit.passes.empty() || !dit.fuse_pays(batch)) return false;
        const auto &ps = dit.plan_for_r2c(batch);
        return !ps.empty() && ps.back().r2c_blocks > 0;
    }
    bool c2r_fuses(size_t batch) const {
        if (dit.passes.empty() || !c2r_fuse_enabled()) return false;
        const auto &ps = dit.plan_for_c2r(batch);
        return !ps.empty() && ps.front().c2r_blocks > 0;
    }
    // r2c.rs:535-593 / 607-662 on device pointers (a _dev call: checks a workspace out for the enqueue)
    int r2c(const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist, size_t out_dist, hipStream_t s,
            PassTimer *timer = nullptr) const {
        if (in_dist & 1) return PHAST_ERR_INVALID_ARG;  // the input is read as (even, odd) pairs
        // (... and large batches too where the twin has a plan ranked for them: f32, plan.hpp: real_batch_plan)
        if (twin && (batch <= Planner<T>::twin_max_batch() || (batch * (n / 2) >= ((size_t)1 << 24) && !twin->dit.passes_r2c_tp.empty())))
            return twin->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, s, timer);
        PHAST_ON_DEVICE(dit.device);
        Lease L;
        if (!dit.passes.empty()) {
            int rc = dit.check_out(L, s);
            if (rc) return rc;
        } else {
            L.stream = s;
        }
        return r2c_in(L, d_in, d_ore, d_oim, batch, in_dist, out_dist, timer);
    }
    int r2c_in(const Lease &L, const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist, size_t out_dist,
               PassTimer *timer = nullptr) const {
        const size_t half = n / 2;
        hipStream_t s = L.stream;
        if (dit.passes.empty())  // N/2 <= 8192: one kernel, the untangle is its epilogue
            return dit.exec_small_real(1, d_in, nullptr, in_dist / 2, d_ore, d_oim, out_dist, batch, 1.0, d_tw3, tw_bits, s);
        const R2cFuse fuse{d_tw3, tw_bits};
        bool fused = false;
        size_t np = 0;
        int rc = dit.exec_in(L, d_in, nullptr, in_dist / 2, 1, d_ore, d_oim, out_dist, 0, batch, 1.0, timer, &fuse, &fused, &np);
        if (rc) return rc;
        if (fused) return PHAST_OK;  // the last pass wrote X[k] and X[h - k] itself (r2c_fused.hpp)
        const int untangle_slot = (int)np;  // timer slot after the passes of the plan that ran
        for (size_t b0 = 0; b0 < batch; b0 += 65535) {
            UntangleArgs ua{};
            ua.re = d_ore + b0 * out_dist;
            ua.im = d_oim + b0 * out_dist;
            ua.tw3 = d_tw3;
            ua.dist = out_dist;
            ua.half = (unsigned)half;
            ua.tw_bits = tw_bits;
            ua.batch = (unsigned)(batch - b0 < 65535 ? batch - b0 : 65535);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timer) PHAST_HIP(timer->pair(untangle_slot, &e0, &e1));
            PHAST_HIP(launch_untangle<T>(ua, s, e0, e1));
        }
        return PHAST_OK;
    }

    // r2c.rs:740-790 / 836-895 on device pointers
    int c2r(const T *d_ire, const T *d_iim, T *d_out, size_t batch, size_t in_dist, size_t out_dist,
            hipStream_t s, PassTimer *timer = nullptr) const {
        if (out_dist & 1) return PHAST_ERR_INVALID_ARG;
        if (twin && (batch <= Planner<T>::twin_max_batch() || (batch * (n / 2) >= ((size_t)1 << 24) && !twin->dit.passes_c2r_tp.empty())))
            return twin->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, s, timer);
        PHAST_ON_DEVICE(dit.device);
        Lease L;
        if (!dit.passes.empty()) {
            int rc = dit.check_out(L, s);
            if (rc) return rc;
        } else {
            L.stream = s;
        }
        return c2r_in(L, d_ire, d_iim, d_out, batch, in_dist, out_dist, timer);
    }
    int c2r_in(const Lease &L, const T *d_ire, const T *d_iim, T *d_out, size_t batch, size_t in_dist, size_t out_dist,
               PassTimer *timer = nullptr) const {
        const size_t half = n / 2;
        hipStream_t s = L.stream;
        if (dit.passes.empty())  // N/2 <= 8192: one kernel, the preprocess is its prologue
            return dit.exec_small_real(2, d_ire, d_iim, in_dist, d_out, nullptr, out_dist / 2, batch, 1.0 / (double)half,
                                       d_tw3, tw_bits, s);
        if (c2r_fuses(batch)) {  // the first pass forms z on load: no preprocess sweep, no workspace (c2r_fused.hpp)
            const R2cFuse fuse{d_tw3, tw_bits};
            return dit.exec_in(L, d_ire, d_iim, in_dist, 3, d_out, nullptr, out_dist / 2, 2, batch, 1.0 / (double)half, timer, &fuse);
        }
        size_t cap = 0;
        int rc = ensure_z(L, batch, &cap);
        if (rc) return rc;
        for (size_t b0 = 0; b0 < batch; b0 += cap) {
            const size_t nb = batch - b0 < cap ? batch - b0 : cap;
            T *z_re = reinterpret_cast<T *>(L.ws->d_z), *z_im = z_re + cap * half;
            C2rPreArgs pa{};
            pa.in_re = d_ire + b0 * in_dist;
            pa.in_im = d_iim + b0 * in_dist;
            pa.z_re = z_re;
            pa.z_im = z_im;
            pa.tw3 = d_tw3;
            pa.in_dist = in_dist;
            pa.z_dist = half;
            pa.half = (unsigned)half;
            pa.tw_bits = tw_bits;
            pa.batch = (unsigned)nb;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            // the sweep's timer slot sits after the inner passes: of the plan a chunk of nb transforms runs (exec_in
            // picks it from nb too)
            if (timer) PHAST_HIP(timer->pair((int)dit.plan_for(nb).size(), &e0, &e1));
            PHAST_HIP(launch_c2r_preprocess<T>(pa, s, e0, e1));
            // inverse by the swap trick (algorithms/dit.rs:297-300): forward FFT of (z_im, z_re), 1/half scale,
            // and the (positional re, positional im) = (caller im, caller re) pair is stored as (im, re)
            rc = dit.exec_in(L, z_im, z_re, half, 0, d_out + b0 * out_dist, nullptr, out_dist / 2, 2, nb, 1.0 / (double)half, timer);
            if (rc) return rc;
        }
        return PHAST_OK;
    }
};

// ------------------------------------------------------------------------------------------------
// helpers shared by the C entry points
// ------------------------------------------------------------------------------------------------
template <typename P> static int planner_new(size_t n, P **out) {  // P = Planner<T> or the C-ABI struct over it
    if (!out) return PHAST_ERR_INVALID_ARG;
    *out = nullptr;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;  // planner.rs:66
    auto *p = new (std::nothrow) P();
    if (!p) return PHAST_ERR_ALLOC;
    int rc = p->init(n);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return PHAST_OK;
}

template <typename P> static int r2c_planner_new(size_t n, P **out) {
    if (!out) return PHAST_ERR_INVALID_ARG;
    *out = nullptr;
    if (n < 4 || !is_pow2(n)) return PHAST_ERR_R2C_N;  // planner.rs:195
    auto *p = new (std::nothrow) P();
    if (!p) return PHAST_ERR_ALLOC;
    int rc = p->init(n);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return PHAST_OK;
}

// algorithms/dit.rs:276-332: asserts, swap trick, transform, 1/N scale -- on device pointers
template <typename T>
static int fft_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, int direction, const Planner<T> *pl,
                   hipStream_t s) {
    if (!pl || !d_re || !d_im) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (batch > 1 && dist < n) return PHAST_ERR_INVALID_ARG;
    if (direction == PHAST_REVERSE) return pl->exec(d_im, d_re, dist, 0, d_im, d_re, dist, 0, batch, 1.0 / (double)n, s);
    return pl->exec(d_re, d_im, dist, 0, d_re, d_im, dist, 0, batch, 1.0, s);
}

// `count` independent transforms at arbitrary addresses, each run exactly as a single-transform call (the plan for ONE
// transform), enqueued back to back by one host call: the host-side cost per transform drops to the kernel launches
// themselves (a Python / FFI caller pays its call overhead once).
template <typename T>
static int fft_dev_many(T *const *d_re, T *const *d_im, size_t count, size_t n, int direction, const Planner<T> *pl,
                        hipStream_t s) {
    if (!pl || (!d_re && count) || (!d_im && count)) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    for (size_t i = 0; i < count; ++i)
        if (!d_re[i] || !d_im[i]) return PHAST_ERR_INVALID_ARG;
    if (count == 0) return PHAST_OK;
    if (pl->twin) pl = pl->twin.get();  // 8192 points: the plan of a single-transform call (Planner::twin)
    PHAST_ON_DEVICE(pl->device);
    typename Planner<T>::Lease L;  // one workspace for the whole list: the transforms follow each other on the stream
    if (!pl->passes.empty()) {
        int rc = pl->check_out(L, s);
        if (rc) return rc;
    } else {
        L.stream = s;
    }
    for (size_t i = 0; i < count; ++i) {
        int rc = direction == PHAST_REVERSE
                     ? pl->exec_in(L, d_im[i], d_re[i], n, 0, d_im[i], d_re[i], n, 0, 1, 1.0 / (double)n)
                     : pl->exec_in(L, d_re[i], d_im[i], n, 0, d_re[i], d_im[i], n, 0, 1, 1.0);
        if (rc) return rc;
    }
    return PHAST_OK;
}

// Strided batches on device pointers: transform b occupies elements b*dist + j*stride, j < n.
//   stride == 1: contiguous transforms `dist` apart (the plain batched path above);
//   dist == 1:   "column FFTs" of a row-major [n][stride] array -- stride and batch powers of two, batch <= stride,
//                n >= 64 (what a four-step split and any multi-dimensional transform need; no reference counterpart,
//                SURVEY.md section 8b suggested the signature).
template <typename T>
static int fft_strided_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, size_t stride, int direction,
                           const Planner<T> *pl, hipStream_t s, size_t tw_n = 0, size_t tw_col0 = 0) {
    if (tw_n && (stride == 1 || !is_pow2(tw_n) || tw_n < n || ilog2(tw_n) > 32 || tw_col0 + batch > tw_n))
        return PHAST_ERR_INVALID_ARG;
    if (stride == 1) return fft_dev<T>(d_re, d_im, n, batch, dist, direction, pl, s);
    if (!pl || !d_re || !d_im) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (dist != 1 || !is_pow2(stride) || !is_pow2(batch) || batch > stride) return PHAST_ERR_INVALID_ARG;
    const unsigned sb = ilog2(stride), bb = ilog2(batch);
    const unsigned gl = tw_n ? ilog2(tw_n) : 0u;
    if (direction == PHAST_REVERSE) return pl->exec_strided(d_im, d_re, sb, bb, 1.0 / (double)n, s, gl, (unsigned)tw_col0);
    return pl->exec_strided(d_re, d_im, sb, bb, 1.0, s, gl, (unsigned)tw_col0);
}

// average kernel duration of every pass over `reps` forward transforms of the same buffers
template <typename T>
static int time_passes(const Planner<T> *pl, T *d_re, T *d_im, size_t batch, size_t dist, int reps, float *pass_ms,
                       int *n_passes, hipStream_t s) {
    if (!pl || !d_re || !d_im || !pass_ms || !n_passes || reps < 1) return PHAST_ERR_INVALID_ARG;
    int np = 1;  // the slots the launches really used (a batch above the scratch runs in chunks, each with its own plan)
    PHAST_ON_DEVICE(pl->device);
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        PassTimer tm;
        int rc = pl->exec(d_re, d_im, dist, 0, d_re, d_im, dist, 0, batch, 1.0, s, &tm);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < tm.pass_of.size(); ++i) {
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
            acc[tm.pass_of[i] & 3] += ms;
            np = std::max(np, (tm.pass_of[i] & 3) + 1);
        }
    }
    for (int i = 0; i < np; ++i) pass_ms[i] = (float)(acc[i] / reps);
    *n_passes = np;
    return PHAST_OK;
}

// the same for a real transform: slots 0..np-1 = passes of the inner N/2-point transform, slot np = the untangle sweep
// (a transform small enough for the one-pass kernel has the untangle fused: one slot)
template <typename T>
static int time_passes_r2c(const PlannerR2c<T> *pl, const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist,
                           size_t out_dist, int reps, float *pass_ms, int *n_passes, hipStream_t s) {
    if (!pl || !d_in || !d_ore || !d_oim || !pass_ms || !n_passes || reps < 1) return PHAST_ERR_INVALID_ARG;
    int np = 1;  // slots the launches really used: the plan and the fuse decision are exec_in's
    PHAST_ON_DEVICE(pl->dit.device);
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        PassTimer tm;
        int rc = pl->dit.passes.empty() ? pl->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, s)
                                        : pl->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, s, &tm);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < tm.pass_of.size(); ++i) {
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
            acc[tm.pass_of[i] & 3] += ms;
            np = std::max(np, (tm.pass_of[i] & 3) + 1);
        }
    }
    for (int i = 0; i < np; ++i) pass_ms[i] = (float)(acc[i] / reps);
    *n_passes = np;
    return PHAST_OK;
}

// ... and for the inverse real transform: slots 0..np-1 = passes of the inner transform (the first one forms z on load
// when its fused form exists, c2r_fused.hpp), slot np = the preprocess sweep otherwise
template <typename T>
static int time_passes_c2r(const PlannerR2c<T> *pl, const T *d_ire, const T *d_iim, T *d_out, size_t batch, size_t in_dist,
                           size_t out_dist, int reps, float *pass_ms, int *n_passes, hipStream_t s) {
    if (!pl || !d_ire || !d_iim || !d_out || !pass_ms || !n_passes || reps < 1) return PHAST_ERR_INVALID_ARG;
    int np = 1;
    PHAST_ON_DEVICE(pl->dit.device);
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        PassTimer tm;
        int rc = pl->dit.passes.empty() ? pl->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, s)
                                        : pl->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, s, &tm);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < tm.pass_of.size(); ++i) {
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
            acc[tm.pass_of[i] & 3] += ms;
            np = std::max(np, (tm.pass_of[i] & 3) + 1);
        }
    }
    for (int i = 0; i < np; ++i) pass_ms[i] = (float)(acc[i] / reps);
    *n_passes = np;
    return PHAST_OK;
}

// lib.rs:41-140 (feature complex-nums): interleaved Complex<T> signal, in place.  The reference copies into two
// planar Vecs and back; here the (de)interleave is the first pass's load and the last pass's store.
template <typename T>
static int fft_interleaved_dev(T *d_signal, size_t n, size_t batch, size_t dist, int direction, const Planner<T> *pl,
                               hipStream_t s) {
    if (!pl || !d_signal) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (batch > 1 && dist < n) return PHAST_ERR_INVALID_ARG;
    if (direction == PHAST_REVERSE)  // swap trick: read (im, re), transform, store (im, re) scaled by 1/N
        return pl->exec(d_signal, nullptr, dist, 2, d_signal, nullptr, dist, 2, batch, 1.0 / (double)n, s);
    return pl->exec(d_signal, nullptr, dist, 1, d_signal, nullptr, dist, 1, batch, 1.0, s);
}

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

// Host slices <-> the leased workspace's device staging buffer, on the workspace's own stream.  `parts` are (host pointer,
// byte offset in the staging buffer, bytes); small totals travel through the pinned mirror (Planner::pinned_max_bytes).
struct HostPart {
    void *host;
    size_t off, bytes;
};
template <typename T>
static int host_in(const Planner<T> *pl, const typename Planner<T>::Lease &L, void *d_stage, const HostPart *parts, int np,
                   size_t total, bool small) {
    if (small) {
        void *pin = nullptr;
        int rc = pl->pinned(L, total, &pin);
        if (rc) return rc;
        size_t lo = total, hi = 0;
        for (int i = 0; i < np; ++i) {
            std::memcpy((char *)pin + parts[i].off, parts[i].host, parts[i].bytes);
            lo = std::min(lo, parts[i].off);
            hi = std::max(hi, parts[i].off + parts[i].bytes);
        }
        if (hi > lo) PHAST_HIP(hipMemcpyAsync((char *)d_stage + lo, (char *)pin + lo, hi - lo, hipMemcpyHostToDevice, L.stream));
        return PHAST_OK;
    }
    // pageable memory: the runtime stages the copy; issued on the workspace's stream so that nothing waits on, or is
    // waited for by, the NULL stream (other threads' calls on this planner run beside this one)
    for (int i = 0; i < np; ++i)
        PHAST_HIP(hipMemcpyAsync((char *)d_stage + parts[i].off, parts[i].host, parts[i].bytes, hipMemcpyHostToDevice, L.stream));
    return PHAST_OK;
}
template <typename T>
static int host_out(const Planner<T> *pl, const typename Planner<T>::Lease &L, void *d_stage, const HostPart *parts, int np,
                    size_t total, bool small) {
    if (small) {
        void *pin = nullptr;
        int rc = pl->pinned(L, total, &pin);
        if (rc) return rc;
        size_t lo = total, hi = 0;
        for (int i = 0; i < np; ++i) {
            lo = std::min(lo, parts[i].off);
            hi = std::max(hi, parts[i].off + parts[i].bytes);
        }
        if (hi > lo) PHAST_HIP(hipMemcpyAsync((char *)pin + lo, (char *)d_stage + lo, hi - lo, hipMemcpyDeviceToHost, L.stream));
        PHAST_HIP(hipStreamSynchronize(L.stream));
        for (int i = 0; i < np; ++i) std::memcpy(parts[i].host, (char *)pin + parts[i].off, parts[i].bytes);
        return PHAST_OK;
    }
    for (int i = 0; i < np; ++i)
        PHAST_HIP(hipMemcpyAsync(parts[i].host, (char *)d_stage + parts[i].off, parts[i].bytes, hipMemcpyDeviceToHost, L.stream));
    PHAST_HIP(hipStreamSynchronize(L.stream));
    return PHAST_OK;
}

static bool zero_copy_small() {  // PHAST_ZERO_COPY=0: small host-slice calls stage through device memory as the large ones do
    static const bool v = [] {
        const char *e = std::getenv("PHAST_ZERO_COPY");
        return !(e && *e == '0');
    }();
    return v;
}

// lib.rs:143-226 on host slices: validate as the reference asserts, stage through device memory.  The call checks a
// workspace out for its whole (blocking) duration and runs on that workspace's own stream: concurrent host threads on
// one planner overlap their copies and kernels (planner.rs:38-39).
template <typename T>
static int fft_host(T *re, size_t re_len, T *im, size_t im_len, int direction, const Planner<T> *pl) {
    if (!pl || (!re && re_len) || (!im && im_len)) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (re_len != im_len) return PHAST_ERR_LEN_MISMATCH;     // dit.rs:284
    if (!is_pow2(re_len)) return PHAST_ERR_NOT_POW2;          // dit.rs:285
    if (ilog2(re_len) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;  // dit.rs:289
    const size_t n = re_len, bytes = n * sizeof(T), total = 2 * bytes;
    if (pl->twin) pl = pl->twin.get();  // ONE transform of 8192 points: the plan (and the bits) of the _dev call
    PHAST_ON_DEVICE(pl->device);
    typename Planner<T>::Lease L;
    int rc = pl->check_out(L, nullptr, 1);
    if (rc) return rc;
    const double scale = direction == PHAST_REVERSE ? 1.0 / (double)n : 1.0;
    const bool small = total <= Planner<T>::pinned_max_bytes();
    const HostPart parts[2] = {{re, 0, bytes}, {im, bytes, bytes}};
    if (small && zero_copy_small()) {
        // Up to the pinned limit (1 MiB of planes: N <= 2^16 in f64) the kernels read and write the pinned mirror themselves
        // over PCIe (pinned host memory is device-accessible): no DMA copy either way, one wait per call, and no device
        // staging buffer at all.  A multi-pass transform touches the mirror in its first load and its last store only
        // (tools/host_call_cost.py: 2^12 43.7 -> 30.3 us per call, 2^14 67 -> 52, 2^16 143 -> 127).
        void *pin = nullptr;
        rc = pl->pinned(L, total, &pin);
        if (rc) return rc;
        T *p_re = reinterpret_cast<T *>(pin), *p_im = p_re + n;
        std::memcpy(p_re, re, bytes);
        std::memcpy(p_im, im, bytes);
        rc = direction == PHAST_REVERSE ? pl->exec_in(L, p_im, p_re, n, 0, p_im, p_re, n, 0, 1, scale)
                                        : pl->exec_in(L, p_re, p_im, n, 0, p_re, p_im, n, 0, 1, scale);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(L.stream));
        std::memcpy(re, p_re, bytes);
        std::memcpy(im, p_im, bytes);
        return PHAST_OK;
    }
    void *stage = nullptr;
    rc = pl->stage(L, total, &stage);
    if (rc) return rc;
    T *d_re = reinterpret_cast<T *>(stage), *d_im = d_re + n;
    rc = host_in(pl, L, stage, parts, 2, total, small);
    if (!rc)
        rc = direction == PHAST_REVERSE ? pl->exec_in(L, d_im, d_re, n, 0, d_im, d_re, n, 0, 1, scale)
                                        : pl->exec_in(L, d_re, d_im, n, 0, d_re, d_im, n, 0, 1, scale);
    if (!rc) rc = host_out(pl, L, stage, parts, 2, total, small);
    return rc;
}

template <typename T> static int fft_interleaved_host(T *signal, size_t n, int direction, const Planner<T> *pl) {
    if (!pl || (!signal && n)) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (pl->twin) pl = pl->twin.get();
    PHAST_ON_DEVICE(pl->device);
    typename Planner<T>::Lease L;
    int rc = pl->check_out(L, nullptr, 1);
    if (rc) return rc;
    const size_t total = 2 * n * sizeof(T);
    const bool small = total <= Planner<T>::pinned_max_bytes();
    const HostPart parts[1] = {{signal, 0, total}};
    // swap trick for the inverse: read (im, re), transform, store (im, re) scaled by 1/N (fft_interleaved_dev)
    const unsigned mode = direction == PHAST_REVERSE ? 2u : 1u;
    const double scale = direction == PHAST_REVERSE ? 1.0 / (double)n : 1.0;
    if (small && zero_copy_small()) {  // the kernels work on the pinned mirror itself, as in fft_host
        void *pin = nullptr;
        rc = pl->pinned(L, total, &pin);
        if (rc) return rc;
        std::memcpy(pin, signal, total);
        rc = pl->exec_in(L, pin, nullptr, n, mode, pin, nullptr, n, mode, 1, scale);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(L.stream));
        std::memcpy(signal, pin, total);
        return PHAST_OK;
    }
    void *stage = nullptr;
    rc = pl->stage(L, total, &stage);
    if (rc) return rc;
    rc = host_in(pl, L, stage, parts, 1, total, small);
    if (!rc) rc = pl->exec_in(L, stage, nullptr, n, mode, stage, nullptr, n, mode, 1, scale);
    if (!rc) rc = host_out(pl, L, stage, parts, 1, total, small);
    return rc;
}

// The planner-less entry points (lib.rs:121,181,224; r2c.rs:522,599,696) make a planner per call in the reference.  Here a
// planner owns device tables, a staging buffer and a scratch -- 0.2-0.8 ms to make plus the allocations, 1.2 ms of a 1.9 ms
// call at N = 2^20 (tools/planner_cost.py) -- so the few most recently used ones are kept per type, size and device.
// Invisible to the caller: same results, same errors (a failed construction is never cached), and a planner is immutable to
// its users (planner.rs:38-39).  Large planners (planes above 64 MiB: the call is PCIe time, not planner time) are made and
// dropped per call as before; PHAST_PLANNER_CACHE=0 turns the cache off.  The cache itself is never destroyed: at process
// exit the HIP runtime may be gone before static destructors run.
template <typename P> struct PlannerCache {
    struct Entry {
        size_t n;
        int device;
        std::shared_ptr<P> pl;
        unsigned long long stamp;
    };
    static constexpr size_t kMaxEntries = 4;
    std::mutex mu;
    std::vector<Entry> entries;
    unsigned long long clock = 0;

    static bool enabled() {
        static const bool v = [] {
            const char *e = std::getenv("PHAST_PLANNER_CACHE");
            return !(e && *e == '0');
        }();
        return v;
    }
    template <typename Make> int get(size_t n, size_t elem_bytes, Make &&make, std::shared_ptr<P> *out) {
        int dev = -1;
        const bool cacheable = enabled() && n != 0 && n <= ((size_t)64 << 20) / elem_bytes && hipGetDevice(&dev) == hipSuccess;
        if (!cacheable) (void)hipGetLastError();
        if (cacheable) {
            std::lock_guard<std::mutex> lk(mu);
            for (Entry &e : entries)
                if (e.n == n && e.device == dev) {
                    e.stamp = ++clock;
                    *out = e.pl;
                    return PHAST_OK;
                }
        }
        P *raw = nullptr;
        int rc = make(n, &raw);
        if (rc) return rc;
        out->reset(raw);
        if (cacheable) {
            std::shared_ptr<P> evicted;  // released outside the lock (frees device memory)
            {
                std::lock_guard<std::mutex> lk(mu);
                for (Entry &e : entries)  // a concurrent miss of the same size got there first: use its planner, drop ours
                    if (e.n == n && e.device == dev) {
                        e.stamp = ++clock;
                        evicted = std::move(*out);
                        *out = e.pl;
                        return PHAST_OK;
                    }
                if (entries.size() >= kMaxEntries) {
                    size_t lru = 0;
                    for (size_t i = 1; i < entries.size(); ++i)
                        if (entries[i].stamp < entries[lru].stamp) lru = i;
                    evicted = std::move(entries[lru].pl);
                    entries.erase(entries.begin() + (long)lru);
                }
                entries.push_back(Entry{n, dev, *out, ++clock});
            }
        }
        return PHAST_OK;
    }
    static PlannerCache &instance() {
        static PlannerCache *c = new PlannerCache();  // see above: deliberately not destroyed
        return *c;
    }
};

template <typename T> static int fft_host_noplanner(T *re, size_t re_len, T *im, size_t im_len, int direction) {
    // lib.rs:180-183: the planner is built from reals.len() first, so a bad length panics in the planner
    std::shared_ptr<Planner<T>> pl;
    int rc = PlannerCache<Planner<T>>::instance().get(re_len, sizeof(T), [](size_t n, Planner<T> **o) { return planner_new(n, o); }, &pl);
    if (rc) return rc;
    return fft_host<T>(re, re_len, im, im_len, direction, pl.get());
}

template <typename T>
static int r2c_host(const T *in, size_t in_len, T *ore, size_t ore_len, T *oim, size_t oim_len,
                    const PlannerR2c<T> *pl) {
    if (!pl || !in || !ore || !oim) return PHAST_ERR_INVALID_ARG;
    const size_t n = pl->n, half = n / 2;
    if (in_len != n) return PHAST_ERR_R2C_INPUT_LEN;
    if (ore_len != half + 1) return PHAST_ERR_R2C_OUT_RE_LEN;
    if (oim_len != half + 1) return PHAST_ERR_R2C_OUT_IM_LEN;
    if (pl->twin) pl = pl->twin.get();
    PHAST_ON_DEVICE(pl->dit.device);
    typename Planner<T>::Lease L;
    int rc = pl->dit.check_out(L, nullptr, 1);
    if (rc) return rc;
    const size_t ob = (half + 1) * sizeof(T), total = n * sizeof(T) + 2 * ob;
    const bool small = total <= Planner<T>::pinned_max_bytes();
    const HostPart pin[1] = {{const_cast<T *>(in), 0, n * sizeof(T)}};
    const HostPart pout[2] = {{ore, n * sizeof(T), ob}, {oim, n * sizeof(T) + ob, ob}};
    if (small && pl->dit.passes.empty() && zero_copy_small()) {  // one kernel on the pinned mirror itself, as fft_host
        void *pm = nullptr;
        rc = pl->dit.pinned(L, total, &pm);
        if (rc) return rc;
        T *p_in = reinterpret_cast<T *>(pm), *p_ore = p_in + n, *p_oim = p_ore + half + 1;
        std::memcpy(p_in, in, n * sizeof(T));
        rc = pl->r2c_in(L, p_in, p_ore, p_oim, 1, n, half + 1);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(L.stream));
        std::memcpy(ore, p_ore, ob);
        std::memcpy(oim, p_oim, ob);
        return PHAST_OK;
    }
    void *stage = nullptr;
    rc = pl->dit.stage(L, total, &stage);
    if (rc) return rc;
    T *d_in = reinterpret_cast<T *>(stage), *d_ore = d_in + n, *d_oim = d_ore + half + 1;
    rc = host_in(&pl->dit, L, stage, pin, 1, total, small);
    if (!rc) rc = pl->r2c_in(L, d_in, d_ore, d_oim, 1, n, half + 1);
    if (!rc) rc = host_out(&pl->dit, L, stage, pout, 2, total, small);
    return rc;
}

template <typename T>
static int c2r_host(const T *ire, size_t ire_len, const T *iim, size_t iim_len, T *out, size_t out_len,
                    const PlannerR2c<T> *pl, bool check_scratch, size_t sre_len, size_t sim_len) {
    if (!pl || !ire || !iim || !out) return PHAST_ERR_INVALID_ARG;
    const size_t n = pl->n, half = n / 2;
    if (out_len != n) return PHAST_ERR_C2R_OUTPUT_LEN;
    if (ire_len != half + 1) return PHAST_ERR_C2R_IN_RE_LEN;
    if (iim_len != half + 1) return PHAST_ERR_C2R_IN_IM_LEN;
    if (check_scratch && sre_len != half) return PHAST_ERR_C2R_SCRATCH_RE;
    if (check_scratch && sim_len != half) return PHAST_ERR_C2R_SCRATCH_IM;
    if (pl->twin) pl = pl->twin.get();
    PHAST_ON_DEVICE(pl->dit.device);
    typename Planner<T>::Lease L;
    int rc = pl->dit.check_out(L, nullptr, 1);
    if (rc) return rc;
    const size_t ib = (half + 1) * sizeof(T), total = n * sizeof(T) + 2 * ib;
    const bool small = total <= Planner<T>::pinned_max_bytes();
    const HostPart pin[2] = {{const_cast<T *>(ire), n * sizeof(T), ib}, {const_cast<T *>(iim), n * sizeof(T) + ib, ib}};
    const HostPart pout[1] = {{out, 0, n * sizeof(T)}};
    if (small && pl->dit.passes.empty() && zero_copy_small()) {  // one kernel on the pinned mirror itself, as fft_host
        void *pm = nullptr;
        rc = pl->dit.pinned(L, total, &pm);
        if (rc) return rc;
        T *p_out = reinterpret_cast<T *>(pm), *p_ire = p_out + n, *p_iim = p_ire + half + 1;
        std::memcpy(p_ire, ire, ib);
        std::memcpy(p_iim, iim, ib);
        rc = pl->c2r_in(L, p_ire, p_iim, p_out, 1, half + 1, n);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(L.stream));
        std::memcpy(out, p_out, n * sizeof(T));
        return PHAST_OK;
    }
    void *stage = nullptr;
    rc = pl->dit.stage(L, total, &stage);
    if (rc) return rc;
    T *d_out = reinterpret_cast<T *>(stage), *d_ire = d_out + n, *d_iim = d_ire + half + 1;
    rc = host_in(&pl->dit, L, stage, pin, 2, total, small);
    if (!rc) rc = pl->c2r_in(L, d_ire, d_iim, d_out, 1, half + 1, n);
    if (!rc) rc = host_out(&pl->dit, L, stage, pout, 1, total, small);
    return rc;
}

template <typename T> static int bitrev_host(T *data, size_t len, unsigned log_n) {
    if (!data && len) return PHAST_ERR_INVALID_ARG;
    if (log_n > 40 || len != ((size_t)1 << log_n)) return PHAST_ERR_INVALID_ARG;  // bravo.rs:228
    int rc = ensure_device();
    if (rc) return rc;
    if (log_n > 31) return PHAST_ERR_INVALID_ARG;
    DevBuf buf;
    rc = buf.alloc(len * sizeof(T));
    if (rc) return rc;
    PHAST_HIP(hipMemcpy(buf.p, data, len * sizeof(T), hipMemcpyHostToDevice));
    PHAST_HIP(launch_bitrev<T>(reinterpret_cast<T *>(buf.p), log_n, 1, len, nullptr));
    PHAST_HIP(hipStreamSynchronize(nullptr));
    PHAST_HIP(hipMemcpy(data, buf.p, len * sizeof(T), hipMemcpyDeviceToHost));
    return PHAST_OK;
}

template <typename T> static int describe_to(const Planner<T> *p, char *buf, size_t len) {
    if (!p || !buf || !len) return PHAST_ERR_INVALID_ARG;
    std::string s = p->describe();
    std::snprintf(buf, len, "%s", s.c_str());
    return PHAST_OK;
}

template <typename T>
static int set_plan_c(Planner<T> *p, const unsigned *log_rows, const unsigned *tile_logs, size_t n_passes,
                      unsigned points_log) {
    if (!p) return PHAST_ERR_INVALID_ARG;
    if (p->passes.empty()) {  // a one-pass size: nothing to plan -- except in the multi-pass twin of 8192 points (tools)
        if (p->twin) return set_plan_c<T>(p->twin.get(), log_rows, tile_logs, n_passes, points_log);
        return n_passes == 0 ? PHAST_OK : PHAST_ERR_INVALID_ARG;
    }
    std::vector<unsigned> lrs, tls;
    if (n_passes == 0) {
        return p->default_plans();
    } else {
        if (!log_rows || !tile_logs) return PHAST_ERR_INVALID_ARG;
        lrs.assign(log_rows, log_rows + n_passes);
        tls.assign(tile_logs, tile_logs + n_passes);
    }
    if ((points_log & 0xfu) < 3 || (points_log & 0xfu) > 5 || (points_log & ~0x1fu)) return PHAST_ERR_INVALID_ARG;
    return p->set_plan(lrs, tls, 0, points_log);
}

}  // namespace phast

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

extern "C" {

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

int phast_hip_graph_upload(void *graph_exec, void *stream) {
    if (!graph_exec) return PHAST_ERR_INVALID_ARG;
    PHAST_HIP(hipGraphUpload(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
    return PHAST_OK;
}

int phast_stream_probe_dev(const void *d_a, void *d_b, size_t bytes, int reps, double *out_gbps, void *stream) {
    if (!d_a || !d_b || !out_gbps || reps < 1 || bytes < ((size_t)1 << 20) || (bytes & 15)) return PHAST_ERR_INVALID_ARG;
    int dev = 0;
    int rc = ensure_device(&dev);
    if (rc) return rc;
    PHAST_HIP(stream_probe(d_a, d_b, bytes, reps, cus_of(dev), out_gbps, static_cast<hipStream_t>(stream)));
    return PHAST_OK;
}

void phast_debug_set_guard_bytes(size_t bytes) { g_guard_bytes = (bytes + 255) & ~(size_t)255; }

void phast_debug_set_wg_per_cu(int wg_per_cu) { g_wg_per_cu_override = wg_per_cu; }
void phast_debug_set_trace(unsigned long long *d_trace) { g_trace = d_trace; }

int phast_device_info(char *name, size_t name_len, int *compute_units, size_t *lds_per_block,
                      size_t *global_mem_bytes) {
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
}

void phast_options_default(phast_options *out) {
    if (!out) return;
    out->multithreaded_bit_reversal = 0;
    out->smallest_parallel_chunk_size = 16384;
}

int phast_options_guess(size_t input_size, phast_options *out) {
    if (!out) return PHAST_ERR_INVALID_ARG;
    if (input_size == 0) return PHAST_ERR_NOT_POW2;  // usize::ilog2(0) panics (options.rs:40)
    phast_options_default(out);
    out->multithreaded_bit_reversal = ilog2(input_size) >= 16;
    return PHAST_OK;
}

#define PHAST_PLANNER_API(SFX, T)                                                                                  \
    int phast_planner_dit##SFX##_new(size_t n, phast_planner_dit##SFX **out) {                                     \
        return planner_new(n, out);                                                                                \
    }                                                                                                              \
    int phast_planner_dit##SFX##_with_mode(size_t n, int mode, phast_planner_dit##SFX **out) {                     \
        if (mode != PHAST_MODE_HEURISTIC && mode != PHAST_MODE_TUNE) return PHAST_ERR_INVALID_ARG;                 \
        return planner_new(n, out);                                                                                \
    }                                                                                                              \
    void phast_planner_dit##SFX##_free(phast_planner_dit##SFX *p) { delete p; }                                    \
    size_t phast_planner_dit##SFX##_device_bytes(const phast_planner_dit##SFX *p) {                                \
        return p ? p->device_bytes() : 0;                                                                          \
    }                                                                                                              \
    int phast_planner_dit##SFX##_debug_check_guards(const phast_planner_dit##SFX *p, size_t *bad_bytes) {          \
        if (!p || !bad_bytes) return PHAST_ERR_INVALID_ARG;                                                        \
        return p->check_guards(bad_bytes);                                                                         \
    }                                                                                                              \
    int phast_planner_dit##SFX##_describe(const phast_planner_dit##SFX *p, char *buf, size_t len) {                \
        return describe_to<T>(p, buf, len);                                                                        \
    }                                                                                                              \
    int phast_planner_dit##SFX##_reserve_batch(phast_planner_dit##SFX *p, size_t max_batch) {                      \
        if (!p || max_batch == 0) return PHAST_ERR_INVALID_ARG;                                                    \
        PHAST_ON_DEVICE(p->device);                                                                                \
        p->reserve = max_batch;                                                                                    \
        Planner<T>::Lease L;                                                                                       \
        int rc = p->check_out(L, nullptr, 2);                                                                      \
        if (rc) return rc;                                                                                         \
        L.stream = nullptr; /* check_out waited for the workspace: whatever is retired below is idle */          \
        size_t cap;                                                                                                \
        return p->ensure_scratch(L, max_batch, &cap);                                                              \
    }                                                                                                              \
    int phast_planner_dit##SFX##_set_plan(phast_planner_dit##SFX *p, const unsigned *lr, const unsigned *tl,       \
                                          size_t np, unsigned points_log) {                                        \
        return set_plan_c<T>(p, lr, tl, np, points_log);                                                           \
    }                                                                                                              \
    int phast_planner_dit##SFX##_time_passes(const phast_planner_dit##SFX *p, T *d_re, T *d_im, size_t batch,       \
                                             size_t dist, int reps, float *pass_ms, int *n_passes, void *stream) { \
        return time_passes<T>(p, d_re, d_im, batch, dist, reps, pass_ms, n_passes,                                 \
                              static_cast<hipStream_t>(stream));                                                   \
    }                                                                                                              \
    int phast_planner_r2c##SFX##_new(size_t n, phast_planner_r2c##SFX **out) {                                     \
        return r2c_planner_new(n, out);                                                                            \
    }                                                                                                              \
    void phast_planner_r2c##SFX##_free(phast_planner_r2c##SFX *p) { delete p; }                                    \
    int phast_planner_r2c##SFX##_time_passes(const phast_planner_r2c##SFX *p, const T *d_in, T *d_ore, T *d_oim,    \
                                             size_t batch, size_t in_dist, size_t out_dist, int reps,              \
                                             float *pass_ms, int *n_passes, void *stream) {                        \
        return time_passes_r2c<T>(p, d_in, d_ore, d_oim, batch, in_dist, out_dist, reps, pass_ms, n_passes,        \
                                  static_cast<hipStream_t>(stream));                                               \
    }                                                                                                              \
    int phast_planner_r2c##SFX##_time_c2r_passes(const phast_planner_r2c##SFX *p, const T *d_ire, const T *d_iim,   \
                                                 T *d_out, size_t batch, size_t in_dist, size_t out_dist, int reps, \
                                                 float *pass_ms, int *n_passes, void *stream) {                     \
        return time_passes_c2r<T>(p, d_ire, d_iim, d_out, batch, in_dist, out_dist, reps, pass_ms, n_passes,        \
                                  static_cast<hipStream_t>(stream));                                               \
    }                                                                                                              \
    int phast_planner_r2c##SFX##_set_inner_plan(phast_planner_r2c##SFX *p, const unsigned *lr, const unsigned *tl,  \
                                                size_t np, unsigned points_log) {                                  \
        if (!p) return PHAST_ERR_INVALID_ARG;                                                                      \
        PlannerR2c<T> *q = (p->dit.passes.empty() && p->twin) ? p->twin.get() : p;                                 \
        int rc = set_plan_c<T>(&q->dit, lr, tl, np, points_log);                                                   \
        if (rc == PHAST_OK && np == 0 && !q->dit.passes.empty()) rc = q->dit.make_c2r_plans();                     \
        return rc;                                                                                                 \
    }                                                                                                              \
    int phast_planner_r2c##SFX##_describe(const phast_planner_r2c##SFX *p, char *buf, size_t len) {                \
        if (!p || !buf || !len) return PHAST_ERR_INVALID_ARG;                                                      \
        std::string s = p->dit.describe();                                                                         \
        if (p->twin) s += " | one transform: " + p->twin->dit.describe();                                          \
        std::snprintf(buf, len, "%s", s.c_str());                                                                  \
        return PHAST_OK;                                                                                           \
    }

PHAST_PLANNER_API(64, double)
PHAST_PLANNER_API(32, float)

#define PHAST_FFT_API(SFX, FS, T)                                                                                   \
    int phast_fft_##SFX##_dit(T *re, size_t re_len, T *im, size_t im_len, int direction) {                          \
        return fft_host_noplanner<T>(re, re_len, im, im_len, direction);                                            \
    }                                                                                                               \
    int phast_fft_##SFX##_dit_with_planner(T *re, size_t re_len, T *im, size_t im_len, int direction,               \
                                           const phast_planner_dit##SFX *pl) {                                      \
        return fft_host<T>(re, re_len, im, im_len, direction, pl);                                                  \
    }                                                                                                               \
    int phast_fft_##SFX##_dit_with_planner_and_opts(T *re, size_t re_len, T *im, size_t im_len, int direction,      \
                                                    const phast_planner_dit##SFX *pl, const phast_options *opts) {  \
        if (!opts) return PHAST_ERR_INVALID_ARG;                                                                    \
        return fft_host<T>(re, re_len, im, im_len, direction, pl);                                                  \
    }                                                                                                               \
    int phast_fft_##SFX##_dit_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, int direction,             \
                                  const phast_planner_dit##SFX *pl, void *stream) {                                 \
        return fft_dev<T>(d_re, d_im, n, batch, dist, direction, pl, static_cast<hipStream_t>(stream));             \
    }                                                                                                               \
    int phast_fft_##SFX##_dit_many_dev(T *const *d_re, T *const *d_im, size_t count, size_t n, int direction,       \
                                       const phast_planner_dit##SFX *pl, void *stream) {                            \
        return fft_dev_many<T>(d_re, d_im, count, n, direction, pl, static_cast<hipStream_t>(stream));              \
    }                                                                                                               \
    int phast_fft_##SFX##_dit_strided_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, size_t stride,     \
                                          int direction, const phast_planner_dit##SFX *pl, void *stream) {          \
        return fft_strided_dev<T>(d_re, d_im, n, batch, dist, stride, direction, pl,                                \
                                  static_cast<hipStream_t>(stream));                                                \
    }                                                                                                               \
    int phast_fft_##SFX##_dit_strided_tw_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, size_t stride,  \
                                             int direction, const phast_planner_dit##SFX *pl, size_t tw_n,          \
                                             size_t tw_col0, void *stream) {                                        \
        if (tw_n == 0) return PHAST_ERR_INVALID_ARG;                                                                \
        return fft_strided_dev<T>(d_re, d_im, n, batch, dist, stride, direction, pl,                                \
                                  static_cast<hipStream_t>(stream), tw_n, tw_col0);                                 \
    }                                                                                                               \
    int phast_fft_##SFX##_interleaved(T *signal, size_t n, int direction) {                                         \
        std::shared_ptr<Planner<T>> pl; /* lib.rs:121: a planner per call -- kept, see PlannerCache */             \
        int rc = PlannerCache<Planner<T>>::instance().get(                                                          \
            n, sizeof(T), [](size_t m, Planner<T> **o) { return planner_new(m, o); }, &pl);                         \
        if (rc) return rc;                                                                                          \
        return fft_interleaved_host<T>(signal, n, direction, pl.get());                                             \
    }                                                                                                               \
    int phast_fft_##SFX##_interleaved_with_planner(T *signal, size_t n, int direction,                              \
                                                   const phast_planner_dit##SFX *pl) {                              \
        return fft_interleaved_host<T>(signal, n, direction, pl);                                                   \
    }                                                                                                               \
    int phast_fft_##SFX##_interleaved_with_planner_and_opts(T *signal, size_t n, int direction,                     \
                                                            const phast_planner_dit##SFX *pl,                       \
                                                            const phast_options *opts) {                            \
        if (!opts) return PHAST_ERR_INVALID_ARG;                                                                    \
        return fft_interleaved_host<T>(signal, n, direction, pl);                                                   \
    }                                                                                                               \
    int phast_fft_##SFX##_interleaved_dev(T *d_signal, size_t n, size_t batch, size_t dist, int direction,          \
                                          const phast_planner_dit##SFX *pl, void *stream) {                         \
        return fft_interleaved_dev<T>(d_signal, n, batch, dist, direction, pl, static_cast<hipStream_t>(stream));   \
    }                                                                                                               \
    int phast_bit_rev_##FS(T *data, size_t len, unsigned log_n) { return bitrev_host<T>(data, len, log_n); }        \
    int phast_bit_rev_##FS##_dev(T *d, unsigned log_n, size_t batch, size_t dist, void *stream) {                   \
        if (!d || log_n > 31 || (batch > 1 && dist < ((size_t)1 << log_n))) return PHAST_ERR_INVALID_ARG;           \
        int rc = ensure_device();                                                                                   \
        if (rc) return rc;                                                                                          \
        PHAST_HIP(launch_bitrev<T>(d, log_n, batch, dist, static_cast<hipStream_t>(stream)));                       \
        return PHAST_OK;                                                                                            \
    }                                                                                                               \
    int phast_r2c_fft_##FS(const T *in, size_t in_len, T *ore, size_t ore_len, T *oim, size_t oim_len) {            \
        std::shared_ptr<PlannerR2c<T>> pl; /* r2c.rs:522: planner from input_re.len() */                           \
        int rc = PlannerCache<PlannerR2c<T>>::instance().get(                                                       \
            in_len, sizeof(T), [](size_t m, PlannerR2c<T> **o) { return r2c_planner_new(m, o); }, &pl);             \
        if (rc) return rc;                                                                                          \
        return r2c_host<T>(in, in_len, ore, ore_len, oim, oim_len, pl.get());                                       \
    }                                                                                                               \
    int phast_r2c_fft_##FS##_with_planner(const T *in, size_t in_len, T *ore, size_t ore_len, T *oim,               \
                                          size_t oim_len, const phast_planner_r2c##SFX *pl) {                       \
        return r2c_host<T>(in, in_len, ore, ore_len, oim, oim_len, pl);                                             \
    }                                                                                                               \
    int phast_r2c_fft_##FS##_dev(const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist, size_t out_dist,  \
                                 const phast_planner_r2c##SFX *pl, void *stream) {                                  \
        if (!pl || !d_in || !d_ore || !d_oim) return PHAST_ERR_INVALID_ARG;                                         \
        if (batch > 1 && (in_dist < pl->n || out_dist < pl->n / 2 + 1)) return PHAST_ERR_INVALID_ARG;               \
        return pl->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, static_cast<hipStream_t>(stream));             \
    }                                                                                                               \
    int phast_c2r_fft_##FS(const T *ire, size_t ire_len, const T *iim, size_t iim_len, T *out, size_t out_len) {    \
        std::shared_ptr<PlannerR2c<T>> pl; /* r2c.rs:696: planner from output.len() */                             \
        int rc = PlannerCache<PlannerR2c<T>>::instance().get(                                                       \
            out_len, sizeof(T), [](size_t m, PlannerR2c<T> **o) { return r2c_planner_new(m, o); }, &pl);            \
        if (rc) return rc;                                                                                          \
        return c2r_host<T>(ire, ire_len, iim, iim_len, out, out_len, pl.get(), false, 0, 0);                        \
    }                                                                                                               \
    int phast_c2r_fft_##FS##_with_planner(const T *ire, size_t ire_len, const T *iim, size_t iim_len, T *out,       \
                                          size_t out_len, const phast_planner_r2c##SFX *pl) {                       \
        return c2r_host<T>(ire, ire_len, iim, iim_len, out, out_len, pl, false, 0, 0);                              \
    }                                                                                                               \
    int phast_c2r_fft_##FS##_with_planner_and_scratch(const T *ire, size_t ire_len, const T *iim, size_t iim_len,   \
                                                      T *out, size_t out_len, const phast_planner_r2c##SFX *pl,     \
                                                      T *sre, size_t sre_len, T *sim, size_t sim_len) {             \
        (void)sre;                                                                                                  \
        (void)sim;                                                                                                  \
        return c2r_host<T>(ire, ire_len, iim, iim_len, out, out_len, pl, true, sre_len, sim_len);                   \
    }                                                                                                               \
    int phast_c2r_fft_##FS##_dev(const T *d_ire, const T *d_iim, T *d_out, size_t batch, size_t in_dist,            \
                                 size_t out_dist, const phast_planner_r2c##SFX *pl, void *stream) {                 \
        if (!pl || !d_ire || !d_iim || !d_out) return PHAST_ERR_INVALID_ARG;                                        \
        if (batch > 1 && (in_dist < pl->n / 2 + 1 || out_dist < pl->n)) return PHAST_ERR_INVALID_ARG;               \
        return pl->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, static_cast<hipStream_t>(stream));            \
    }                                                                                                               \
    int phast_fill_##FS##_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, unsigned long long seed,       \
                              unsigned long long first_id, void *stream) {                                          \
        if (!d_re) return PHAST_ERR_INVALID_ARG;                                                                    \
        int rc = ensure_device();                                                                                   \
        if (rc) return rc;                                                                                          \
        PHAST_HIP(launch_fill<T>(d_re, d_im, n, batch, dist, seed, first_id, static_cast<hipStream_t>(stream)));    \
        return PHAST_OK;                                                                                            \
    }                                                                                                               \
    int phast_digest_##FS##_dev(const T *d_re, const T *d_im, size_t n, size_t batch, size_t dist, size_t probe,    \
                                double *d_digest, void *stream) {                                                   \
        if (!d_re || !d_im || !d_digest) return PHAST_ERR_INVALID_ARG;                                              \
        int rc = ensure_device();                                                                                   \
        if (rc) return rc;                                                                                          \
        PHAST_HIP(launch_digest<T>(d_re, d_im, n, batch, dist, probe, d_digest, static_cast<hipStream_t>(stream))); \
        return PHAST_OK;                                                                                            \
    }

PHAST_FFT_API(64, f64, double)
PHAST_FFT_API(32, f32, float)

#define PHAST_TWIDDLE_API(SFX, T)                                                                                   \
    int phast_twiddle_grid##SFX##_new(size_t n, phast_twiddle_grid##SFX **out) {                                    \
        return planner_new(n, out);                                                                                 \
    }                                                                                                               \
    void phast_twiddle_grid##SFX##_free(phast_twiddle_grid##SFX *g) { delete g; }                                   \
    int phast_twiddle_grid##SFX##_apply_dev(const phast_twiddle_grid##SFX *g, T *d_re, T *d_im, size_t rows,        \
                                            size_t cols, size_t row_pitch, size_t row0, size_t col0, void *stream) { \
        if (!g) return PHAST_ERR_INVALID_ARG;                                                                       \
        return g->apply(d_re, d_im, rows, cols, row_pitch, row0, col0, static_cast<hipStream_t>(stream));           \
    }
PHAST_TWIDDLE_API(64, double)
PHAST_TWIDDLE_API(32, float)

}  // extern "C"
