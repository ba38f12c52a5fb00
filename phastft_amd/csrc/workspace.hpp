// workspace.hpp -- what ONE call sequence on a planner works in: inter-pass scratch, staging buffer, pinned mirror, the
// unfused C2R workspace; the pool of them lives in Planner (planner.hpp, planner_pool.hpp).
#pragma once

#include <thread>

#include "host_util.hpp"

namespace phast {

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
    // Who touches what: the thread that has the workspace checked out (`busy`, set and cleared under the planner's `mu`) owns
    // every field; other threads look at a workspace only under `mu` and only after seeing `busy == false`.  The exceptions
    // are the byte counts and `captured`, which device_bytes() and the pool's head count read at any time: atomics (ADVICE
    // r04: they were plain fields read under `mu` while the holder wrote them outside it).
    void *d_scratch = nullptr;  // [cap][2][stride]: re plane then im plane per transform (typed by the planner)
    std::atomic<size_t> cap{0};
    size_t guard = 0;           // bytes of guard band before and after the scratch (debug hook, normally 0)
    std::atomic<size_t> per{0}; // bytes per transform the scratch was cut for (2 * stride * sizeof(T))
    void *d_stage = nullptr;    // device staging of the host-slice entry points (grow-only)
    std::atomic<size_t> stage_bytes{0};
    void *h_pin = nullptr;      // pinned host mirror of the staging buffer for SMALL host-slice calls
    size_t pin_bytes = 0;
    void *d_z = nullptr;        // unfused C2R: the preprocess workspace [z_cap][2][n/2] (PlannerR2c)
    size_t z_cap = 0;
    std::atomic<size_t> z_bytes{0};
    hipStream_t stream = nullptr;  // the stream the last work of this workspace went to (valid while `pending`)
    bool pending = false;          // work may still be running on `stream`
    hipStream_t own = nullptr;     // the non-blocking stream of host-slice calls (created on first use)
    hipEvent_t idle = nullptr;     // recorded behind the last _dev call's work (Planner::check_in); owned by the workspace
    bool busy = false;             // checked out by a host thread
    std::atomic<bool> captured{false};         // used under stream capture: pinned to the captured graphs (see above)
    std::thread::id last_thread;   // the host thread whose call used it last (who may capture over its own warm-up, check_out)
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
    std::atomic<size_t> retired_dev_bytes{0};

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
        cap = 0;
        per = 0;
        stage_bytes = 0;
        z_bytes = 0;
        retired_dev_bytes = 0;
        z_cap = pin_bytes = 0;
    }
};

}  // namespace phast
