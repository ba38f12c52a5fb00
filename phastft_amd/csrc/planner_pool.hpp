// planner_pool.hpp -- Planner<T>: checking workspaces out and in, and the scratch inside them (see workspace.hpp).
#pragma once

#include "planner.hpp"

namespace phast {

// which = 0: a _dev call on `stream`; 1: a host-slice call (runs on the workspace's own stream); 2: bookkeeping only
// (reserve_batch: any free eager workspace, nothing is enqueued).
// The library never touches a caller's stream handle after the call that was given it has returned -- the caller may
// destroy the stream the moment its work is done (round 4's first version asked hipStreamQuery about the stream a
// workspace had last served: a use-after-free inside the HIP runtime once that stream was gone, found by the ASan pass of
// tests/cpp/concurrent_planner_test.cpp).  What outlives a call is the workspace's OWN event, recorded behind the call's
// work while the stream is certainly alive (check_in): "has this workspace drained?" is hipEventQuery(idle), "order that
// stream behind it" is hipStreamWaitEvent(stream, idle).
template <typename T> int Planner<T>::check_out(Lease &L, hipStream_t stream, int which) const {
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
    const std::thread::id me = std::this_thread::get_id();
    // Under capture nothing executes now and nothing may be allocated or queried.  What the graph's replays will work in must
    // not be in use by anybody else when they run: a workspace with nothing in flight, or whose work in flight is on this very
    // stream, or is THIS thread's own (its warm-up call, usually on another stream than the capture's -- the caller orders that
    // before the first replay, as every capture of already-running work requires).  Never one that another thread's stream is
    // still working in: nothing would order that work before the replays.
    auto capture_can_take = [&](const Workspace &w) {
        const bool own_ws = w.pending && w.stream == stream;
        return (!w.pending || own_ws || w.last_thread == me) && (!w.captured || own_ws);
    };
    for (;;) {
        if (which == 0 && !cap)  // 1. the workspace this stream used last: stream order protects its buffers
            for (auto &w : pool)
                if (!w->busy && w->pending && w->stream == stream && !w->captured) {
                    pick = w.get();
                    break;
                }
        if (cap) {
            // 1b. the largest that FITS (a larger batch runs in fewer chunks; the stream's own on a tie) -- it belongs to the
            // graph from here on, so no eager call can meet a replay in it
            for (auto &w : pool) {
                if (w->busy || !fits(*w) || !capture_can_take(*w)) continue;
                const bool own_ws = w->pending && w->stream == stream;
                if (!pick || w->cap > pick->cap || (w->cap == pick->cap && own_ws)) pick = w.get();
            }
            // 1c. none fits (no eager call since the plan changed, or none at all): one this call may take anyway --
            // ensure_scratch will have to allocate and the capture fails with the runtime's message, as documented
            // ("HIP graphs": run the call once eagerly before capturing it)
            if (!pick)
                for (auto &w : pool)
                    if (!w->busy && !w->captured && capture_can_take(*w) && (!pick || w->cap > pick->cap)) pick = w.get();
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
        pick->last_thread = me;
        if (!cap) pick->reap(work);
    }
    return PHAST_OK;
}

// the call has enqueued everything (a host-slice call: and waited for it)
template <typename T> void Planner<T>::check_in(Workspace *ws, hipStream_t stream, bool host_synchronised) const {
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

// scratch for `want` transforms in flight (capped by the target footprint, at least 1) in the leased workspace.  Grows
// geometrically: a sequence of slowly growing batches retires O(log) buffers whose sizes sum to less than the live one
// (and reap() frees them as soon as the work that used them is done).  Out of device memory: idle retired buffers are
// released, then the request is halved (exec() loops over chunks) down to the reserved batch.
template <typename T> int Planner<T>::ensure_scratch(const Lease &L, size_t batch, size_t *cap_out, bool exact) const {
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
    if (w.cap < want && !exact && w.cap >= std::max<size_t>(reserve.load(), 1) && capturing(stream)) {
        *cap_out = w.cap;  // under capture nothing may be allocated: the batch runs in the chunks this scratch allows
        return PHAST_OK;
    }
    if (w.cap < want) {
        if (!exact && w.cap && want < 2 * w.cap) want = std::min<size_t>(2 * w.cap, std::max(target, want));
        const size_t floor_cap = exact ? want : std::max<size_t>(std::max<size_t>(reserve.load(), 1), w.cap + 1);
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
            if (e == hipErrorOutOfMemory && !exact && w.cap >= std::max<size_t>(reserve.load(), 1)) {
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
template <typename T> int Planner<T>::check_guards(size_t *bad_out) const {
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

// Can a call on `stream` be CAPTURED on this planner right now without allocating -- is there a scratch, cut for the present
// plans, that check_out would hand a capture on that stream (the same rule: nothing in flight, or this stream's own -- incl. the
// one an earlier call of the same capture already took --, or this thread's warm-up)?  (Planner::exec: the twin under capture.)
template <typename T> bool Planner<T>::capture_ready(hipStream_t stream) const {
    std::shared_lock<std::shared_mutex> plans(plan_mu);
    const size_t per_now = 2 * sstride() * sizeof(T);
    const std::thread::id me = std::this_thread::get_id();
    std::lock_guard<std::mutex> lk(mu);
    for (auto &w : pool) {
        if (w->busy || w->cap == 0 || w->per != per_now) continue;
        const bool own_ws = w->pending && w->stream == stream;
        if ((!w->pending || own_ws || w->last_thread == me) && (!w->captured || own_ws)) return true;
    }
    return false;
}

// scratch for up to `max_batch` transforms per launch, allocated now (a capture allocates nothing); a one-pass size forwards to
// its multi-pass twin, which is what serves its small batches
template <typename T> int Planner<T>::reserve_batch(size_t max_batch) {
    if (max_batch == 0) return PHAST_ERR_INVALID_ARG;
    if (passes.empty()) return twin ? twin->reserve_batch(std::min(max_batch, twin_max_batch())) : PHAST_OK;
    PHAST_ON_DEVICE(device);
    reserve = max_batch;
    Lease L;
    int rc = check_out(L, nullptr, 2);
    if (rc) return rc;
    L.stream = nullptr;  // check_out waited for the workspace: whatever is retired below is idle
    size_t cap;
    return ensure_scratch(L, max_batch, &cap);
}

// The workspaces captured graphs work in are kept until the planner goes -- the library cannot know when a graph dies.  A
// long-lived planner that is captured again and again may hand them back: the caller promises that every graph captured on
// this planner so far has been destroyed (or will never be launched again).  Returns the bytes released.
template <typename T> size_t Planner<T>::release_graph_workspaces() {
    DeviceGuard on(device);
    std::vector<std::unique_ptr<Workspace>> gone;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < pool.size();)
            if (!pool[i]->busy && pool[i]->captured) {
                gone.push_back(std::move(pool[i]));
                pool.erase(pool.begin() + (long)i);
            } else {
                ++i;
            }
    }
    size_t bytes = 0;
    for (auto &w : gone) {
        bytes += w->device_bytes();
        w->release();  // hipFree waits for the device
    }
    if (twin) bytes += twin->release_graph_workspaces();
    return bytes;
}

}  // namespace phast
