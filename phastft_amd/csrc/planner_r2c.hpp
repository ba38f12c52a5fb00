// planner_r2c.hpp -- PlannerR2c<T>: the real transforms around the inner N/2-point planner.
#pragma once

#include "planner.hpp"

namespace phast {

// ------------------------------------------------------------------------------------------------
// R2C planner (planner.rs:164-212)
// ------------------------------------------------------------------------------------------------
template <typename T> struct PlannerR2c {
    using Lease = typename Planner<T>::Lease;
    using value_type = T;
    unsigned wisdom_log_n() const { return ilog2(n); }
    int device_of() const { return dit.device; }
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
        dit.is_inner_of_real = true;
        int rc = dit.init(n / 2, force_multi, false);
        if (rc) return rc;
        if (!dit.passes.empty()) {
            rc = dit.make_c2r_plans();
            if (rc == PHAST_OK) rc = dit.apply_wisdom(ilog2(n));  // r2c / c2r wisdom is keyed by the real length
            if (rc) return rc;
        }
        tw_bits = tw3_bits_for(ilog2(n));
        rc = upload<T>(host_tw3<T>(ilog2(n), tw_bits), &d_tw3);
        if (rc == PHAST_OK && !force_multi && dit.passes.empty() && dit.log_n >= kTwinMinLog && Planner<T>::twin_enabled()) {
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

    // the planner a call of ONE (or a few) real transforms runs on (see Planner::kTwinMinLogR2c / kTwinMinLogC2r)
    const PlannerR2c<T> *route_small(bool c2r, size_t batch = 1) const {
        const unsigned min_log = c2r ? Planner<T>::kTwinMinLogC2r : Planner<T>::kTwinMinLogR2c;
        return (twin && dit.log_n >= min_log && batch <= Planner<T>::twin_max_batch()) ? twin.get() : this;
    }
    // PlannerMode::Tune for the real transforms (tune.hpp)
    int tune(int kind, size_t batch, typename Planner<T>::TuneReport *rep);
    // C2R: does the first pass of `ch` form z on load (c2r_fused.hpp), or does the preprocess run as a sweep of its own?
    static bool c2r_fuses(const typename Planner<T>::Choice &ch) {
        return c2r_fuse_enabled() && ch.passes && !ch.passes->empty() && ch.passes->front().c2r_blocks > 0;
    }
    // r2c.rs:535-593 / 607-662 on device pointers (a _dev call: checks a workspace out for the enqueue)
    int r2c(const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist, size_t out_dist, hipStream_t s,
            PassTimer *timer = nullptr) const {
        if (in_dist & 1) return PHAST_ERR_INVALID_ARG;  // the input is read as (even, odd) pairs
        // (... and large batches too where the twin has a plan ranked for them: f32, plan.hpp: real_batch_plan)
        if (route_small(false, batch) != this || (twin && batch * (n / 2) >= ((size_t)1 << 24) && !twin->dit.passes_r2c_tp.empty()))
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
               PassTimer *timer = nullptr, const typename Planner<T>::Choice *forced = nullptr) const {
        const size_t half = n / 2;
        hipStream_t s = L.stream;
        if (dit.passes.empty())  // N/2 <= 8192: one kernel, the untangle is its epilogue
            return dit.exec_small_real(1, d_in, nullptr, in_dist / 2, d_ore, d_oim, out_dist, batch, 1.0, d_tw3, tw_bits, s);
        const R2cFuse fuse{d_tw3, tw_bits};
        bool fused = false;
        size_t np = 0;
        int rc = dit.exec_in(L, d_in, nullptr, in_dist / 2, 1, d_ore, d_oim, out_dist, 0, batch, 1.0, timer, &fuse, &fused, &np, forced);
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
        if (route_small(true, batch) != this || (twin && batch * (n / 2) >= ((size_t)1 << 24) && !twin->dit.passes_c2r_tp.empty()))
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
               PassTimer *timer = nullptr, const typename Planner<T>::Choice *forced = nullptr) const {
        const size_t half = n / 2;
        hipStream_t s = L.stream;
        if (dit.passes.empty())  // N/2 <= 8192: one kernel, the preprocess is its prologue
            return dit.exec_small_real(2, d_ire, d_iim, in_dist, d_out, nullptr, out_dist / 2, batch, 1.0 / (double)half,
                                       d_tw3, tw_bits, s);
        const typename Planner<T>::Choice ch = forced ? *forced : dit.choose(kC2R, batch, batch);
        if (c2r_fuses(ch)) {  // the first pass forms z on load: no preprocess sweep, no workspace (c2r_fused.hpp)
            const R2cFuse fuse{d_tw3, tw_bits};
            return dit.exec_in(L, d_ire, d_iim, in_dist, 3, d_out, nullptr, out_dist / 2, 2, batch, 1.0 / (double)half, timer, &fuse,
                               nullptr, nullptr, &ch);
        }
        // no fused form of that plan's first pass: the preprocess sweep, then the inner transform -- on a measured C2R plan if
        // there is one (it was measured running exactly this), else on the C2C choice for the chunk
        const typename Planner<T>::Choice *inner = (forced || ch.tuned) ? &ch : nullptr;
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
            if (timer) PHAST_HIP(timer->pair((int)(inner ? inner->passes->size() : dit.choose(kC2CI, nb, nb).passes->size()), &e0, &e1));
            PHAST_HIP(launch_c2r_preprocess<T>(pa, s, e0, e1));
            // inverse by the swap trick (algorithms/dit.rs:297-300): forward FFT of (z_im, z_re), 1/half scale,
            // and the (positional re, positional im) = (caller im, caller re) pair is stored as (im, re)
            rc = dit.exec_in(L, z_im, z_re, half, 0, d_out + b0 * out_dist, nullptr, out_dist / 2, 2, nb, 1.0 / (double)half, timer,
                             nullptr, nullptr, nullptr, inner);
            if (rc) return rc;
        }
        return PHAST_OK;
    }
};

}  // namespace phast
