// planner.hpp -- Planner<T>: the tables, plans and workspace pool behind PlannerDit64 / PlannerDit32 (planner.rs:34-114).
// The long member functions live next door: planner_pool.hpp (workspaces), planner_plans.hpp (plans and tables), exec.hpp
// (the launches).
#pragma once

#include <map>
#include <thread>

#include "wisdom.hpp"
#include "workspace.hpp"

namespace phast {

// ------------------------------------------------------------------------------------------------
// planner
// ------------------------------------------------------------------------------------------------
struct PassDesc : PassGeom {
    // twiddle tables: entries of the planner's table cache (Planner::table) -- shared between plans, released with the planner
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

static const char *test_perturb_hook() {  // PHAST_TEST_PERTURB_TW3: tests only (Planner::table)
    static const char *v = [] {
        const char *e = std::getenv("PHAST_TEST_PERTURB_TW3");
        return (e && *e) ? e : nullptr;
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
    else return launch_wave_f32(transpose, s, a, q, b, l, e0, e1);
}
template <typename T> static hipError_t launch_quad(unsigned grid, hipStream_t s, const TileArgs &a, bool q, int *b, size_t *l,
                                                    hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if constexpr (sizeof(T) == 8) return launch_quad_f64(grid, s, a, q, b, l, e0, e1);
    else return launch_quad_f32(grid, s, a, q, b, l, e0, e1);
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

template <typename T> struct Planner {
    using value_type = T;
    unsigned wisdom_log_n() const { return log_n; }
    int device_of() const { return device; }
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
    bool is_inner_of_real = false;  // the N/2-point planner inside a PlannerR2c: its wisdom is keyed by the real length
    // elements per transform and plane in the scratch: n plus the padding of the intermediate layouts (plan.hpp:
    // scratch_pad_bytes); the largest over this planner's plans, fixed before the first allocation grows past it
    size_t scratch_stride = 0;
    size_t sstride() const { return scratch_stride ? scratch_stride : n; }
    // the widest (padded) intermediate layout of any plan installed right now -- static, forced or tuned; plan_mu held
    size_t installed_pitch_locked() const {
        size_t need = n;
        for (const std::vector<PassDesc> *v : {&passes, &passes_lat, &passes_mid, &passes_one, &passes_c2r_one, &passes_c2r_lat, &passes_r2c,
                                               &passes_r2c_tp, &passes_c2r_tp})
            for (const PassDesc &p : *v) need = std::max(need, (size_t)p.scratch_dist);
        for (const auto &t : tuned)
            for (const PassDesc &p : t->passes) need = std::max(need, (size_t)p.scratch_dist);
        return need;
    }
    mutable std::atomic<size_t> reserve{1};
    int device = -1;  // the device this planner's tables and scratch live on (current at creation); calls run there
    // Twiddle tables, keyed by what they hold: every plan of this planner -- the static ones, those a tuning run tries, those
    // set_plan installs -- shares them; uploaded once, released with the planner.  (Until round 5 every plan owned copies and a
    // replaced plan's tables were parked until the planner died, counted twice: ADVICE r04.)  A few dozen tables of at most
    // 96 KiB each.
    enum TableKind : unsigned { kTwr = 1, kTwq = 2, kTw3 = 3, kTwu = 4 };
    mutable std::mutex tables_mu;
    mutable std::map<unsigned long long, void *> tables;
    mutable std::atomic<size_t> table_bytes{0};
    int table(TableKind kind, unsigned a, unsigned b, void **out) const;
    // the workspace pool (see Workspace); `mu` guards the pool's bookkeeping, never a launch
    mutable std::mutex mu;
    mutable std::condition_variable cv;
    mutable std::vector<std::unique_ptr<Workspace>> pool;
    // plans are read by every call and replaced by set_plan (a tuning hook): shared for the enqueue, exclusive to swap
    mutable std::shared_mutex plan_mu;
    // Plans a tuning run measured (tune.hpp) or wisdom supplied (wisdom.hpp), per call kind and batch bucket: looked up before
    // the static rules (choose).  Entries are added, replaced and removed under the EXCLUSIVE hold of plan_mu only (a Choice that
    // points into one lives inside a call's shared hold); kernels in flight or captured hold their arguments by value.
    struct TunedPlan {
        int kind = kC2C;
        unsigned bucket = 0;
        bool fuse = false;  // r2c: run the fused last pass
        std::vector<PassDesc> passes;
        PlanSpec spec;
        float us = 0, us_heur = 0;
    };
    mutable std::vector<std::unique_ptr<TunedPlan>> tuned;
    mutable std::mutex tune_mu;  // one tuning run at a time per planner
    // set by the public plan hook (phast_planner_*_set_plan / _set_inner_plan with a plan), cleared by the hook's restore form
    // and by a later tuning run: while it is set, calls run the plan that was forced -- not a tuned or built-in-wisdom plan
    // that happens to exist for the (kind, bucket) (a forced plan was silently ignored there).  Guarded by plan_mu.
    bool forced = false;
    void set_forced(bool f) {
        std::unique_lock<std::shared_mutex> plans(plan_mu);
        forced = f;
    }
    const TunedPlan *tuned_for(int kind, size_t batch) const {
        if (forced) return nullptr;
        const unsigned b = batch_bucket(batch);
        for (const auto &t : tuned)
            if (t->kind == kind && t->bucket == b) return t.get();
        return nullptr;
    }
    // The plan a call runs and whether an R2C call fuses its untangle into the last pass: ONE decision per call, made here
    // (until round 5 exec_in and PlannerR2c each had their own copy of the fuse rule: ADVICE r03).  `batch` = transforms of the
    // call, `chunk` = transforms per launch (the scratch may hold fewer than the call brings).
    struct Choice {
        const std::vector<PassDesc> *passes = nullptr;
        bool r2c_fuse = false;
        const TunedPlan *tuned = nullptr;
    };
    Choice choose(int kind, size_t batch, size_t chunk, bool use_tuned = true) const;
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
    // workspaces (planner_pool.hpp)
    int check_out(Lease &L, hipStream_t stream, int which = 0) const;
    void check_in(Workspace *ws, hipStream_t stream, bool host_synchronised) const;
    int ensure_scratch(const Lease &L, size_t batch, size_t *cap_out, bool exact = false) const;
    int check_guards(size_t *bad_out) const;
    bool capture_ready(hipStream_t stream) const;
    int reserve_batch(size_t max_batch);
    size_t release_graph_workspaces();

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
    // (plans do not own their tables -- see `tables`: dropping a plan is dropping its descriptors; kernels already enqueued or
    // captured keep reading tables that live as long as the planner)
    static void retire_passes(std::vector<PassDesc> &v) { v.clear(); }
    void release_passes() {
        strided_plans.clear();
        tuned.clear();
        for (auto *v : {&passes, &passes_lat, &passes_mid, &passes_one, &passes_c2r_one, &passes_c2r_lat, &passes_r2c_tp, &passes_c2r_tp, &passes_r2c})
            v->clear();
        for (auto &kv : tables) hipFree(kv.second);
        tables.clear();
        table_bytes = 0;
    }
    void release() {
        DeviceGuard on(device);
        release_passes();
        if (d_small_tw) hipFree(d_small_tw);
        d_small_tw = nullptr;
        for (auto &w : pool) w->release();
        pool.clear();
    }

    // plans and tables (planner_plans.hpp)
    const std::vector<PassDesc> &plan_for(size_t batch) const;
    const std::vector<PassDesc> &plan_for_r2c(size_t batch, bool fusing = true) const;
    const std::vector<PassDesc> &plan_for_c2r(size_t batch) const;
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

    int set_plan(const std::vector<unsigned> &lrs, const std::vector<unsigned> &tls, int which = 0, unsigned lp = 4);

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
    // Round 5: 4096 points as well.  Measured per call kind (profiles/r05_small_twin_4096.log, one transform, graph over a cold
    // ring): C2R of 8192 real points 12.5 -> 8.7 us in f64, 11.1 -> 6.7 in f32 (two passes with the preprocess fused into the
    // first against one workgroup's chain) -- adopted; R2C 9.8 -> 10.4 / 8.1 -> 8.8 (the untangle becomes a third kernel) -- not;
    // C2C: 8.48 -> 7.67 us (f64), 6.72 -> 6.50 (f32) once the capture fix let the twin run (call 4's A/B) -- adopted; 2^11 stays
    // in one workgroup (7.5 us).  PHAST_SMALL_TWIN_MIN_LOG=13 restores round 4's threshold for C2C (tools: A/B).
    static unsigned twin_min_log_c2c() {
        static const unsigned v = [] {
            const char *e = std::getenv("PHAST_SMALL_TWIN_MIN_LOG");
            const unsigned m = (e && *e) ? (unsigned)std::atoi(e) : kTwinMinLog;
            return m < kTwinMinLog ? kTwinMinLog : m;
        }();
        return v;
    }
    static constexpr unsigned kTwinMinLogR2c = kSmallMaxLog, kTwinMinLogC2r = kTwinMinLog;  // log2 of the INNER length
    // the planner a call of ONE (or a few) C2C transforms runs on: the multi-pass twin where this length has one in use
    const Planner<T> *route_small(size_t batch = 1) const {
        return (twin && log_n >= twin_min_log_c2c() && batch <= twin_max_batch()) ? twin.get() : this;
    }
    static size_t twin_max_batch() {
        static const size_t v = [] {
            const char *e = std::getenv("PHAST_SMALL_TWIN_MAX_BATCH");
            return (e && *e) ? (size_t)std::atoll(e) : (size_t)128;
        }();
        return v;
    }

    int init(size_t num_points, bool force_multi = false, bool with_twin = true);
    int default_plans();
    int make_c2r_plans();
    int prepare_passes(std::vector<PassDesc> &ps) const;
    int build_plan(const PlanSpec &spec, std::vector<PassDesc> &ps, size_t *scratch_need) const;
    // wisdom -> tuned plans; measuring (tune.hpp).  `real_log_n`: log2 of the caller's length for kR2C / kC2R (this planner is the
    // inner one), else 0
    int apply_wisdom(unsigned real_log_n);
    int install_tuned(int kind, unsigned bucket, const PlanSpec &spec, bool fuse, float us, float us_heur);
    struct TuneReport {
        int adopted = 0;  // a measured plan replaced the static rule's
        int rejected = 0; // the fastest plan computed something else than the static rule's (never adopted; a bug to report)
        unsigned candidates = 0;
        float us_heuristic = 0, us_best = 0;
        double seconds = 0;
        std::string plan;
    };
    int install_built(int kind, unsigned bucket, const PlanSpec &spec, bool fuse, std::vector<PassDesc> &&ps, float us, float us_heur);
    void remove_tuned(int kind, unsigned bucket);
    template <typename Run, typename Refill, typename Digest>
    int tune_core(int kind, size_t batch, unsigned wisdom_log_n, int ring, bool grows, Run &&run, Refill &&refill, Digest &&digest,
                  TuneReport *rep);
    int tune(int kind, size_t batch, TuneReport *rep);
    std::string describe() const;
    std::string describe_call(int kind, size_t batch) const;
    // live tables and scratch plus what is retired but not yet released
    size_t device_bytes() const {
        std::lock_guard<std::mutex> lk(mu);
        size_t b = table_bytes;
        for (auto &w : pool) b += w->device_bytes();
        if (twin) b += twin->device_bytes();
        return b;
    }

    // launches (exec.hpp)
    hipError_t launch_pass(const PassDesc &p, const TileArgs &ta, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) const;
    int exec_strided(T *re, T *im, unsigned s_bits, unsigned sb_bits, double scale, hipStream_t stream, unsigned grid_log_n = 0,
                         unsigned grid_col0 = 0) const;
    int exec_small_real(unsigned mode, const void *in_a, const void *in_b, size_t in_dist, void *out_a, void *out_b, size_t out_dist,
                            size_t batch, double scale, const void *rtw3, unsigned rtw_bits, hipStream_t stream) const;
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
    int exec(const void *in_re, const void *in_im, size_t in_dist, unsigned in_mode, void *out_re, void *out_im, size_t out_dist,
                 unsigned out_mode, size_t batch, double scale, hipStream_t stream, PassTimer *timer = nullptr) const;
    int exec_in(const Lease &L, const void *in_re, const void *in_im, size_t in_dist, unsigned in_mode, void *out_re, void *out_im,
                    size_t out_dist, unsigned out_mode, size_t batch, double scale, PassTimer *timer = nullptr,
                    const R2cFuse *fuse = nullptr, bool *fused_out = nullptr, size_t *np_out = nullptr,
                    const Choice *forced = nullptr) const;
};

}  // namespace phast
