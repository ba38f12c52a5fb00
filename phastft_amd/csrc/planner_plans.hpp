// planner_plans.hpp -- Planner<T>: which plan serves which call, building plans and their tables.
#pragma once

#include "planner.hpp"

namespace phast {

// the plan for `batch` transforms in flight: throughput when its tiles fill the chip, else the mid plan for
// more than one transform (where there is one), else the latency plan
template <typename T> const std::vector<PassDesc> &Planner<T>::plan_for(size_t batch) const {
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
template <typename T> const std::vector<PassDesc> &Planner<T>::plan_for_r2c(size_t batch, bool fusing) const {
    if (batch <= 2 && !passes_r2c.empty()) return passes_r2c;  // ranked for R2C itself (plan.hpp: real_plan)
    const std::vector<PassDesc> &ps = plan_for(batch);
    if (&ps == &passes && !passes_r2c_tp.empty()) return passes_r2c_tp;  // ... and for batches of them (real_batch_plan)
    if (fusing && !ps.empty() && ps.back().r2c_blocks == 0 && log_n <= 25 && !passes_lat.empty() && passes_lat.back().r2c_blocks > 0 &&
        r2c_lat_ok())
        return passes_lat;
    return ps;
}

// C2R, the same on the other side: a plan whose FIRST pass is a wave tile has no fused form of it (c2r_fused.hpp)
template <typename T> const std::vector<PassDesc> &Planner<T>::plan_for_c2r(size_t batch) const {
    const std::vector<PassDesc> &ps0 = plan_for(batch);
    if (&ps0 == &passes && batch > 2 && !passes_c2r_tp.empty() && passes_c2r_tp.front().c2r_blocks > 0) return passes_c2r_tp;
    const std::vector<PassDesc> &ps = (batch <= 2 && !passes_c2r_one.empty())                 ? passes_c2r_one
                                      : (&ps0 == &passes_lat && !passes_c2r_lat.empty()) ? passes_c2r_lat
                                                                                         : ps0;
    if (!ps.empty() && ps.front().c2r_blocks == 0 && !passes_lat.empty() && passes_lat.front().c2r_blocks > 0 && c2r_lat_ok())
        return passes_lat;
    return ps;
}

// which: 0 = one plan for every batch size, 1 = throughput plan only, 2 = latency plan only, 3 = mid plan only,
// 4 = the plan for one transform;
// lp = log2(points per thread)
template <typename T> int Planner<T>::set_plan(const std::vector<unsigned> &lrs, const std::vector<unsigned> &tls, int which, unsigned lp) {
    if (lrs.size() < 2 || lrs.size() > 3 || (tls.size() != 1 && tls.size() != lrs.size())) return PHAST_ERR_INVALID_ARG;
    PlanSpec spec;
    spec.np = (unsigned)lrs.size();
    for (unsigned i = 0; i < spec.np; ++i) {
        spec.lr[i] = lrs[i];
        spec.tl[i] = tls.size() == 1 ? tls[0] : tls[i];
    }
    spec.lp = lp;
    std::vector<PassDesc> ps;
    size_t need = 0;
    {
        int rc = build_plan(spec, ps, &need);
        if (rc) return rc;
    }
    // every call reads the pass vectors under a shared hold of plan_mu for as long as it enqueues: a plan is never
    // swapped under a launch sequence; kernels already enqueued (or captured) keep reading the tables, which live as long as
    // the planner (Planner::table)
    std::unique_lock<std::shared_mutex> plans(plan_mu);
    if (which == 2) {
        retire_passes(passes_lat);
        retire_passes(passes_c2r_lat);  // derived from the plan that goes
        passes_lat = std::move(ps);
    } else if (which == 3) {
        passes_mid = std::move(ps);
    } else if (which == 4) {
        retire_passes(passes_c2r_one);
        passes_one = std::move(ps);
    } else if (which >= 5 && which <= 9) {  // the real transforms' own plans
        std::vector<PassDesc> &dst = which == 5   ? passes_c2r_one
                                     : which == 6 ? passes_c2r_lat
                                     : which == 7 ? passes_r2c
                                     : which == 8 ? passes_c2r_tp
                                                  : passes_r2c_tp;
        dst = std::move(ps);
    } else {
        passes = std::move(ps);
        if (which == 0)  // one plan for every batch size
            for (auto *v : {&passes_lat, &passes_mid, &passes_one, &passes_c2r_one, &passes_c2r_lat, &passes_r2c, &passes_r2c_tp, &passes_c2r_tp})
                retire_passes(*v);
    }
    // a plan with wider pitches than the scratches were cut for: every workspace re-cuts its scratch on next use
    // (ensure_scratch compares Workspace::per)
    if (need > sstride()) scratch_stride = need;
    return PHAST_OK;
}

// descriptors, tables and launch parameters of the plan `spec` names; *scratch_need = elements per transform and plane its
// (padded) intermediate layout takes.  PHAST_ERR_INVALID_ARG: not a plan of this length, or a tile that does not fit the LDS.
template <typename T> int Planner<T>::build_plan(const PlanSpec &spec, std::vector<PassDesc> &ps, size_t *scratch_need) const {
    std::vector<PassGeom> geo;
    if (!make_passes(log_n, spec.lrs(), spec.tls(), geo, spec.lp, sizeof(T))) return PHAST_ERR_INVALID_ARG;
    PHAST_ON_DEVICE(device);
    ps.assign(geo.size(), PassDesc());
    for (size_t i = 0; i < geo.size(); ++i) static_cast<PassGeom &>(ps[i]) = geo[i];
    if (scratch_need) *scratch_need = (size_t)scratch_elems(geo, log_n);
    return prepare_passes(ps);
}

// One of this planner's twiddle tables, uploaded on first use:
//   kTwr (rows): W_rows^j two-level table of a tile FFT (plan.hpp: host_twr);  kTwq: the four-wave kernel's step twiddles;
//   kTw3 (log_mod, bits): the three-level W_{2^log_mod} table of an inter-pass twiddle;  kTwu (lr): W_{2 * 2^lr}^k, the real
//   transforms' fused passes.
template <typename T> int Planner<T>::table(TableKind kind, unsigned a, unsigned b, void **out) const {
    const unsigned long long key = ((unsigned long long)kind << 48) | ((unsigned long long)a << 24) | b;
    std::lock_guard<std::mutex> lk(tables_mu);
    auto it = tables.find(key);
    if (it != tables.end()) {
        *out = it->second;
        return PHAST_OK;
    }
    std::vector<cx_t<T>> h;
    switch (kind) {
    case kTwr: h = host_twr<T>(a); break;
    case kTwq: h = host_twq<T>(); break;
    case kTw3: h = host_tw3<T>(a, b); break;
    case kTwu:
        h.resize((size_t)1 << a);
        for (size_t k = 0; k < h.size(); ++k) h[k] = twiddle_t<T>(k, 2ull << a);
        break;
    }
    if (const char *hook = test_perturb_hook()) {
        // test-only (tests/test_gpu_parity_r5.py): PHAST_TEST_PERTURB_TW3=<relative error> moves entry 5 of level 0 of every
        // three-level table -- the parity gates must be tight enough to notice a twiddle that is off by 1e-9
        if (kind == kTw3 && h.size() > 5) h[5].x = (T)((double)h[5].x * (1.0 + std::atof(hook)));
    }
    void *d = nullptr;
    int rc = upload<T>(h, &d);
    if (rc) return rc;
    tables.emplace(key, d);
    table_bytes += h.size() * sizeof(cx_t<T>);
    *out = d;
    return PHAST_OK;
}

// ---- which plan a call runs ----
template <typename T> typename Planner<T>::Choice Planner<T>::choose(int kind, size_t batch, size_t chunk, bool use_tuned) const {
    Choice c;
    if (const TunedPlan *t = use_tuned ? tuned_for(kind, batch) : nullptr) {  // measured (tune.hpp) or supplied as wisdom
        c.passes = &t->passes;
        c.r2c_fuse = kind == kR2C && t->fuse && r2c_fuse_enabled();
        c.tuned = t;
        return c;
    }
    switch (kind) {
    case kR2C:
        // fused or not is decided from the size of a full chunk -- a smaller tail chunk follows the others (the caller runs
        // the untangle sweep over the whole batch or not at all)
        c.r2c_fuse = fuse_pays(chunk);
        c.passes = &plan_for_r2c(batch, c.r2c_fuse);
        break;
    case kC2R: c.passes = &plan_for_c2r(batch); break;
    default: c.passes = &plan_for(batch); break;
    }
    return c;
}

// a measured (or imported) plan for (kind, bucket); replaces an earlier one for the same key
template <typename T> int Planner<T>::install_tuned(int kind, unsigned bucket, const PlanSpec &spec, bool fuse, float us, float us_heur) {
    std::vector<PassDesc> ps;
    int rc = build_plan(spec, ps, nullptr);
    if (rc) return rc;
    return install_built(kind, bucket, spec, fuse, std::move(ps), us, us_heur);
}
template <typename T>
int Planner<T>::install_built(int kind, unsigned bucket, const PlanSpec &spec, bool fuse, std::vector<PassDesc> &&ps, float us, float us_heur) {
    std::unique_ptr<TunedPlan> t(new (std::nothrow) TunedPlan());
    if (!t || ps.empty()) return PHAST_ERR_ALLOC;
    size_t need = n;
    for (const PassDesc &p : ps) need = std::max(need, (size_t)p.scratch_dist);
    // an R2C plan marked `fuse` must have the fused last pass (a C2R plan without the fused first pass falls back to the
    // preprocess sweep: legal, and measured as such by a tuning run)
    t->fuse = kind == kR2C && fuse && ps.back().r2c_blocks > 0;
    t->kind = kind;
    t->bucket = bucket;
    t->passes = std::move(ps);
    t->spec = spec;
    t->us = us;
    t->us_heur = us_heur;
    // exclusive: no call is enqueueing with a Choice that points into the entry that goes (calls hold plan_mu shared while
    // they enqueue; kernels in flight read tables, which are the planner's)
    std::unique_lock<std::shared_mutex> plans(plan_mu);
    forced = false;  // a tuning run (or an import) after set_plan: the newest instruction wins
    if (need > sstride()) scratch_stride = need;
    for (auto &e : tuned)
        if (e->kind == kind && e->bucket == bucket) {
            e = std::move(t);
            return PHAST_OK;
        }
    tuned.push_back(std::move(t));
    return PHAST_OK;
}
template <typename T> void Planner<T>::remove_tuned(int kind, unsigned bucket) {
    std::unique_lock<std::shared_mutex> plans(plan_mu);
    for (size_t i = 0; i < tuned.size(); ++i)
        if (tuned[i]->kind == kind && tuned[i]->bucket == bucket) {
            tuned.erase(tuned.begin() + (long)i);
            return;
        }
}

// the wisdom store's entries for this planner's type and length become tuned plans (entries that name the static rule's plan
// install nothing).  Entries that no longer build -- a tile shape that went -- are skipped.
template <typename T> int Planner<T>::apply_wisdom(unsigned real_log_n) {
    if (passes.empty()) return PHAST_OK;
    const int cus = cus_of(device), arch = arch_of(device);
    for (int kind : {(int)kC2C, (int)kC2CI, (int)kR2C, (int)kC2R}) {
        const bool real = kind == kR2C || kind == kC2R;
        if (real != (real_log_n != 0)) continue;
        for (const auto &kv : WisdomStore::instance().lookup_all(sizeof(T), kind, real ? real_log_n : log_n, cus, arch)) {
            const WisdomEntry &e = kv.second;
            if (e.heuristic) continue;
            int rc = install_tuned(kind, kv.first, e.spec, e.fuse, e.us, e.us_heur);
            if (rc != PHAST_OK && rc != PHAST_ERR_INVALID_ARG) return rc;
        }
    }
    return PHAST_OK;
}

template <typename T> int Planner<T>::init(size_t num_points, bool force_multi, bool with_twin) {
    n = num_points;
    log_n = ilog2(n);
    int rc = ensure_device(&device);
    if (rc) return rc;
    if (log_n <= kSmallMaxLog && !(force_multi && log_n >= kTwinMinLog)) {
        std::vector<cx_t<T>> h = host_twr<T>((unsigned)n);  // W_N^j two-level table of the one-pass kernel
        rc = upload<T>(h, &d_small_tw);
        if (rc == PHAST_OK) table_bytes += h.size() * sizeof(cx_t<T>);
        if (rc == PHAST_OK && with_twin && log_n >= kTwinMinLog && twin_enabled()) {
            twin.reset(new (std::nothrow) Planner<T>());
            if (twin && twin->init(n, true) != PHAST_OK) twin.reset();  // an optimisation: without it the one-pass kernel serves
        }
        return rc;
    }
    rc = default_plans();
    if (rc == PHAST_OK && !is_inner_of_real) rc = apply_wisdom(0);
    return rc;
}

// the library's own two plans (plan.hpp: heuristic_plan)
template <typename T> int Planner<T>::default_plans() {
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
template <typename T> int Planner<T>::make_c2r_plans() {
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

// tables + launch parameters of a pass list (shared by build_plan and the strided plans)
template <typename T> int Planner<T>::prepare_passes(std::vector<PassDesc> &ps) const {
    for (size_t i = 0; i < ps.size(); ++i) {
        PassDesc &q = ps[i];
        int rc = q.quad ? table(kTwq, 0, 0, &q.d_twr) : table(kTwr, 1u << q.lr, 0, &q.d_twr);
        if (rc == PHAST_OK && q.pre_tw) rc = table(kTw3, q.log_mod(), q.tw_bits, &q.d_tw3);
        if (rc) return rc;
        TileArgs ta{};
        ta.tw_bits = q.tw_bits;
        hipError_t e = q.wave        ? launch_wave<T>(q.transpose, nullptr, ta, true, &q.blocks_per_cu, &q.lds)
                       : q.quad      ? launch_quad<T>(0, nullptr, ta, true, &q.blocks_per_cu, &q.lds)
                       : q.transpose ? Types<T>::launch_a(q.lr, q.lc, (int)q.lp, 0, nullptr, ta, true, &q.blocks_per_cu, &q.lds)
                                     : Types<T>::launch_bc(q.lr, q.lc, (int)q.lp, 0, nullptr, ta, true, &q.blocks_per_cu, &q.lds);
        if (e != hipSuccess) return hip_fail(e, "occupancy query");
        if (q.blocks_per_cu < 1) q.blocks_per_cu = 1;
        if (q.lds > 160 * 1024) return PHAST_ERR_INVALID_ARG;  // tile does not fit one CU's LDS
        // the fused R2C form of a LAST pass: generic tiles with <= 16 points per thread whose columns span at least
        // two tiles (r2c_fused.hpp); anything else keeps the separate untangle sweep
        if (i + 1 == ps.size() && i > 0 && !q.wave && !q.quad && !q.strided && q.pre_tw && q.log_s_in >= q.lc + 1 &&
            r2c_shape_ok(q.lr, q.lc, q.lp, sizeof(T))) {
            rc = table(kTwu, q.lr, 0, &q.d_twu);
            if (rc) return rc;
            R2cFuseArgs fa{};
            int b = 0;
            hipError_t e2 = launch_r2c_last<T>((int)q.lr, (int)q.lc, (int)q.lp, 0, nullptr, ta, fa, true, &b);
            q.r2c_blocks = e2 == hipSuccess ? b : 0;
            (void)hipGetLastError();
        }
        // the fused C2R form of a FIRST pass of a contiguous transform: generic tiles, at least two of them per transform
        if (i == 0 && ps.size() > 1 && !q.wave && !q.quad && !q.strided && q.transpose && !q.pre_tw && q.log_s_in >= q.lc + 1 &&
            c2r_shape_ok(q.lr, q.lc, q.lp, sizeof(T))) {
            rc = table(kTwu, q.lr, 0, &q.d_twu);
            if (rc) return rc;
            C2rFuseArgs fa{};
            int b = 0;
            hipError_t e2 = launch_c2r_first<T>((int)q.lr, (int)q.lc, (int)q.lp, 0, nullptr, ta, fa, true, &b);
            q.c2r_blocks = e2 == hipSuccess ? b : 0;
            (void)hipGetLastError();
        }
    }
    return PHAST_OK;
}

// "<which plan> [rows x cols ...][...]" of the call (kind, batch): what bench.py and the tools label their numbers with -- the
// library's own answer (choose), not a copy of its rules
template <typename T> std::string Planner<T>::describe_call(int kind, size_t batch) const {
    if (passes.empty()) return route_small(batch) != this ? twin->describe_call(kind, batch) : std::string("one-pass");
    std::shared_lock<std::shared_mutex> plans(plan_mu);
    const Choice c = choose(kind, batch ? batch : 1, batch ? batch : 1);
    const std::vector<PassDesc> *v = c.passes;
    std::string s = c.tuned                  ? "tuned"
                    : forced                 ? "forced"
                    : v == &passes           ? "throughput"
                    : v == &passes_mid       ? "mid"
                    : v == &passes_lat       ? "latency"
                    : v == &passes_one       ? "single"
                    : v == &passes_c2r_one   ? "c2r-single"
                    : v == &passes_c2r_lat   ? "c2r-latency"
                    : v == &passes_r2c       ? "r2c-single"
                    : v == &passes_r2c_tp    ? "r2c-batch"
                    : v == &passes_c2r_tp    ? "c2r-batch"
                                             : "?";
    s += " ";
    char buf[96];
    for (const PassDesc &p : *v) {
        std::snprintf(buf, sizeof buf, "[%ux%u%s %s%u]", 1u << p.lr, 1u << p.lc, p.transpose ? "A" : "", p.wave ? "w" : p.quad ? "q" : "p",
                      1u << p.lp);
        s += buf;
    }
    if (kind == kR2C && c.r2c_fuse && !v->empty() && v->back().r2c_blocks > 0) s += " untangle-fused";
    if (kind == kC2R && !v->empty() && v->front().c2r_blocks > 0 && c2r_fuse_enabled()) s += " preprocess-fused";
    return s;
}

template <typename T> std::string Planner<T>::describe() const {
    // shared hold: a tuning run beside this call installs / removes entries of `tuned` under the exclusive lock
    std::shared_lock<std::shared_mutex> plans(plan_mu);
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
        if (route_small() != this) add("single", twin->plan_for(1));
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
    for (const auto &t : tuned)
        add((std::string("tuned:") + kind_name(t->kind) + "/b" + std::to_string(t->bucket) + (t->fuse ? "/fused" : "")).c_str(), t->passes);
    return s;
}

}  // namespace phast
