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

template <typename T> int Planner<T>::init(size_t num_points, bool force_multi, bool with_twin) {
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

// tables + launch parameters of a pass list (shared by set_plan and the strided plans)
template <typename T> int Planner<T>::prepare_passes(std::vector<PassDesc> &ps, size_t *table_bytes_out) const {
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

template <typename T> std::string Planner<T>::describe() const {
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

}  // namespace phast
