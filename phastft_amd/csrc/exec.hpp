// exec.hpp -- Planner<T>: the launches of one (batched) transform.
#pragma once

#include "planner.hpp"

namespace phast {

template <typename T>
hipError_t Planner<T>::launch_pass(const PassDesc &p, const TileArgs &ta, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) const {
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
template <typename T>
int Planner<T>::exec_strided(T *re, T *im, unsigned s_bits, unsigned sb_bits, double scale, hipStream_t stream, unsigned grid_log_n,
                             unsigned grid_col0) const {
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
        rc = prepare_passes(sp->passes);
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(mu);
        for (const auto &q : strided_plans)  // another thread may have built the same plan meanwhile
            if (q->s == s_bits && q->sb == sb_bits && q->grid_log_n == grid_log_n) plan = q.get();
        if (!plan) {
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

// Small real transforms (the N-point core runs in the one-pass kernel): R2C untangle / C2R preprocess fused
// into that kernel (row_fft.hpp, RowArgs::real_mode).  rtw3 = W_{2N} tables of the R2C planner.
template <typename T>
int Planner<T>::exec_small_real(unsigned mode, const void *in_a, const void *in_b, size_t in_dist, void *out_a, void *out_b,
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

// a _dev call: checks a workspace out for the enqueue
template <typename T>
int Planner<T>::exec(const void *in_re, const void *in_im, size_t in_dist, unsigned in_mode, void *out_re, void *out_im,
                     size_t out_dist, unsigned out_mode, size_t batch, double scale, hipStream_t stream, PassTimer *timer) const {
    if (batch == 0) return PHAST_OK;
    // one 8192-point transform: two passes over the whole chip instead of one workgroup -- unless the call is being captured
    // and the twin has no scratch yet: the one-pass kernel needs none, so a capture without a warm-up call keeps working as
    // it did before the twin existed
    if (route_small(batch) != this && !(capturing(stream) && !twin->capture_ready(stream)))
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
template <typename T>
int Planner<T>::exec_in(const Lease &L, const void *in_re, const void *in_im, size_t in_dist, unsigned in_mode, void *out_re,
                        void *out_im, size_t out_dist, unsigned out_mode, size_t batch, double scale, PassTimer *timer,
                        const R2cFuse *fuse, bool *fused_out, size_t *np_out, const Choice *forced) const {
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
    // which plan, and (R2C) fused or not: ONE decision per call (Planner::choose) -- or the caller's (a tuning run's candidate)
    const int kind = in_mode == 3 ? kC2R : fuse ? kR2C : (in_mode || out_mode) ? kC2CI : kC2C;
    const Choice ch = forced ? *forced : choose(kind, batch, batch < cap ? batch : cap);
    const bool r2c_fuse = fuse && in_mode != 3 && ch.r2c_fuse;
    const std::vector<PassDesc> &passes = *ch.passes;
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

}  // namespace phast
