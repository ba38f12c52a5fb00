// emu.hip -- CPU emulation of the multi-pass tile FFT.  TEST INFRASTRUCTURE (libphastft_emu.so), not
// part of libphastft_hip.so: it runs the SAME TileBody phase functions and the SAME plan geometry on
// host memory, thread by thread, so the kernels' index arithmetic, LDS layouts, twiddle tables and
// pass plans can be checked against the oracle in the GPU-less build container
// (tests/test_emulator.py).  It is never loaded by the product package.
#include <cstring>
#include <vector>

#include "plan.hpp"
#include "row_fft.hpp"
#include "tile_fft.hpp"
#include "wave_fft.hpp"
#include "quad_fft.hpp"
#include "r2c_fused.hpp"
#include "c2r_fused.hpp"
#include "tile_dispatch.hpp"

namespace phast {

// A plan named as the wisdom names it ("9,9@13,13:p32", plan.hpp: PlanSpec) -- tile logs PER PASS and the wave flag, which
// the (lrs, tile_log | points << 8) arguments of the entry points below cannot carry.  While one is set
// (phast_emu_set_plan) it replaces those arguments in emu_exec / emu_r2c_fused / emu_c2r_fused.
inline PlanSpec g_plan;
inline bool g_have_plan = false;
static bool forced_plan(std::vector<unsigned> &lrs, std::vector<unsigned> &tls, unsigned &lp) {
    if (!g_have_plan) return false;
    lrs = g_plan.lrs();
    tls = g_plan.tls();
    lp = g_plan.lp;
    return true;
}

template <typename T> static bool emu_pass(const PassGeom &p, const TileArgs &a) {
    if (p.quad) {
        emulate_quad_pass<T>(a);
        return true;
    }
    if (p.wave) {
        if (p.transpose) emulate_wave_pass<T, false, true>(a);
        else emulate_wave_pass<T, true, false>(a);
        return true;
    }
#define PHAST_EMU(LR_, LC_, LP_)                                                                   \
    if (p.lr == LR_ && p.lc == LC_ && p.lp == LP_) {                                                   \
        if (p.transpose)                                                                               \
            emulate_tile_pass<T, LR_, LC_, LP_, false, true, plane_seq_v<T, LP_>>(a);         \
        else                                                                                           \
            emulate_tile_pass<T, LR_, LC_, LP_, true, false, plane_seq_v<T, LP_>>(a);         \
        return true;                                                                                   \
    }
    PHAST_TILE_SHAPES(PHAST_EMU)
    if constexpr (sizeof(T) == 4) {
        PHAST_TILE_SHAPES_F32(PHAST_EMU)
    }
#undef PHAST_EMU
    return false;
}

// in -> out through the same pass sequence as Planner<T>::exec_in (csrc/exec.hpp)
template <typename T>
static int emu_exec(const void *in_re, const void *in_im, unsigned in_mode, void *out_re, void *out_im,
                    unsigned out_mode, unsigned log_n, size_t batch, size_t in_dist, size_t out_dist, double scale,
                    const unsigned *lrs_in, size_t np_in, unsigned tile_log_and_lp) {
    const size_t n = (size_t)1 << log_n;
    const unsigned tile_log = tile_log_and_lp & 0xff;
    unsigned lp = (tile_log_and_lp >> 8) ? (tile_log_and_lp >> 8) : 4;
    std::vector<unsigned> lrs(lrs_in, lrs_in + np_in), tls(1, tile_log);
    if (forced_plan(lrs, tls, lp)) {
    } else if (lrs.empty()) {  // the library's own plans: tile_log 0 = latency plan, 1 = the plan for one transform, else throughput
        if (tile_log != 1 || !single_plan<T>(log_n, lrs, tls, lp)) heuristic_plan<T>(log_n, tile_log <= 1, lrs, tls, lp);
    }
    std::vector<PassGeom> ps;
    if (!make_passes(log_n, lrs, tls, ps, lp, sizeof(T))) return 1;
    const size_t sd = (size_t)scratch_elems(ps, log_n);  // per transform and plane: n + the padding of the intermediate layouts
    std::vector<T> s_re(sd * batch), s_im(sd * batch);
    for (size_t i = 0; i < ps.size(); ++i) {
        const PassGeom &p = ps[i];
        std::vector<cx_t<T>> twr = p.quad ? host_twq<T>() : host_twr<T>(1u << p.lr), tw3;
        if (p.pre_tw) tw3 = host_tw3<T>(p.log_mod(), p.tw_bits);
        TileArgs ta{};
        const bool first = i == 0, last = i + 1 == ps.size();
        ta.in_re = first ? in_re : s_re.data();
        ta.in_im = first ? in_im : s_im.data();
        ta.in_dist = first ? in_dist : sd;
        ta.in_interleaved = first ? in_mode : 0;
        ta.out_re = last ? out_re : s_re.data();
        ta.out_im = last ? out_im : s_im.data();
        ta.out_dist = last ? out_dist : sd;
        ta.out_interleaved = last ? out_mode : 0;
        ta.scale = last ? scale : 1.0;
        ta.tw3 = tw3.data();
        ta.twr = twr.data();
        geom_to_args(p, log_n, batch, ta);
        if (!emu_pass<T>(p, ta)) return 2;
    }
    return 0;
}

// real transform of 2^log_n points through the FUSED last pass (r2c_fused.hpp): the inner 2^(log_n - 1)-point transform's
// passes as emu_exec runs them (interleaved load in the first), the last one with the untangle in it.  Returns 3 when the
// plan's last pass has no fused form (the library then runs the separate untangle sweep).
template <typename T> static bool emu_r2c_last(const PassGeom &p, const TileArgs &a, const R2cFuseArgs &f) {
#define PHAST_EMU_R2C(LR_, LC_, LP_)                                                        \
    if constexpr (r2c_shape_fits(LR_, LC_, LP_, sizeof(T))) {                               \
        if (p.lr == LR_ && p.lc == LC_ && p.lp == LP_) {                                    \
            emulate_r2c_last_pass<T, LR_, LC_, LP_, plane_seq_v<T, LP_>>(a, f);             \
            return true;                                                                    \
        }                                                                                   \
    }
    PHAST_TILE_SHAPES(PHAST_EMU_R2C)
#undef PHAST_EMU_R2C
    return false;
}
template <typename T>
static int emu_r2c_fused(const T *in, unsigned log_n, T *ore, T *oim, const unsigned *lrs_in, size_t np_in, unsigned tile_log_and_lp) {
    const unsigned L = log_n - 1;
    const size_t h = (size_t)1 << L;
    const unsigned tile_log = tile_log_and_lp & 0xff;
    unsigned lp = (tile_log_and_lp >> 8) ? (tile_log_and_lp >> 8) : 4;
    std::vector<unsigned> lrs(lrs_in, lrs_in + np_in), tls(1, tile_log);
    if (forced_plan(lrs, tls, lp)) {
    } else if (lrs.empty()) {  // 0 = latency plan, 1 = the plan for one transform, 2 / 3 = the R2C tables (real_plan / real_batch_plan)
        if (tile_log == 2 || tile_log == 3) {
            if (!(tile_log == 2 ? real_plan<T>(L, false, lrs, tls, lp) : real_batch_plan<T>(L, false, lrs, tls, lp))) return 3;
            lp &= ~kFuseBelow;
        } else if (tile_log != 1 || !single_plan<T>(L, lrs, tls, lp)) {
            heuristic_plan<T>(L, tile_log <= 1, lrs, tls, lp);
        }
    }
    std::vector<PassGeom> ps;
    if (!make_passes(L, lrs, tls, ps, lp, sizeof(T))) return 1;
    const PassGeom &q = ps.back();
    if (q.wave || q.quad || !r2c_shape_ok(q.lr, q.lc, q.lp, sizeof(T)) || q.log_s_in < q.lc + 1) return 3;
    const size_t sd = (size_t)scratch_elems(ps, L);
    std::vector<T> s_re(sd), s_im(sd);
    for (size_t i = 0; i < ps.size(); ++i) {
        const PassGeom &p = ps[i];
        std::vector<cx_t<T>> twr = p.quad ? host_twq<T>() : host_twr<T>(1u << p.lr), tw3;
        if (p.pre_tw) tw3 = host_tw3<T>(p.log_mod(), p.tw_bits);
        TileArgs ta{};
        const bool first = i == 0, last = i + 1 == ps.size();
        ta.in_re = first ? (const void *)in : (const void *)s_re.data();
        ta.in_im = first ? nullptr : s_im.data();
        ta.in_dist = first ? h : sd;
        ta.in_interleaved = first ? 1 : 0;
        ta.out_re = last ? ore : s_re.data();
        ta.out_im = last ? oim : s_im.data();
        ta.out_dist = last ? h + 1 : sd;
        ta.scale = 1.0;
        ta.tw3 = tw3.data();
        ta.twr = twr.data();
        geom_to_args(p, L, 1, ta);
        if (!last) {
            if (!emu_pass<T>(p, ta)) return 2;
            continue;
        }
        const unsigned nb = tw3_bits_for(log_n);
        const std::vector<cx_t<T>> tw3n = host_tw3<T>(log_n, nb);
        std::vector<cx_t<T>> twu((size_t)1 << p.lr);
        for (size_t k = 0; k < twu.size(); ++k) twu[k] = twiddle_t<T>(k, 2ull << p.lr);
        R2cFuseArgs fa{};
        fa.tw3n = tw3n.data();
        fa.twn_bits = nb;
        fa.twu = twu.data();
        fa.tiles_per_xform = (1u << (p.log_s_in - p.lc - 1)) + 1u;
        fa.tiles_total = fa.tiles_per_xform;
        fa.pair_tiles = fa.tiles_per_xform - 1u;
        if (!emu_r2c_last<T>(p, ta, fa)) return 3;
    }
    return 0;
}

// inverse real transform (half-spectrum of 2^(log_n - 1) + 1 bins -> 2^log_n reals) through the FUSED first pass
// (c2r_fused.hpp): the inner transform's passes as emu_exec runs them, the first one forming z on load, the last one storing
// (im, re) pairs scaled by 1/h.  Returns 3 when the plan's first pass has no fused form.
template <typename T> static bool emu_c2r_first(const PassGeom &p, const TileArgs &a, const C2rFuseArgs &f) {
#define PHAST_EMU_C2R(LR_, LC_, LP_)                                                        \
    if constexpr (c2r_shape_fits(LR_, LC_, LP_, sizeof(T))) {                               \
        if (p.lr == LR_ && p.lc == LC_ && p.lp == LP_) {                                    \
            emulate_c2r_first_pass<T, LR_, LC_, LP_, plane_seq_v<T, LP_>>(a, f);            \
            return true;                                                                    \
        }                                                                                   \
    }
    PHAST_TILE_SHAPES(PHAST_EMU_C2R)
    if constexpr (sizeof(T) == 4) {
        PHAST_TILE_SHAPES_F32(PHAST_EMU_C2R)
    }
#undef PHAST_EMU_C2R
    return false;
}
template <typename T>
static int emu_c2r_fused(const T *ire, const T *iim, unsigned log_n, T *out, size_t batch, size_t in_dist, const unsigned *lrs_in,
                         size_t np_in, unsigned tile_log_and_lp) {
    const unsigned L = log_n - 1;
    const size_t h = (size_t)1 << L;
    const unsigned tile_log = tile_log_and_lp & 0xff;
    unsigned lp = (tile_log_and_lp >> 8) ? (tile_log_and_lp >> 8) : 4;
    std::vector<unsigned> lrs(lrs_in, lrs_in + np_in), tls(1, tile_log);
    if (forced_plan(lrs, tls, lp)) {
    } else if (lrs.empty()) {  // 0 = latency plan, 1 = the plan for one transform, 2 / 3 = the C2R tables (real_plan / real_batch_plan)
        if (tile_log == 2 || tile_log == 3) {
            if (!(tile_log == 2 ? real_plan<T>(L, true, lrs, tls, lp) : real_batch_plan<T>(L, true, lrs, tls, lp))) return 3;
            lp &= ~kFuseBelow;
        } else if (tile_log != 1 || !single_plan<T>(L, lrs, tls, lp)) {
            heuristic_plan<T>(L, tile_log <= 1, lrs, tls, lp);
        }
    }
    std::vector<PassGeom> ps;
    if (!make_passes(L, lrs, tls, ps, lp, sizeof(T))) return 1;
    const PassGeom &q = ps.front();
    if (ps.size() < 2 || q.wave || q.quad || !q.transpose || !c2r_shape_ok(q.lr, q.lc, q.lp, sizeof(T)) || q.log_s_in < q.lc + 1) return 3;
    const size_t sd = (size_t)scratch_elems(ps, L);
    std::vector<T> s_re(sd * batch), s_im(sd * batch);
    for (size_t i = 0; i < ps.size(); ++i) {
        const PassGeom &p = ps[i];
        std::vector<cx_t<T>> twr = p.quad ? host_twq<T>() : host_twr<T>(1u << p.lr), tw3;
        if (p.pre_tw) tw3 = host_tw3<T>(p.log_mod(), p.tw_bits);
        TileArgs ta{};
        const bool first = i == 0, last = i + 1 == ps.size();
        ta.in_re = first ? (const void *)ire : (const void *)s_re.data();
        ta.in_im = first ? (const void *)iim : (const void *)s_im.data();
        ta.in_dist = first ? in_dist : sd;
        ta.out_re = last ? out : s_re.data();
        ta.out_im = last ? nullptr : s_im.data();
        ta.out_dist = last ? h : sd;  // in (im, re) pairs
        ta.out_interleaved = last ? 2 : 0;
        ta.scale = last ? 1.0 / (double)h : 1.0;
        ta.tw3 = tw3.data();
        ta.twr = twr.data();
        geom_to_args(p, L, batch, ta);
        if (!first) {
            if (!emu_pass<T>(p, ta)) return 2;
            continue;
        }
        const unsigned nb = tw3_bits_for(log_n);
        const std::vector<cx_t<T>> tw3n = host_tw3<T>(log_n, nb);
        std::vector<cx_t<T>> twu((size_t)1 << p.lr);
        for (size_t k = 0; k < twu.size(); ++k) twu[k] = twiddle_t<T>(k, 2ull << p.lr);
        C2rFuseArgs fa{};
        fa.tw3n = tw3n.data();
        fa.twn_bits = nb;
        fa.twu = twu.data();
        if (!emu_c2r_first<T>(p, ta, fa)) return 3;
    }
    return 0;
}

}  // namespace phast

namespace phast {

// LDS audit of one tile shape: every exchange must (1) stay inside the buffer, (2) write each used word exactly
// once, (3) read only written words; and the worst-case bank-conflict degree per wave instruction is
// reported using the gfx950 rules of MI355X_MICROARCH.md section LDS (reads: 32-lane groups, 32 cells of
// sizeof(T) [b32] or 8 bytes [b64]; b64 writes: 16-lane groups over 16 eight-byte cells; b32 writes: 32-lane
// groups over 32 cells).
template <typename T, int LR, int LC, int LP, bool PRE_TW, bool TRANSPOSE> static int audit_shape(int *max_read_ways, int *max_write_ways) {
    using Body = TileBody<T, LR, LC, LP, PRE_TW, TRANSPOSE, plane_seq_v<T, LP>>;
    constexpr int NT = Body::NT;
    int errors = 0, rw = 1, ww = 1;
    auto audit = [&](auto e) {
        constexpr int E = decltype(e)::value;
        std::vector<int> written(Body::EXCH, 0);
        constexpr int PP = Body::P;
        std::vector<std::vector<int>> wa(PP, std::vector<int>(NT)), ra(PP, std::vector<int>(NT));
        for (int t = 0; t < NT; ++t)
            static_for<0, PP>([&](auto P) {
                wa[P][t] = Body::template waddr<E, decltype(P)::value>(t);
                ra[P][t] = Body::template raddr<E, decltype(P)::value>(t);
            });
        for (int p = 0; p < PP; ++p)
            for (int t = 0; t < NT; ++t) {
                if (wa[p][t] < 0 || wa[p][t] >= Body::EXCH) { ++errors; continue; }
                if (written[wa[p][t]]++) ++errors;
            }
        for (int p = 0; p < PP; ++p)
            for (int t = 0; t < NT; ++t)
                if (ra[p][t] < 0 || ra[p][t] >= Body::EXCH || !written[ra[p][t]]) ++errors;
        auto ways = [&](const std::vector<int> &addr, int group, int cells) {
            int worst = 1;
            for (int base = 0; base < NT; base += group) {
                int cnt[32] = {0};
                std::vector<int> seen;
                for (int l = 0; l < group; ++l) {
                    const int a = addr[base + l];
                    bool dup = false;
                    for (int s_ : seen) dup |= (s_ == a);
                    if (dup) continue;  // identical addresses broadcast
                    seen.push_back(a);
                    const int w = ++cnt[a & (cells - 1)];
                    if (w > worst) worst = w;
                }
            }
            return worst;
        };
        // MI355X_MICROARCH.md, LDS: ds_read_b32/b64 serve 2 x 32 lanes over 32 / 64 banks (32 cells of sizeof(T));
        // ds_write_b32 2 x 32 lanes over 32 banks; ds_write_b64 4 x 16 lanes over 32 banks = 16 eight-byte cells
        for (int p = 0; p < PP; ++p) {
            const int r_ = ways(ra[p], 32, 32), w_ = sizeof(T) == 8 ? ways(wa[p], 16, 16) : ways(wa[p], 32, 32);
            if (r_ > rw) rw = r_;
            if (w_ > ww) ww = w_;
        }
    };
    static_for<1, Body::S>([&](auto e) { audit(e); });
    if constexpr (TRANSPOSE && !Body::DIRECT) audit(std::integral_constant<int, Body::S>{});
    *max_read_ways = rw;
    *max_write_ways = ww;
    return errors;
}

// the small-transform kernel (row_fft.hpp): its exchanges are TileBody's, plus the parked tile -- park() must write
// every word it later pick()s exactly once, inside the buffer, without bank conflicts either way
template <typename T, int LR, int LC, int LP> static int audit_small_shape(int *max_read_ways, int *max_write_ways) {
    using RB = RowBody<T, LR, LC, LP>;
    int errors = audit_shape<T, LR, LC, LP, false, true>(max_read_ways, max_write_ways);
    constexpr int NT = RB::NT, PP = RB::P;
    std::vector<int> written(RB::PLANE, 0);
    std::vector<std::vector<int>> wa(PP, std::vector<int>(NT)), ra(PP, std::vector<int>(NT));
    for (int t = 0; t < NT; ++t)
        for (int i = 0; i < PP; ++i) {
            const int f = i * NT + t;
            wa[i][t] = (f >> LR) * RB::PITCH + (f & (RB::ROWS - 1));
            ra[i][t] = RB::Body::col_of(t) * RB::PITCH + RB::Body::tau_of(t) + i * RB::M;
        }
    for (int i = 0; i < PP; ++i)
        for (int t = 0; t < NT; ++t) {
            if (wa[i][t] < 0 || wa[i][t] >= RB::PLANE) { ++errors; continue; }
            if (written[wa[i][t]]++) ++errors;
        }
    for (int i = 0; i < PP; ++i)
        for (int t = 0; t < NT; ++t)
            if (ra[i][t] < 0 || ra[i][t] >= RB::PLANE || !written[ra[i][t]]) ++errors;
    auto ways = [&](const std::vector<int> &addr, int group, int cells) {
        int worst = 1;
        for (int base = 0; base < NT; base += group) {
            int cnt[32] = {0};
            std::vector<int> seen;
            for (int l = 0; l < group; ++l) {
                const int a = addr[base + l];
                bool dup = false;
                for (int s_ : seen) dup |= (s_ == a);
                if (dup) continue;
                seen.push_back(a);
                const int w = ++cnt[a & (cells - 1)];
                if (w > worst) worst = w;
            }
        }
        return worst;
    };
    for (int i = 0; i < PP; ++i) {
        const int r_ = ways(ra[i], 32, 32), w_ = sizeof(T) == 8 ? ways(wa[i], 16, 16) : ways(wa[i], 32, 32);
        if (r_ > *max_read_ways) *max_read_ways = r_;
        if (w_ > *max_write_ways) *max_write_ways = w_;
    }
    return errors;
}

}  // namespace phast

#if !defined(EMU_PART) || EMU_PART == 3
extern "C" int phast_emu_audit_lds(int is_f64, unsigned lr, unsigned lc, unsigned lp, int transpose, int *max_read_ways,
                                   int *max_write_ways) {
#define PHAST_AUD(LR_, LC_, LP_)                                                                                       \
    if (lr == LR_ && lc == LC_ && lp == LP_) {                                                                         \
        if (is_f64)                                                                                                    \
            return transpose ? phast::audit_shape<double, LR_, LC_, LP_, false, true>(max_read_ways, max_write_ways)   \
                             : phast::audit_shape<double, LR_, LC_, LP_, true, false>(max_read_ways, max_write_ways);  \
        return transpose ? phast::audit_shape<float, LR_, LC_, LP_, false, true>(max_read_ways, max_write_ways)        \
                         : phast::audit_shape<float, LR_, LC_, LP_, true, false>(max_read_ways, max_write_ways);       \
    }
    PHAST_TILE_SHAPES(PHAST_AUD)
#undef PHAST_AUD
#define PHAST_AUD32(LR_, LC_, LP_)                                                                                     \
    if (!is_f64 && lr == LR_ && lc == LC_ && lp == LP_)                                                                \
        return transpose ? phast::audit_shape<float, LR_, LC_, LP_, false, true>(max_read_ways, max_write_ways)        \
                         : phast::audit_shape<float, LR_, LC_, LP_, true, false>(max_read_ways, max_write_ways);
    PHAST_TILE_SHAPES_F32(PHAST_AUD32)
#undef PHAST_AUD32
    return -1;
}
#endif

extern "C" {
// planar in-place forward / inverse (swap trick + 1/N as algorithms/dit.rs:297-300,325-331)
#if !defined(EMU_PART) || EMU_PART == 1
// every transform that follows runs the plan `spec` names (nullptr: back to the arguments' plan); 1 = not a plan text
int phast_emu_set_plan(const char *spec) {
    phast::g_have_plan = false;
    if (!spec) return 0;
    if (!phast::spec_from_string(spec, phast::g_plan)) return 1;
    phast::g_have_plan = true;
    return 0;
}
int phast_emu_fft_f64(double *re, double *im, unsigned log_n, size_t batch, int direction, const unsigned *lrs,
                      size_t np, unsigned tile_log) {
    const size_t n = (size_t)1 << log_n;
    if (direction < 0) return phast::emu_exec<double>(im, re, 0, im, re, 0, log_n, batch, n, n, 1.0 / (double)n, lrs, np, tile_log);
    return phast::emu_exec<double>(re, im, 0, re, im, 0, log_n, batch, n, n, 1.0, lrs, np, tile_log);
}
#endif
#if !defined(EMU_PART) || EMU_PART == 2
int phast_emu_fft_f32(float *re, float *im, unsigned log_n, size_t batch, int direction, const unsigned *lrs,
                      size_t np, unsigned tile_log) {
    const size_t n = (size_t)1 << log_n;
    if (direction < 0) return phast::emu_exec<float>(im, re, 0, im, re, 0, log_n, batch, n, n, 1.0 / (double)n, lrs, np, tile_log);
    return phast::emu_exec<float>(re, im, 0, re, im, 0, log_n, batch, n, n, 1.0, lrs, np, tile_log);
}
// interleaved input -> planar output (the R2C inner transform) and planar -> interleaved (im, re) (C2R)
int phast_emu_fft_f32_modes(const float *in_re, const float *in_im, unsigned in_mode, float *out_re, float *out_im,
                            unsigned out_mode, unsigned log_n, double scale, const unsigned *lrs, size_t np,
                            unsigned tile_log) {
    const size_t n = (size_t)1 << log_n;
    return phast::emu_exec<float>(in_re, in_im, in_mode, out_re, out_im, out_mode, log_n, 1, n, n, scale, lrs, np, tile_log);
}
#endif
#if !defined(EMU_PART) || EMU_PART == 4
int phast_emu_r2c_fused_f32(const float *in, unsigned log_n, float *ore, float *oim, const unsigned *lrs, size_t np, unsigned tile_log) {
    return phast::emu_r2c_fused<float>(in, log_n, ore, oim, lrs, np, tile_log);
}
int phast_emu_c2r_fused_f32(const float *ire, const float *iim, unsigned log_n, float *out, size_t batch, size_t in_dist,
                            const unsigned *lrs, size_t np, unsigned tile_log) {
    return phast::emu_c2r_fused<float>(ire, iim, log_n, out, batch, in_dist, lrs, np, tile_log);
}
int phast_emu_c2r_fused_f64(const double *ire, const double *iim, unsigned log_n, double *out, size_t batch, size_t in_dist,
                            const unsigned *lrs, size_t np, unsigned tile_log) {
    return phast::emu_c2r_fused<double>(ire, iim, log_n, out, batch, in_dist, lrs, np, tile_log);
}
int phast_emu_r2c_fused_f64(const double *in, unsigned log_n, double *ore, double *oim, const unsigned *lrs, size_t np, unsigned tile_log) {
    return phast::emu_r2c_fused<double>(in, log_n, ore, oim, lrs, np, tile_log);
}
// batches of small transforms (N = 2..2048) through the one-pass kernel's body (row_fft.hpp); modes as above
static int emu_small(int is_f64, const void *in_re, const void *in_im, unsigned in_mode, void *out_re, void *out_im,
                     unsigned out_mode, unsigned log_n, size_t batch, size_t in_dist, size_t out_dist, double scale,
                     unsigned real_mode = 0) {
    phast::RowArgs r{};
    const unsigned rbits = phast::tw3_bits_for(log_n + 1);
    const std::vector<phast::cx_t<double>> r64 = phast::host_tw3<double>(log_n + 1, rbits);
    const std::vector<phast::cx_t<float>> r32 = phast::host_tw3<float>(log_n + 1, rbits);
    r.real_mode = real_mode;
    r.rtw_bits = rbits;
    r.rtw3 = is_f64 ? (const void *)r64.data() : (const void *)r32.data();
    r.in_re = in_re;
    r.in_im = in_im;
    r.out_re = out_re;
    r.out_im = out_im;
    r.in_dist = in_dist;
    r.out_dist = out_dist;
    r.batch = batch;
    r.in_interleaved = in_mode;
    r.out_interleaved = out_mode;
    r.scale = scale;
    const unsigned lc = phast::row_tile_cols_log(log_n);
    r.tiles_total = (unsigned)((batch + ((1ull << lc) - 1)) >> lc);
    const std::vector<phast::cx_t<double>> t64 = phast::host_twr<double>(1u << log_n);
    const std::vector<phast::cx_t<float>> t32 = phast::host_twr<float>(1u << log_n);
    r.twr = is_f64 ? (const void *)t64.data() : (const void *)t32.data();
    if (log_n == 5 && real_mode) {  // as launch_small_fft: the 64-point real transforms' core runs 16 points per thread
        if (is_f64) phast::emulate_row_fft<double, 5, 7, phast::kRealRow5LP>(r);
        else phast::emulate_row_fft<float, 5, 7, phast::kRealRow5LP>(r);
        return 0;
    }
#define PHAST_ROW_EMU(LR_, LC_, LP_)                                      \
    if (log_n == LR_) {                                                   \
        if (is_f64) phast::emulate_row_fft<double, LR_, LC_, LP_>(r);     \
        else phast::emulate_row_fft<float, LR_, LC_, LP_>(r);             \
        return 0;                                                         \
    }
    PHAST_ROW_SHAPES(PHAST_ROW_EMU)
#undef PHAST_ROW_EMU
    return 1;
}
int phast_emu_small_fft(int is_f64, const void *in_re, const void *in_im, unsigned in_mode, void *out_re, void *out_im,
                        unsigned out_mode, unsigned log_n, size_t batch, size_t in_dist, size_t out_dist, double scale) {
    return emu_small(is_f64, in_re, in_im, in_mode, out_re, out_im, out_mode, log_n, batch, in_dist, out_dist, scale);
}
// real transforms of n = 2 * 2^log_half points through the fused kernel: mode 1 = R2C (in: n reals per transform,
// in_dist apart; out: planar n/2 + 1), mode 2 = C2R (in: planar n/2 + 1; out: n reals, scaled by 1/(n/2))
int phast_emu_small_real(int is_f64, unsigned mode, const void *in_a, const void *in_b, void *out_a, void *out_b,
                         unsigned log_half, size_t batch, size_t in_dist, size_t out_dist) {
    if (mode == 1)
        return emu_small(is_f64, in_a, nullptr, 1, out_a, out_b, 0, log_half, batch, in_dist / 2, out_dist, 1.0, 1);
    return emu_small(is_f64, in_a, in_b, 0, out_a, nullptr, 2, log_half, batch, in_dist, out_dist / 2,
                     1.0 / (double)((size_t)1 << log_half), 2);
}
// LDS audit of the small-transform kernel: park / pick and every exchange in bounds, permutations, conflict-free
int phast_emu_audit_small(int is_f64, unsigned log_n, int *max_read_ways, int *max_write_ways) {
#define PHAST_ROW_AUD(LR_, LC_, LP_)                                                                          \
    if (log_n == LR_)                                                                                         \
        return is_f64 ? phast::audit_small_shape<double, LR_, LC_, LP_>(max_read_ways, max_write_ways)        \
                      : phast::audit_small_shape<float, LR_, LC_, LP_>(max_read_ways, max_write_ways);
    PHAST_ROW_SHAPES(PHAST_ROW_AUD)
#undef PHAST_ROW_AUD
    return -1;
}

#endif
#if !defined(EMU_PART) || EMU_PART == 1
// strided batch (column FFTs of a row-major [2^log_n][2^s] array, first 2^sb columns), in place, forward
int phast_emu_fft_strided_tw_f64(double *re, double *im, unsigned log_n, unsigned s, unsigned sb, unsigned grid_log_n,
                                 unsigned col0);
int phast_emu_fft_strided_f64(double *re, double *im, unsigned log_n, unsigned s, unsigned sb) {
    return phast_emu_fft_strided_tw_f64(re, im, log_n, s, sb, 0, 0);
}
// ... with the input twiddle W_{2^grid_log_n}^(j (col0 + c)) fused into the first pass (grid_log_n = 0: none)
int phast_emu_fft_strided_tw_f64(double *re, double *im, unsigned log_n, unsigned s, unsigned sb, unsigned grid_log_n,
                                 unsigned col0) {
    using namespace phast;
    std::vector<PassGeom> ps;
    if (!make_strided_passes(log_n, s, sb, sizeof(double), ps, grid_log_n)) return 1;
    const size_t total = (size_t)1 << (log_n + s);
    std::vector<double> t_re(total), t_im(total);
    for (size_t i = 0; i < ps.size(); ++i) {
        const PassGeom &p = ps[i];
        std::vector<cx_t<double>> twr = host_twr<double>(1u << p.lr), tw3 = host_tw3<double>(p.log_mod(), p.tw_bits);
        TileArgs ta{};
        const size_t np = ps.size();
        const bool from_x = (i % 2) == 0 || (i + 1 == np && np == 3);
        const bool to_x = (i % 2) == 1 || i + 1 == np;
        ta.in_re = from_x ? re : t_re.data();
        ta.in_im = from_x ? im : t_im.data();
        ta.out_re = to_x ? re : t_re.data();
        ta.out_im = to_x ? im : t_im.data();
        ta.scale = 1.0;
        ta.tw3 = tw3.data();
        ta.twr = twr.data();
        geom_to_args(p, log_n, 1, ta);
        ta.grid_col0 = col0;
        if (!emu_pass<double>(p, ta)) return 2;
    }
    return 0;
}

// the default plan of the library for (type, log_n): fills lrs[3], returns the number of passes
int phast_emu_default_plan(int is_f64, int latency, unsigned log_n, unsigned *lrs, unsigned *tile_log,
                           unsigned *points_log) {
    std::vector<unsigned> v, tl;
    if (is_f64) phast::heuristic_plan<double>(log_n, latency != 0, v, tl, *points_log);
    else phast::heuristic_plan<float>(log_n, latency != 0, v, tl, *points_log);
    *tile_log = tl[0];
    for (size_t i = 0; i < v.size(); ++i) lrs[i] = v[i];
    return (int)v.size();
}

// Every entry of the plan tables (plan.hpp: single_plan, real_plan, real_batch_plan) must be a plan that EXISTS: rows adding
// up to the length, every pass a shape that is instantiated, the geometry accepted by make_passes.  A typo in a table would
// otherwise only show on the GPU as a planner that silently keeps its heuristic plan (set_plan -> INVALID_ARG is not an error
// for an optional plan).  Returns the number of bad entries, *n_entries = entries seen.
}  // extern "C"
template <typename T> static int check_tables(int *n_entries) {
    using namespace phast;
    int bad = 0;
    for (unsigned L = kTwinMinLog; L <= 30; ++L)
        for (int which = 0; which < 5; ++which) {
            std::vector<unsigned> lrs, tls;
            unsigned lp = 4;
            const bool have = which == 0   ? single_plan<T>(L, lrs, tls, lp)
                              : which <= 2 ? real_plan<T>(L, which == 2, lrs, tls, lp)
                                           : real_batch_plan<T>(L, which == 4, lrs, tls, lp);
            if (!have) continue;
            ++*n_entries;
            unsigned sum = 0;
            for (unsigned r : lrs) sum += r;
            std::vector<PassGeom> geo;
            const bool ok = sum == L && lrs.size() == tls.size() && make_passes(L, lrs, tls, geo, lp & ~kFuseBelow, sizeof(T));
            if (!ok) {
                std::fprintf(stderr, "plan table %d, %zu-byte elements, L = %u: not a plan\n", which, sizeof(T), L);
                ++bad;
            }
        }
    return bad;
}
extern "C" {
int phast_emu_check_plan_tables(int *n_entries) {
    *n_entries = 0;
    return check_tables<double>(n_entries) + check_tables<float>(n_entries);
}
// plan.hpp: digests_agree -- the result check of a tuning run
int phast_emu_digests_agree(const double *a, const double *c, size_t batch, size_t n, size_t elem_bytes) {
    return phast::digests_agree(a, c, batch, n, elem_bytes) ? 1 : 0;
}
// The candidate set of a tuning run (plan.hpp: enumerate_plans, tune_tile_range): every entry must be a plan (make_passes), must
// survive the round trip through its text form (the wisdom format), and -- returned -- how many there are; *has_spec = whether
// `spec` (e.g. the hand-ranked table entry of that length) is among them.
int phast_emu_enumerate_plans(unsigned L, size_t elem_bytes, size_t batch, const char *spec, int *has_spec, int *bad) {
    using namespace phast;
    unsigned tl_lo, tl_hi;
    tune_tile_range(batch << L, elem_bytes, tl_lo, tl_hi);
    std::vector<PlanSpec> specs;
    enumerate_plans(L, elem_bytes, batch, tl_lo, tl_hi, specs);
    PlanSpec want;
    const bool have_want = spec && spec_from_string(spec, want);
    *has_spec = 0;
    *bad = 0;
    std::vector<PassGeom> geo;
    for (const PlanSpec &s : specs) {
        PlanSpec back;
        if (!spec_from_string(spec_to_string(s).c_str(), back) || !(back == s)) ++*bad;
        if (!make_passes(L, s.lrs(), s.tls(), geo, s.lp, elem_bytes)) ++*bad;
        if (have_want && s == want) *has_spec = 1;
    }
    return (int)specs.size();
}
#endif
}
