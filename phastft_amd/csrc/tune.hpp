// tune.hpp -- PlannerMode::Tune (planner.rs:18-32: "benchmarks both paths at plan time and picks whichever is faster, at the
// cost of additional planning time").  Here "both paths" are the plans of plan.hpp: a tuning run enumerates every plan of the
// length that exists as kernels (enumerate_plans: factorisation x tile size per pass x points per thread x wave / quad
// tiles; for R2C also the untangle fused or as a sweep), times each on THIS device for one call kind and batch size, and
// installs the winner for that (kind, batch bucket) -- what tools/sweep_*.py + a hand-edited table in plan.hpp did until
// round 5.  How it measures (the protocol the round-4 sweeps converged on, tools/confirm_single.py):
//   * a COLD ring of distinct buffer sets (>= 1.25 GiB where the device has it): a buffer timed in place sits in the 256 MiB
//     Infinity Cache from 2^21 to 2^24 points and ranks plans for a situation a caller's data is never in;
//   * screening: every candidate once, over a slice of the ring that moves on with every candidate, after one untimed call;
//   * finals: the best few and the static rule's plan INTERLEAVED (A B C A B C ...) over the whole ring, median of the rounds --
//     boxes drift by a few per cent within seconds, so only alternating measurements compare;
//   * the winner's RESULT is compared with the static rule's plan's before anything is adopted (digests of one set of inputs,
//     plan.hpp: digests_agree): the candidates are compositions no parity test enumerates;
//   * a measured plan is adopted only if it beats the static rule's by more than 3 % (PHAST_TUNE_MIN_GAIN): below that the
//     ranking is noise (where a buffer landed is worth +-5 % at 2^25 points and beyond, profiles/r04_placement_probe.log).
// Eager launches on a private stream, never graphs: a workspace that was captured belongs to its graph for good.
// The result goes to the wisdom store (wisdom.hpp) so that the next planner of this length -- or process, PHAST_WISDOM -- starts
// with it.
#pragma once

#include <chrono>
#include <cmath>

#include "planner_r2c.hpp"

namespace phast {

struct TuneKnobs {
    static long env_long(const char *name, long dflt) {
        const char *e = std::getenv(name);
        return (e && *e) ? std::atol(e) : dflt;
    }
    static size_t max_candidates() { return (size_t)env_long("PHAST_TUNE_MAX_CANDIDATES", 1000); }
    static int finals() { return (int)env_long("PHAST_TUNE_FINALS", 4); }
    static int rounds() { return (int)env_long("PHAST_TUNE_ROUNDS", 5); }
    static double min_gain() { return (double)env_long("PHAST_TUNE_MIN_GAIN_PERMILLE", 30) / 1000.0; }
    static double budget_s() { return (double)env_long("PHAST_TUNE_BUDGET_MS", 20000) / 1000.0; }
    static size_t ring_bytes() { return (size_t)env_long("PHAST_TUNE_RING_MB", 1280) << 20; }
};

// the cold ring of a tuning run: `ring` sets of `set_bytes`
struct TuneRing {
    char *base = nullptr;
    size_t set_bytes = 0;
    int ring = 0;
    ~TuneRing() {
        if (base) hipFree(base);
    }
    int alloc(size_t set_bytes_) {
        set_bytes = (set_bytes_ + 255) & ~(size_t)255;
        size_t want = (TuneKnobs::ring_bytes() + set_bytes - 1) / set_bytes;
        ring = (int)(want < 3 ? 3 : want > 64 ? 64 : want);
        for (;;) {
            hipError_t e = hipMalloc((void **)&base, (size_t)ring * set_bytes);
            if (e == hipSuccess) return PHAST_OK;
            (void)hipGetLastError();
            base = nullptr;
            if (e != hipErrorOutOfMemory || ring <= 2) return hip_fail(e, "hipMalloc(tuning ring)");
            ring = ring > 4 ? ring / 2 : ring - 1;
        }
    }
    char *set(int i) const { return base + (size_t)i * set_bytes; }
};

// run(L, choice, set) enqueues one call of the kind being tuned on L.stream; refill(stream) rewrites the ring with inputs (in
// place transforms grow their data by ~sqrt(N) per call: `grows`); digest(set, probe, d_out, stream) enqueues the per-transform digests
// {sum re, sum im, energy, one probed value} of the call's OUTPUT in `set` (4 doubles per transform).
template <typename T>
template <typename Run, typename Refill, typename Digest>
int Planner<T>::tune_core(int kind, size_t batch, unsigned wisdom_log_n, int ring, bool grows, Run &&run, Refill &&refill, Digest &&digest,
                          TuneReport *rep) {
    using clock = std::chrono::steady_clock;
    const auto t_start = clock::now();
    auto elapsed_s = [&] { return std::chrono::duration<double>(clock::now() - t_start).count(); };
    std::lock_guard<std::mutex> one_at_a_time(tune_mu);
    PHAST_ON_DEVICE(device);
    const unsigned bucket = batch_bucket(batch);

    // ---- 1. the candidates, built (tables come from the planner's cache: a candidate costs two occupancy queries) ----
    struct Cand {
        PlanSpec spec;
        std::vector<PassDesc> passes;
        bool fuse = false, heuristic = false;
        float us = 1e30f;
        std::vector<float> rounds;
    };
    std::vector<Cand> cands;
    size_t need_max = sstride();
    {
        unsigned tl_lo, tl_hi;
        tune_tile_range(batch * n, sizeof(T), tl_lo, tl_hi);
        std::vector<PlanSpec> specs;
        enumerate_plans(log_n, sizeof(T), batch, tl_lo, tl_hi, specs);
        if (specs.size() > TuneKnobs::max_candidates()) specs.resize(TuneKnobs::max_candidates());  // (sorted by the cost model)
        for (const PlanSpec &sp : specs) {
            Cand c;
            c.spec = sp;
            size_t need = 0;
            int rc = build_plan(sp, c.passes, &need);
            if (rc == PHAST_ERR_INVALID_ARG) continue;  // a tile that does not fit the LDS with this length's tables
            if (rc) return rc;
            need_max = std::max(need_max, need);
            if (kind == kR2C) {
                // the untangle in the last pass, or as a sweep of its own: both where the static rule's threshold says it could
                // go either way (below 2^23 complex points in flight), else the fused form wherever it exists
                const bool fusable = c.passes.back().r2c_blocks > 0 && r2c_fuse_enabled();
                if (fusable && batch * n < ((size_t)1 << 23)) {
                    cands.push_back(c);  // (the unfused twin)
                }
                c.fuse = fusable;
            }
            cands.push_back(std::move(c));
        }
    }
    // One scratch pitch for every candidate, before a workspace is cut for it -- for the duration of the run only.  The pitch
    // goes back to what the installed plans need when the run ends (the static plans, the tuned ones and an adopted winner):
    // left at the maximum over up to a thousand candidates that were never adopted, every later scratch was larger than any
    // plan in force needs and every existing workspace had to be re-cut.  A tuning run still re-cuts the workspace
    // it used: like set_plan, tune() is followed by one eager call before a capture (phastft_hip.h).
    struct RestorePitch {
        Planner<T> *p;
        size_t before;
        ~RestorePitch() {  // (what the plans installed NOW need: a set_plan that slipped in after the lease went is counted)
            std::unique_lock<std::shared_mutex> plans(p->plan_mu);
            const size_t want = std::max(before, p->installed_pitch_locked());
            p->scratch_stride = want > p->n ? want : 0;
        }
    } restore_pitch{this, scratch_stride};
    {
        std::unique_lock<std::shared_mutex> plans(plan_mu);
        if (need_max > sstride()) scratch_stride = need_max;
    }

    // ---- 2. a private stream, a workspace for the whole run ----
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    PHAST_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Cleanup {
        hipStream_t &st;
        hipEvent_t &e0, &e1;
        ~Cleanup() {
            if (st) (void)hipStreamSynchronize(st);
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            if (st) hipStreamDestroy(st);
        }
    } cleanup{st, e0, e1};
    PHAST_HIP(hipEventCreate(&e0));
    PHAST_HIP(hipEventCreate(&e1));

    Cand heur;
    heur.heuristic = true;
    int best = -1;
    float med_heur = 0, med_best = 0;
    {
        Lease L;  // (holds the plans shared: set_plan waits for the run; the static rule's plan stays where `heur` points)
        int rc = check_out(L, st);
        if (rc) return rc;
        const Choice h_choice = choose(kind, batch, batch, false);
        std::vector<int> uses((size_t)ring, 0);
        const int growth_limit = (int)((sizeof(T) == 8 ? 900 : 100) / std::max(1u, (log_n + 1) / 2));
        int cursor = 0;
        // one timed run of `count` calls on consecutive sets from `cursor` on (after `warm` untimed calls); *us = per call
        auto timed = [&](const Choice &c, int warm, int count, float *us) -> int {
            if (grows) {
                bool hot = false;
                for (int i = 0; i < warm + count; ++i) hot = hot || uses[(size_t)((cursor + i) % ring)] + 1 > growth_limit;
                if (hot) {
                    int rf = refill(st);
                    if (rf) return rf;
                    std::fill(uses.begin(), uses.end(), 0);
                }
            }
            for (int i = 0; i < warm; ++i, cursor = (cursor + 1) % ring) {
                int r = run(L, c, cursor);
                if (r) return r;
                ++uses[(size_t)cursor];
            }
            PHAST_HIP(hipEventRecord(e0, st));
            for (int i = 0; i < count; ++i, cursor = (cursor + 1) % ring) {
                int r = run(L, c, cursor);
                if (r) return r;
                ++uses[(size_t)cursor];
            }
            PHAST_HIP(hipEventRecord(e1, st));
            PHAST_HIP(hipEventSynchronize(e1));
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, e0, e1));
            *us = 1e3f * ms / (float)count;
            return PHAST_OK;
        };
        auto choice_of = [&](const Cand &c) {
            if (c.heuristic) return h_choice;
            Choice ch;
            ch.passes = &c.passes;
            ch.r2c_fuse = c.fuse;
            return ch;
        };
        rc = refill(st);
        if (rc) return rc;
        // the static rule's plan first: the yardstick, and what sizes the screening slices (>= ~150 us of work per candidate)
        rc = timed(h_choice, 1, ring, &heur.us);
        if (rc) return rc;
        int slice = (int)(150.0f / std::max(heur.us, 1.0f)) + 1;
        slice = slice < 2 ? 2 : slice > ring ? ring : slice;
        // ---- 3. screening ----
        auto same_plan = [](const std::vector<PassDesc> &a, const std::vector<PassDesc> &b) {
            if (a.size() != b.size()) return false;
            for (size_t i = 0; i < a.size(); ++i)
                if (a[i].lr != b[i].lr || a[i].lc != b[i].lc || a[i].lp != b[i].lp || a[i].wave != b[i].wave || a[i].quad != b[i].quad ||
                    a[i].scratch_dist != b[i].scratch_dist)
                    return false;
            return true;
        };
        size_t screened = 0;
        for (Cand &c : cands) {
            if (elapsed_s() > TuneKnobs::budget_s()) break;
            // the static rule's own plan is the yardstick, not a candidate (two timings of one plan differ by a few per cent of
            // launch jitter at the 8 us end: round 5's first built-in wisdom "adopted" 6,6@10,10:p8 over itself at 2^12)
            if (same_plan(c.passes, *h_choice.passes) && (kind != kR2C || c.fuse == (h_choice.r2c_fuse && c.passes.back().r2c_blocks > 0))) continue;
            int r = timed(choice_of(c), 1, slice, &c.us);
            if (r) {  // a candidate the device refuses is not a candidate; anything else would have failed for the yardstick too
                (void)hipGetLastError();
                (void)hipStreamSynchronize(st);
                c.us = 1e30f;
            }
            ++screened;
        }
        // ---- 4. finals: the best few and the yardstick, interleaved over the whole ring ----
        std::vector<Cand *> fin;
        {
            std::vector<Cand *> order;
            for (Cand &c : cands)
                if (c.us < 1e29f) order.push_back(&c);
            std::stable_sort(order.begin(), order.end(), [](const Cand *a, const Cand *b) { return a->us < b->us; });
            for (size_t i = 0; i < order.size() && (int)i < TuneKnobs::finals(); ++i) fin.push_back(order[i]);
        }
        fin.push_back(&heur);
        // (short calls get more rounds: the medians of 5 eager runs of an 8 us call scatter by ~5 %, those of 11 by ~2 %)
        const int rounds = heur.us < 100.0f ? 2 * TuneKnobs::rounds() + 1 : TuneKnobs::rounds();
        for (int r = 0; r < rounds; ++r)
            for (Cand *c : fin) {
                float us = 0;
                int e = timed(choice_of(*c), r == 0 ? 1 : 0, ring, &us);
                if (e) return e;
                c->rounds.push_back(us);
            }
        auto median = [](std::vector<float> v) {
            std::sort(v.begin(), v.end());
            return v.empty() ? 1e30f : v[v.size() / 2];
        };
        med_heur = median(heur.rounds);
        med_best = 1e30f;
        for (size_t i = 0; i + 1 < fin.size(); ++i) {
            const float m = median(fin[i]->rounds);
            if (m < med_best) {
                med_best = m;
                best = (int)(fin[i] - cands.data());
            }
        }
        if (rep) rep->candidates = (unsigned)screened;
        // ---- 4b. the winner must compute what the static rule's plan computes.  Every candidate is built from kernels that are
        // parity-tested on their own, but the tuner composes them in combinations no test enumerates: before a plan is adopted
        // its output for one set of inputs is compared with the static rule's, transform by transform, through the digests
        // (energy, sums, a probed bin) -- far coarser than the parity gates, and enough to stop a plan whose geometry is wrong.
        if (best >= 0 && med_best < (float)(1.0 - TuneKnobs::min_gain()) * med_heur) {
            DevBuf d_dig;  // [static rule | winner] x [two probed bins] x batch x 4
            rc = d_dig.alloc(4 * batch * 4 * sizeof(double));
            if (rc) return rc;
            double *d_all = reinterpret_cast<double *>(d_dig.p);
            const Choice w_choice = choice_of(cands[(size_t)best]);
            const size_t probes[2] = {1, n / 2 + 1};
            for (int pass = 0; pass < 2; ++pass) {
                rc = refill(st);
                if (rc == PHAST_OK) rc = run(L, pass == 0 ? h_choice : w_choice, 0);
                for (int k = 0; k < 2 && rc == PHAST_OK; ++k) rc = digest(0, probes[k], d_all + (size_t)(2 * pass + k) * batch * 4, st);
                if (rc) return rc;
            }
            std::vector<double> h(4 * batch * 4);
            PHAST_HIP(hipMemcpyAsync(h.data(), d_all, h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            PHAST_HIP(hipStreamSynchronize(st));
            const bool same = digests_agree(&h[0], &h[8 * batch], batch, n, sizeof(T)) &&
                              digests_agree(&h[4 * batch], &h[12 * batch], batch, n, sizeof(T));
            if (!same) {
                best = -1;  // not adopted, and said so: this is a bug in plan.hpp's geometry, not a slow plan
                if (rep) rep->rejected = 1;
            }
        }
        PHAST_HIP(hipStreamSynchronize(st));
    }  // the lease goes: the plans may change now

    // ---- 5. adopt, remember ----
    const bool adopted = best >= 0 && med_best < (float)(1.0 - TuneKnobs::min_gain()) * med_heur;
    WisdomEntry we;
    we.heuristic = !adopted;
    we.us = adopted ? med_best : med_heur;
    we.us_heur = med_heur;
    we.cus = cus_of(device);
    we.arch = arch_of(device);
    we.lib = kWisdomLib;
    we.layer = 3;
    if (adopted) {
        Cand &w = cands[(size_t)best];
        we.spec = w.spec;
        we.fuse = w.fuse;
        int rc = install_built(kind, bucket, w.spec, w.fuse, std::move(w.passes), med_best, med_heur);
        if (rc) return rc;
    } else {
        remove_tuned(kind, bucket);  // (an imported plan that this device does not confirm)
    }
    WisdomStore::instance().record(sizeof(T), kind, wisdom_log_n, bucket, we);
    if (rep) {
        rep->adopted = adopted ? 1 : 0;
        rep->us_heuristic = med_heur;
        rep->us_best = adopted ? med_best : med_heur;
        rep->plan = adopted ? spec_to_string(we.spec) + (we.fuse ? " fused" : "") : std::string(rep->rejected ? "heuristic (the fastest plan FAILED the result check)" : "heuristic");
        rep->seconds = elapsed_s();
    }
    return PHAST_OK;
}

// PlannerDit*: C2C on planar arrays (kC2C) or on Complex<T> pairs (kC2CI), `batch` transforms per call
template <typename T> int Planner<T>::tune(int kind, size_t batch, TuneReport *rep) {
    if ((kind != kC2C && kind != kC2CI) || batch == 0) return PHAST_ERR_INVALID_ARG;
    if (rep) *rep = TuneReport();
    if (passes.empty()) {  // whole transforms on chip: one kernel, nothing to choose -- except in the 8192-point twin
        if (route_small(batch) != this) return twin->tune(kind, batch, rep);
        if (rep) rep->plan = "one pass";
        return PHAST_OK;
    }
    PHAST_ON_DEVICE(device);
    TuneRing ring;
    int rc = ring.alloc(2 * batch * n * sizeof(T));
    if (rc) return rc;
    const size_t plane = batch * n;
    auto run = [&](const Lease &L, const Choice &c, int set) {
        T *re = reinterpret_cast<T *>(ring.set(set)), *im = re + plane;
        return kind == kC2C ? exec_in(L, re, im, n, 0, re, im, n, 0, batch, 1.0, nullptr, nullptr, nullptr, nullptr, &c)
                            : exec_in(L, re, nullptr, n, 1, re, nullptr, n, 1, batch, 1.0, nullptr, nullptr, nullptr, nullptr, &c);
    };
    auto refill = [&](hipStream_t st) {
        const size_t total = (size_t)ring.ring * ring.set_bytes / sizeof(T);
        PHAST_HIP(launch_fill<T>(reinterpret_cast<T *>(ring.base), nullptr, total, 1, total, 0xCAFEull, 0, st));
        return (int)PHAST_OK;
    };
    auto digest = [&](int set, size_t probe, double *d_out, hipStream_t st) {
        T *re = reinterpret_cast<T *>(ring.set(set)), *im = re + plane;
        if (kind == kC2C) PHAST_HIP(launch_digest<T>(re, im, n, batch, n, probe, d_out, st));
        else PHAST_HIP(launch_digest<T>(re, re + 1, 2 * n - 1, batch, 2 * n, 2 * probe, d_out, st));  // pairs: the pair array as two shifted planes
        return (int)PHAST_OK;
    };
    return tune_core(kind, batch, log_n, ring.ring, true, run, refill, digest, rep);
}

// PlannerR2c*: r2c_fft (kR2C) or c2r_fft (kC2R), `batch` real transforms of n points per call.  The plans are the inner
// N/2-point planner's; the wisdom is keyed by the real length.
template <typename T> int PlannerR2c<T>::tune(int kind, size_t batch, typename Planner<T>::TuneReport *rep) {
    if ((kind != kR2C && kind != kC2R) || batch == 0) return PHAST_ERR_INVALID_ARG;
    if (rep) *rep = typename Planner<T>::TuneReport();
    if (dit.passes.empty()) {
        if (route_small(kind == kC2R, batch) != this) return twin->tune(kind, batch, rep);
        if (rep) rep->plan = "one pass";
        return PHAST_OK;
    }
    PHAST_ON_DEVICE(dit.device);
    const size_t half = n / 2, hp = half + 1;
    // the three arrays of a set start on 256-byte boundaries, as a caller's separately allocated slices do (the half-spectrum
    // has an odd length: packed back to back its imaginary plane would start 4 or 8 bytes off a 16-byte boundary)
    auto pad = [](size_t elems) { return (elems + 63) & ~(size_t)63; };
    const size_t x_elems = pad(batch * n), s_elems = pad(batch * hp);
    TuneRing ring;
    int rc = ring.alloc((x_elems + 2 * s_elems) * sizeof(T));
    if (rc) return rc;
    auto run = [&](const Lease &L, const typename Planner<T>::Choice &c, int set) {
        T *x = reinterpret_cast<T *>(ring.set(set)), *sr = x + x_elems, *si = sr + s_elems;
        return kind == kR2C ? r2c_in(L, x, sr, si, batch, n, hp, nullptr, &c) : c2r_in(L, sr, si, x, batch, hp, n, nullptr, &c);
    };
    auto refill = [&](hipStream_t st) {
        const size_t total = (size_t)ring.ring * ring.set_bytes / sizeof(T);
        PHAST_HIP(launch_fill<T>(reinterpret_cast<T *>(ring.base), nullptr, total, 1, total, 0xCAFEull, 0, st));
        return (int)PHAST_OK;
    };
    auto digest = [&](int set, size_t probe, double *d_out, hipStream_t st) {
        T *x = reinterpret_cast<T *>(ring.set(set)), *sr = x + x_elems, *si = sr + s_elems;
        if (kind == kR2C) PHAST_HIP(launch_digest<T>(sr, si, hp, batch, hp, probe, d_out, st));
        else PHAST_HIP(launch_digest<T>(x, x, n, batch, n, 2 * probe + 1, d_out, st));
        return (int)PHAST_OK;
    };
    return dit.tune_core(kind, batch, ilog2(n), ring.ring, false, run, refill, digest, rep);
}

}  // namespace phast
