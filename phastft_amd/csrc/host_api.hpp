// host_api.hpp -- the host-slice entry points (the reference's calling convention): staging, the pinned mirror, the planner cache.
#pragma once

#include "entry.hpp"

namespace phast {


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
    pl = pl->route_small();  // ONE transform of 8192 points: the plan (and the bits) of the _dev call
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
    pl = pl->route_small();
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
    pl = pl->route_small(false);
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
    pl = pl->route_small(true);
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

// complex_nums.rs:11-17: `input.chunks_exact(2)` -- an odd last element is dropped; the reference returns two new Vecs of
// len / 2, here the caller brings them (their lengths are checked: a C caller cannot receive a Vec)
template <typename T> static int deinterleave_host(const T *in, size_t len, T *a, size_t a_len, T *b, size_t b_len) {
    const size_t pairs = len / 2;
    if (a_len != pairs || b_len != pairs) return PHAST_ERR_LEN_MISMATCH;
    if (pairs == 0) return PHAST_OK;
    if (!in || !a || !b) return PHAST_ERR_INVALID_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf buf;
    rc = buf.alloc(4 * pairs * sizeof(T));
    if (rc) return rc;
    T *d_in = reinterpret_cast<T *>(buf.p), *d_a = d_in + 2 * pairs, *d_b = d_a + pairs;
    PHAST_HIP(hipMemcpy(d_in, in, 2 * pairs * sizeof(T), hipMemcpyHostToDevice));
    PHAST_HIP(launch_deinterleave<T>(d_in, d_a, d_b, pairs, nullptr));
    PHAST_HIP(hipStreamSynchronize(nullptr));
    PHAST_HIP(hipMemcpy(a, d_a, pairs * sizeof(T), hipMemcpyDeviceToHost));
    PHAST_HIP(hipMemcpy(b, d_b, pairs * sizeof(T), hipMemcpyDeviceToHost));
    return PHAST_OK;
}

// complex_nums.rs:47-56: `assert_eq!(reals.len(), imags.len())`
template <typename T> static int combine_host(const T *re, size_t re_len, const T *im, size_t im_len, T *out, size_t out_len) {
    if (re_len != im_len || out_len != 2 * re_len) return PHAST_ERR_LEN_MISMATCH;
    if (re_len == 0) return PHAST_OK;
    if (!re || !im || !out) return PHAST_ERR_INVALID_ARG;
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf buf;
    rc = buf.alloc(4 * re_len * sizeof(T));
    if (rc) return rc;
    T *d_re = reinterpret_cast<T *>(buf.p), *d_im = d_re + re_len, *d_out = d_im + re_len;
    PHAST_HIP(hipMemcpy(d_re, re, re_len * sizeof(T), hipMemcpyHostToDevice));
    PHAST_HIP(hipMemcpy(d_im, im, re_len * sizeof(T), hipMemcpyHostToDevice));
    PHAST_HIP(launch_combine<T>(d_re, d_im, d_out, re_len, nullptr));
    PHAST_HIP(hipStreamSynchronize(nullptr));
    PHAST_HIP(hipMemcpy(out, d_out, 2 * re_len * sizeof(T), hipMemcpyDeviceToHost));
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
    if (n_passes == 0) {  // the restore form: the library's own plans, tuned / wisdom plans in force again
        const int rc = p->default_plans();
        if (rc == PHAST_OK) p->set_forced(false);
        return rc;
    } else {
        if (!log_rows || !tile_logs) return PHAST_ERR_INVALID_ARG;
        lrs.assign(log_rows, log_rows + n_passes);
        tls.assign(tile_logs, tile_logs + n_passes);
    }
    if ((points_log & 0xfu) < 3 || (points_log & 0xfu) > 5 || (points_log & ~0x1fu)) return PHAST_ERR_INVALID_ARG;
    const int rc = p->set_plan(lrs, tls, 0, points_log);
    if (rc == PHAST_OK) p->set_forced(true);  // the caller's plan runs for every call kind and batch, whatever wisdom says
    return rc;
}

}  // namespace phast
