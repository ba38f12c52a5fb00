// entry.hpp -- the device-pointer entry points: argument checks as the reference asserts, then the planner.
#pragma once

#include "planner_r2c.hpp"

namespace phast {

// ------------------------------------------------------------------------------------------------
// helpers shared by the C entry points
// ------------------------------------------------------------------------------------------------
template <typename P> static int planner_new(size_t n, P **out) {  // P = Planner<T> or the C-ABI struct over it
    if (!out) return PHAST_ERR_INVALID_ARG;
    *out = nullptr;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;  // planner.rs:66
    auto *p = new (std::nothrow) P();
    if (!p) return PHAST_ERR_ALLOC;
    int rc = p->init(n);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return PHAST_OK;
}

template <typename P> static int r2c_planner_new(size_t n, P **out) {
    if (!out) return PHAST_ERR_INVALID_ARG;
    *out = nullptr;
    if (n < 4 || !is_pow2(n)) return PHAST_ERR_R2C_N;  // planner.rs:195
    auto *p = new (std::nothrow) P();
    if (!p) return PHAST_ERR_ALLOC;
    int rc = p->init(n);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return PHAST_OK;
}

// algorithms/dit.rs:276-332: asserts, swap trick, transform, 1/N scale -- on device pointers
template <typename T>
static int fft_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, int direction, const Planner<T> *pl,
                   hipStream_t s) {
    if (!pl || !d_re || !d_im) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (batch > 1 && dist < n) return PHAST_ERR_INVALID_ARG;
    if (direction == PHAST_REVERSE) return pl->exec(d_im, d_re, dist, 0, d_im, d_re, dist, 0, batch, 1.0 / (double)n, s);
    return pl->exec(d_re, d_im, dist, 0, d_re, d_im, dist, 0, batch, 1.0, s);
}

// `count` independent transforms at arbitrary addresses, each run exactly as a single-transform call (the plan for ONE
// transform), enqueued back to back by one host call: the host-side cost per transform drops to the kernel launches
// themselves (a Python / FFI caller pays its call overhead once).
template <typename T>
static int fft_dev_many(T *const *d_re, T *const *d_im, size_t count, size_t n, int direction, const Planner<T> *pl,
                        hipStream_t s) {
    if (!pl || (!d_re && count) || (!d_im && count)) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    for (size_t i = 0; i < count; ++i)
        if (!d_re[i] || !d_im[i]) return PHAST_ERR_INVALID_ARG;
    if (count == 0) return PHAST_OK;
    pl = pl->route_small();  // 4096 / 8192 points: the plan of a single-transform call (Planner::twin)
    PHAST_ON_DEVICE(pl->device);
    typename Planner<T>::Lease L;  // one workspace for the whole list: the transforms follow each other on the stream
    if (!pl->passes.empty()) {
        int rc = pl->check_out(L, s);
        if (rc) return rc;
    } else {
        L.stream = s;
    }
    for (size_t i = 0; i < count; ++i) {
        int rc = direction == PHAST_REVERSE
                     ? pl->exec_in(L, d_im[i], d_re[i], n, 0, d_im[i], d_re[i], n, 0, 1, 1.0 / (double)n)
                     : pl->exec_in(L, d_re[i], d_im[i], n, 0, d_re[i], d_im[i], n, 0, 1, 1.0);
        if (rc) return rc;
    }
    return PHAST_OK;
}

// Strided batches on device pointers: transform b occupies elements b*dist + j*stride, j < n.
//   stride == 1: contiguous transforms `dist` apart (the plain batched path above);
//   dist == 1:   "column FFTs" of a row-major [n][stride] array -- stride and batch powers of two, batch <= stride,
//                n >= 64 (what a four-step split and any multi-dimensional transform need; no reference counterpart,
//                SURVEY.md section 8b suggested the signature).
template <typename T>
static int fft_strided_dev(T *d_re, T *d_im, size_t n, size_t batch, size_t dist, size_t stride, int direction,
                           const Planner<T> *pl, hipStream_t s, size_t tw_n = 0, size_t tw_col0 = 0) {
    if (tw_n && (stride == 1 || !is_pow2(tw_n) || tw_n < n || ilog2(tw_n) > 32 || tw_col0 + batch > tw_n))
        return PHAST_ERR_INVALID_ARG;
    if (stride == 1) return fft_dev<T>(d_re, d_im, n, batch, dist, direction, pl, s);
    if (!pl || !d_re || !d_im) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (dist != 1 || !is_pow2(stride) || !is_pow2(batch) || batch > stride) return PHAST_ERR_INVALID_ARG;
    const unsigned sb = ilog2(stride), bb = ilog2(batch);
    const unsigned gl = tw_n ? ilog2(tw_n) : 0u;
    if (direction == PHAST_REVERSE) return pl->exec_strided(d_im, d_re, sb, bb, 1.0 / (double)n, s, gl, (unsigned)tw_col0);
    return pl->exec_strided(d_re, d_im, sb, bb, 1.0, s, gl, (unsigned)tw_col0);
}

// average kernel duration of every pass over `reps` forward transforms of the same buffers
template <typename T>
static int time_passes(const Planner<T> *pl, T *d_re, T *d_im, size_t batch, size_t dist, int reps, float *pass_ms,
                       int *n_passes, hipStream_t s) {
    if (!pl || !d_re || !d_im || !pass_ms || !n_passes || reps < 1) return PHAST_ERR_INVALID_ARG;
    int np = 1;  // the slots the launches really used (a batch above the scratch runs in chunks, each with its own plan)
    PHAST_ON_DEVICE(pl->device);
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        PassTimer tm;
        int rc = pl->exec(d_re, d_im, dist, 0, d_re, d_im, dist, 0, batch, 1.0, s, &tm);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < tm.pass_of.size(); ++i) {
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
            acc[tm.pass_of[i] & 3] += ms;
            np = std::max(np, (tm.pass_of[i] & 3) + 1);
        }
    }
    for (int i = 0; i < np; ++i) pass_ms[i] = (float)(acc[i] / reps);
    *n_passes = np;
    return PHAST_OK;
}

// the same for a real transform: slots 0..np-1 = passes of the inner N/2-point transform, slot np = the untangle sweep
// (a transform small enough for the one-pass kernel has the untangle fused: one slot)
template <typename T>
static int time_passes_r2c(const PlannerR2c<T> *pl, const T *d_in, T *d_ore, T *d_oim, size_t batch, size_t in_dist,
                           size_t out_dist, int reps, float *pass_ms, int *n_passes, hipStream_t s) {
    if (!pl || !d_in || !d_ore || !d_oim || !pass_ms || !n_passes || reps < 1) return PHAST_ERR_INVALID_ARG;
    int np = 1;  // slots the launches really used: the plan and the fuse decision are exec_in's
    PHAST_ON_DEVICE(pl->dit.device);
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        PassTimer tm;
        int rc = pl->dit.passes.empty() ? pl->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, s)
                                        : pl->r2c(d_in, d_ore, d_oim, batch, in_dist, out_dist, s, &tm);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < tm.pass_of.size(); ++i) {
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
            acc[tm.pass_of[i] & 3] += ms;
            np = std::max(np, (tm.pass_of[i] & 3) + 1);
        }
    }
    for (int i = 0; i < np; ++i) pass_ms[i] = (float)(acc[i] / reps);
    *n_passes = np;
    return PHAST_OK;
}

// ... and for the inverse real transform: slots 0..np-1 = passes of the inner transform (the first one forms z on load
// when its fused form exists, c2r_fused.hpp), slot np = the preprocess sweep otherwise
template <typename T>
static int time_passes_c2r(const PlannerR2c<T> *pl, const T *d_ire, const T *d_iim, T *d_out, size_t batch, size_t in_dist,
                           size_t out_dist, int reps, float *pass_ms, int *n_passes, hipStream_t s) {
    if (!pl || !d_ire || !d_iim || !d_out || !pass_ms || !n_passes || reps < 1) return PHAST_ERR_INVALID_ARG;
    int np = 1;
    PHAST_ON_DEVICE(pl->dit.device);
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        PassTimer tm;
        int rc = pl->dit.passes.empty() ? pl->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, s)
                                        : pl->c2r(d_ire, d_iim, d_out, batch, in_dist, out_dist, s, &tm);
        if (rc) return rc;
        PHAST_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < tm.pass_of.size(); ++i) {
            float ms = 0;
            PHAST_HIP(hipEventElapsedTime(&ms, tm.ev[2 * i], tm.ev[2 * i + 1]));
            acc[tm.pass_of[i] & 3] += ms;
            np = std::max(np, (tm.pass_of[i] & 3) + 1);
        }
    }
    for (int i = 0; i < np; ++i) pass_ms[i] = (float)(acc[i] / reps);
    *n_passes = np;
    return PHAST_OK;
}

// lib.rs:41-140 (feature complex-nums): interleaved Complex<T> signal, in place.  The reference copies into two
// planar Vecs and back; here the (de)interleave is the first pass's load and the last pass's store.
template <typename T>
static int fft_interleaved_dev(T *d_signal, size_t n, size_t batch, size_t dist, int direction, const Planner<T> *pl,
                               hipStream_t s) {
    if (!pl || !d_signal) return PHAST_ERR_INVALID_ARG;
    if (direction != PHAST_FORWARD && direction != PHAST_REVERSE) return PHAST_ERR_INVALID_ARG;
    if (!is_pow2(n)) return PHAST_ERR_NOT_POW2;
    if (ilog2(n) != pl->log_n) return PHAST_ERR_PLANNER_SIZE;
    if (batch > 1 && dist < n) return PHAST_ERR_INVALID_ARG;
    if (direction == PHAST_REVERSE)  // swap trick: read (im, re), transform, store (im, re) scaled by 1/N
        return pl->exec(d_signal, nullptr, dist, 2, d_signal, nullptr, dist, 2, batch, 1.0 / (double)n, s);
    return pl->exec(d_signal, nullptr, dist, 1, d_signal, nullptr, dist, 1, batch, 1.0, s);
}

}  // namespace phast
