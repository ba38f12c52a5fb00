"""The tile-FFT kernels' index arithmetic, LDS layouts, twiddle tables and pass plans, executed on the
CPU: tests/emu/libphastft_emu.so runs the SAME `TileBody` phase functions (csrc/tile_fft.hpp) and the SAME plan
geometry (csrc/plan.hpp) as the GPU kernels, thread by thread, and is compared with the oracle.  This is
what keeps kernel edits honest in the GPU-less build container; the `-m gpu` tests check the real thing."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    from tests.emu import build_emulator

    lib = C.CDLL(build_emulator())
    lib.phast_emu_fft_f32_modes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint,
                                            C.c_double, C.c_void_p, C.c_size_t, C.c_uint]
    return lib


def run(emu, re, im, direction=1, lrs=(), tile_log=12, points_log=4):
    L = int(np.log2(re.size))
    arr = (C.c_uint * max(1, len(lrs)))(*lrs)
    fn = emu.phast_emu_fft_f64 if re.dtype == np.float64 else emu.phast_emu_fft_f32
    return fn(re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p), C.c_uint(L), C.c_size_t(1),
              C.c_int(direction), arr, C.c_size_t(len(lrs)), C.c_uint(tile_log | (points_log << 8)))


# (log2 rows, log2 cols, log2 points per thread) -- PHAST_TILE_SHAPES of csrc/plan.hpp
SHAPES = [(6, 6, 4), (7, 5, 4), (8, 4, 4), (9, 3, 4), (10, 2, 4), (7, 6, 4), (8, 5, 4), (9, 4, 4), (10, 3, 4), (8, 6, 4),
          (9, 5, 4), (10, 4, 4), (6, 6, 3), (7, 5, 3), (8, 4, 3), (9, 3, 3), (10, 2, 3),
          (10, 4, 5), (9, 5, 5), (8, 6, 5), (10, 3, 5), (9, 4, 5), (8, 5, 5), (10, 2, 5), (11, 3, 5),
          (6, 5, 3), (7, 4, 3), (8, 3, 3), (6, 4, 3), (7, 3, 3), (6, 5, 4), (7, 4, 4), (8, 3, 4), (6, 4, 4), (7, 3, 4), (7, 4, 5)]
SHAPES_F32_ONLY = [(10, 5, 5), (9, 6, 5), (8, 7, 5), (11, 4, 5)]  # PHAST_TILE_SHAPES_F32: 32768-point tiles


@pytest.mark.parametrize("is_f64", [1, 0])
def test_lds_exchanges_in_bounds_permutations_and_conflict_free(emu, is_f64):
    for lr, lc, lp in SHAPES + ([] if is_f64 else SHAPES_F32_ONLY):
        for transpose in (1, 0):
            r, w = C.c_int(), C.c_int()
            errors = emu.phast_emu_audit_lds(is_f64, lr, lc, lp, transpose, C.byref(r), C.byref(w))
            assert errors == 0, (lr, lc, lp, transpose)
            assert r.value == 1 and w.value == 1, (lr, lc, lp, transpose, r.value, w.value)


PLANS = [(12, (6, 6), 12, 4), (13, (7, 6), 12, 4), (15, (8, 7), 12, 4), (16, (8, 8), 13, 4), (17, (9, 8), 13, 4),
         (18, (9, 9), 12, 4), (19, (10, 9), 13, 4), (20, (10, 10), 12, 4), (20, (10, 10), 13, 4), (20, (10, 10), 14, 4),
         (20, (7, 7, 6), 12, 4), (18, (6, 6, 6), 12, 4), (21, (7, 7, 7), 13, 4), (22, (8, 7, 7), 13, 4),
         # 8 points per thread (latency tiles): every 4096-point shape, 2..4 radix steps
         (12, (6, 6), 12, 3), (13, (7, 6), 12, 3), (15, (8, 7), 12, 3), (17, (9, 8), 12, 3), (19, (10, 9), 12, 3),
         (20, (10, 10), 12, 3), (21, (7, 7, 7), 12, 3),
         # 32 points per thread: 32x32, 32x16 and 32x8 chains, 16384-/8192-/4096-point tiles
         (20, (10, 10), 14, 5), (19, (10, 9), 14, 5), (18, (9, 9), 13, 5), (16, (8, 8), 13, 5), (20, (10, 10), 12, 5),
         (20, (10, 10), 13, 5),
         # a 2048-point tile FFT (32 x 32 x 2) as second or first pass
         (21, (10, 11), 14, 5), (21, (11, 10), 14, 5),
         # round 2: 1024- and 2048-point tiles (several workgroups per CU)
         (18, (6, 6, 6), 10, 3), (18, (6, 6, 6), 10, 4), (20, (7, 7, 6), 11, 3), (20, (7, 6, 7), 11, 4), (21, (7, 7, 7), 11, 5),
         # WAVE tiles (points code | 0x10: wave_fft.hpp, the cross-lane swaps emulated): all three passes
         (18, (6, 6, 6), 10, 4 | 0x10),
         # round 3: the forced plans that replaced the uninstantiable (9, 8, 3) of tests/test_gpu_parity.py
         (20, (8, 6, 6), 12, 4), (20, (6, 6, 8), 12, 4)]


@pytest.mark.parametrize("L,lrs,tl,lp", PLANS)
def test_forced_plans_vs_oracle_f64(emu, oracle, L, lrs, tl, lp):
    n = 1 << L
    re, im = oracle.fill(n, np.float64, transform_id=L)
    a, b = re.copy(), im.copy()
    assert run(emu, a, b, 1, lrs, tl, lp) == 0
    oracle.fft_64_dit(re, im, oracle.FORWARD)
    err = np.sqrt(np.sum((a - re) ** 2 + (b - im) ** 2) / np.sum(re ** 2 + im ** 2))
    assert err <= 1e-13, err


@pytest.mark.parametrize("L", list(range(14, 23)))  # up to 2^13 the one-pass small-transform kernel runs
def test_default_plans_both_types_and_inverse(emu, oracle, L):
    n = 1 << L
    for is_f64, dtype, tol, ofn in ((1, np.float64, 1e-13, oracle.fft_64_dit), (0, np.float32, 1e-5, oracle.fft_32_dit)):
        for latency in (0, 1, 2):  # throughput plan, latency plan, the plan for ONE transform (wave / quad tiles; 2^19..2^23)
            lrs = (C.c_uint * 3)()
            tl, lp = C.c_uint(), C.c_uint()
            npass = emu.phast_emu_default_plan(is_f64, latency, L, lrs, C.byref(tl), C.byref(lp))
            assert npass in (2, 3)
            re, im = oracle.fill(n, dtype, transform_id=7 * L + latency)
            a, b = re.copy(), im.copy()
            direction = -1 if latency else 1
            # lrs = () lets the emulator take the library's own heuristic plan (incl. per-pass tile sizes and wave
            # tiles): tile_log 0 selects the latency plan, anything else the throughput plan
            assert run(emu, a, b, direction, (), (12, 0, 1)[latency], 0) == 0
            ofn(re, im, oracle.REVERSE if latency else oracle.FORWARD)
            err = np.sqrt(np.sum((a.astype(np.float64) - re) ** 2 + (b.astype(np.float64) - im) ** 2) /
                          np.sum(re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2))
            assert err <= tol, (L, is_f64, latency, err)


@pytest.mark.parametrize("L,lrs", [(20, (10, 10)), (19, (10, 9)), (17, (9, 8)), (16, (8, 8))])
def test_f32_32768_point_tiles_vs_oracle(emu, oracle, L, lrs):
    n = 1 << L
    re, im = oracle.fill(n, np.float32, transform_id=L)
    a, b = re.copy(), im.copy()
    assert run(emu, a, b, 1, lrs, 15, 5) == 0
    oracle.fft_32_dit(re, im, oracle.FORWARD)
    err = np.sqrt(np.sum((a.astype(np.float64) - re) ** 2 + (b.astype(np.float64) - im) ** 2) /
                  np.sum(re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2))
    assert err <= 1e-5, err


def test_interleaved_load_and_swapped_interleaved_store(emu, oracle):
    """The R2C deinterleave fused into the first pass's load and the C2R (im, re) interleave fused into the
    last pass's store (r2c.rs:73-128, 446-489; algorithms/dit.rs:297-300)."""
    L = 14
    n = 1 << L
    re, im = oracle.fill(n, np.float32, transform_id=3)
    z = np.empty(2 * n, np.float32)
    z[0::2], z[1::2] = re, im
    out_re, out_im = np.zeros(n, np.float32), np.zeros(n, np.float32)
    lrs = (C.c_uint * 2)(7, 7)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert emu.phast_emu_fft_f32_modes(p(z), None, 1, p(out_re), p(out_im), 0, L, 1.0, lrs, 2, 12) == 0
    r, m = re.copy(), im.copy()
    oracle.fft_32_dit(r, m, oracle.FORWARD)
    assert np.sqrt(np.sum((out_re - r) ** 2 + (out_im - m) ** 2) / np.sum(r ** 2 + m ** 2)) < 1e-5
    # inverse by the swap trick, stored as (im, re) pairs == interleaved (re, im) of the true inverse
    zz = np.zeros(2 * n, np.float32)
    assert emu.phast_emu_fft_f32_modes(p(im), p(re), 0, p(zz), None, 2, L, 1.0 / n, lrs, 2, 12) == 0
    r, m = re.copy(), im.copy()
    oracle.fft_32_dit(r, m, oracle.REVERSE)
    assert np.max(np.abs(zz[0::2] - r)) < 1e-6 and np.max(np.abs(zz[1::2] - m)) < 1e-6


# ---------------------------------------------------------------- small transforms (row_fft.hpp)
def _small(emu, is_f64, in_re, in_im, in_mode, out_re, out_im, out_mode, log_n, batch, in_dist, out_dist, scale=1.0):
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    emu.phast_emu_small_fft.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint,
                                        C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double]
    return emu.phast_emu_small_fft(is_f64, p(in_re), p(in_im), in_mode, p(out_re), p(out_im), out_mode, log_n, batch,
                                   in_dist, out_dist, scale)


@pytest.mark.parametrize("is_f64", [1, 0])
def test_small_kernel_lds_audit(emu, is_f64):
    for log_n in range(1, 14):
        r, w = C.c_int(), C.c_int()
        errors = emu.phast_emu_audit_small(is_f64, log_n, C.byref(r), C.byref(w))
        assert errors == 0, log_n
        # N <= 16: a pad word every N < 32 words leaves 2-way conflicts; N = 4096 and 8192 (one transform per workgroup at
        # 16 points per thread, a 32-lane group spans 32 rows): one bank pair collides in the first exchange -- measured
        # and accepted for 8192 in round 4 (16 points per thread on 512 threads beat 32 on 256 by 10-22 %, r04_row13_ab.log)
        worst = 2 if (log_n <= 4 or log_n >= 12) else 1
        assert r.value <= worst and w.value <= worst, (log_n, r.value, w.value)


@pytest.mark.parametrize("log_n", list(range(1, 14)))
def test_small_transforms_vs_oracle_ragged_batches(emu, oracle, log_n):
    """Every small size, in place, with a batch that does not fill the last tile and transforms `dist` apart."""
    n = 1 << log_n
    for dtype, is_f64, tol, ofn in ((np.float64, 1, 1e-13, oracle.fft_64_dit), (np.float32, 0, 1e-5, oracle.fft_32_dit)):
        lc_full = {1: 256, 2: 256, 3: 256, 4: 256, 5: 128}.get(log_n, max(1, 4096 // n))
        batch = lc_full + max(1, lc_full // 2) + 1
        dist = n + 3
        re = np.full(batch * dist, 7.0, dtype)
        im = np.full(batch * dist, -7.0, dtype)
        want = []
        for b in range(batch):
            r, m = oracle.fill(n, dtype, transform_id=100 * log_n + b)
            re[b * dist:b * dist + n], im[b * dist:b * dist + n] = r, m
            if b in (0, 1, lc_full - 1, lc_full, batch - 1):
                ofn(r, m, oracle.FORWARD)
                want.append((b, r, m))
        assert _small(emu, is_f64, re, im, 0, re, im, 0, log_n, batch, dist, dist) == 0
        for b, r, m in want:
            gr, gm = re[b * dist:b * dist + n].astype(np.float64), im[b * dist:b * dist + n].astype(np.float64)
            err = np.sqrt(np.sum((gr - r) ** 2 + (gm - m) ** 2) / np.sum(r.astype(np.float64) ** 2 + m.astype(np.float64) ** 2))
            assert err <= tol, (log_n, b, err)
        gaps = np.concatenate([re[b * dist + n:(b + 1) * dist] for b in range(batch)])
        assert np.all(gaps == 7.0)


def test_small_transforms_interleaved_modes(emu, oracle):
    """(re, im) pairs in -> planar out (R2C's inner transform) and planar in -> (im, re) pairs out (C2R's)."""
    log_n, batch = 7, 45
    n = 1 << log_n
    z = np.empty(2 * n * batch, np.float64)
    refs = []
    for b in range(batch):
        r, m = oracle.fill(n, np.float64, transform_id=b)
        z[2 * b * n:2 * (b + 1) * n:2], z[2 * b * n + 1:2 * (b + 1) * n:2] = r, m
        oracle.fft_64_dit(r, m, oracle.FORWARD)
        refs.append((r, m))
    out_re, out_im = np.zeros(n * batch), np.zeros(n * batch)
    assert _small(emu, 1, z, None, 1, out_re, out_im, 0, log_n, batch, n, n) == 0
    for b, (r, m) in enumerate(refs):
        assert np.max(np.abs(out_re[b * n:(b + 1) * n] - r)) < 1e-11 and np.max(np.abs(out_im[b * n:(b + 1) * n] - m)) < 1e-11
    zz = np.zeros(2 * n * batch)
    assert _small(emu, 1, out_re, out_im, 0, zz, None, 2, log_n, batch, n, n, 0.5) == 0
    back_im, back_re = zz[0::2], zz[1::2]
    # same arithmetic through planar output, scaled the same way
    pr, pi = np.zeros(n * batch), np.zeros(n * batch)
    assert _small(emu, 1, out_re, out_im, 0, pr, pi, 0, log_n, batch, n, n, 0.5) == 0
    assert np.array_equal(back_re, pr) and np.array_equal(back_im, pi)


@pytest.mark.parametrize("log_half", list(range(1, 14)))
def test_small_real_transforms_fused_untangle_and_preprocess(emu, oracle, log_half):
    """R2C with the untangle as the one-pass kernel's epilogue and C2R with the preprocess as its prologue
    (n = 4 .. 16384 real points), ragged batches, against the oracle's r2c / c2r."""
    emu.phast_emu_small_real.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
                                         C.c_size_t, C.c_size_t, C.c_size_t]
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    n = 2 << log_half
    half = n // 2
    per_tile = {1: 256, 2: 256, 3: 256, 4: 256, 5: 128}.get(log_half, max(1, 4096 // half))
    batch = per_tile + max(1, per_tile // 2) + 1
    for dtype, is_f64, tol, r2c, c2r in ((np.float64, 1, 1e-9, oracle.r2c_fft_f64, oracle.c2r_fft_f64),
                                         (np.float32, 0, 1e-5, oracle.r2c_fft_f32, oracle.c2r_fft_f32)):
        x = np.empty(batch * n, dtype)
        for b in range(batch):
            x[b * n:(b + 1) * n] = oracle.fill(n, dtype, transform_id=31 * log_half + b)[0]
        ore = np.zeros(batch * (half + 1), dtype)
        oim = np.zeros(batch * (half + 1), dtype)
        assert emu.phast_emu_small_real(is_f64, 1, p(x), None, p(ore), p(oim), log_half, batch, n, half + 1) == 0
        for b in (0, per_tile - 1, per_tile, batch - 1):
            rr, ri = np.zeros(half + 1, dtype), np.zeros(half + 1, dtype)
            r2c(x[b * n:(b + 1) * n].copy(), rr, ri)
            sl = slice(b * (half + 1), (b + 1) * (half + 1))
            num = np.sqrt(np.sum((ore[sl].astype(np.float64) - rr) ** 2 + (oim[sl].astype(np.float64) - ri) ** 2))
            den = np.sqrt(np.sum(rr.astype(np.float64) ** 2 + ri.astype(np.float64) ** 2))
            assert num / den <= tol, (log_half, b, num / den)
        back = np.zeros(batch * n, dtype)
        assert emu.phast_emu_small_real(is_f64, 2, p(ore), p(oim), p(back), None, log_half, batch, half + 1, n) == 0
        assert np.max(np.abs(back - x)) < (1e-12 if is_f64 else 2e-5), log_half
        b = batch - 1  # and against the oracle's own c2r of the oracle's spectrum
        rr, ri = np.zeros(half + 1, dtype), np.zeros(half + 1, dtype)
        r2c(x[b * n:(b + 1) * n].copy(), rr, ri)
        want = np.zeros(n, dtype)
        c2r(rr, ri, want)
        got = np.zeros(n, dtype)
        assert emu.phast_emu_small_real(is_f64, 2, p(rr), p(ri), p(got), None, log_half, 1, half + 1, n) == 0
        assert np.max(np.abs(got - want)) < (1e-12 if is_f64 else 2e-5)


@pytest.mark.parametrize("L,s,sb", [(6, 4, 4), (8, 5, 5), (11, 4, 4), (12, 6, 5), (12, 5, 4), (16, 4, 4), (21, 4, 4)])
def test_strided_batch_geometry_vs_oracle(emu, oracle, L, s, sb):
    """Strided batches (plan.hpp: make_strided_passes -- column FFTs of a row-major [2^L][2^s] array, first 2^sb
    columns): one, two and three passes, none of them transposing, the batch index as the contiguous dimension; every
    transformed column against the oracle, every other column untouched."""
    n, stride, batch = 1 << L, 1 << s, 1 << sb
    rng = np.random.default_rng(L * 10 + s)
    re, im = rng.uniform(-1, 1, n * stride), rng.uniform(-1, 1, n * stride)
    a, b = re.copy(), im.copy()
    assert emu.phast_emu_fft_strided_f64(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), L, s, sb) == 0
    A, B, R, I = (x.reshape(n, stride) for x in (a, b, re, im))
    for c in sorted({0, 1, batch // 2, batch - 1}):
        r, m = np.ascontiguousarray(R[:, c]), np.ascontiguousarray(I[:, c])
        oracle.fft_64_dit(r, m, oracle.FORWARD)
        assert np.sqrt(np.sum((A[:, c] - r) ** 2 + (B[:, c] - m) ** 2) / np.sum(r ** 2 + m ** 2)) <= 1e-13, c
    if batch < stride:
        assert np.array_equal(A[:, batch:], R[:, batch:]) and np.array_equal(B[:, batch:], I[:, batch:])


@pytest.mark.parametrize("L,lrs,tl", [(18, (6, 6, 6), 11), (17, (6, 6, 5), 11)])
def test_f32_wave_tiles_vs_oracle(emu, oracle, L, lrs, tl):
    """The f32 wave tiles (round 6: 64 rows x 32 columns, a lane holds float2 COLUMN PAIRS -- the f64 tile's lane layout,
    exchanges and instruction count with every register carrying two columns; wave_fft.hpp) -- all passes of 2^18 as wave
    tiles, forward and inverse, against the oracle AND float64 pocketfft; a plan whose last pass is not 64 rows long must be
    refused or run the generic tiles for it."""
    n = 1 << L
    for direction, odir in ((1, oracle.FORWARD), (-1, oracle.REVERSE)):
        re, im = oracle.fill(n, np.float32, transform_id=L)
        z = re.astype(np.float64) + 1j * im.astype(np.float64)
        a, b = re.copy(), im.copy()
        rc = run(emu, a, b, direction, lrs, tl, 3 | 0x10)
        if lrs[-1] != 6:
            assert rc != 0  # (5, 6, 3) is not a tile shape: the plan is refused, nothing runs
            return
        assert rc == 0
        oracle.fft_32_dit(re, im, odir)
        err = np.sqrt(np.sum((a.astype(np.float64) - re) ** 2 + (b.astype(np.float64) - im) ** 2) /
                      np.sum(re.astype(np.float64) ** 2 + im.astype(np.float64) ** 2))
        assert err <= 1e-5, err
        want = np.fft.fft(z) if direction == 1 else np.fft.ifft(z)
        got = a.astype(np.float64) + 1j * b.astype(np.float64)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1.5e-7 * L, (direction, L)


@pytest.mark.parametrize("plan,L", [("6,8,6@11,13,11:p8w", 20), ("6,8,6@12,13,11:p8w", 20), ("6,8@11,13:p8w", 14), ("7,8,6@12,13,11:p8w", 21),
                                    ("6,8,8@11,13,13:p8w", 22)])
def test_f32_four_wave_pass_vs_oracle(emu, oracle, plan, L):
    """round 6: the f32 four-wave 256-row pass (quad_fft.hpp on float2 column pairs: 256 rows x 32 columns) between / behind
    f32 wave tiles and generic tiles -- the plan of ONE f32 transform of 2^20 points [64x32A w][256x32 q][64x32 w] among them;
    every output against float64 pocketfft at the f32 gate of tests/tolerances.py, forward and inverse."""
    emu.phast_emu_set_plan.argtypes = [C.c_char_p]
    n = 1 << L
    try:
        assert emu.phast_emu_set_plan(plan.encode()) == 0, plan
        for direction in (1, -1):
            re, im = oracle.fill(n, np.float32, transform_id=L + 40)
            z = re.astype(np.float64) + 1j * im.astype(np.float64)
            a, b = re.copy(), im.copy()
            assert run(emu, a, b, direction) == 0, plan
            want = np.fft.fft(z) if direction == 1 else np.fft.ifft(z)
            got = a.astype(np.float64) + 1j * b.astype(np.float64)
            assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1.5e-7 * L, (plan, direction)
            assert np.max(np.abs(got - want)) / np.sqrt(np.mean(np.abs(want) ** 2)) <= 2e-6 * L, (plan, direction)
    finally:
        emu.phast_emu_set_plan(None)


def _emu_r2c(emu, x, lrs=(), tile_log=0, points_log=0):
    n = x.size
    L = int(np.log2(n))
    ore, oim = np.zeros(n // 2 + 1, x.dtype), np.zeros(n // 2 + 1, x.dtype)
    arr = (C.c_uint * max(1, len(lrs)))(*lrs)
    fn = emu.phast_emu_r2c_fused_f64 if x.dtype == np.float64 else emu.phast_emu_r2c_fused_f32
    rc = fn(x.ctypes.data_as(C.c_void_p), C.c_uint(L), ore.ctypes.data_as(C.c_void_p), oim.ctypes.data_as(C.c_void_p), arr,
            C.c_size_t(len(lrs)), C.c_uint(tile_log | (points_log << 8)))
    return rc, ore, oim


@pytest.mark.parametrize("log_n", [15, 16, 17, 18, 19, 20, 21])
def test_r2c_fused_last_pass_vs_oracle_and_rfft(emu, oracle, log_n):
    """round 3 (r2c_fused.hpp): the inner transform's LAST pass computes every column twice -- once on column g, once on
    the conjugate of the mirrored column M - g -- and untangles thread-locally; X[k] and X[h - k] both stored.  The
    library's latency plan, its plan for one transform and forced plans (two and three passes, 8 and 16 points per
    thread), both types: every output against the oracle's r2c and an independent real FFT, incl. X[0], X[h/2], X[h]."""
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    ran = 0
    # forced plans of the INNER 2^(log_n - 1)-point transform: (rows per pass, log2 tile points, log2 points per thread)
    forced = {15: [((7, 7), 12, 3), ((7, 7), 13, 4)], 16: [((8, 7), 12, 3)], 17: [((8, 8), 13, 4), ((8, 8), 12, 3)],
              18: [((9, 8), 12, 3)], 19: [((6, 6, 6), 12, 4), ((9, 9), 14, 5)], 20: [((7, 6, 6), 12, 3)],
              21: [((7, 7, 6), 12, 3), ((10, 10), 14, 5)]}[log_n]  # (.., 14, 5): 32 points per thread -- fused in f32 only
    for dtype, tol_or, tol_np in ((np.float64, 1e-9, 1e-13), (np.float32, 1e-5, 1e-5)):
        x = rng.uniform(-1, 1, n).astype(dtype)
        o_re, o_im = np.zeros(n // 2 + 1, dtype), np.zeros(n // 2 + 1, dtype)
        (oracle.r2c_fft_f64 if dtype == np.float64 else oracle.r2c_fft_f32)(x.copy(), o_re, o_im)
        ref = np.fft.rfft(x.astype(np.float64))
        for lrs, tl, lp in [((), 0, 0), ((), 1, 0), ((), 2, 0), ((), 3, 0)] + forced:  # 2 / 3: plan.hpp's real_plan / real_batch_plan
            rc, ore, oim = _emu_r2c(emu, x, lrs, tl, lp)
            if rc == 3:  # this plan's last pass has no fused form (wave / quad tiles, 32 points per thread)
                continue
            assert rc == 0, (rc, lrs, tl, lp)
            ran += 1
            got = ore.astype(np.float64) + 1j * oim.astype(np.float64)
            den = np.sqrt(np.sum(np.abs(ref) ** 2))
            assert np.sqrt(np.sum(np.abs(got - ref) ** 2)) / den <= tol_np, (dtype, lrs, tl, lp)
            assert np.sqrt(np.sum(np.abs(got - (o_re.astype(np.float64) + 1j * o_im.astype(np.float64))) ** 2)) / den <= tol_or
            assert oim[0] == 0 and oim[-1] == 0  # r2c.rs:161-166: exact zeros
            assert np.max(np.abs(got - ref)) <= (1e-11 if dtype == np.float64 else 2e-3) * np.sqrt(n), "a single bin is off"
    assert ran >= 4, ran


def _emu_c2r(emu, ire, iim, n, batch=1, in_dist=None, lrs=(), tile_log=0, points_log=0):
    L = int(np.log2(n))
    out = np.zeros(n * batch, ire.dtype)
    arr = (C.c_uint * max(1, len(lrs)))(*lrs)
    fn = emu.phast_emu_c2r_fused_f64 if ire.dtype == np.float64 else emu.phast_emu_c2r_fused_f32
    rc = fn(ire.ctypes.data_as(C.c_void_p), iim.ctypes.data_as(C.c_void_p), C.c_uint(L), out.ctypes.data_as(C.c_void_p),
            C.c_size_t(batch), C.c_size_t(in_dist or n // 2 + 1), arr, C.c_size_t(len(lrs)), C.c_uint(tile_log | (points_log << 8)))
    return rc, out


@pytest.mark.parametrize("log_n", [15, 16, 17, 18, 19, 20, 21])
def test_c2r_fused_first_pass_vs_oracle_and_irfft(emu, oracle, log_n):
    """round 3 (c2r_fused.hpp): the inner transform's FIRST pass loads X[k] and its partner X[h - k] (mirrored column, rows
    reversed; X[h] for k = 0) and forms z in registers -- no preprocess sweep, no workspace.  The library's plans and
    forced ones (two and three passes; 8, 16 and 32 points per thread -- incl. the chunked loads of the wide shapes),
    both types: every output against the oracle's c2r and numpy's irfft; a ragged batch of two (in_dist > h + 1)."""
    n = 1 << log_n
    h1 = n // 2 + 1
    rng = np.random.default_rng(100 + log_n)
    ran = 0
    forced = {15: [((7, 7), 12, 3), ((7, 7), 13, 4)], 16: [((8, 7), 12, 3), ((8, 7), 13, 5)], 17: [((8, 8), 13, 4), ((8, 8), 12, 3)],
              18: [((9, 8), 12, 3), ((9, 8), 14, 4)], 19: [((6, 6, 6), 12, 4), ((9, 9), 14, 5)], 20: [((7, 6, 6), 12, 3), ((10, 9), 14, 4)],
              21: [((7, 7, 6), 12, 3), ((10, 10), 14, 5), ((10, 10), 15, 5)]}[log_n]
    for dtype, tol_or, tol_np in ((np.float64, 1e-9, 1e-13), (np.float32, 1e-5, 1e-5)):
        x = rng.uniform(-1, 1, n)
        spec = np.fft.rfft(x)
        spec.imag[0] = spec.imag[-1] = 0.0
        ire, iim = spec.real.astype(dtype), spec.imag.astype(dtype)
        want = np.zeros(n, dtype)
        (oracle.c2r_fft_f64 if dtype == np.float64 else oracle.c2r_fft_f32)(ire.copy(), iim.copy(), want)
        ref = np.fft.irfft(ire.astype(np.float64) + 1j * iim.astype(np.float64), n)
        for lrs, tl, lp in [((), 0, 0), ((), 1, 0), ((), 2, 0), ((), 3, 0), ((), 14, 0)] + forced:  # 2 / 3: real_plan / real_batch_plan
            rc, out = _emu_c2r(emu, ire, iim, n, lrs=lrs, tile_log=tl, points_log=lp)
            if rc in (1, 3):  # not a plan of this type (32768-point tiles are f32 only) / first pass without a fused form
                continue
            assert rc == 0, (rc, lrs, tl, lp)
            ran += 1
            den = np.sqrt(np.sum(ref ** 2))
            assert np.sqrt(np.sum((out - ref) ** 2)) / den <= tol_np, (dtype, lrs, tl, lp)
            assert np.sqrt(np.sum((out.astype(np.float64) - want) ** 2)) / den <= tol_or, (dtype, lrs, tl, lp)
            assert np.max(np.abs(out - ref)) <= (1e-13 if dtype == np.float64 else 1e-5) * np.sqrt(n), "a single sample is off"
        if log_n == 16:  # two transforms, spectra 7 elements further apart than they are long
            dist = h1 + 7
            bre, bim = np.zeros(2 * dist, dtype), np.zeros(2 * dist, dtype)
            spec2 = np.fft.rfft(rng.uniform(-1, 1, n))
            spec2.imag[0] = spec2.imag[-1] = 0.0
            bre[:h1], bim[:h1] = ire, iim
            bre[dist:dist + h1], bim[dist:dist + h1] = spec2.real.astype(dtype), spec2.imag.astype(dtype)
            rc, out = _emu_c2r(emu, bre, bim, n, batch=2, in_dist=dist)
            assert rc == 0
            ref2 = np.fft.irfft(bre[dist:dist + h1].astype(np.float64) + 1j * bim[dist:dist + h1].astype(np.float64), n)
            for got, r in ((out[:n], ref), (out[n:], ref2)):
                assert np.sqrt(np.sum((got - r) ** 2)) / np.sqrt(np.sum(r ** 2)) <= tol_np
    assert ran >= 6, ran


def _two_pass_plans_through(shape, shapes, first):
    """A forced two-pass plan (rows of pass A, rows of pass B) of the inner transform whose first (or last) pass is `shape`:
    both passes share the tile size and the points per thread, so the other pass's shape must exist too."""
    lr, lc, lp = shape
    tl = lr + lc
    for other in (8, 7, 9, 6, 10, 11):
        if (other, tl - other, lp) in shapes and tl - other >= 2:
            lrs = (lr, other) if first else (other, lr)
            if lrs[1] + 1 <= 12 and lr + other <= 21:  # (columns of the last pass = rows of the first: keep the sizes testable)
                return lrs, tl, lp
    return None


@pytest.mark.parametrize("is_f64", [1, 0])
def test_every_instantiated_shape_as_fused_first_pass_of_c2r_and_last_pass_of_r2c(emu, oracle, is_f64):
    """GPU tests reach the shapes the default plans use; the fused real-transform passes are instantiated for every tile
    shape (tile_f*_c2r.hip, tile_f*_r2c.hip).  Each of them as the first pass of a forced two-pass C2R and as the last pass
    of a forced two-pass R2C, thread by thread: every output against numpy (and the C2R's chunked partner loads -- 2, 4, 8
    rows at a time depending on the shape's register budget -- with it)."""
    dtype = np.float64 if is_f64 else np.float32
    shapes = set(SHAPES + ([] if is_f64 else SHAPES_F32_ONLY))
    tol = 1e-13 if is_f64 else 1e-5
    ran_c2r = ran_r2c = 0
    for shape in sorted(shapes):
        for first in (True, False):
            plan = _two_pass_plans_through(shape, shapes, first)
            if plan is None:
                continue
            lrs, tl, lp = plan
            n = 2 << (lrs[0] + lrs[1])
            rng = np.random.default_rng(sum(shape) * 7 + first)
            x = rng.uniform(-1, 1, n)
            if first:
                spec = np.fft.rfft(x)
                spec.imag[0] = spec.imag[-1] = 0.0
                ire, iim = spec.real.astype(dtype), spec.imag.astype(dtype)
                rc, out = _emu_c2r(emu, ire, iim, n, lrs=lrs, tile_log=tl, points_log=lp)
                if rc == 3:  # no fused form of this shape (c2r_shape_fits)
                    continue
                assert rc == 0, (rc, shape, plan)
                ref = np.fft.irfft(ire.astype(np.float64) + 1j * iim.astype(np.float64), n)
                assert np.sqrt(np.sum((out - ref) ** 2) / np.sum(ref ** 2)) <= tol, (shape, plan)
                ran_c2r += 1
            else:
                rc, ore, oim = _emu_r2c(emu, x.astype(dtype), lrs, tl, lp)
                if rc == 3:  # r2c_shape_fits: 32 points per thread on 1024 threads, f64 above 16 points
                    continue
                assert rc == 0, (rc, shape, plan)
                ref = np.fft.rfft(x.astype(dtype).astype(np.float64))
                got = ore.astype(np.float64) + 1j * oim.astype(np.float64)
                assert np.sqrt(np.sum(np.abs(got - ref) ** 2) / np.sum(np.abs(ref) ** 2)) <= tol, (shape, plan)
                ran_r2c += 1
    assert ran_c2r >= (35 if is_f64 else 40) and ran_r2c >= (24 if is_f64 else 36), (ran_c2r, ran_r2c)


def test_every_plan_table_entry_is_a_plan_that_exists(emu):
    """plan.hpp: single_plan / real_plan / real_batch_plan (round 4: 77 entries ranked on the GPU) -- rows add up to the length,
    every pass is an instantiated shape, make_passes accepts the geometry.
    A bad entry would not fail anywhere else: an optional plan that cannot be built is skipped silently."""
    n = C.c_int()
    emu.phast_emu_check_plan_tables.argtypes = [C.POINTER(C.c_int)]
    emu.phast_emu_check_plan_tables.restype = C.c_int
    assert emu.phast_emu_check_plan_tables(C.byref(n)) == 0
    assert n.value >= 70, n.value


def test_tuning_candidates_are_plans_and_cover_the_hand_ranked_tables(emu):
    """PlannerMode::Tune (csrc/tune.hpp) times the plans plan.hpp: enumerate_plans lists.  Every one of them must be a plan that
    exists as kernels and must survive its text form (the wisdom format); and the set must CONTAIN what rounds 2-4 found by
    hand-driven sweeps -- the single-transform table entries -- or the tuner could never rediscover them."""
    emu.phast_emu_enumerate_plans.argtypes = [C.c_uint, C.c_size_t, C.c_size_t, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    emu.phast_emu_enumerate_plans.restype = C.c_int
    has, bad = C.c_int(), C.c_int()
    # (length, element bytes, batch, a plan that must be among the candidates: plan.hpp single_plan / real_plan entries)
    cases = [(20, 8, 1, b"6,8,6@10,12,10:p8w"), (14, 8, 1, b"6,8@10,12:p16w"), (22, 8, 1, b"8,7,7@13,12,13:p16"),
             (19, 4, 1, b"6,7,6@11,11,11:p8"), (25, 4, 1, b"8,9,8@14,13,14:p16"), (20, 8, 8, b"10,10@13,13:p16"),
             (13, 8, 1, b"6,7@10,11:p8w"), (26, 8, 1, None), (28, 4, 1, None), (16, 4, 64, None)]
    for L, eb, batch, spec in cases:
        cnt = emu.phast_emu_enumerate_plans(L, eb, batch, spec, C.byref(has), C.byref(bad))
        assert bad.value == 0, (L, eb, batch)
        assert 8 <= cnt <= 4000, (L, eb, batch, cnt)
        if spec:
            assert has.value == 1, (L, eb, batch, spec)
    # nothing to enumerate below the multi-pass lengths
    assert emu.phast_emu_enumerate_plans(10, 8, 1, None, C.byref(has), C.byref(bad)) == 0


def test_tuning_result_check_stops_a_wrong_plan(emu):
    """Before a tuning run adopts a plan it compares the plan's output with the static rule's through per-transform digests
    (plan.hpp: digests_agree).  Two correct plans differ by rounding; a plan with a wrong geometry does not survive."""
    emu.phast_emu_digests_agree.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_size_t, C.c_size_t]
    emu.phast_emu_digests_agree.restype = C.c_int
    rng = np.random.default_rng(5)
    n, batch = 1 << 12, 3
    digs = []
    for b in range(batch):
        x = np.fft.fft(rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n))
        digs.append([x.real.sum(), x.imag.sum(), float(np.sum(np.abs(x) ** 2)), x.real[1]])
    a = np.array(digs, np.float64)

    def agree(c, eb):
        return emu.phast_emu_digests_agree(a.ctypes.data_as(C.POINTER(C.c_double)), np.ascontiguousarray(c).ctypes.data_as(C.POINTER(C.c_double)),
                                           batch, n, eb)

    assert agree(a.copy(), 8) == 1 and agree(a.copy(), 4) == 1
    r = a * (1 + 1e-12 * rng.standard_normal(a.shape))      # rounding-level differences: another plan, the same transform
    assert agree(r, 8) == 1
    r32 = a * (1 + 1e-6 * rng.standard_normal(a.shape))
    assert agree(r32, 4) == 1 and agree(r32, 8) == 0         # (f64 results that differ by 1e-6 are not the same transform)
    bad = a.copy(); bad[1, 2] *= 1.01                        # one transform's energy off by 1 %
    assert agree(bad, 8) == 0 and agree(bad, 4) == 0
    bad = a.copy(); bad[2, 3] += 0.5 * np.sqrt(a[2, 2] / n)  # one probed bin off by half an rms bin
    assert agree(bad, 4) == 0
    bad = a.copy(); bad[0, 0] = np.nan
    assert agree(bad, 8) == 0
    swapped = a[::-1].copy()                                 # the right transforms in the wrong places
    assert agree(swapped, 4) == 0


def _builtin_wisdom_plans(max_log=21, stride_above=3, cap_log=24):
    """(type, kind, caller's log2 length, plan text, fuse) of csrc/builtin_wisdom.inc -- every distinct plan up to 2^max_log,
    every stride_above-th of the longer ones up to 2^cap_log (emulating a thread at a time, a 2^20-point plan takes 0.2 s)."""
    import os
    import re
    path = os.path.join(os.path.dirname(__file__), "..", "phastft_amd", "csrc", "builtin_wisdom.inc")
    seen, out, long_ones = set(), [], 0
    for m in re.finditer(r'^"(f64|f32) (c2c|c2ci|r2c|c2r) (\d+) \d+ (\S+) fuse=([01])', open(path).read(), re.M):
        ty, kind, L, plan, fuse = m.group(1), m.group(2), int(m.group(3)), m.group(4), int(m.group(5))
        key = (ty, kind, L, plan, fuse)
        if plan == "heuristic" or key in seen or L > cap_log:
            continue
        seen.add(key)
        if L > max_log:
            long_ones += 1
            if long_ones % stride_above:
                continue
        out.append(key)
    return out


def test_builtin_wisdom_plans_thread_by_thread_vs_numpy(emu):
    """round 5: the tuner composes passes the hand-ranked tables never did (a tile log per pass: "8,8@12,13:p16"), and what it
    adopted on the GPU ships as built-in wisdom.  On the GPU a plan is adopted only after its output agreed with the static
    rule's (digests at two bins); here every distinct built-in plan up to 2^21 points and a third of those up to 2^24 runs
    through the kernels' own phase functions thread by thread, against numpy in float64: C2C plans in place, R2C plans
    through the fused last pass where the entry says so (else the inner transform from interleaved pairs), C2R plans through
    the fused first pass where that pass has one (else the inner transform)."""
    emu.phast_emu_set_plan.argtypes = [C.c_char_p]
    plans = _builtin_wisdom_plans()
    assert len(plans) >= 200, len(plans)
    ran = {"c2c": 0, "c2ci": 0, "r2c": 0, "c2r": 0, "r2c_fused": 0, "c2r_fused": 0}
    try:
        for ty, kind, L, plan, fuse in plans:
            dtype = np.float64 if ty == "f64" else np.float32
            tol = 1e-13 if ty == "f64" else 1e-5
            rng = np.random.default_rng(L * 131 + len(plan))
            assert emu.phast_emu_set_plan(plan.encode()) == 0, plan
            n = 1 << L

            def inner_c2c(m, interleaved_in):  # the plan as the complex transform of 2^m points it is inside a real one
                z = rng.uniform(-1, 1, 1 << m) + 1j * rng.uniform(-1, 1, 1 << m)
                a, b = z.real.astype(dtype), z.imag.astype(dtype)
                ref = np.fft.fft(a.astype(np.float64) + 1j * b.astype(np.float64))
                if interleaved_in and dtype == np.float32:
                    pairs = np.empty(2 << m, dtype)
                    pairs[0::2], pairs[1::2] = a, b
                    ore, oim = np.zeros(1 << m, dtype), np.zeros(1 << m, dtype)
                    p = lambda v: v.ctypes.data_as(C.c_void_p)
                    rc = emu.phast_emu_fft_f32_modes(p(pairs), None, 1, p(ore), p(oim), 0, m, 1.0, None, 0, 0)
                    a, b = ore, oim
                else:
                    rc = run(emu, a, b, 1)
                assert rc == 0, (ty, kind, L, plan, rc)
                got = a.astype(np.float64) + 1j * b.astype(np.float64)
                assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= tol, (ty, kind, L, plan)

            if kind in ("c2c", "c2ci"):
                inner_c2c(L, kind == "c2ci")
                ran[kind] += 1
            elif kind == "r2c":
                x = rng.uniform(-1, 1, n).astype(dtype)
                if fuse:
                    rc, ore, oim = _emu_r2c(emu, x)
                    assert rc == 0, (ty, kind, L, plan, rc)  # an entry that says fuse=1 names a plan whose last pass has the form
                    ref = np.fft.rfft(x.astype(np.float64))
                    got = ore.astype(np.float64) + 1j * oim.astype(np.float64)
                    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= tol, (ty, kind, L, plan)
                    assert oim[0] == 0 and oim[-1] == 0
                    ran["r2c_fused"] += 1
                else:
                    inner_c2c(L - 1, True)
                ran[kind] += 1
            else:
                spec = np.fft.rfft(rng.uniform(-1, 1, n))
                spec.imag[0] = spec.imag[-1] = 0.0
                ire, iim = spec.real.astype(dtype), spec.imag.astype(dtype)
                rc, out = _emu_c2r(emu, ire, iim, n)
                assert rc in (0, 3), (ty, kind, L, plan, rc)
                if rc == 0:
                    ref = np.fft.irfft(ire.astype(np.float64) + 1j * iim.astype(np.float64), n)
                    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= tol, (ty, kind, L, plan)
                    ran["c2r_fused"] += 1
                else:  # no fused form of its first pass: the library runs the preprocess sweep, then this plan
                    inner_c2c(L - 1, False)
                ran[kind] += 1
    finally:
        emu.phast_emu_set_plan(None)
    assert min(ran["c2c"], ran["r2c"], ran["c2r"]) >= 10 and ran["r2c_fused"] >= 3 and ran["c2r_fused"] >= 3, ran


def test_every_builtin_wisdom_line_is_a_candidate_of_its_tuning_run(emu):
    """csrc/builtin_wisdom.inc against csrc/plan.hpp, without a GPU: every line (all 425, every batch bucket) must name a plan the
    tuner would enumerate for that type, length and batch TODAY -- a plan that exists as kernels and fits the LDS.  On the
    GPU a line that no longer builds is skipped silently when a planner applies the wisdom (its speed-up is lost, nothing
    fails); here a tile shape removed from plan.hpp, or a wisdom file regenerated by another build, shows up as a failure."""
    import os
    import re

    emu.phast_emu_enumerate_plans.argtypes = [C.c_uint, C.c_size_t, C.c_size_t, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    emu.phast_emu_enumerate_plans.restype = C.c_int
    path = os.path.join(os.path.dirname(__file__), "..", "phastft_amd", "csrc", "builtin_wisdom.inc")
    has, bad = C.c_int(), C.c_int()
    lines = 0
    for m in re.finditer(r'^"(f64|f32) (c2c|c2ci|r2c|c2r) (\d+) (\d+) (\S+) fuse=([01])', open(path).read(), re.M):
        ty, kind, L, bucket, plan = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), m.group(5)
        if plan == "heuristic":
            continue
        inner = L - 1 if kind in ("r2c", "c2r") else L   # the real transforms' plans are the inner N/2-point transform's
        cnt = emu.phast_emu_enumerate_plans(inner, 8 if ty == "f64" else 4, 1 << bucket, plan.encode(), C.byref(has), C.byref(bad))
        assert cnt > 0 and bad.value == 0 and has.value == 1, (m.group(0), cnt, bad.value)
        lines += 1
    assert lines >= 300, lines
