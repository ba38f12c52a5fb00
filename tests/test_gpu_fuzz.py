"""Seeded random sweep over what the parametrised parity tests fix by hand: size, batch, distance between transforms,
precision and direction of device-resident batches -- in particular the batch counts around which the planner switches
between its single / latency / mid / throughput plans (planner_plans.hpp: Planner::plan_for) -- each against numpy's pocketfft
in double precision (forward unnormalised, reverse scaled by 1/N: algorithms/dit.rs:297-331)."""
import os

import numpy as np
import pytest

from tests import tolerances as tol

pytestmark = pytest.mark.gpu

# PHAST_FUZZ_SEED=<int> shifts every seed, PHAST_FUZZ_SCALE=<int> multiplies the number of cases per chunk: the suite runs seed 0 x 1;
# tools/extended_fuzz.sh runs other seeds at x 4 (profiles/r06_extended_fuzz.log)
SEED = int(os.environ.get("PHAST_FUZZ_SEED", "0"))
SCALE = int(os.environ.get("PHAST_FUZZ_SCALE", "1"))


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        k = int(rng.integers(1, 23))
        cap = max(1, min(40, (1 << 24) >> k))          # at most 2^24 points per case
        batch = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 40])) if rng.random() < 0.7 else int(rng.integers(1, 41))
        batch = min(batch, cap)
        pad = int(rng.choice([0, 0, 0, 2, 8, 64]))      # distance between transforms = n + pad
        out.append((k, batch, pad, "f64" if rng.random() < 0.6 else "f32", bool(rng.random() < 0.35)))
    return out


@pytest.mark.parametrize("chunk", range(6))
def test_random_batches_against_pocketfft(gpu, chunk):
    import torch

    planners = {}
    for k, batch, pad, dt, reverse in _cases(0xF022 + chunk + 1000 * SEED, 30 * SCALE):
        n = 1 << k
        dist = n + pad
        np_t, t_t = (np.float64, torch.float64) if dt == "f64" else (np.float32, torch.float32)
        rng = np.random.default_rng(k * 1000 + batch * 10 + pad)
        total = (batch - 1) * dist + n
        re = rng.uniform(-1, 1, total).astype(np_t)
        im = rng.uniform(-1, 1, total).astype(np_t)
        d_re, d_im = torch.from_numpy(re.copy()).cuda(), torch.from_numpy(im.copy()).cuda()
        key = (k, dt)
        if key not in planners:
            planners[key] = (gpu.PlannerDit64 if dt == "f64" else gpu.PlannerDit32)(n)
        gpu.fft_dit_batched(d_re, d_im, n, gpu.Direction.Reverse if reverse else gpu.Direction.Forward, planners[key], dist=dist)
        g_re, g_im = d_re.cpu().numpy(), d_im.cpu().numpy()
        for b in range(batch):
            sl = slice(b * dist, b * dist + n)
            z = re[sl].astype(np.float64) + 1j * im[sl].astype(np.float64)
            want = np.fft.ifft(z) if reverse else np.fft.fft(z)
            got = g_re[sl].astype(np.float64) + 1j * g_im[sl].astype(np.float64)
            # against float64 pocketfft nothing of the reference's needs absorbing: the measured-error formulas, every bin
            tol.check_c(f"fuzz_c2c k={k} batch={batch} pad={pad} rev={int(reverse)} b={b}", dt, k, got, want)
        # the padding between transforms is not touched
        if pad:
            for b in range(batch - 1):
                gap = slice(b * dist + n, (b + 1) * dist)
                assert np.array_equal(g_re[gap], re[gap]) and np.array_equal(g_im[gap], im[gap]), (k, batch, pad, dt, b)


@pytest.mark.parametrize("chunk", range(3))
def test_random_real_batches_against_pocketfft(gpu, chunk):
    """R2C and C2R batches (inputs n apart, spectra n/2 + 1 apart): rfft / irfft in double precision.  The reference's
    R2C twiddles are an f64 rotation recurrence (planner.rs:120-162) -- the GPU tables are exact to rounding, so f64
    stays within the f64 formula of tests/tolerances.py of pocketfft (tests/test_gpu_parity.py compares with the oracle's drift
    separately)."""
    import torch

    rng0 = np.random.default_rng(0xBEA1 + chunk + 1000 * SEED)
    for _ in range(20 * SCALE):
        k = int(rng0.integers(2, 22))
        n = 1 << k
        batch = int(min(rng0.choice([1, 2, 3, 5, 8, 16, 17, 33]), max(1, (1 << 23) >> k)))
        dt = "f64" if rng0.random() < 0.6 else "f32"
        np_t = np.float64 if dt == "f64" else np.float32
        rng = np.random.default_rng(k * 100 + batch)
        x = rng.uniform(-1, 1, batch * n).astype(np_t)
        planner = (gpu.PlannerR2c64 if dt == "f64" else gpu.PlannerR2c32)(n)
        half = n // 2 + 1
        d_x = torch.from_numpy(x.copy()).cuda()
        d_re = torch.empty(batch * half, dtype=d_x.dtype, device="cuda")
        d_im = torch.empty_like(d_re)
        gpu.r2c_fft_batched(d_x, d_re, d_im, planner, batch)
        assert torch.equal(d_x.cpu(), torch.from_numpy(x)), "R2C must not modify its input (r2c.rs:535)"
        g_re, g_im = d_re.cpu().numpy().astype(np.float64), d_im.cpu().numpy().astype(np.float64)
        for b in range(batch):
            want = np.fft.rfft(x[b * n:(b + 1) * n].astype(np.float64))
            got = g_re[b * half:(b + 1) * half] + 1j * g_im[b * half:(b + 1) * half]
            tol.check_c(f"fuzz_r2c k={k} batch={batch} b={b}", dt, k, got, want)
        # C2R of a random Hermitian-consistent spectrum
        s_re = rng.uniform(-1, 1, batch * half).astype(np_t)
        s_im = rng.uniform(-1, 1, batch * half).astype(np_t)
        s_im[0::half] = 0   # irfft ignores Im(DC) and Im(Nyquist); the reference's formulas use them (r2c.rs:263-489,
        s_im[half - 1::half] = 0  # covered against the oracle in tests/test_gpu_parity_r2.py): keep the comparison Hermitian
        d_out = torch.empty(batch * n, dtype=d_x.dtype, device="cuda")
        gpu.c2r_fft_batched(torch.from_numpy(s_re.copy()).cuda(), torch.from_numpy(s_im.copy()).cuda(), d_out, planner, batch)
        g = d_out.cpu().numpy().astype(np.float64)
        for b in range(batch):
            spec = s_re[b * half:(b + 1) * half].astype(np.float64) + 1j * s_im[b * half:(b + 1) * half].astype(np.float64)
            want = np.fft.irfft(spec, n)
            tol.check_real(f"fuzz_c2r k={k} batch={batch} b={b}", dt, k, g[b * n:(b + 1) * n], want)
