"""Seeded random sweep over what the parametrised parity tests fix by hand: size, batch, distance between transforms,
precision and direction of device-resident batches -- in particular the batch counts around which the planner switches
between its single / latency / mid / throughput plans (api.hip: Planner::plan_for) -- each against numpy's pocketfft
in double precision (forward unnormalised, reverse scaled by 1/N: algorithms/dit.rs:297-331)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F64_REL, F32_REL = 1e-13, 1e-5


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
    for k, batch, pad, dt, reverse in _cases(0xF022 + chunk, 30):
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
        tol = F64_REL if dt == "f64" else F32_REL
        for b in range(batch):
            sl = slice(b * dist, b * dist + n)
            z = re[sl].astype(np.float64) + 1j * im[sl].astype(np.float64)
            want = np.fft.ifft(z) if reverse else np.fft.fft(z)
            got = g_re[sl].astype(np.float64) + 1j * g_im[sl].astype(np.float64)
            err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300)
            assert err <= tol, (k, batch, pad, dt, reverse, b, err)
        # the padding between transforms is not touched
        if pad:
            for b in range(batch - 1):
                gap = slice(b * dist + n, (b + 1) * dist)
                assert np.array_equal(g_re[gap], re[gap]) and np.array_equal(g_im[gap], im[gap]), (k, batch, pad, dt, b)
