"""Round-4 GPU tests ("Next round" items 1, 3, 6 and ADVICE r03).

* `python bench.py --gpus N` launches its own ranks and can no longer report a one-GPU number as an N-GPU one.
* the fused C2R first pass at 2^24 ... 2^26 on ARBITRARY half-spectra (Im X[0], Im X[h] != 0): the reference's formula is
  defined for any input (r2c.rs:263-347) and the k = 0 <-> X[h] and k = h/2 address special cases of c2r_fused.hpp are
  where a non-Hermitian input bites.
* one planner under concurrent callers (planner.rs:38-39: the planner is a `&`-shared value).
* HIP graphs captured through a planner survive buffer growth on that planner.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain_env():
    """no launcher variables: what a user's shell (or the driver's N = 1 command) looks like"""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


# ---------------------------------------------------------------- bench.py --gpus N
def test_bench_self_launches_its_ranks(gpu):
    """plain `python bench.py --gpus 2 --same-gpu --backend gloo` (no torchrun): bench.py re-executes itself under
    torch.distributed.run with two ranks, both run BASELINE configs[4]'s code path on their shard, rank 0 prints ONE line
    with n_gpus = 2 (the size of the process group), digest_ok = the MIN all-reduce over both ranks' checks."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--backend", "gloo",
                        "--shard", "64", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       env=_plain_env(), timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["digest_ok"] is True, out["config"]
    assert out["config"]["transforms_per_step"] == 128 and out["steps"] == 3 and out["value"] > 1.0
    assert out["scaling"] == "weak" and "roofline" in out


def test_bench_refuses_more_gpus_than_the_box_has(gpu):
    import torch

    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 7), "--steps", "1"],
                       capture_output=True, text=True, env=_plain_env(), timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert f"{have} GPU(s) visible" in r.stderr and "nothing measured" in r.stderr, r.stderr[-2000:]
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


# ---------------------------------------------------------------- fused C2R on arbitrary half-spectra
def _c2r_model_f64(x_re, x_im, n):
    """float64 restatement of preprocess + inverse FFT + interleave (r2c.rs:263-489) with exact twiddles, for any input"""
    half = n // 2
    k = np.arange(half)
    first = x_re[:half] + 1j * x_im[:half]
    second = x_re[half - k] - 1j * x_im[half - k]
    w = 0.5 * np.exp(-2j * np.pi * k / n)
    zx = 0.5 * (first + second)
    d = first - second
    zy = (w.real * d.real + w.imag * d.imag) + 1j * (w.real * d.imag - w.imag * d.real)
    z = (zx.real - zy.imag) + 1j * (zx.imag + zy.real)
    zz = np.fft.ifft(z)
    out = np.empty(n)
    out[0::2] = zz.real
    out[1::2] = zz.imag
    return out


@pytest.mark.parametrize("k,batch,dt", [(24, 1, "f32"), (24, 1, "f64"), (26, 1, "f32"), (25, 1, "f64"), (23, 2, "f32"),
                                        (20, 32, "f32"), (19, 32, "f64"), (16, 512, "f32"), (21, 1, "f64")])
def test_c2r_fused_first_pass_non_hermitian_vs_oracle(gpu, oracle, k, batch, dt):
    """Half-spectra with random imaginary parts EVERYWHERE, incl. X[0] and X[h] (round 3 fed the fused kernel spectra of
    real signals only): every output of the first and last transform against the oracle's c2r (which accepts any input) and
    an independent float64 model, rel-L2 and the worst single sample."""
    import torch

    n = 1 << k
    h1 = n // 2 + 1
    ndt, tdt = (np.float64, torch.float64) if dt == "f64" else (np.float32, torch.float32)
    pl = (gpu.PlannerR2c64 if dt == "f64" else gpu.PlannerR2c32)(n)
    rng = np.random.default_rng(4000 + k)
    h_re = rng.uniform(-1, 1, h1 * batch).astype(ndt)
    h_im = rng.uniform(-1, 1, h1 * batch).astype(ndt)
    assert h_im[0] != 0 and h_im[h1 - 1] != 0
    ire, iim = torch.from_numpy(h_re).cuda(), torch.from_numpy(h_im).cuda()
    y = torch.full((n * batch,), float("nan"), dtype=tdt, device="cuda")
    gpu.c2r_fft_batched(ire, iim, y, pl, batch)
    torch.cuda.synchronize()
    assert np.array_equal(ire.cpu().numpy(), h_re) and np.array_equal(iim.cpu().numpy(), h_im)  # `&[T]`: untouched
    inner = pl.describe()
    ms = pl.time_c2r_passes(ire[:h1], iim[:h1], torch.empty(n, dtype=tdt, device="cuda"), reps=1)
    assert len(ms) <= 3 and (len(ms) == 2 or "3p[" in inner), (ms, inner)   # no preprocess sweep: the fused path ran
    tol_or, tol_m, tol_s = (1e-9, 1e-13, 1e-11) if dt == "f64" else (1e-5, 1e-5, 2e-3)
    for b in (0, batch - 1):
        s_re, s_im = h_re[b * h1:(b + 1) * h1], h_im[b * h1:(b + 1) * h1]
        want = np.zeros(n, ndt)
        (oracle.c2r_fft_f64 if dt == "f64" else oracle.c2r_fft_f32)(s_re.copy(), s_im.copy(), want)
        got = y[b * n:(b + 1) * n].cpu().numpy().astype(np.float64)
        model = _c2r_model_f64(s_re.astype(np.float64), s_im.astype(np.float64), n)
        den = np.sqrt(np.sum(model ** 2))
        rms = den / np.sqrt(n)
        assert np.sqrt(np.sum((got - want) ** 2)) / den <= tol_or, (b, inner)
        assert np.sqrt(np.sum((got - model) ** 2)) / den <= tol_m, (b, inner)
        assert np.max(np.abs(got - model)) / rms <= tol_s, (b, inner)
        assert np.max(np.abs(got - want.astype(np.float64))) / rms <= max(tol_s, 1e-7), (b, inner)
