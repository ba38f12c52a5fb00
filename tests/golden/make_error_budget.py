#!/usr/bin/env python3
"""Measures what the parity gates of tests/tolerances.py are tied to ("Next round" #3): per type, log2 N and entry
point the WORST rel-L2 and worst-bin error of the HIP path over several seeds and batch positions, on the GPU box:

    f64 against a long-double FFT (numpy pocketfft in np.longdouble) up to 2^22, against the oracle beyond;
    f32 against float64 pocketfft (an independent reference with nothing to absorb) and against the f32 oracle.

    python tests/golden/make_error_budget.py [out.json]      (default: tests/golden/error_budget.json; needs the GPU)

The committed error_budget.json is this script's output on an MI355X; tests/test_gpu_parity_r5.py::test_error_budget_holds
re-measures a sample of it and fails if the HIP path's error has grown past twice the committed value (or past the gate).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import phastft_amd as P  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import tolerances as tol  # noqa: E402


def ref_fft(h_re, h_im, L):
    if L <= 22:
        z = np.fft.fft(h_re.astype(np.longdouble) + 1j * h_im.astype(np.longdouble))
    else:
        z = np.fft.fft(h_re.astype(np.float64) + 1j * h_im.astype(np.float64))
    return np.asarray(z.real, np.float64), np.asarray(z.imag, np.float64)


def measure(lo=4, hi=26, seeds=(1, 2, 3)):
    out = {}

    def put(key, L, rel, worst):
        e = out.setdefault(key, {}).setdefault(str(L), {"rel": 0.0, "bin": 0.0})
        e["rel"], e["bin"] = max(e["rel"], float(rel)), max(e["bin"], float(worst))

    for dt in ("f64", "f32"):
        f64 = dt == "f64"
        ndt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
        for L in range(lo, hi + 1):
            n = 1 << L
            pl = (P.PlannerDit64 if f64 else P.PlannerDit32)(n)
            fft = P.fft_64_dit_with_planner if f64 else P.fft_32_dit_with_planner
            for seed in seeds if L <= 22 else seeds[:1]:
                rng = np.random.default_rng(seed * 1000 + L)
                h_re, h_im = rng.uniform(-1, 1, n).astype(ndt), rng.uniform(-1, 1, n).astype(ndt)
                r_re, r_im = ref_fft(h_re, h_im, L)
                # one transform (the single / latency plans) and the last of a batch (the mid / throughput plans)
                d_re, d_im = torch.from_numpy(h_re.copy()).cuda(), torch.from_numpy(h_im.copy()).cuda()
                fft(d_re, d_im, P.Direction.Forward, pl)
                g_re, g_im = d_re.cpu().numpy(), d_im.cpu().numpy()
                put(f"c2c_{dt}_single", L, tol.rel_l2(g_re, g_im, r_re, r_im), tol.max_bin_err(g_re, g_im, r_re, r_im))
                # inverse of the forward result against the input (round trip, lib.rs:381-425)
                fft(d_re, d_im, P.Direction.Reverse, pl)
                put(f"c2c_{dt}_roundtrip", L, tol.rel_l2(d_re.cpu().numpy(), d_im.cpu().numpy(), h_re, h_im),
                    tol.max_bin_err(d_re.cpu().numpy(), d_im.cpu().numpy(), h_re.astype(np.float64), h_im.astype(np.float64)))
                batch = max(2, min(1 << 12, (1 << 25) // n))
                re = torch.from_numpy(np.tile(h_re, batch)).cuda()
                im = torch.from_numpy(np.tile(h_im, batch)).cuda()
                P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
                g_re, g_im = re[-n:].cpu().numpy(), im[-n:].cpu().numpy()
                put(f"c2c_{dt}_batch", L, tol.rel_l2(g_re, g_im, r_re, r_im), tol.max_bin_err(g_re, g_im, r_re, r_im))
                del re, im
                if not f64 and L <= 24:  # against the f32 oracle (its 3.5-ulp planner twiddles are in this number)
                    o_re, o_im = h_re.copy(), h_im.copy()
                    O.fft_32_dit(o_re, o_im, O.FORWARD)
                    d_re.copy_(torch.from_numpy(h_re)); d_im.copy_(torch.from_numpy(h_im))
                    fft(d_re, d_im, P.Direction.Forward, pl)
                    g_re, g_im = d_re.cpu().numpy(), d_im.cpu().numpy()
                    put("c2c_f32_vs_oracle", L, tol.rel_l2(g_re, g_im, o_re, o_im), tol.max_bin_err(g_re, g_im, o_re, o_im))
                if f64 and L <= 24 and seed == seeds[0]:
                    o_re, o_im = h_re.copy(), h_im.copy()
                    O.fft_64_dit(o_re, o_im, O.FORWARD)
                    d_re.copy_(torch.from_numpy(h_re)); d_im.copy_(torch.from_numpy(h_im))
                    fft(d_re, d_im, P.Direction.Forward, pl)
                    g_re, g_im = d_re.cpu().numpy(), d_im.cpu().numpy()
                    put("c2c_f64_vs_oracle", L, tol.rel_l2(g_re, g_im, o_re, o_im), tol.max_bin_err(g_re, g_im, o_re, o_im))
                # real transforms of n points (n >= 4): r2c against rfft, c2r against irfft
                if L >= 2:
                    x = rng.uniform(-1, 1, n).astype(ndt)
                    X = np.fft.rfft(x.astype(np.longdouble if L <= 22 else np.float64))
                    X = np.asarray(X.real, np.float64) + 1j * np.asarray(X.imag, np.float64)
                    h = n // 2 + 1
                    rp = (P.PlannerR2c64 if f64 else P.PlannerR2c32)(n)
                    dx = torch.from_numpy(x.copy()).cuda()
                    sr = torch.empty(h, dtype=tdt, device="cuda"); si = torch.empty_like(sr)
                    P.r2c_fft_batched(dx, sr, si, rp, 1)
                    put(f"r2c_{dt}", L, tol.rel_l2(sr.cpu().numpy(), si.cpu().numpy(), X.real, X.imag),
                        tol.max_bin_err(sr.cpu().numpy(), si.cpu().numpy(), X.real, X.imag))
                    sr.copy_(torch.from_numpy(X.real.astype(ndt))); si.copy_(torch.from_numpy(X.imag.astype(ndt)))
                    want = np.fft.irfft(X.real.astype(ndt).astype(np.float64) + 1j * X.imag.astype(ndt).astype(np.float64), n)
                    P.c2r_fft_batched(sr, si, dx, rp, 1)
                    z = np.zeros(n)
                    put(f"c2r_{dt}", L, tol.rel_l2(dx.cpu().numpy(), z, want, z), tol.max_bin_err(dx.cpu().numpy(), z, want, z))
            print(dt, L, {k: v[str(L)] for k, v in out.items() if str(L) in v and dt in k}, flush=True)
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "error_budget.json")
    lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 26)
    res = {"device": P.device_info()["name"], "reference": "numpy pocketfft, long double up to 2^22 for f64, float64 for f32 and beyond",
           "inputs": "uniform [-1, 1), seeds 1..3 (one beyond 2^22)", "budget": measure(lo, hi)}
    with open(path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote", path)
