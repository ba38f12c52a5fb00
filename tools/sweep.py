#!/usr/bin/env python3
"""Plan / residency sweep on the GPU: per-pass HIP-event times for forced plans and workgroups-per-CU.

    python tools/sweep.py [--what batch20|big26|single20|all]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402
from phastft_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="all")
ap.add_argument("--dtype", default="f64")
a = ap.parse_args()
lib = _lib.lib()
dt = torch.float64 if a.dtype == "f64" else torch.float32
Planner = P.PlannerDit64 if a.dtype == "f64" else P.PlannerDit32
bps = 32 if a.dtype == "f64" else 16


def run(log_n, batch, plans, wgs, reps):
    n = 1 << log_n
    re = torch.empty(n * batch, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    for plan in plans:
        lrs, tl = plan[0], plan[1]
        lp = plan[2] if len(plan) > 2 else 4
        pl = Planner(n)
        try:
            if lrs:
                pl.set_plan(lrs, tl, lp)
        except P.PhastPanic:
            print(f"  plan {lrs}@{tl}: not available")
            continue
        for wg in wgs:
            lib.phast_debug_set_wg_per_cu(wg)
            P.fill_uniform(re, im, n)
            pl.time_passes(re, im, n, reps=1)  # warm (scratch, attributes)
            P.fill_uniform(re, im, n)
            ms = pl.time_passes(re, im, n, reps=reps)
            tot = sum(ms)
            gbs = [bps * n * batch / (m * 1e-3) / 1e9 for m in ms]
            print(f"  2^{log_n} x{batch} plan={lrs}@{tl}p{1 << lp} wg/cu={wg or 'auto'}: pass_ms={[round(m, 4) for m in ms]} "
                  f"pass_GB/s={[int(g) for g in gbs]} total={tot:.4f} ms {n * batch / tot / 1e6:.1f} GS/s "
                  f"| {pl.describe() if wg == wgs[0] else ''}", flush=True)
    lib.phast_debug_set_wg_per_cu(0)
    del re, im


if a.what == "p32":  # 32 points per thread (16384-point tiles with 512 threads) against the defaults
    print("batch of 256 x 2^20")
    run(20, 256, [((), 12), ((10, 10), 14, 5), ((10, 10), 13, 5), ((10, 10), 14, 4)], [0], 3)
    print("single 2^20")
    run(20, 1, [((), 12), ((10, 10), 12, 5), ((10, 10), 13, 5), ((10, 10), 14, 5)], [0], 20)
    print("batch of 64 x 2^18, 2^19")
    run(18, 1024, [((), 12), ((9, 9), 14, 5), ((9, 9), 13, 5), ((10, 8), 14, 5)], [0], 3)
    run(19, 512, [((), 12), ((10, 9), 14, 5), ((10, 9), 13, 5)], [0], 3)
    print("single 2^26, 2^24")
    run(26, 1, [((), 12), ((9, 9, 8), 14, 5), ((10, 8, 8), 14, 5), ((9, 9, 8), 13, 5)], [0], 3)
    run(24, 1, [((), 12), ((8, 8, 8), 14, 5), ((8, 8, 8), 13, 5)], [0], 3)
if a.what == "lr11":  # two passes for 2^21 / 2^22 with a 2048-point tile FFT
    run(21, 32, [((), 12), ((11, 10), 14, 5), ((10, 11), 14, 5), ((11, 10), 15, 5)], [0], 3)
    run(22, 16, [((), 12), ((11, 11), 14, 5), ((11, 11), 15, 5)], [0], 3)
if a.what in ("batch20", "all"):
    print("batch of 256 x 2^20")
    run(20, 256, [((), 12), ((10, 10), 13), ((7, 7, 6), 12), ((8, 6, 6), (13, 12, 12))], [0], 3)
if a.what in ("single20", "all"):
    print("single 2^20")
    run(20, 1, [((), 12), ((10, 10), 12, 3), ((10, 10), 12, 4), ((10, 10), 13, 4), ((7, 7, 6), 12, 3), ((7, 7, 6), 12, 4)], [0], 20)
if a.what in ("big26", "all"):
    print("single 2^26")
    run(26, 1, [((), 12), ((9, 9, 8), 13), ((10, 8, 8), 13), ((8, 9, 9), 13), ((9, 9, 8), 12)], [0], 3)
