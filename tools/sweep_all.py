#!/usr/bin/env python3
"""Exhaustive plan sweep on the GPU: for every log2 N, every factorisation into 2 or 3 tile FFTs and every
(tile size, points per thread) that exists as a kernel; prints the ranking per size.

    python tools/sweep_all.py [--dtype f64|f32] [--lo 12] [--hi 27] [--single] [--total 26]
"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f64")
ap.add_argument("--lo", type=int, default=12)
ap.add_argument("--hi", type=int, default=27)
ap.add_argument("--single", action="store_true", help="one transform per size (latency regime)")
ap.add_argument("--total", type=int, default=26, help="log2 of the points per batch in throughput mode")
ap.add_argument("--top", type=int, default=4)
a = ap.parse_args()
dt = torch.float64 if a.dtype == "f64" else torch.float32
Planner = P.PlannerDit64 if a.dtype == "f64" else P.PlannerDit32

SHAPES = {(4, 12): range(6, 11), (4, 13): range(7, 11), (4, 14): range(8, 11), (3, 12): range(6, 11),
          (5, 14): range(8, 11), (5, 13): range(8, 11), (5, 12): range(10, 11)}
if a.dtype == "f32":
    SHAPES[(5, 15)] = range(8, 11)

for L in range(a.lo, a.hi + 1):
    n = 1 << L
    batch = 1 if a.single or L >= a.total else 1 << (a.total - L)
    re = torch.empty(n * batch, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    res = []
    pl = Planner(n)
    P.fill_uniform(re, im, n)
    pl.time_passes(re, im, n, reps=1)
    P.fill_uniform(re, im, n)
    reps = 20 if batch * n <= (1 << 22) else 3
    ms = pl.time_passes(re, im, n, reps=reps)
    res.append((sum(ms), "default", ms, pl.describe()))
    splits = [s for k in (2, 3) for s in itertools.product(range(6, 11), repeat=k) if sum(s) == L]
    for (lp, tl), lrs_ok in SHAPES.items():
        for s in splits:
            if any(lr not in lrs_ok for lr in s):
                continue
            pl = Planner(n)
            try:
                pl.set_plan(s, tl, lp)
            except (P.PhastPanic, P.PhastHipError):
                continue
            P.fill_uniform(re, im, n)
            pl.time_passes(re, im, n, reps=1)
            P.fill_uniform(re, im, n)
            ms = pl.time_passes(re, im, n, reps=reps)
            res.append((sum(ms), f"{s}@{tl}p{1 << lp}", ms, ""))
    res.sort(key=lambda r: r[0])
    dflt = [r for r in res if r[1] == "default"][0]
    print(f"2^{L} x{batch} {a.dtype}: default {dflt[0]:.4f} ms = {n * batch / dflt[0] / 1e6:.1f} GS/s  [{dflt[3]}]")
    for tot, name, ms, _ in res[:a.top]:
        print(f"    {name:28s} {tot:.4f} ms {n * batch / tot / 1e6:6.1f} GS/s  passes={[round(m, 4) for m in ms]}", flush=True)
    del re, im
