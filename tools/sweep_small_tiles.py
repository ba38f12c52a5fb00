#!/usr/bin/env python3
"""Single-transform plan sweep with SMALL tiles (1024 / 2048 / 4096 points, several workgroups per CU):
per-pass kernel times (HIP events) and the whole transform replayed from a HIP graph on a cold ring.

    python tools/sweep_small_tiles.py [--log-n 20] [--dtype f64]
"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--top", type=int, default=12)
ap.add_argument("--ring", type=int, default=40)
a = ap.parse_args()
dt = torch.float64 if a.dtype == "f64" else torch.float32
Planner = P.PlannerDit64 if a.dtype == "f64" else P.PlannerDit32
fwd = P.fft_64_dit_with_planner if a.dtype == "f64" else P.fft_32_dit_with_planner
L, n, ring = a.log_n, 1 << a.log_n, a.ring

re = torch.empty(ring * n, dtype=dt, device="cuda")
im = torch.empty_like(re)
views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]


def graph_us(pl):
    """whole transforms back to back from a HIP graph, every one on a fresh buffer of the ring"""
    P.fill_uniform(re, im, n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd(*views[0], P.Direction.Forward, pl)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    P.fill_uniform(re, im, n)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(ring):
            fwd(*views[i], P.Direction.Forward, pl)
    best = 1e9
    for _ in range(3):
        P.fill_uniform(re, im, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / ring)
    return best


def passes_us(pl):
    P.fill_uniform(re, im, n)
    acc = None
    for i in range(ring):
        ms = pl.time_passes(views[i][0], views[i][1], n, reps=1)
        acc = ms if acc is None else [x + y for x, y in zip(acc, ms)]
    return [1e3 * x / ring for x in acc]


res = []
pl = Planner(n)
res.append((graph_us(pl), "default", passes_us(pl), pl.describe()))
print(f"default: {res[0][0]:.2f} us  passes {[round(x, 2) for x in res[0][2]]}  {res[0][3]}", flush=True)
splits = [s for k in (2, 3) for s in itertools.product(range(6, 11), repeat=k) if sum(s) == L]
for s in splits:
    for tls in itertools.product((10, 11, 12), repeat=len(s)):
        for lp in (3, 4, 5):
            pl = Planner(n)
            try:
                pl.set_plan(s, tls, lp)
            except (P.PhastPanic, P.PhastHipError):
                continue
            try:
                res.append((graph_us(pl), f"{s}@{tls}p{1 << lp}", passes_us(pl), pl.describe()))
            except Exception as e:  # noqa: BLE001
                print("failed", s, tls, lp, e, flush=True)
res.sort(key=lambda r: r[0])
for tot, name, ps, desc in res[:a.top]:
    print(f"  {name:40s} graph {tot:6.2f} us = {n / tot / 1e3:6.1f} GS/s   passes {[round(x, 2) for x in ps]} sum {sum(ps):.2f}")
    print(f"      {desc}", flush=True)
