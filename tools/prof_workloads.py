#!/usr/bin/env python3
"""Small fixed workloads for `rocprofv3 --kernel-trace --stats` / `--pmc` runs (profiles/).

    python tools/prof_workloads.py single|batch|big|r2c|c2r|bitrev [--plan 10,10 --tile-log 13] [--iters K]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["single", "batch", "big", "r2c", "c2r", "bitrev"])
ap.add_argument("--plan", default="")
ap.add_argument("--tile-log", type=int, default=12)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="f64")
a = ap.parse_args()
dt = torch.float64 if a.dtype == "f64" else torch.float32
Planner = P.PlannerDit64 if a.dtype == "f64" else P.PlannerDit32
plan = tuple(int(x) for x in a.plan.split(",")) if a.plan else ()

if a.what in ("single", "batch", "big"):
    n = 1 << (26 if a.what == "big" else 20)
    batch = a.batch if a.what == "batch" else 1
    ring = a.iters if a.what == "single" else 1
    pl = Planner(n)
    if plan:
        pl.set_plan(plan, a.tile_log)
    print(pl.describe())
    re = torch.empty(n * batch * ring, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    P.fill_uniform(re, im, n)
    for i in range(a.iters):
        if a.what == "single":
            P.fft_dit_batched(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n], n, P.Direction.Forward, pl)
        else:
            if i:
                P.fill_uniform(re, im, n)
            P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
    torch.cuda.synchronize()
elif a.what == "r2c":
    n = 1 << 24
    pl = P.PlannerR2c32(n)
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    P.fill_uniform(x, None, n)
    ore = torch.empty(n // 2 + 1, dtype=torch.float32, device="cuda")
    oim = torch.empty_like(ore)
    for i in range(a.iters):
        P.r2c_fft_f32_with_planner(x, ore, oim, pl)
    torch.cuda.synchronize()
elif a.what == "c2r":
    n = 1 << 24
    pl = P.PlannerR2c32(n)
    ire = torch.empty(n // 2 + 1, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    iim = torch.empty_like(ire).uniform_(-1, 1)
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    for i in range(a.iters):
        P.c2r_fft_f32_with_planner(ire, iim, y, pl)
    torch.cuda.synchronize()
else:
    n = 26
    x = torch.arange(1 << n, dtype=dt, device="cuda")
    for i in range(a.iters):
        (P.bit_rev_bravo_f64 if a.dtype == "f64" else P.bit_rev_bravo_f32)(x, n)
    torch.cuda.synchronize()
print("done")
