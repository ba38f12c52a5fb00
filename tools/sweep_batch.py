#!/usr/bin/env python3
"""Where does the throughput plan start to beat the latency plan?  2^L transforms in batches of 1..64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
dt = torch.float64 if dtype == "f64" else torch.float32
Planner = P.PlannerDit64 if dtype == "f64" else P.PlannerDit32
n = 1 << L
plans = {20: [((6, 8, 6), (10, 12, 10), 3 | 16), ((6, 8, 6), 12, 3), ((10, 10), 13, 4), ((10, 10), 14, 5), ((10, 10), 14, 4), ((7, 7, 6), 12, 4)],
         19: [((6, 7, 6), (10, 11, 10), 3 | 16), ((10, 9), 12, 3), ((10, 9), 13, 4), ((10, 9), 14, 5)],
         22: [((8, 8, 6), (13, 12, 10), 4 | 16), ((8, 7, 7), 12, 3), ((8, 7, 7), 13, 4), ((8, 7, 7), 14, 5)],
         18: [((9, 9), 12, 3), ((9, 9), 13, 4), ((9, 9), 14, 5), ((6, 6, 6), 12, 3)],
         24: [((8, 8, 8), 12, 3), ((8, 8, 8), 13, 4), ((8, 8, 8), 14, 5), ((8, 9, 7), 12, 3)]}[L]
for batch in (1, 2, 3, 4, 6, 8, 16, 32, 64):
    if batch * n > (1 << 28):
        break
    re = torch.empty(n * batch, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    row = []
    for lrs, tl, lp in plans:
        pl = Planner(n)
        pl.set_plan(lrs, tl, lp)
        P.fill_uniform(re, im, n)
        pl.time_passes(re, im, n, reps=1)
        P.fill_uniform(re, im, n)
        ms = pl.time_passes(re, im, n, reps=10)
        row.append(f"{lrs}@{tl}p{1 << (lp & 15)}{chr(119) if lp & 16 else chr(32)}: {n * batch / sum(ms) / 1e6:5.1f}")
    print(f"2^{L} x{batch:3d} {dtype} GS/s  " + "  ".join(row), flush=True)
