#!/usr/bin/env python3
"""One f64 transform of 2^27 / 2^28 / 2^29 points: forced three-pass plans (rows per pass, tile size, points per thread)
against the library's own choice -- where do the very large transforms lose their per-pass rate?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
for L in [int(x) for x in sys.argv[1:]] or [28]:
    n = 1 << L
    re = torch.empty(n, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
    plans = [((), 0, 0)]
    for lrs in ((10, 9, 9), (9, 10, 9), (9, 9, 10), (10, 10, 8), (10, 8, 10), (8, 10, 10), (9, 9, 9, ), (10, 10, 7), (10, 9, 8),
                (9, 9, 8), (9, 8, 9), (8, 9, 9), (8, 10, 8), (10, 8, 8), (8, 8, 10), (9, 10, 7), (7, 10, 9), (10, 6, 10), (9, 7, 10), (10, 7, 9),
                (8, 8, 8), (9, 8, 8), (8, 9, 8), (8, 8, 9), (9, 9, 7), (9, 7, 9), (7, 9, 9)):
        if sum(lrs) != L:
            continue
        for tl, lp in ((14, 5), (13, 4), (13, 5)):
            plans.append((lrs, tl, lp))
    for lrs, tl, lp in plans:
        pl = P.PlannerDit64(n)
        try:
            if lrs:
                pl.set_plan(lrs, tl, lp)
        except P.PhastPanic:
            continue
        P.fill_uniform(re, im, n)
        pl.time_passes(re, im, n, reps=1)
        P.fill_uniform(re, im, n)
        ms = pl.time_passes(re, im, n, reps=2)
        print(f"2^{L} plan={lrs or 'default'}@{tl}p{1 << lp if lp else ''}: {[round(m, 3) for m in ms]} ms  "
              f"{[int(32 * n / m / 1e6) for m in ms]} GB/s  total {sum(ms):.3f} ms = {n / sum(ms) / 1e6:.1f} GS/s  {pl.describe()[:160] if not lrs else ''}", flush=True)
        del pl
