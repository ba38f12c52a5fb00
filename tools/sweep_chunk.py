#!/usr/bin/env python3
"""Batch-chunk sweep: does keeping the inter-pass scratch inside the 256 MiB Infinity Cache pay?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

n, batch = 1 << 20, 256
re = torch.empty(n * batch, dtype=torch.float64, device="cuda")
im = torch.empty_like(re)
for plan, tl, lp in (((10, 10), 14, 5), ((10, 10), 13, 4), ((10, 10), 12, 3)):
    for mb in (16, 32, 64, 128, 256, 512, 4096):
        os.environ["PHAST_SCRATCH_MB"] = str(mb)
        pl = P.PlannerDit64(n)
        pl.set_plan(plan, tl, lp)
        P.fill_uniform(re, im, n)
        pl.time_passes(re, im, n, reps=1)
        P.fill_uniform(re, im, n)
        ms = pl.time_passes(re, im, n, reps=2)
        tot = sum(ms)
        print(f"plan={plan}@{tl}p{1 << lp} scratch={mb:5d} MiB ({mb // 16} transforms/chunk): pass_ms={[round(m, 3) for m in ms]} "
              f"total={tot:.3f} ms  {n * batch / tot / 1e6:.1f} GS/s  frac={32 * n * batch / (tot * 1e-3) / 8e12:.3f}", flush=True)
        del pl
