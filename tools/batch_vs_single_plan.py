#!/usr/bin/env python3
"""Batches of large f64 / f32 transforms: the library's choice for the batch (throughput plan) against the plan ranked for
ONE transform of that size (plan.hpp: single_plan), forced for the whole batch.  Interleaved, two planners per variant.
    python tools/batch_vs_single_plan.py f64 22 23 24 25 [--total 27]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

argv = sys.argv[1:]
total = 27
if "--total" in argv:
    i = argv.index("--total")
    total = int(argv[i + 1])
    del argv[i:i + 2]
dt_s = argv[0]
es = 8 if dt_s == "f64" else 4
dt = torch.float64 if es == 8 else torch.float32
Planner = P.PlannerDit64 if es == 8 else P.PlannerDit32
SINGLE = {("f64", 21): ((6, 8, 7), (10, 12, 12), 3 | 16), ("f64", 22): ((8, 7, 7), (13, 12, 13), 4), ("f64", 23): ((7, 9, 7), (13, 12, 13), 4),
          ("f64", 24): ((8, 9, 7), (13, 12, 13), 4), ("f64", 25): ((8, 9, 8), (12, 12, 14), 4), ("f32", 22): ((7, 8, 7), (13, 13, 13), 4),
          ("f32", 23): ((8, 8, 7), (12, 12, 12), 4), ("f32", 25): ((8, 9, 8), (14, 13, 14), 4)}
for L in [int(a) for a in argv[1:]]:
    n = 1 << L
    for batch in sorted({4, 1 << max(2, total - L)}):
        re = torch.empty(batch * n, dtype=dt, device="cuda")
        im = torch.empty_like(re)
        planners = []
        for c in range(2):
            planners.append(("library", Planner(n)))
            pl = Planner(n)
            lrs, tls, lp = SINGLE[(dt_s, L)]
            pl.set_plan(lrs, list(tls), lp)
            planners.append(("single plan", pl))
        times = [[] for _ in planners]
        for rnd in range(5):
            for k, (name, pl) in enumerate(planners):
                P.fill_uniform(re, im, n)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[k].append(1e3 * e0.elapsed_time(e1))
        for k in range(2):
            meds = [statistics.median(times[c * 2 + k]) for c in range(2)]
            print(f"{dt_s} 2^{L} x {batch:4d}: {planners[k][0]:12s} {meds[0]:9.1f} {meds[1]:9.1f} us  = {batch * n / min(meds) / 1e3:6.1f} GS/s", flush=True)
        del re, im, planners
        torch.cuda.empty_cache()
