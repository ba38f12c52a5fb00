#!/usr/bin/env python3
"""Confirm plan candidates against the library's default: interleaved repetitions in ONE process (fresh planner per
candidate, all measured `rounds` times alternately), median of the totals.  python tools/confirm_plans.py [f64|f32]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

F32 = len(sys.argv) > 1 and sys.argv[1] == "f32"
CAND = {
    False: {14: [((8, 6), 12, 3), ((7, 7), 13, 4)], 17: [((9, 8), 13, 5), ((7, 10), 13, 4)], 22: [((7, 8, 7), 13, 4), ((7, 9, 6), 12, 4)],
            23: [((7, 9, 7), 13, 4), ((7, 9, 7), 12, 4)], 24: [((7, 10, 7), 13, 4), ((7, 9, 8), 13, 4)],
            25: [((8, 9, 8), 13, 4), ((8, 10, 7), 13, 4), ((7, 10, 8), 13, 4)], 26: [((8, 10, 8), 13, 4), ((8, 10, 8), 14, 5), ((9, 9, 8), 13, 4)],
            27: [((9, 10, 8), 13, 4), ((8, 10, 9), 14, 5)], 28: [((8, 10, 10), 14, 5), ((9, 10, 9), 14, 5)]},
    True: {14: [((7, 7), 13, 4), ((8, 6), 12, 3)], 18: [((9, 9), 15, 5), ((10, 8), 15, 5)], 21: [((7, 8, 6), 12, 3), ((7, 7, 7), 13, 4)],
           24: [((7, 9, 8), 13, 4), ((7, 10, 7), 13, 4)], 25: [((8, 9, 8), 13, 4), ((7, 10, 8), 13, 4)], 26: [((8, 10, 8), 13, 4), ((8, 10, 8), 14, 5)]},
}[F32]
dt = torch.float32 if F32 else torch.float64
Planner = P.PlannerDit32 if F32 else P.PlannerDit64
for L, cands in CAND.items():
    n = 1 << L
    batch = max(1, (1 << 26) // n)
    re = torch.empty(n * batch, dtype=dt, device="cuda"); im = torch.empty_like(re)
    planners = [("default", Planner(n))]
    for lrs, tl, lp in cands:
        pl = Planner(n)
        try:
            pl.set_plan(lrs, tl, lp)
        except P.PhastPanic:
            continue
        planners.append((f"{lrs}@{tl}p{1 << lp}", pl))
    tot = {name: [] for name, _ in planners}
    last = {}
    for rnd in range(4):
        for name, pl in planners:
            P.fill_uniform(re, im, n)
            ms = pl.time_passes(re, im, n, reps=2)
            if rnd:
                tot[name].append(sum(ms))
            last[name] = ms
    print(f"2^{L} x{batch} {'f32' if F32 else 'f64'}:", flush=True)
    for name, _ in planners:
        med = statistics.median(tot[name])
        print(f"    {name:24s} {n * batch / med / 1e6:7.1f} GS/s  (median of 3; last passes {[round(x, 4) for x in last[name]]})", flush=True)
    del re, im, planners
