#!/usr/bin/env python3
"""What a planner costs to make, with the built-in wisdom (its plans are built with the planner) and without:
    python tools/planner_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, phastft_amd as P
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for L in (12, 14, 16, 18, 20, 22, 24, 26):
    n = 1 << L
    row = []
    for name, ctor in (("PlannerDit64", P.PlannerDit64), ("PlannerDit32", P.PlannerDit32), ("PlannerR2c64", P.PlannerR2c64), ("PlannerR2c32", P.PlannerR2c32)):
        cell = []
        for wisdom in (True, False):
            P.wisdom_builtin(wisdom)
            pl = ctor(n)  # first of its size: kernels' attributes looked up once per process
            tuned = pl.describe().count("tuned:")
            del pl
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); pl = ctor(n); ts.append(time.perf_counter() - t0); del pl
            cell.append((1e3 * min(ts), tuned))
        P.wisdom_builtin(True)
        row.append(f"{name} {cell[0][0]:6.2f} ms with {cell[0][1]:2d} wisdom plans / {cell[1][0]:5.2f} ms without")
    print(f"2^{L}: " + " | ".join(row), flush=True)
