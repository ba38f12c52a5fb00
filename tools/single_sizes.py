#!/usr/bin/env python3
"""One f64 transform at a time, N = 2^L for L in argv (default 16..24): us per transform on a cold ring (>= 1 GiB of
distinct buffers), eager launches timed with one event pair around the whole ring (launch gaps included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
Ls = [int(a) for a in sys.argv[1:]] or list(range(16, 25))
for L in Ls:
    n = 1 << L
    ring = max(8, min(256, (1 << 30) // (16 * n)))
    pl = P.PlannerDit64(n)
    re = torch.empty(ring * n, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
    P.fill_uniform(re, im, n)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]
    best = 1e9
    for rep in range(4):
        P.fill_uniform(re, im, n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r, m in views:
            P.fft_64_dit_with_planner(r, m, P.Direction.Forward, pl)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / ring)
    d = pl.describe()
    kind = "single" if "single=" in d else "latency"
    print(f"2^{L}: {best:8.2f} us  {n / best / 1e3:6.1f} GS/s   {kind}={d.split(kind + '=')[1].split(' latency=')[0].split(' single=')[0] if kind + '=' in d else d}", flush=True)
