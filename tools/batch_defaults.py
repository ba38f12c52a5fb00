#!/usr/bin/env python3
"""The library's DEFAULT plan selection across batch sizes (f64): GSamples/s from per-pass HIP events, each batch on a cold
ring of buffers (no buffer transformed twice in a row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
for L in (19, 20, 21, 22):
    n = 1 << L
    pl = P.PlannerDit64(n)
    out = []
    for batch in (1, 2, 3, 4, 8, 16, 32):
        ring = max(2, min(16, (1 << 29) // (16 * n * batch)))
        re = torch.empty(ring * batch * n, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
        P.fill_uniform(re, im, n)
        acc = 0.0
        for rep in range(2):
            for i in range(ring):
                v = slice(i * batch * n, (i + 1) * batch * n)
                ms = pl.time_passes(re[v], im[v], n, reps=1)
                if rep: acc += sum(ms)
        out.append(f"x{batch}: {n * batch * ring / acc / 1e6:5.1f}")
        del re, im
    print(f"2^{L}: " + "  ".join(out) + "   " + pl.describe()[:60], flush=True)
