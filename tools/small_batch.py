#!/usr/bin/env python3
"""Batched throughput of the small-transform kernel (N <= 2^11, one pass through LDS) and of the sizes just above."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

for dtype, dt, Planner, bps in (("f64", torch.float64, P.PlannerDit64, 32), ("f32", torch.float32, P.PlannerDit32, 16)):
    for L in range(4, 14):
        n = 1 << L
        batch = (1 << 26) // n
        re = torch.empty(n * batch, dtype=dt, device="cuda")
        im = torch.empty_like(re)
        P.fill_uniform(re, im, n)
        pl = Planner(n)
        P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        P.fill_uniform(re, im, n)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{dtype} 2^{L} x{batch}: {ms:.4f} ms  {n * batch / ms / 1e6:7.1f} GS/s  {bps * n * batch / ms / 1e6:6.0f} GB/s algorithmic", flush=True)
