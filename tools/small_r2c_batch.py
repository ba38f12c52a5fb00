#!/usr/bin/env python3
"""Batched small real transforms (R2C then C2R): throughput in real samples per second."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

for dtype, dt, Planner in (("f32", torch.float32, P.PlannerR2c32), ("f64", torch.float64, P.PlannerR2c64)):
    for L in (6, 8, 10, 12, 14, 15):
        n = 1 << L
        batch = (1 << 26) // n
        x = torch.empty(n * batch, dtype=dt, device="cuda")
        P.fill_uniform(x, None, n)
        ore = torch.empty(batch * (n // 2 + 1), dtype=dt, device="cuda")
        oim = torch.empty_like(ore)
        back = torch.empty_like(x)
        pl = Planner(n)
        P.r2c_fft_batched(x, ore, oim, pl, batch)
        P.c2r_fft_batched(ore, oim, back, pl, batch)
        err = float((back - x).abs().max())
        res = []
        for fn in (lambda: P.r2c_fft_batched(x, ore, oim, pl, batch), lambda: P.c2r_fft_batched(ore, oim, back, pl, batch)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5)
        sz = 4 if dtype == "f32" else 8
        alg = sz * n + 2 * sz * (n // 2 + 1)
        print(f"{dtype} n=2^{L} x{batch}: r2c {res[0]:.4f} ms {n * batch / res[0] / 1e6:7.1f} GS/s ({alg * batch / res[0] / 1e6:5.0f} GB/s)  "
              f"c2r {res[1]:.4f} ms {n * batch / res[1] / 1e6:7.1f} GS/s  roundtrip err {err:.2e}", flush=True)
