#!/usr/bin/env python3
"""f32 plan sweep (C2C batch 2^20, single 2^23 = the R2C N=2^24 inner transform) and R2C/C2R end-to-end timing."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402


def run(log_n, batch, plans, reps):
    n = 1 << log_n
    re = torch.empty(n * batch, dtype=torch.float32, device="cuda")
    im = torch.empty_like(re)
    for lrs, tl in plans:
        pl = P.PlannerDit32(n)
        try:
            if lrs:
                pl.set_plan(lrs, tl)
        except P.PhastPanic:
            print(f"  plan {lrs}@{tl}: not available")
            continue
        P.fill_uniform(re, im, n)
        pl.time_passes(re, im, n, reps=1)
        P.fill_uniform(re, im, n)
        ms = pl.time_passes(re, im, n, reps=reps)
        tot = sum(ms)
        gbs = [16 * n * batch / (m * 1e-3) / 1e9 for m in ms]
        print(f"  f32 2^{log_n} x{batch} plan={lrs}@{tl}: pass_ms={[round(m, 4) for m in ms]} GB/s={[int(g) for g in gbs]} "
              f"total={tot:.4f} ms {n * batch / tot / 1e6:.1f} GS/s | {pl.describe()}", flush=True)


print("f32 batch 64 x 2^20")
run(20, 64, [((), 12), ((10, 10), 13), ((10, 10), 14), ((7, 7, 6), 13), ((7, 7, 6), 12), ((8, 6, 6), (13, 12, 12)),
             ((7, 7, 6), (14, 14, 12))], 3)
print("f32 single 2^23")
run(23, 1, [((), 12), ((8, 8, 7), 13), ((8, 8, 7), 14), ((9, 7, 7), 13), ((8, 8, 7), 12), ((9, 9, 5), 13)], 3)

print("R2C f32 N=2^24 end to end")
n = 1 << 24
x = torch.empty(n, dtype=torch.float32, device="cuda")
P.fill_uniform(x, None, n)
ore = torch.empty(n // 2 + 1, dtype=torch.float32, device="cuda")
oim = torch.empty_like(ore)
back = torch.empty(n, dtype=torch.float32, device="cuda")
pl = P.PlannerR2c32(n)
for name, fn in (("r2c", lambda: P.r2c_fft_f32_with_planner(x, ore, oim, pl)),
                 ("c2r", lambda: P.c2r_fft_f32_with_planner(ore, oim, back, pl))):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    alg = 4 * n + 8 * (n // 2 + 1)
    print(f"  {name}: {ms:.4f} ms  {n / ms / 1e6:.1f} GSamples/s (real samples)  algorithmic {alg / ms / 1e6:.0f} GB/s "
          f"frac {alg / (ms * 1e-3) / 8e12:.3f}")
