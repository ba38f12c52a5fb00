#!/usr/bin/env python3
"""Per-kernel times of r2c_fft / c2r_fft on a cold ring (HIP events bound to the dispatches) for the library named by
PHASTFT_HIP_LIB -- the A/B companion of tools/cmp_throughput.py for the real transforms.
    python tools/real_passes.py [f32:24 f64:24 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

specs = [a for a in sys.argv[1:]] or ["f32:24", "f64:24"]
for spec in specs:
    dt_s, L = spec.split(":")
    n = 1 << int(L)
    dt = torch.float32 if dt_s == "f32" else torch.float64
    pl = (P.PlannerR2c32 if dt_s == "f32" else P.PlannerR2c64)(n)
    ring = max(3, min(16, (3 << 29) // (n * (4 if dt_s == "f32" else 8))))
    h1 = n // 2 + 1
    pitch = (h1 + 63) // 64 * 64
    x = torch.empty(ring * n, dtype=dt, device="cuda").uniform_(-1, 1)
    a = torch.empty(ring * pitch, dtype=dt, device="cuda").uniform_(-1, 1)
    b = torch.empty_like(a).uniform_(-1, 1)
    sets = [(x[i * n:(i + 1) * n], a[i * pitch:i * pitch + h1], b[i * pitch:i * pitch + h1]) for i in range(ring)]
    r2c = P.r2c_fft_f32_with_planner if dt_s == "f32" else P.r2c_fft_f64_with_planner
    c2r = P.c2r_fft_f32_with_planner if dt_s == "f32" else P.c2r_fft_f64_with_planner
    for name in ("r2c", "c2r"):
        def call(s):
            if name == "r2c":
                r2c(s[0], s[1], s[2], pl)
            else:
                c2r(s[1], s[2], s[0], pl)
        call(sets[0])
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for s in sets:
                call(s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / ring)
        acc = None
        for s in sets:
            t = pl.time_passes(s[0], s[1], s[2], reps=1) if name == "r2c" else pl.time_c2r_passes(s[1], s[2], s[0], reps=1)
            acc = t if acc is None else [p + q for p, q in zip(acc, t)]
        ms = [round(1e3 * v / ring, 1) for v in acc]
        print(f"{name} {dt_s} 2^{L}: {1e3 * best:8.1f} us = {n / best / 1e6:7.1f} GS/s  kernels {ms}", flush=True)
    del x, a, b, sets, pl
    torch.cuda.empty_cache()
