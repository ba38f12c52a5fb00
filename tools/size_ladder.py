#!/usr/bin/env python3
"""Every entry point across the sizes: ONE transform at a time (C2C forward, R2C, C2R; f64 and f32), N = 2^L, timed as a
HIP-graph replay over a cold ring of distinct buffers (>= 1 GiB, so nothing is served from the L2 / Infinity Cache at the
large sizes).  Prints us per transform, GSamples/s and the algorithmic HBM rate of ONE read + ONE write of the data
(the one-pass ideal): the place to look for a size whose plan is out of line with its neighbours.
    python tools/size_ladder.py [lo hi]          # default 10 28"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 28)


def graph_time(calls, refill):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        calls[0]()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c in calls:
            c()
    g.replay()
    best = 1e9
    for _ in range(3):
        refill()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / len(calls))
    del g
    return best


print(f"{'N':>5} | " + " | ".join(f"{name:>24}" for name in ("c2c f64", "c2c f32", "r2c f64", "c2r f64", "r2c f32", "c2r f32")))
print(f"{'':>5} | " + " | ".join(f"{'us':>8} {'GS/s':>7} {'TB/s':>7}" for _ in range(6)))
for L in range(lo, hi + 1):
    n = 1 << L
    cells = []
    for kind, dt in (("c2c", torch.float64), ("c2c", torch.float32), ("r2c", torch.float64), ("c2r", torch.float64),
                     ("r2c", torch.float32), ("c2r", torch.float32)):
        es = 8 if dt == torch.float64 else 4
        sfx = "64" if es == 8 else "32"
        try:
            if kind == "c2c":
                ring = max(2, min(128, (1 << 30) // (2 * es * n)))
                re = torch.empty(ring * n, dtype=dt, device="cuda")
                im = torch.empty_like(re)
                pl = (P.PlannerDit64 if es == 8 else P.PlannerDit32)(n)
                fn = P.fft_64_dit_with_planner if es == 8 else P.fft_32_dit_with_planner
                calls = [(lambda i=i: fn(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n], P.Direction.Forward, pl)) for i in range(ring)]
                us = graph_time(calls, lambda: P.fill_uniform(re, im, n))
                bytes_ = 4 * es * n
            else:
                h = n // 2 + 1
                ring = max(2, min(128, (1 << 30) // (2 * es * n)))
                x = torch.empty(ring * n, dtype=dt, device="cuda")
                sr = torch.empty(ring * h, dtype=dt, device="cuda")
                si = torch.empty_like(sr)
                pl = (P.PlannerR2c64 if es == 8 else P.PlannerR2c32)(n)
                if kind == "r2c":
                    fn = P.r2c_fft_f64_with_planner if es == 8 else P.r2c_fft_f32_with_planner
                    calls = [(lambda i=i: fn(x[i * n:(i + 1) * n], sr[i * h:(i + 1) * h], si[i * h:(i + 1) * h], pl)) for i in range(ring)]
                    us = graph_time(calls, lambda: x.uniform_(-1, 1))
                else:
                    fn = P.c2r_fft_f64_with_planner if es == 8 else P.c2r_fft_f32_with_planner
                    calls = [(lambda i=i: fn(sr[i * h:(i + 1) * h], si[i * h:(i + 1) * h], x[i * n:(i + 1) * n], pl)) for i in range(ring)]

                    def refill():
                        sr.uniform_(-1, 1)
                        si.uniform_(-1, 1)
                    us = graph_time(calls, refill)
                bytes_ = 2 * es * n
            cells.append(f"{us:8.2f} {n / us / 1e3:7.1f} {bytes_ / us / 1e6:7.2f}")
            del calls, pl
        except Exception as e:  # sizes an entry point refuses (R2C needs N >= 4 ...) or that do not fit
            cells.append(f"{'-':>8} {'-':>7} {type(e).__name__[:7]:>7}")
        torch.cuda.empty_cache()
    print(f"2^{L:<3} | " + " | ".join(cells), flush=True)
