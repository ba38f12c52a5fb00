#!/usr/bin/env python3
"""The batched counterpart of tools/size_ladder.py: 2^27 points in flight per call (2 GiB of f64 planes: far beyond the
caches), every transform length N = 2^1 .. 2^24, C2C forward / R2C / C2R in f64 and f32: GSamples/s and the algorithmic
rate of ONE read + ONE write of the data.  Where is a length out of line with its neighbours?
    python tools/batch_ladder.py [lo hi]          (LADDER_TOTAL=22: 2^22 points in flight, timed as a graph over a cold ring)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 24)
TOTAL = int(os.environ.get("LADDER_TOTAL", "27"))  # log2 of the points in flight per call (22: the regime between one transform and a full chip)


RING = 1 if TOTAL >= 26 else 16   # small totals: 16 distinct buffer sets per graph, so that nothing is served from the caches


def timed(fn, refill):
    if RING == 1:
        fn(0)
        run = lambda: fn(0)
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(RING):
                fn(i)
        g.replay()
        run = g.replay
    best = 1e9
    for _ in range(3):
        refill()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / RING)
    return best


print(f"{'N':>5} | " + " | ".join(f"{name:>20}" for name in ("c2c f64", "c2c f32", "r2c f64", "c2r f64", "r2c f32", "c2r f32")))
print(f"{'':>5} | " + " | ".join(f"{'us':>8} {'GS/s':>6} {'TB/s':>4}" for _ in range(6)))
for L in range(lo, hi + 1):
    n = 1 << L
    batch = 1 << (TOTAL - L)
    cells = []
    for kind, dt in (("c2c", torch.float64), ("c2c", torch.float32), ("r2c", torch.float64), ("c2r", torch.float64),
                     ("r2c", torch.float32), ("c2r", torch.float32)):
        es = 8 if dt == torch.float64 else 4
        try:
            if kind == "c2c":
                m = batch * n
                re = torch.empty(RING * m, dtype=dt, device="cuda")
                im = torch.empty_like(re)
                pl = (P.PlannerDit64 if es == 8 else P.PlannerDit32)(n)
                us = timed(lambda i: P.fft_dit_batched(re[i * m:(i + 1) * m], im[i * m:(i + 1) * m], n, P.Direction.Forward, pl),
                           lambda: P.fill_uniform(re, im, n))
                bytes_ = 4 * es * n * batch
                del re, im
            else:
                if n < 4:
                    raise ValueError
                h = n // 2 + 1
                mx, mh = batch * n, batch * h
                x = torch.empty(RING * mx, dtype=dt, device="cuda")
                sr = torch.empty(RING * mh, dtype=dt, device="cuda")
                si = torch.empty_like(sr)
                pl = (P.PlannerR2c64 if es == 8 else P.PlannerR2c32)(n)
                if kind == "r2c":
                    us = timed(lambda i: P.r2c_fft_batched(x[i * mx:(i + 1) * mx], sr[i * mh:(i + 1) * mh], si[i * mh:(i + 1) * mh], pl, batch),
                               lambda: x.uniform_(-1, 1))
                else:
                    def refill():
                        sr.uniform_(-1, 1)
                        si.uniform_(-1, 1)
                    us = timed(lambda i: P.c2r_fft_batched(sr[i * mh:(i + 1) * mh], si[i * mh:(i + 1) * mh], x[i * mx:(i + 1) * mx], pl, batch), refill)
                bytes_ = 2 * es * n * batch
                del x, sr, si
            cells.append(f"{us:8.1f} {n * batch / us / 1e3:6.1f} {bytes_ / us / 1e6:4.2f}")
            del pl
        except Exception as e:
            cells.append(f"{'-':>8} {'-':>6} {'-':>4}")
        torch.cuda.empty_cache()
    print(f"2^{L:<3} | " + " | ".join(cells), flush=True)
