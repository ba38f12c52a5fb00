#!/usr/bin/env python3
"""The batched counterpart of tools/size_ladder.py: 2^27 points in flight per call (2 GiB of f64 planes: far beyond the
caches), every transform length N = 2^1 .. 2^24, C2C forward / R2C / C2R in f64 and f32: GSamples/s and the algorithmic
rate of ONE read + ONE write of the data.  Where is a length out of line with its neighbours?
    python tools/batch_ladder.py [lo hi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 24)
TOTAL = 27


def timed(fn, refill):
    fn()
    best = 1e9
    for _ in range(3):
        refill()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1))
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
                re = torch.empty(batch * n, dtype=dt, device="cuda")
                im = torch.empty_like(re)
                pl = (P.PlannerDit64 if es == 8 else P.PlannerDit32)(n)
                us = timed(lambda: P.fft_dit_batched(re, im, n, P.Direction.Forward, pl), lambda: P.fill_uniform(re, im, n))
                bytes_ = 4 * es * n * batch
                del re, im
            else:
                if n < 4:
                    raise ValueError
                h = n // 2 + 1
                x = torch.empty(batch * n, dtype=dt, device="cuda")
                sr = torch.empty(batch * h, dtype=dt, device="cuda")
                si = torch.empty_like(sr)
                pl = (P.PlannerR2c64 if es == 8 else P.PlannerR2c32)(n)
                if kind == "r2c":
                    us = timed(lambda: P.r2c_fft_batched(x, sr, si, pl, batch), lambda: x.uniform_(-1, 1))
                else:
                    def refill():
                        sr.uniform_(-1, 1)
                        si.uniform_(-1, 1)
                    us = timed(lambda: P.c2r_fft_batched(sr, si, x, pl, batch), refill)
                bytes_ = 2 * es * n * batch
                del x, sr, si
            cells.append(f"{us:8.1f} {n * batch / us / 1e3:6.1f} {bytes_ / us / 1e6:4.2f}")
            del pl
        except Exception as e:
            cells.append(f"{'-':>8} {'-':>6} {'-':>4}")
        torch.cuda.empty_cache()
    print(f"2^{L:<3} | " + " | ".join(cells), flush=True)
