#!/usr/bin/env python3
"""Burst vs sustained: a d2d copy and the 1024 x 2^20 f64 batch, timed per repetition over ~0.5 s of back-to-back work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
dev = torch.device("cuda")
src = torch.empty(1 << 30, dtype=torch.float64, device=dev).fill_(1.0)   # 8 GiB
dst = torch.empty_like(src)
def timed(fn, reps):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
import time
time.sleep(2.0)
t = timed(lambda: dst.copy_(src), 40)
print("copy 8 GiB, GB/s per repetition:", [round(2 * src.numel() * 8 / x / 1e6) for x in t])
del src, dst
N = 1 << 20
pl = P.PlannerDit64(N)
re = torch.empty(1024 * N, dtype=torch.float64, device=dev); im = torch.empty_like(re)
P.fill_uniform(re, im, N)
P.fft_dit_batched(re, im, N, P.Direction.Forward, pl)
P.fill_uniform(re, im, N)
torch.cuda.synchronize()
time.sleep(2.0)
t = timed(lambda: P.fft_dit_batched(re, im, N, P.Direction.Forward, pl), 24)
print("1024 x 2^20 batch, GSamples/s per repetition:", [round(1024 * N / x / 1e6, 1) for x in t])
