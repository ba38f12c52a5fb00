#!/usr/bin/env python3
"""Does the distance between the transforms of a batch matter?  1024 x 2^20 f64 with dist = n (packed) against
dist = n + pad elements (every transform starts a few 128-byte lines later than a multiple of 8 MiB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as C
import phastft_amd as P
from phastft_amd import _lib

n, batch = 1 << 20, 1024
pl = P.PlannerDit64(n)
for pad in (0, 16, 48, 80, 272, 0, 48):
    dist = n + pad
    re = torch.empty(dist * (batch - 1) + n, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
    re.uniform_(-1, 1); im.uniform_(-1, 1)
    P.fft_dit_batched(re, im, n, P.Direction.Forward, pl, dist=dist)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        re.uniform_(-1, 1); im.uniform_(-1, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        P.fft_dit_batched(re, im, n, P.Direction.Forward, pl, dist=dist)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"dist = n + {pad:4d}: {best:.3f} ms = {n * batch / best / 1e6:.1f} GS/s", flush=True)
    del re, im
