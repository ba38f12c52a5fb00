#!/usr/bin/env python3
"""Why do replays of the same K-step graph differ (f32 2^20: min 19.3 us per step, median 22.5 in tools/ab_single.py)?
Times ten replays of one graph (K = 20 single transforms on a cold ring) in three settings:
  a. synchronize, replay                 (the GPU idles for the host's round trip before every replay)
  b. re-fill the ring, synchronize, replay (bench.py's and ab_single's protocol)
  c. re-fill the ring, replay behind it    (no idle gap: the graph is queued while the fill kernel runs)
  d. re-fill, then READ 768 MiB of another buffer (the fill's dirty lines leave the Infinity Cache), synchronize, replay
  e. the same without the synchronize
    python tools/replay_variance.py [f32|f64] [log2 N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
from bench import capture_steps

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << L
f64 = dt == "f64"
tdt = torch.float64 if f64 else torch.float32
pl = (P.PlannerDit64 if f64 else P.PlannerDit32)(n)
fft = P.fft_64_dit_with_planner if f64 else P.fft_32_dit_with_planner
K = 20
ring = (640 << 20) // (2 * (8 if f64 else 4) * n) + 1
re = torch.empty(ring * n, dtype=tdt, device="cuda"); im = torch.empty_like(re)
P.fill_uniform(re, im, n, seed=1, first_id=0)
views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]
step = lambda i: fft(*views[i % ring], P.Direction.Forward, pl)
for i in range(3): step(i)
torch.cuda.synchronize()
g, _ = capture_steps(torch, P, step, 3, K, touch=lambda: step(0))


def timed(pre):
    out = []
    for _ in range(10):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / K)
    return " ".join(f"{x:6.2f}" for x in out)


refill = lambda: P.fill_uniform(re, im, n, seed=1, first_id=0)
dn = 1 << 20
drain_re = torch.zeros(48 * dn * (1 if f64 else 2), dtype=tdt, device="cuda"); drain_im = torch.zeros_like(drain_re)   # 2 x 384 MiB, read-only
drain = lambda: P.digest(drain_re, drain_im, dn)
print(dt, L, pl.describe_call(1, 0))
print("a. sync, replay           :", timed(torch.cuda.synchronize))
print("b. refill, sync, replay   :", timed(lambda: (refill(), torch.cuda.synchronize())))
print("c. refill, replay queued  :", timed(refill))
print("a. sync, replay           :", timed(torch.cuda.synchronize))
print("d. refill, drain, sync, replay :", timed(lambda: (refill(), drain(), torch.cuda.synchronize())))
print("e. refill, drain, replay queued:", timed(lambda: (refill(), drain())))
print("b. refill, sync, replay   :", timed(lambda: (refill(), torch.cuda.synchronize())))
