"""How much of the headline's K-step timed region is the graph's first launch?  (the driver's
20-step protocol gave 26.25 us per step where 200 steps give 23.9.)  Same protocol as bench.py's headline, K steps
captured into one graph, timed wall-clock between synchronisations:
    cold    -- capture, instantiate, timed replay                     (round 2's protocol)
    upload  -- + hipGraphUpload before the timed region               (bench.py now)
    warm    -- + one untimed replay before the timed one              (the floor: nothing one-time left)
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

import phastft_amd as P

N = 1 << 20
torch.cuda.set_device(0)
planner = P.PlannerDit64(N)
ring = 240
re = torch.empty(ring * N, dtype=torch.float64, device="cuda")
im = torch.empty_like(re)
P.fill_uniform(re, im, N)
views = [(re[i * N:(i + 1) * N], im[i * N:(i + 1) * N]) for i in range(ring)]


def step(i):
    r, m = views[i % ring]
    P.fft_64_dit_with_planner(r, m, P.Direction.Forward, planner)


for i in range(5):
    step(i)
torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step(0)
torch.cuda.synchronize()

tl = P.TransformList(views, N, planner)
for steps in (20, 200):
    res = []
    for rep in range(7):
        P.fill_uniform(re, im, N)
        tl.run(P.Direction.Forward, 5, steps)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tl.run(P.Direction.Forward, 5, steps)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    res.sort()
    print(f"K={steps:4d} eager-C: us/step min {res[0]:.2f} median {res[len(res)//2]:.2f} max {res[-1]:.2f}  (one call into the library enqueues the K transforms)", flush=True)
for steps in (20, 200):
    for mode in ("cold", "upload", "warm"):
        res = []
        for rep in range(7):
            P.fill_uniform(re, im, N)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(steps):
                    step(5 + i)
            if mode == "upload":
                assert P.graph_upload(g)
            if mode == "warm":
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e6)
            del g
        res.sort()
        print(f"K={steps:4d} {mode:7s}: us/step min {res[0]:.2f} median {res[len(res)//2]:.2f} max {res[-1]:.2f}", flush=True)
