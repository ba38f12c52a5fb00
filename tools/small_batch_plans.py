#!/usr/bin/env python3
"""Small batches (3 .. 64 transforms) of small transforms: the library's choice (latency / throughput plan by points in
flight) against the single-transform plan forced for the batch -- where is the crossover?  Graph replay over a cold ring.
    python tools/small_batch_plans.py f64 16 17 [--batches 3,4,8,16,32,64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

argv = sys.argv[1:]
batches = (3, 4, 8, 16, 32, 64)
if "--batches" in argv:
    i = argv.index("--batches")
    batches = tuple(int(b) for b in argv[i + 1].split(","))
    del argv[i:i + 2]
dt_s = argv[0]
es = 8 if dt_s == "f64" else 4
dt = torch.float64 if es == 8 else torch.float32
Planner = P.PlannerDit64 if es == 8 else P.PlannerDit32
SINGLE = {("f64", 14): ((6, 8), (10, 12), 4 | 16), ("f64", 15): ((7, 8), (10, 12), 3 | 16), ("f64", 16): ((8, 8), (11, 11), 3), ("f64", 17): ((8, 9), (11, 12), 3),
          ("f64", 19): ((6, 7, 6), (10, 11, 10), 3 | 16), ("f64", 20): ((6, 8, 6), (10, 12, 10), 3 | 16),
          ("f32", 14): ((7, 7), (11, 11), 3), ("f32", 15): ((8, 7), (12, 11), 3), ("f32", 19): ((6, 7, 6), (11, 11, 11), 3)}
for L in [int(a) for a in argv[1:]]:
    n = 1 << L
    for batch in batches:
        ring = max(2, min(64, (1 << 30) // (2 * es * n * batch)))
        re = torch.empty(ring * batch * n, dtype=dt, device="cuda")
        im = torch.empty_like(re)
        res = []
        for name in ("library", "single plan"):
            pl = Planner(n)
            if name != "library":
                lrs, tls, lp = SINGLE[(dt_s, L)]
                pl.set_plan(lrs, list(tls), lp)
            m = batch * n
            views = [(re[i * m:(i + 1) * m], im[i * m:(i + 1) * m]) for i in range(ring)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                P.fft_dit_batched(*views[0], n, P.Direction.Forward, pl)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for a, b in views:
                    P.fft_dit_batched(a, b, n, P.Direction.Forward, pl)
            g.replay()
            best = 1e9
            for _ in range(3):
                P.fill_uniform(re, im, n)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, 1e3 * e0.elapsed_time(e1) / ring)
            del g
            res.append(best)
        print(f"{dt_s} 2^{L} x {batch:3d}: library {res[0]:8.2f} us  single plan {res[1]:8.2f} us  ({100 * (res[0] / res[1] - 1):+5.1f} %)", flush=True)
        del re, im
