#!/usr/bin/env python3
"""One f32 transform of 2^20 / 2^21 points on 2048-point tiles (64 x 32, 128 x 16 at 8 points per thread: twice the tiles of the
4096-point plans) against the library's plan: HIP-graph time per transform on a cold ring + kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

CASES = {20: [((6, 8, 6), (11, 12, 11), 3), ((7, 6, 7), (11, 11, 11), 3), ((7, 6, 7), (12, 11, 12), 3), ((6, 8, 6), (11, 11, 11), 3),
              ((7, 7, 6), (11, 11, 11), 3), ((6, 7, 7), (11, 11, 11), 3), ((6, 8, 6), (12, 11, 12), 3)],
         21: [((7, 7, 7), (11, 11, 11), 3), ((7, 7, 7), (12, 11, 12), 3)]}
for L, plans in CASES.items():
    n = 1 << L
    ring = max(4, min(48, (1 << 30) // (8 * n)))
    re = torch.empty(ring * n, dtype=torch.float32, device="cuda")
    im = torch.empty_like(re)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]

    def measure(pl):
        P.fill_uniform(re, im, n)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            P.fft_32_dit_with_planner(*views[0], P.Direction.Forward, pl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(ring):
                P.fft_32_dit_with_planner(*views[i], P.Direction.Forward, pl)
        best = 1e9
        for _ in range(4):
            P.fill_uniform(re, im, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / ring)
        P.fill_uniform(re, im, n)
        acc = None
        for i in range(min(ring, 16)):
            t = pl.time_passes(views[i][0], views[i][1], n, reps=1)
            acc = t if acc is None else [a + b for a, b in zip(acc, t)]
        return best, [round(1e3 * a / min(ring, 16), 2) for a in acc]

    for rnd in range(2):
        pl = P.PlannerDit32(n)
        us, ks = measure(pl)
        print(f"2^{L} f32 default: {us:7.2f} us = {n / us / 1e3:6.1f} GS/s kernels {ks}", flush=True)
        for lrs, tls, lp in plans:
            pl = P.PlannerDit32(n)
            try:
                pl.set_plan(lrs, list(tls), lp)
            except Exception as e:
                print("   ", lrs, tls, "refused")
                continue
            us, ks = measure(pl)
            print(f"    {lrs}@{tls}: {us:7.2f} us = {n / us / 1e3:6.1f} GS/s kernels {ks}", flush=True)
