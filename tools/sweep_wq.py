#!/usr/bin/env python3
"""Single-transform f64: the library's default plan against forced wave/quad-tile plans (HIP-graph timing on a cold ring)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

CASES = {20: [((8, 6, 6), (12, 10, 10), 3 | 16), ((8, 6, 6), (12, 10, 10), 4 | 16), ((6, 8, 6), (10, 12, 10), 3 | 16), ((7, 7, 6), (11, 11, 10), 3 | 16),
              ((7, 6, 7), (11, 10, 11), 3 | 16), ((6, 7, 7), (10, 11, 11), 3 | 16), ((6, 7, 7), (10, 11, 11), 4 | 16), ((7, 7, 6), (11, 11, 10), 4 | 16)],
         19: [((6, 7, 6), (10, 11, 10), 3 | 16), ((7, 6, 6), (11, 10, 10), 3 | 16)],
         21: [((7, 8, 6), (11, 12, 10), 3 | 16), ((8, 7, 6), (12, 11, 10), 3 | 16), ((7, 8, 6), (12, 12, 10), 3 | 16), ((8, 7, 6), (12, 12, 10), 3 | 16)],
         22: [((8, 8, 6), (12, 12, 10), 3 | 16), ((8, 8, 6), (13, 12, 10), 4 | 16)],
         23: [((8, 8, 7), (12, 12, 11), 3 | 16), ((8, 8, 7), (12, 12, 12), 3 | 16), ((7, 8, 8), (12, 12, 12), 3 | 16), ((8, 7, 8), (12, 12, 12), 3 | 16)]}
ONLY = [int(a) for a in sys.argv[1:]]
for L, plans in CASES.items():
    if ONLY and L not in ONLY:
        continue
    n = 1 << L
    ring = max(4, min(40, (1 << 30) // (16 * n)))
    re = torch.empty(ring * n, dtype=torch.float64, device="cuda")
    im = torch.empty_like(re)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]

    def graph_us(pl):
        P.fill_uniform(re, im, n)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            P.fft_64_dit_with_planner(*views[0], P.Direction.Forward, pl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(4):
                for i in range(ring):
                    P.fft_64_dit_with_planner(*views[i], P.Direction.Forward, pl)
        best = 1e9
        for _ in range(3):
            P.fill_uniform(re, im, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / (4 * ring))
        return best

    pl = P.PlannerDit64(n)
    print(f"2^{L}: default {graph_us(pl):8.2f} us  {pl.describe()}", flush=True)
    for lrs, tls, lp in plans:
        pl = P.PlannerDit64(n)
        try:
            pl.set_plan(lrs, tls, lp)
        except (P.PhastPanic, P.PhastHipError) as e:
            print("   ", lrs, tls, hex(lp), "not instantiable", e)
            continue
        print(f"    {lrs} {tls} {hex(lp)}: {graph_us(pl):8.2f} us  {pl.describe()}", flush=True)
    del re, im, views
