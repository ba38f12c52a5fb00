#!/usr/bin/env python3
"""Single-transform f32: the library's default plan against forced plans built around the f32 WAVE tiles (64 rows x 32
columns, wave_fft.hpp; points code | 0x10; needs the experimental build: python -m phastft_amd.build --experimental, PHASTFT_HIP_LIB=phastft_amd/lib/libphastft_hip_exp.so) -- HIP-graph timing on a cold ring, as tools/sweep_wq.py does for f64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402

W = 16
CASES = {18: [((6, 6, 6), (11, 11, 11), 3 | W), ((9, 9), (12, 12), 3)],
         19: [((6, 7, 6), (11, 12, 11), 3 | W), ((6, 7, 6), (11, 11, 11), 3 | W), ((6, 7, 6), (11, 12, 11), 4 | W), ((7, 6, 6), (12, 11, 11), 3 | W)],
         20: [((6, 8, 6), (11, 12, 11), 3 | W), ((6, 8, 6), (11, 13, 11), 4 | W), ((6, 8, 6), (11, 12, 11), 4 | W), ((7, 7, 6), (12, 12, 11), 3 | W),
              ((6, 7, 7), (11, 12, 12), 3 | W), ((7, 6, 7), (12, 11, 12), 3 | W), ((6, 8, 6), (12, 12, 12), 3)],
         21: [((7, 8, 6), (12, 12, 11), 3 | W), ((8, 7, 6), (12, 12, 11), 3 | W), ((6, 9, 6), (11, 13, 11), 4 | W), ((7, 8, 6), (12, 13, 11), 4 | W),
              ((6, 8, 7), (11, 12, 12), 3 | W), ((6, 8, 7), (11, 13, 12), 4 | W)],
         22: [((8, 8, 6), (12, 12, 11), 3 | W), ((8, 8, 6), (13, 13, 11), 4 | W), ((6, 8, 8), (11, 13, 13), 4 | W), ((7, 8, 7), (13, 13, 13), 4)],
         23: [((8, 9, 6), (13, 13, 11), 4 | W), ((6, 9, 8), (11, 13, 13), 4 | W), ((8, 8, 7), (12, 12, 12), 4), ((8, 8, 7), (13, 13, 13), 4)]}
ONLY = [int(a) for a in sys.argv[1:]]
for L, plans in CASES.items():
    if ONLY and L not in ONLY:
        continue
    n = 1 << L
    ring = max(4, min(80, (1 << 30) // (8 * n)))
    re = torch.empty(ring * n, dtype=torch.float32, device="cuda")
    im = torch.empty_like(re)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]

    def graph_us(pl):
        P.fill_uniform(re, im, n)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            P.fft_32_dit_with_planner(*views[0], P.Direction.Forward, pl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(4):
                for i in range(ring):
                    P.fft_32_dit_with_planner(*views[i], P.Direction.Forward, pl)
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

    pl = P.PlannerDit32(n)
    print(f"2^{L}: default {graph_us(pl):8.2f} us  {pl.describe()}", flush=True)
    for lrs, tls, lp in plans:
        pl = P.PlannerDit32(n)
        try:
            pl.set_plan(lrs, tls, lp)
        except (P.PhastPanic, P.PhastHipError) as e:
            print("   ", lrs, tls, hex(lp), "not instantiable", e)
            continue
        ms = pl.time_passes(views[0][0], views[0][1], n, reps=3)
        print(f"    {lrs} {tls} {hex(lp)}: {graph_us(pl):8.2f} us  passes {[round(1e3 * x, 2) for x in ms]}  {pl.describe()}", flush=True)
    del re, im, views
