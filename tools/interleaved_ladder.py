#!/usr/bin/env python3
"""ONE transform per call, planar (fft_64_dit / fft_32_dit) against interleaved (fft_64_interleaved / fft_32_interleaved:
Complex<T> pairs in place), forward and reverse, N = 2^lo .. 2^hi: graph replay over a cold ring.  The plans were ranked on
planar data -- is any length out of line on pairs?      python tools/interleaved_ladder.py [lo hi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 26)


def graph_time(calls, refill):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        calls[0]()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c in calls:
            c()
    g.replay()
    best = 1e9
    for _ in range(3):
        refill()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / len(calls))
    return best


print(f"{'N':>5} | {'f64 planar fwd':>15} {'planar rev':>11} {'pairs fwd':>10} {'pairs rev':>10} | {'f32 planar fwd':>15} {'planar rev':>11} {'pairs fwd':>10} {'pairs rev':>10}   (us)")
for L in range(lo, hi + 1):
    n = 1 << L
    row = []
    for es, dt, cdt, Pl, fp, fi in ((8, torch.float64, torch.complex128, P.PlannerDit64, P.fft_64_dit_with_planner, P.fft_64_interleaved_with_planner),
                                    (4, torch.float32, torch.complex64, P.PlannerDit32, P.fft_32_dit_with_planner, P.fft_32_interleaved_with_planner)):
        ring = max(2, min(64, (1 << 30) // (2 * es * n)))
        re = torch.empty(ring * n, dtype=dt, device="cuda")
        im = torch.empty_like(re)
        z = torch.empty(ring * n, dtype=cdt, device="cuda")
        pl = Pl(n)
        for d in (P.Direction.Forward, P.Direction.Reverse):
            row.append(graph_time([(lambda i=i: fp(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n], d, pl)) for i in range(ring)],
                                  lambda: P.fill_uniform(re, im, n)))
        for d in (P.Direction.Forward, P.Direction.Reverse):
            row.append(graph_time([(lambda i=i: fi(z[i * n:(i + 1) * n], d, pl)) for i in range(ring)],
                                  lambda: torch.view_as_real(z).uniform_(-1, 1)))
        del re, im, z, pl
        torch.cuda.empty_cache()
    print(f"2^{L:<3} | {row[0]:15.2f} {row[1]:11.2f} {row[2]:10.2f} {row[3]:10.2f} | {row[4]:15.2f} {row[5]:11.2f} {row[6]:10.2f} {row[7]:10.2f}", flush=True)
