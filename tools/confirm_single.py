#!/usr/bin/env python3
"""Interleaved confirmation of single-transform plans: the library's choice (A) against forced plans (B, C ...), graphs over
one cold ring replayed alternately (A B C A B C ...), min and median of the rounds.  Cases are given as
    dtype:L:a,b,c@ta,tb,tc:lp[;a,b,c@...:lp]          e.g.  f64:26:9,9,8@14,14,14:5"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

rounds = 6
argv = sys.argv[1:]
sleep_ms, copies = 0, 1
while argv and argv[0].startswith("--"):
    if argv[0].startswith("--sleep="):  # idle gap before every timed replay: does a rested chip rank the plans differently?
        sleep_ms = int(argv[0].split("=")[1])
    elif argv[0].startswith("--copies="):  # planners per plan, each with its own scratch allocation: the time of a large transform
        copies = int(argv[0].split("=")[1])  # moves by +-5 % with WHERE the scratch landed (profiles/r04_placement_probe.log)
    argv = argv[1:]
import time
for case in argv:
    dt_s, L, rest = case.split(":", 2)
    L = int(L)
    es = 8 if dt_s == "f64" else 4
    dt = torch.float64 if es == 8 else torch.float32
    Planner = P.PlannerDit64 if es == 8 else P.PlannerDit32
    fft = P.fft_64_dit_with_planner if es == 8 else P.fft_32_dit_with_planner
    n = 1 << L
    ring = max(3, min(64, (3 << 29) // (2 * es * n)))
    re = torch.empty(ring * n, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]
    planners = []
    for c in range(copies):
        planners.append(("library", Planner(n)))
        for spec in rest.split(";"):
            geo, lp = spec.rsplit(":", 1)
            lrs, tls = geo.split("@")
            pl = Planner(n)
            pl.set_plan(tuple(int(x) for x in lrs.split(",")), [int(x) for x in tls.split(",")], int(lp))
            planners.append((spec, pl))
    graphs = []
    for name, pl in planners:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fft(*views[0], P.Direction.Forward, pl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for r, m in views:
                fft(r, m, P.Direction.Forward, pl)
        g.replay()
        graphs.append(g)
    times = [[] for _ in planners]
    for _ in range(rounds):
        for k, g in enumerate(graphs):
            P.fill_uniform(re, im, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            if sleep_ms:
                time.sleep(sleep_ms / 1e3)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(1e3 * e0.elapsed_time(e1) / ring)
    print(f"2^{L} {dt_s} (ring {ring}, {rounds} rounds, {sleep_ms} ms idle before each)")
    per_plan = len(planners) // copies
    for k in range(per_plan):
        name, pl = planners[k]
        d = pl.describe()
        shown = d.split("single=")[1] if name == "library" and "single=" in d else name
        meds = [statistics.median(times[c * per_plan + k]) for c in range(copies)]
        m = statistics.median(meds)
        print(f"   medians {' '.join(f'{x:9.2f}' for x in meds)}  -> {m:9.2f} us = {n / m / 1e3:6.1f} GS/s   {name if name != 'library' else 'library: ' + shown[:120]}", flush=True)
    del graphs, planners, re, im, views
    torch.cuda.empty_cache()
