#!/usr/bin/env python3
"""A batch of transforms: the library's plan against forced plans, interleaved (A B C A B C ...) on one buffer that is far
larger than the caches, two planners (scratch allocations) per plan.
    python tools/confirm_batch.py f64:20:256:10,10@14,13:5;10,10@13,13:4  [more cases]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

for case in sys.argv[1:]:
    dt_s, L, batch, rest = case.split(":", 3)
    L, batch = int(L), int(batch)
    es = 8 if dt_s == "f64" else 4
    dt = torch.float64 if es == 8 else torch.float32
    Planner = P.PlannerDit64 if es == 8 else P.PlannerDit32
    n = 1 << L
    re = torch.empty(batch * n, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    planners = []
    for c in range(2):
        planners.append(("library", Planner(n)))
        for spec in rest.split(";"):
            geo, lp = spec.rsplit(":", 1)
            lrs, tls = geo.split("@")
            pl = Planner(n)
            try:
                pl.set_plan(tuple(int(x) for x in lrs.split(",")), [int(x) for x in tls.split(",")], int(lp))
            except Exception:
                pl = None
            planners.append((spec, pl))
    times = [[] for _ in planners]
    for rnd in range(4):
        for k, (name, pl) in enumerate(planners):
            if pl is None:
                continue
            P.fill_uniform(re, im, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(1e3 * e0.elapsed_time(e1))
    per = len(planners) // 2
    print(f"{dt_s} 2^{L} x {batch}")
    for k in range(per):
        if planners[k][1] is None:
            print(f"   {planners[k][0]}: no such kernels")
            continue
        meds = [statistics.median(times[c * per + k]) for c in range(2)]
        name = planners[k][0]
        if name == "library":
            name = "library: " + planners[k][1].describe().split("throughput=")[1].split(" latency=")[0].split(" mid=")[0][:110]
        print(f"   {meds[0]:10.1f} {meds[1]:10.1f} us = {batch * n / min(meds) / 1e3:6.1f} GS/s   {name}", flush=True)
    del re, im, planners
    torch.cuda.empty_cache()
