#!/usr/bin/env python3
"""Persistent workgroups (blocks_per_cu x 256 CUs, each walking its run of tiles) against one workgroup per tile dispatched in
order (phast_debug_set_wg_per_cu: the grid is capped at the number of tiles) for the throughput plans: round 6 found the
plain copy 12 % and the plain write 33 % faster in the second form (csrc/probe.hip).  Per-pass event times.
    python tools/grid_order_probe.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
from phastft_amd import _lib

lib = _lib.lib()
for dt, L, batch in (("f64", 26, 1), ("f32", 26, 1), ("f64", 20, 256), ("f64", 24, 1), ("f32", 24, 4)):
    n = 1 << L
    tdt = torch.float64 if dt == "f64" else torch.float32
    pl = (P.PlannerDit64 if dt == "f64" else P.PlannerDit32)(n)
    re = torch.empty(batch * n, dtype=tdt, device="cuda").uniform_(-1, 1); im = torch.empty_like(re).uniform_(-1, 1)
    for wg in (0, 2, 4, 8, 16, 64, 4096):
        lib.phast_debug_set_wg_per_cu(C.c_int(wg))
        best = None
        for rep in range(4):
            re.uniform_(-1, 1); im.uniform_(-1, 1)
            torch.cuda.synchronize()
            ms = pl.time_passes(re, im, n, reps=1) if batch == 1 else None
            if ms is None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); P.fft_dit_batched(re, im, n, P.Direction.Forward, pl); e1.record(); torch.cuda.synchronize()
                ms = [e0.elapsed_time(e1)]
            if best is None or sum(ms) < sum(best):
                best = ms
        print(f"{dt} 2^{L} x {batch} wg/CU {wg or 'default':>7}: total {1e3 * sum(best):9.1f} us  passes " + " ".join(f"{1e3 * x:8.1f}" for x in best) + f"   {pl.describe_call(batch, 0)[:60]}", flush=True)
    lib.phast_debug_set_wg_per_cu(C.c_int(0))
    del re, im, pl
    torch.cuda.empty_cache()
