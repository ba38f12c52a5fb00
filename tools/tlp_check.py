#!/usr/bin/env python3
"""Does a second resident workgroup per CU hide the issue stalls of the latency plan?  batch = 1, 2, 3, 4 with 1024x4 tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P
from phastft_amd import _lib
lib = _lib.lib()
n = 1 << 20
for tl in (12, 13):
    pl = P.PlannerDit64(n)
    pl.set_plan((10, 10), tl)
    for batch in (1, 2, 3, 4, 8):
        for wg in (0, 2, 4):
            lib.phast_debug_set_wg_per_cu(wg)
            re = torch.empty(n * batch, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
            P.fill_uniform(re, im, n); pl.time_passes(re, im, n, reps=2); P.fill_uniform(re, im, n)
            ms = pl.time_passes(re, im, n, reps=10)
            print(f"tile_log={tl} batch={batch} wg/cu={wg or 'auto'}: pass_us={[round(m*1e3,1) for m in ms]} total {sum(ms)*1e3:.1f} us  per-transform {sum(ms)*1e3/batch:.1f} us", flush=True)
lib.phast_debug_set_wg_per_cu(0)
