#!/usr/bin/env python3
"""Phase timeline of the tile kernels for one transform (s_memtime stamps).

    python -m phastft_amd.build --trace && PHASTFT_HIP_LIB=phastft_amd/lib/libphastft_hip_trace.so python tools/trace_tile.py 20
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402
from phastft_amd import _lib  # noqa: E402

lib = _lib.lib()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
plans = [((10, 10), 12), ((10, 10), 13), ((7, 7, 6), 12)] if log_n == 20 else [((), 12)]
n = 1 << log_n
for lrs, tl in plans:
    pl = P.PlannerDit64(n)
    if lrs:
        pl.set_plan(lrs, tl)
    re = torch.empty(n * 8, dtype=torch.float64, device="cuda")
    im = torch.empty_like(re)
    P.fill_uniform(re, im, n)
    for i in range(3):
        P.fft_dit_batched(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n], n, P.Direction.Forward, pl)
    torch.cuda.synchronize()
    trace = torch.zeros(3 * 4096 * 16, dtype=torch.int64, device="cuda")
    lib.phast_debug_set_trace(C.c_void_p(trace.data_ptr()))
    P.fft_dit_batched(re[4 * n:5 * n], im[4 * n:5 * n], n, P.Direction.Forward, pl)
    torch.cuda.synchronize()
    lib.phast_debug_set_trace(C.c_void_p(0))
    t = trace.cpu().numpy().reshape(3, 4096, 16).astype("float64")
    print(pl.describe())
    for p in range(3):
        tp = t[p]
        tp = tp[tp[:, 0] != 0]
        if not len(tp):
            continue
        nst = int((tp[0] != 0).sum())
        tp = tp[:, :nst]
        t0 = tp[:, 0].min()
        d = tp[:, 1:] - tp[:, :-1]  # per-phase ticks (s_memtime = shader cycles on gfx950)
        print(f" pass {p}: {len(tp)} workgroups, entry spread {tp[:, 0].max() - t0:.0f} ticks, "
              f"kernel span {tp[:, -1].max() - t0:.0f} ticks")
        print(f"   phase ticks mean: {[int(x) for x in d.mean(0)]}")
        print(f"   phase ticks max : {[int(x) for x in d.max(0)]}")
