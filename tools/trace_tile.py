#!/usr/bin/env python3
"""Phase timeline of the tile kernels for one transform (s_memtime stamps of every workgroup's FIRST tile).

    python -m phastft_amd.build --trace
    PHASTFT_HIP_LIB=phastft_amd/lib/libphastft_hip_trace.so python tools/trace_tile.py [--f32] 20 "7,6,7@11,10,11p8" ...

Prints, per pass, when each phase boundary is reached (ticks since the first workgroup entered the kernel; min /
mean / max over the workgroups): stamp 0 = entry, 1 = tables in LDS, 2 = tile loaded (+ pre-twiddle), then one per
radix step / exchange, last = stores retired.  The stamps drain vmcnt/lgkmcnt, so the instrumented kernel is slower
than the product kernel; the RELATIVE sizes of the phases are what this is for.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402
from phastft_amd import _lib  # noqa: E402

lib = _lib.lib()
F32 = "--f32" in sys.argv
ARGV = [a for a in sys.argv[1:] if a != "--f32"]
log_n = int(ARGV[0]) if ARGV else 20
specs = ARGV[1:] or ["default"]
DT = torch.float32 if F32 else torch.float64
n = 1 << log_n
for spec in specs:
    pl = (P.PlannerDit32 if F32 else P.PlannerDit64)(n)
    if spec != "default":
        lrs_s, rest = spec.split("@")
        tl_s, p_s = rest.split("p")
        pl.set_plan(tuple(int(x) for x in lrs_s.split(",")), tuple(int(x) for x in tl_s.split(",")),
                    {8: 3, 16: 4, 32: 5}[int(p_s)])
    ring = 40
    re = torch.empty(n * ring, dtype=DT, device="cuda")
    im = torch.empty_like(re)
    P.fill_uniform(re, im, n)
    for i in range(3):
        P.fft_dit_batched(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n], n, P.Direction.Forward, pl)
    torch.cuda.synchronize()
    trace = torch.zeros(3 * 4096 * 16, dtype=torch.int64, device="cuda")
    lib.phast_debug_set_trace(C.c_void_p(trace.data_ptr()))
    P.fft_dit_batched(re[(ring - 1) * n:ring * n], im[(ring - 1) * n:ring * n], n, P.Direction.Forward, pl)
    torch.cuda.synchronize()
    lib.phast_debug_set_trace(C.c_void_p(0))
    t = trace.cpu().numpy().reshape(3, 4096, 16).astype("float64")
    print(spec, pl.describe())
    for p in range(3):
        tp = t[p]
        tp = tp[tp[:, 0] != 0]
        if not len(tp):
            continue
        nst = int((tp[0] != 0).sum())
        tp = tp[:, :nst]
        t0 = tp[:, 0].min()
        rel = tp - t0
        print(f" pass {p}: {len(tp)} workgroups traced; stamp: min / mean / max ticks since first entry")
        for s in range(nst):
            print(f"   stamp {s:2d}: {rel[:, s].min():7.0f} {rel[:, s].mean():7.0f} {rel[:, s].max():7.0f}"
                  f"   (phase mean {0 if s == 0 else (tp[:, s] - tp[:, s - 1]).mean():6.0f})")
