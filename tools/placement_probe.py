#!/usr/bin/env python3
"""Where does the run-to-run spread of one large transform come from?  One process: rings of buffers at different
addresses x planners of the SAME plan (each with its own scratch allocation), every (ring, planner) pair timed as a
graph replay over the ring.  Spread over planners = scratch placement, over rings = buffer placement, over processes
(run it twice) = the rest.  Rings come from torch's allocator ("t") or from hipExtMallocWithFlags(hipDeviceMallocContiguous)
("c": physically contiguous); every planner allocates its own scratch (hipMalloc).  Round 4 also ran planners with a
physically contiguous scratch through a library switch that is gone again (profiles/r04_placement_probe.log: always
slower than an average hipMalloc one).
    python tools/placement_probe.py f64 26 [rings=ttcc] [planners=4]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phastft_amd as P

dt_s, L = sys.argv[1], int(sys.argv[2])
ring_kinds = sys.argv[3] if len(sys.argv) > 3 else "ttcc"
planner_kinds = "m" * (int(sys.argv[4]) if len(sys.argv) > 4 else 4)
es = 8 if dt_s == "f64" else 4
dt = torch.float64 if es == 8 else torch.float32
Planner = P.PlannerDit64 if es == 8 else P.PlannerDit32
fft = P.fft_64_dit_with_planner if es == 8 else P.fft_32_dit_with_planner
n = 1 << L
sets = max(3, (3 << 29) // (2 * es * n))
hip = C.CDLL("libamdhip64.so")
torch.zeros(1, device="cuda")


class Ext:  # a device allocation torch can view (the CUDA array interface)
    def __init__(self, elems):
        self.p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(self.p), C.c_size_t(elems * es), C.c_uint(0x4))  # hipDeviceMallocContiguous
        if rc != 0:
            raise RuntimeError(f"hipExtMallocWithFlags(contiguous) -> {rc}")
        self.__cuda_array_interface__ = {"shape": (elems,), "typestr": "<f8" if es == 8 else "<f4", "data": (self.p.value, False), "version": 2}


keep, rings = [], []
for r, kind in enumerate(ring_kinds):
    elems = sets * n + 1024 * r   # different sizes: the allocator cannot hand a freed block back
    if kind == "c":
        a, b = Ext(elems), Ext(elems)
        keep += [a, b]
        re, im = torch.as_tensor(a, device="cuda"), torch.as_tensor(b, device="cuda")
    else:
        re, im = torch.empty(elems, dtype=dt, device="cuda"), torch.empty(elems, dtype=dt, device="cuda")
    rings.append((kind, re, im))
planners = []
for kind in planner_kinds:
    pl = Planner(n)
    fft(rings[0][1][:n], rings[0][2][:n], P.Direction.Forward, pl)   # the scratch is allocated by the first call
    torch.cuda.synchronize()
    planners.append((kind, pl))
print(f"2^{L} {dt_s}: {sets} sets per ring; plan {planners[0][1].describe().split('single=')[-1][:110]}")
for r, (kind, re, im) in enumerate(rings):
    print(f"ring {r} ({kind}): re at {re.data_ptr():#x} im at {im.data_ptr():#x}")
for rep in range(2):
    for r, (rk, re, im) in enumerate(rings):
        row = []
        for pk, pl in planners:
            views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(sets)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fft(*views[0], P.Direction.Forward, pl)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for a, b in views:
                    fft(a, b, P.Direction.Forward, pl)
            g.replay()
            best = 1e9
            for _ in range(3):
                P.fill_uniform(re, im, n)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, 1e3 * e0.elapsed_time(e1) / sets)
            del g
            row.append(best)
        print(f"rep {rep} ring {r} ({rk}): " + "  ".join(f"scratch {pk}: {t:8.1f}" for (pk, _), t in zip(planners, row)), flush=True)
