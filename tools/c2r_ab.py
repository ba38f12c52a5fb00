"""C2R of one large half-spectrum (or a batch) on a cold ring: whole-call time with the preprocess fused into the first
pass (default) or as a sweep of its own (PHAST_C2R_FUSE=0).  tools/ -- run both inside ONE gpurun call:
   for i in 1 2; do PHAST_C2R_FUSE=1 python tools/c2r_ab.py 20 24; PHAST_C2R_FUSE=0 python tools/c2r_ab.py 20 24; done
arguments: log2 n ...  [xB: batch B of each]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, phastft_amd as P
args = sys.argv[1:] or ["20", "22", "24", "26"]
for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    for arg in args:
        L, _, B = arg.partition("x")
        L, B = int(L), int(B or 1)
        n = 1 << L
        pl = (P.PlannerR2c64 if name == "f64" else P.PlannerR2c32)(n)
        h1 = n // 2 + 1
        pitch = (h1 + 63) // 64 * 64
        bytes_per = (2 * pitch + n) * B * (8 if name == "f64" else 4)
        ring = max(2, min(9, (1 << 30) // bytes_per))
        a = torch.empty(ring * B * pitch, dtype=dt, device="cuda").uniform_(-1, 1)
        b = torch.empty_like(a).uniform_(-1, 1)
        y = torch.empty(ring * B * n, dtype=dt, device="cuda")
        span = B * h1  # transforms of a batch h1 apart (the batched entry point's layout); ring sets 256-byte aligned
        setp = (span + 63) // 64 * 64
        def call(i):
            P.c2r_fft_batched(a[i * setp:i * setp + span], b[i * setp:i * setp + span], y[i * B * n:(i + 1) * B * n], pl, B)
        call(0)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(ring): call(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / ring)
        print(f"C2R_FUSE={os.environ.get('PHAST_C2R_FUSE','1')} {name} 2^{L} x{B}: {1e3*best:.1f} us = {B*n/best/1e6:.1f} GS/s")
        del a, b, y, pl
