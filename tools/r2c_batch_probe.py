"""Batched R2C on a cold ring (arguments LxB: B transforms of 2^L real points); A/B through the environment
(PHAST_R2C_LAT, PHAST_R2C_FUSE) inside one gpurun call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, phastft_amd as P
for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    for arg in sys.argv[1:]:
        L, _, B = arg.partition("x")
        L, B = int(L), int(B or 1)
        n = 1 << L
        h1 = n // 2 + 1
        pl = (P.PlannerR2c64 if name == "f64" else P.PlannerR2c32)(n)
        bytes_per = (2 * h1 + n) * B * (8 if name == "f64" else 4)
        ring = max(2, min(9, (1 << 30) // bytes_per))
        setp = (B * h1 + 63) // 64 * 64
        x = torch.empty(ring * B * n, dtype=dt, device="cuda").uniform_(-1, 1)
        a = torch.empty(ring * setp, dtype=dt, device="cuda"); b = torch.empty_like(a)
        def call(i):
            P.r2c_fft_batched(x[i * B * n:(i + 1) * B * n], a[i * setp:i * setp + B * h1], b[i * setp:i * setp + B * h1], pl, B)
        call(0)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(ring): call(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / ring)
        print(f"LAT={os.environ.get('PHAST_R2C_LAT','1')} FUSE={os.environ.get('PHAST_R2C_FUSE','1')} {name} 2^{L} x{B}: {1e3*best:.1f} us = {B*n/best/1e6:.1f} GS/s")
        del x, a, b, pl
