"""R2C of one large real signal on a cold ring: whole-call time and the per-kernel times (PHAST_R2C_LAT=0: the plan of
the C2C transform even where its last pass has no fused form; PHAST_R2C_FUSE=0: never fused).  tools/ -- A/B inside one
gpurun call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, phastft_amd as P
for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    for L in [int(a) for a in sys.argv[1:]] or (24, 25):
        n = 1 << L
        pl = (P.PlannerR2c64 if name == "f64" else P.PlannerR2c32)(n)
        r2c = P.r2c_fft_f64_with_planner if name == "f64" else P.r2c_fft_f32_with_planner
        ring = 5 if L < 27 else 2
        pitch = (n // 2 + 1 + 63) // 64 * 64
        x = torch.empty(ring * n, dtype=dt, device="cuda").uniform_(-1, 1)
        a = torch.empty(ring * pitch, dtype=dt, device="cuda"); b = torch.empty_like(a)
        sets = [(x[i * n:(i + 1) * n], a[i * pitch:i * pitch + n // 2 + 1], b[i * pitch:i * pitch + n // 2 + 1]) for i in range(ring)]
        r2c(*sets[0], pl)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for s in sets: r2c(*s, pl)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / ring)
        ms = pl.time_passes(*sets[1], reps=3)
        print(f"LAT={os.environ.get('PHAST_R2C_LAT','1')} FUSE={os.environ.get('PHAST_R2C_FUSE','1')} {name} 2^{L}: {1e3*best:.1f} us = {n/best/1e6:.1f} GS/s kernels {[round(1e3*m,1) for m in ms]}")
        del x, a, b, sets, pl
