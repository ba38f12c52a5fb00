#!/usr/bin/env python3
"""R2C with the untangle fused into the last pass (default) against the separate sweep (PHAST_R2C_FUSE=0 in a child
process): one transform per size on a cold ring, HIP events."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import phastft_amd as P
    for dt, Pl, fn in ((torch.float32, P.PlannerR2c32, P.r2c_fft_f32_with_planner), (torch.float64, P.PlannerR2c64, P.r2c_fft_f64_with_planner)):
        for L in (16, 18, 20, 22, 24, 26):
            n = 1 << L
            pl = Pl(n)
            ring = max(3, min(64, (1 << 30) // (n * (4 if dt == torch.float32 else 8))))
            pitch = (n // 2 + 1 + 63) // 64 * 64
            x = torch.empty(ring * n, dtype=dt, device="cuda").uniform_(-1, 1)
            a = torch.empty(ring * pitch, dtype=dt, device="cuda"); b = torch.empty_like(a)
            sets = [(x[i * n:(i + 1) * n], a[i * pitch:i * pitch + n // 2 + 1], b[i * pitch:i * pitch + n // 2 + 1]) for i in range(ring)]
            fn(*sets[0], pl)
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for s in sets:
                    fn(*s, pl)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / ring)
            ms = pl.time_passes(*sets[1], reps=3)
            print(f"  {'f32' if dt == torch.float32 else 'f64'} 2^{L}: {1e3 * best:9.1f} us = {n / best / 1e6:7.1f} GS/s   kernels {[round(1e3 * m, 1) for m in ms]}", flush=True)
            del x, a, b, sets, pl
else:
    for fuse in ("1", "0", "1", "0"):
        print(f"PHAST_R2C_FUSE={fuse}", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PHAST_R2C_FUSE=fuse))
