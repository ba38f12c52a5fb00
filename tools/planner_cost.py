"""What a planner costs to make (the planner-less entry points make one per call, lib.rs:181,224 / r2c.rs:522,599)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, phastft_amd as P
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for L in (10, 14, 16, 20, 24, 26):
    n = 1 << L
    for name, ctor in (("PlannerDit64", P.PlannerDit64), ("PlannerDit32", P.PlannerDit32), ("PlannerR2c64", P.PlannerR2c64), ("PlannerR2c32", P.PlannerR2c32)):
        ctor(n)  # first of its size: kernels' attributes looked up once per process
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); pl = ctor(n); ts.append(time.perf_counter() - t0); del pl
        print(f"{name}(2^{L}): {1e3*min(ts):.3f} ms (best of 5; max {1e3*max(ts):.3f})")
    if L <= 20:
        re, im = np.random.rand(n), np.random.rand(n)
        P.fft_64_dit(re, im, P.Direction.Forward)
        t0 = time.perf_counter()
        for _ in range(5): P.fft_64_dit(re, im, P.Direction.Forward)
        print(f"fft_64_dit(host slices, no planner) 2^{L}: {1e3*(time.perf_counter()-t0)/5:.3f} ms per call")
