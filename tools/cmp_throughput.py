import sys, os
sys.path.insert(0, os.getcwd())
import torch, phastft_amd as P
for L, batch in ((20, 256), (26, 1), (24, 4), (16, 4096)):
    n = 1 << L
    pl = P.PlannerDit64(n)
    re = torch.empty(n * batch, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
    P.fill_uniform(re, im, n)
    pl.time_passes(re, im, n, reps=1)
    best = None
    for _ in range(3):
        P.fill_uniform(re, im, n)
        ms = pl.time_passes(re, im, n, reps=3)
        if best is None or sum(ms) < sum(best): best = ms
    print(f"2^{L} x{batch}: {[round(x,4) for x in best]} sum {sum(best):.4f} ms = {n*batch/sum(best)/1e6:.1f} GS/s")
    del re, im
