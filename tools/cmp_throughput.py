import sys, os
sys.path.insert(0, os.getcwd())
import torch, phastft_amd as P
F32 = '--f32' in sys.argv
ARGS = [a for a in sys.argv[1:] if a != '--f32']
SIZES = ((20, 256), (26, 1), (24, 4), (16, 4096)) if not ARGS else tuple((int(a.split('x')[0]), int(a.split('x')[1])) for a in ARGS)
for L, batch in SIZES:
    n = 1 << L
    pl = (P.PlannerDit32 if F32 else P.PlannerDit64)(n)
    re = torch.empty(n * batch, dtype=torch.float32 if F32 else torch.float64, device="cuda"); im = torch.empty_like(re)
    P.fill_uniform(re, im, n)
    pl.time_passes(re, im, n, reps=1)
    best = None
    for _ in range(3):
        P.fill_uniform(re, im, n)
        ms = pl.time_passes(re, im, n, reps=3)
        if best is None or sum(ms) < sum(best): best = ms
    print(pl.describe() if hasattr(pl, "describe") else "")
    print(f"2^{L} x{batch}: {[round(x,4) for x in best]} sum {sum(best):.4f} ms = {n*batch/sum(best)/1e6:.1f} GS/s")
    del re, im
