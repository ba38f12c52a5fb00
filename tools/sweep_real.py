#!/usr/bin/env python3
"""Plans of the inner transform of a REAL transform, ranked by the time of c2r_fft and r2c_fft (cold ring, HIP events):
the C2C plans were chosen for planar input and output; R2C reads (re, im) pairs and C2R writes them, and the fused first /
last passes have costs of their own (c2r_fused.hpp, r2c_fused.hpp).
    python tools/sweep_real.py f32 24 [top]
SWEEP_TLS=11,12,13 widens the tile sizes tried per pass (default 12,13); with PHAST_R2C_FUSE_MIN_LOG=20 the fused last pass
runs below its usual threshold -- does it pay there on 2048-point tiles (twice the workgroups)?"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

dt_s, L = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n = 1 << L
Li = L - 1
dt = torch.float32 if dt_s == "f32" else torch.float64
pl = (P.PlannerR2c32 if dt_s == "f32" else P.PlannerR2c64)(n)
ring = max(3, min(12, (1 << 30) // (n * (4 if dt_s == "f32" else 8))))
h1 = n // 2 + 1
pitch = (h1 + 63) // 64 * 64
x = torch.empty(ring * n, dtype=dt, device="cuda").uniform_(-1, 1)
a = torch.empty(ring * pitch, dtype=dt, device="cuda").uniform_(-1, 1)
b = torch.empty_like(a).uniform_(-1, 1)
sets = [(x[i * n:(i + 1) * n], a[i * pitch:i * pitch + h1], b[i * pitch:i * pitch + h1]) for i in range(ring)]
r2c = P.r2c_fft_f32_with_planner if dt_s == "f32" else P.r2c_fft_f64_with_planner
c2r = P.c2r_fft_f32_with_planner if dt_s == "f32" else P.c2r_fft_f64_with_planner


def timed(fn):
    fn(sets[0])
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for s in sets:
            fn(s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / ring)
    return 1e3 * best


def measure(tag):
    t_c = timed(lambda s: c2r(s[1], s[2], s[0], pl))
    t_r = timed(lambda s: r2c(s[0], s[1], s[2], pl))
    pc = [round(1e3 * v, 1) for v in pl.time_c2r_passes(sets[1][1], sets[1][2], sets[1][0], reps=3)]
    pr = [round(1e3 * v, 1) for v in pl.time_passes(sets[1][0], sets[1][1], sets[1][2], reps=3)]
    return (t_c, t_r, tag, pc, pr)


res = [measure("library default: " + pl.describe())]
print("default: c2r %.1f us %s   r2c %.1f us %s" % (res[0][0], res[0][3], res[0][1], res[0][4]), flush=True)
cands = set()
for np_ in (2, 3):
    for lrs in itertools.product(range(6, 11), repeat=np_):
        if sum(lrs) != Li:
            continue
        for lp in (3, 4):
            for tls in itertools.product(tuple(int(t) for t in os.environ.get("SWEEP_TLS", "12,13").split(",")), repeat=np_):
                if all(tl - lr >= 3 and tl - lr <= 7 for lr, tl in zip(lrs, tls)):
                    cands.add((lrs, tls, lp))
for lrs, tls, lp in sorted(cands):
    try:
        pl.set_plan(lrs, list(tls), lp)
        res.append(measure(f"{lrs}@{tls}p{1 << lp}"))
    except Exception as e:
        continue
pl.set_plan(())
print(f"{len(res) - 1} plans measured; best by c2r:")
for r in sorted(res, key=lambda r: r[0])[:top]:
    print("  c2r %7.1f us %-24s r2c %7.1f us %-24s %s" % (r[0], r[3], r[1], r[4], r[2][:80]))
print("best by r2c:")
for r in sorted(res, key=lambda r: r[1])[:top]:
    print("  r2c %7.1f us %-24s c2r %7.1f us %-24s %s" % (r[1], r[4], r[0], r[3], r[2][:80]))
