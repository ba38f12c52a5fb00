#!/usr/bin/env python3
"""BATCHES of real transforms (2^27 real samples in flight): every plan of the inner N/2-point transform ranked by the time
of r2c_fft_batched and c2r_fft_batched.  The C2C throughput plans end in 32-point-per-thread passes that have no fused
untangle / preprocess form; a plan whose first / last pass fuses saves a whole sweep.
    python tools/sweep_real_batch.py f32|f64 L [top]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

dt_s, L = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 6
es = 8 if dt_s == "f64" else 4
dt = torch.float64 if es == 8 else torch.float32
n, Li = 1 << L, L - 1
TOTAL = int(os.environ.get("SWEEP_TOTAL", "27"))   # log2 of the real samples in flight per call (22: between one transform and a full chip)
RING = 1 if TOTAL >= 26 else 16                      # small totals: a graph over 16 distinct buffer sets, nothing served from the caches
batch = 1 << (TOTAL - L)
h = n // 2 + 1
mx, mh = batch * n, batch * h
x = torch.empty(RING * mx, dtype=dt, device="cuda")
sr = torch.empty(RING * mh, dtype=dt, device="cuda")
si = torch.empty_like(sr)
pl = (P.PlannerR2c64 if es == 8 else P.PlannerR2c32)(n)


def timed(fn, refill):
    if RING == 1:
        fn(0)
        run = lambda: fn(0)
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(RING):
                fn(i)
        g.replay()
        run = g.replay
    best = 1e9
    for _ in range(2):
        refill()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / RING)
    return best


def refill_s():
    sr.uniform_(-1, 1)
    si.uniform_(-1, 1)


def measure(tag):
    t_r = timed(lambda i: P.r2c_fft_batched(x[i * mx:(i + 1) * mx], sr[i * mh:(i + 1) * mh], si[i * mh:(i + 1) * mh], pl, batch), lambda: x.uniform_(-1, 1))
    t_c = timed(lambda i: P.c2r_fft_batched(sr[i * mh:(i + 1) * mh], si[i * mh:(i + 1) * mh], x[i * mx:(i + 1) * mx], pl, batch), refill_s)
    return (t_r, t_c, tag)


res = [measure("library: " + pl.describe())]
print(f"{dt_s} 2^{L} x {batch}: library r2c {res[0][0]:.1f} us = {n * batch / res[0][0] / 1e3:.1f} GS/s, c2r {res[0][1]:.1f} us = {n * batch / res[0][1] / 1e3:.1f} GS/s", flush=True)
tile_logs = (12, 13, 14) if es == 8 else (12, 13, 14, 15)
if os.environ.get("SWEEP_TLS"):
    tile_logs = tuple(int(t) for t in os.environ["SWEEP_TLS"].split(","))
count = 0
for np_ in (2, 3):
    for lrs in itertools.product(range(6, 11), repeat=np_):
        if sum(lrs) != Li:
            continue
        for tls in itertools.product(tile_logs, repeat=np_):
            if any(tl - lr < (3 if es == 8 else 4) or tl - lr > 7 for lr, tl in zip(lrs, tls)):
                continue
            for lp in (3, 4, 5):
                try:
                    pl.set_plan(lrs, list(tls), lp)
                except Exception:
                    continue
                res.append(measure(f"{lrs}@{tls}p{1 << lp}"))
                count += 1
pl.set_plan(())
print(f"{count} plans; best by r2c:")
for r in sorted(res, key=lambda r: r[0])[:top]:
    print(f"  r2c {r[0]:8.1f} us {n * batch / r[0] / 1e3:6.1f} GS/s   c2r {r[1]:8.1f}   {r[2][:100]}")
print("best by c2r:")
for r in sorted(res, key=lambda r: r[1])[:top]:
    print(f"  c2r {r[1]:8.1f} us {n * batch / r[1] / 1e3:6.1f} GS/s   r2c {r[0]:8.1f}   {r[2][:100]}")
