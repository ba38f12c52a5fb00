#!/usr/bin/env python3
"""ONE C2C transform of 2^L points, every plan that exists as kernels -- factorisation into 2 or 3 passes, tile size PER
PASS (4096 ... 32768 points; f64 also the one-wave 64 x 16 tiles and the four-wave 256 x 16 pass) and points per thread --
ranked by the time per transform of a HIP graph over a COLD ring of distinct buffers (>= 1.5 GiB: tools/sweep_all.py times
one buffer in place, which from 2^21 to 2^24 points is resident in the 256 MiB Infinity Cache and ranks plans for a
situation a caller's first transform is never in).
    python tools/sweep_single_cold.py f64|f32 L [L ...] [--top K]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phastft_amd as P

argv = sys.argv[1:]
top = 6
if "--top" in argv:
    i = argv.index("--top")
    top = int(argv[i + 1])
    del argv[i:i + 2]
dt_s, Ls = argv[0], [int(a) for a in argv[1:]]
es = 8 if dt_s == "f64" else 4
dt = torch.float64 if es == 8 else torch.float32
Planner = P.PlannerDit64 if es == 8 else P.PlannerDit32
fft = P.fft_64_dit_with_planner if es == 8 else P.fft_32_dit_with_planner
WAVE = 0x10

for L in Ls:
    n = 1 << L
    ring = max(3, min(64, (3 << 29) // (2 * es * n)))
    re = torch.empty(ring * n, dtype=dt, device="cuda")
    im = torch.empty_like(re)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]

    def measure(pl):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fft(*views[0], P.Direction.Forward, pl)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for r, m in views:
                fft(r, m, P.Direction.Forward, pl)
        g.replay()
        best = 1e9
        for _ in range(2 if L >= 24 else 3):
            P.fill_uniform(re, im, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / ring)
        del g
        return best

    pl = Planner(n)
    d0 = measure(pl)
    res = [(d0, "default: " + pl.describe())]
    tile_logs = (10, 12, 13, 14) if es == 8 else (12, 13, 14, 15)
    if os.environ.get("SWEEP_TLS"):  # e.g. SWEEP_TLS=10,11,12,13: 2048-point tiles for the small sizes
        tile_logs = tuple(int(t) for t in os.environ["SWEEP_TLS"].split(","))
    lps = (3, 4, 5, 3 | WAVE, 4 | WAVE) if es == 8 else (3, 4, 5)
    count = 0
    for k in (2, 3):
        for lrs in itertools.product(range(6, 11), repeat=k):
            if sum(lrs) != L:
                continue
            for tls in itertools.product(tile_logs, repeat=k):
                if any(tl - lr < (3 if es == 8 else 4) or tl - lr > 7 for lr, tl in zip(lrs, tls)):
                    continue
                for lp in lps:
                    wave_shaped = any((lr == 6 and tl == 10) or (lr == 8 and tl == 12 and i > 0) for i, (lr, tl) in enumerate(zip(lrs, tls)))
                    if (lp & WAVE) and not wave_shaped:
                        continue  # the same plan as without the flag
                    if not (lp & WAVE) and any(tl == 10 for tl in tls):
                        continue
                    pl = Planner(n)
                    try:
                        pl.set_plan(lrs, list(tls), lp)
                    except (P.PhastPanic, P.PhastHipError):
                        continue
                    res.append((measure(pl), f"{lrs}@{tls}p{1 << (lp & 15)}{'w' if lp & WAVE else ''}"))
                    count += 1
    res.sort(key=lambda r: r[0])
    print(f"2^{L} {dt_s} ring {ring}: default {d0:.2f} us = {n / d0 / 1e3:.1f} GS/s; {count} plans measured", flush=True)
    for us, name in res[:top]:
        print(f"  {us:9.2f} us {n / us / 1e3:6.1f} GS/s  {name[:170]}", flush=True)
    del re, im, views
    torch.cuda.empty_cache()
