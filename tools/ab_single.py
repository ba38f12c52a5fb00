#!/usr/bin/env python3
"""A/B of library variants (or environment switches) on SINGLE transforms, inside ONE gpurun call, alternating
(boxes of the pool differ by 3-8 %: numbers from different calls do not compare).

    python tools/ab_single.py --libs "_prev,''" --cases f64:20,f32:20,f64:26 --rounds 3 [--env PHAST_X=1]

For every (variant, case) a child process builds a planner, fills a cold ring (> 640 MiB of distinct buffers), captures
K = 20 transforms into one HIP graph, replays it once untimed and then times `reps` replays with HIP events: us per
transform (min and median over the replays), plus the per-pass kernel times from the planner's event timer and the
worst rel-L2 of three ring buffers against numpy's FFT (f64: pocketfft in double; f32: against the f64 transform of the
same input), so a variant that is fast and wrong is seen at once.  Variants are `phastft_amd/lib/libphastft_hip<suffix>.so`
(built with `phastft_amd.build.build(extra=..., tag=...)`)."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import phastft_amd as P
    from bench import capture_steps, settle

    dt, log_n = a.child.split(":")
    log_n = int(log_n)
    n = 1 << log_n
    f64 = dt == "f64"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tdt = torch.float64 if f64 else torch.float32
    pl = (P.PlannerDit64 if f64 else P.PlannerDit32)(n)
    if a.plan:
        lrs_s, rest = a.plan.split("@")
        tl_s, p_s = rest.split("p")
        tls = tuple(int(x) for x in tl_s.split(","))
        flags = int(p_s, 0) if p_s.startswith("0x") else {8: 3, 16: 4, 32: 5}[int(p_s.rstrip("w"))] | (0x10 if p_s.endswith("w") else 0)
        pl.set_plan(tuple(int(x) for x in lrs_s.split(",")), tls if len(tls) > 1 else tls[0], flags)
    steps = 20 if log_n <= 22 else 5
    esz = 8 if f64 else 4
    ring = max(steps + 3, (640 << 20) // (2 * esz * n) + 1) if log_n < 26 else steps + 1
    re = torch.empty(ring * n, dtype=tdt, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]
    fft = P.fft_64_dit_with_planner if f64 else P.fft_32_dit_with_planner
    # correctness first, on three buffers (they are re-filled afterwards)
    worst = 0.0
    for i in (0, ring // 2, ring - 1):
        x = (views[i][0].cpu().numpy().astype(np.float64) + 1j * views[i][1].cpu().numpy().astype(np.float64))
        fft(views[i][0], views[i][1], P.Direction.Forward, pl)
        torch.cuda.synchronize()
        y = views[i][0].cpu().numpy().astype(np.float64) + 1j * views[i][1].cpu().numpy().astype(np.float64)
        want = np.fft.fft(x)
        worst = max(worst, float(np.linalg.norm(y - want) / np.linalg.norm(want)))
    P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)

    def step(i):
        r, m = views[i % ring]
        fft(r, m, P.Direction.Forward, pl)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    g, _ = capture_steps(torch, P, step, 3, steps, touch=lambda: fft(*views[0], P.Direction.Forward, pl))
    us = []
    for _ in range(a.reps):
        # inputs as generated, and the fill's dirty lines out of the Infinity Cache before the region (bench.py: settle)
        settle(torch, P, refill=lambda: P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if g is not None:
            g.replay()
        else:
            for i in range(steps):
                step(3 + i)
        e1.record()
        torch.cuda.synchronize()
        us.append(1e3 * e0.elapsed_time(e1) / steps)
    us.sort()
    P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)
    torch.cuda.synchronize()
    acc = None
    reps = min(ring, 32)
    for i in range(reps):
        ms = pl.time_passes(views[i][0], views[i][1], n, reps=1)
        acc = ms if acc is None else [x + y for x, y in zip(acc, ms)]
    print(json.dumps({"us_min": us[0], "us_med": us[len(us) // 2], "pass_us": [1e3 * x / reps for x in acc], "rel_l2": worst,
                      "plan": pl.describe_call() if hasattr(pl, "describe_call") else ""}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="''", help="comma-separated library suffixes ('' = product)")
    ap.add_argument("--envs", default="", help="semicolon-separated environment variants 'A=1 B=2;A=0' (crossed with --libs)")
    ap.add_argument("--cases", default="f64:20")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--plan", default="", help="forced plan, bench.py syntax (e.g. 6,8,6@10,12,10p16w)")
    ap.add_argument("--child", default="")
    a = ap.parse_args()
    if a.child:
        child(a)
        return
    libs = [("" if v in ("''", '""') else v) for v in a.libs.split(",")]
    envs = [e for e in a.envs.split(";")] if a.envs else [""]
    for r in range(a.rounds):
        for lib in libs:
            for ev in envs:
                for case in a.cases.split(","):
                    env = dict(os.environ, PHASTFT_HIP_LIB=os.path.join(ROOT, "phastft_amd", "lib", f"libphastft_hip{lib}.so"))
                    for kv in ev.split():
                        k, v = kv.split("=", 1)
                        env[k] = v
                    cmd = [sys.executable, os.path.abspath(__file__), "--child", case, "--reps", str(a.reps)]
                    if a.plan:
                        cmd += ["--plan", a.plan]
                    p = subprocess.run(cmd, env=env, capture_output=True, text=True)
                    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                    if p.returncode or not line:
                        print(f"round {r} [{lib}] {ev} {case}: FAILED rc={p.returncode} {p.stderr[-400:]}", flush=True)
                        continue
                    d = json.loads(line[-1])
                    print(f"round {r} [{lib or 'product'}] {ev} {case}: {d['us_min']:8.2f} us min {d['us_med']:8.2f} med | passes "
                          + " ".join(f"{x:7.2f}" for x in d["pass_us"]) + f" | rel_l2 {d['rel_l2']:.2e} | {d['plan'][:70]}", flush=True)


if __name__ == "__main__":
    main()
