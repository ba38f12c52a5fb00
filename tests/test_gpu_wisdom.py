"""The built-in wisdom must be SAFE on a box it was not measured on.

csrc/builtin_wisdom.inc is one box's opinion: plans the in-library tuner measured > 3 % faster than the static rules of plan.hpp
on the MI355X that generated the table.  Boxes of the pool differ by 3-8 % in clocks and copy bandwidth, and buffer placement
alone moves a large transform by +-5 %.  This test runs, on whatever box the suite lands on, an INTERLEAVED A/B of wisdom-on
against `phast_wisdom_builtin(0)` (the static rules alone) on a cold ring for

  * the BASELINE configurations that can pick up a built-in plan (one f64 transform of 2^20 and of 2^26 points, r2c / c2r f32
    at 2^24, the 1024-transform f64 shard), and
  * the twenty built-in lines with the largest claimed gain (bounded to 2^26 points in flight),

and fails if the built-in plan is SLOWER than the static rule's by more than 5 % (+ 0.5 us for the 10-us transforms, whose
graph replays scatter by that much).  A line that fails here on a healthy box should be dropped from the table, not waived.
"""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "phastft_amd", "csrc", "builtin_wisdom.inc")
MARGIN, SLACK_US = 1.05, 0.5


def builtin_lines():
    out = []
    for m in re.finditer(r'^"(f64|f32) (c2c|c2ci|r2c|c2r) (\d+) (\d+) (\S+) fuse=(\d) us=([\d.]+) heur=([\d.]+)\\n"$', open(INC).read(), flags=re.M):
        dt, kind, L, bucket, plan, fuse, us, heur = m.groups()
        out.append((dt, kind, int(L), int(bucket), plan, float(us), float(heur)))
    return out


def cases():
    lines = builtin_lines()
    keyed = {(dt, kind, L, b): (plan, us, heur) for dt, kind, L, b, plan, us, heur in lines}
    sel = []
    for key in (("f64", "c2c", 20, 0), ("f64", "c2c", 26, 0), ("f32", "r2c", 24, 0), ("f32", "c2r", 24, 0), ("f64", "c2c", 20, 10),
                ("f32", "c2c", 20, 0), ("f32", "c2c", 26, 0)):
        sel.append(key + ("baseline",))
    # real transforms: L is the REAL length; points in flight = 2^(L + bucket) (complex) or half of it (real)
    ranked = sorted((ln for ln in lines if ln[2] + ln[3] <= 26), key=lambda ln: -(ln[6] / ln[5]))
    for dt, kind, L, b, plan, us, heur in ranked[:20]:
        if (dt, kind, L, b, "baseline") not in sel:
            sel.append((dt, kind, L, b, f"claims {100 * (heur / us - 1):.0f} %"))
    return sel, keyed


def _bench_call(gpu, torch, dt, kind, L, bucket, wisdom_on, reuse=None):
    """(callable running ONE graph replay of `steps` calls on a cold ring, steps, describe_call, keep-alive) for a fresh planner.
    `reuse`: the keep-alive tuple of an earlier call for the same key -- the second planner then runs on the SAME buffers (from 2^25
    points in flight on, where a buffer happens to lie moves a call by +-5 %: two allocations would compare placements, not plans)"""
    import sys

    sys.path.insert(0, ROOT)
    from bench import capture_steps

    n, batch = 1 << L, 1 << bucket
    f64 = dt == "f64"
    tdt = torch.float64 if f64 else torch.float32
    esz = 8 if f64 else 4
    was = gpu.wisdom_builtin(wisdom_on)
    try:
        if kind in ("c2c", "c2ci"):
            pl = (gpu.PlannerDit64 if f64 else gpu.PlannerDit32)(n)
        else:
            pl = (gpu.PlannerR2c64 if f64 else gpu.PlannerR2c32)(n)
    finally:
        gpu.wisdom_builtin(was)
    set_bytes = 2 * batch * n * esz
    steps = int(max(2, min(20, (1 << 30) // set_bytes)))
    kind_id = {"c2c": 0, "c2ci": 1, "r2c": 2, "c2r": 3}[kind]
    bufs = reuse[1] if reuse is not None else None
    if kind == "c2c":
        re, im = bufs if bufs else (torch.empty(steps * batch * n, dtype=tdt, device="cuda"), torch.empty(steps * batch * n, dtype=tdt, device="cuda"))
        re.uniform_(-1, 1); im.uniform_(-1, 1)

        def step(i):
            s = slice((i % steps) * batch * n, ((i % steps) + 1) * batch * n)
            gpu.fft_dit_batched(re[s], im[s], n, gpu.Direction.Forward, pl)
        keep = (re, im)
    elif kind == "c2ci":
        z = bufs[0] if bufs else torch.empty(steps * batch * n, dtype=torch.complex128 if f64 else torch.complex64, device="cuda")
        torch.view_as_real(z).uniform_(-1, 1)
        fft = gpu.fft_64_interleaved_with_planner if f64 else gpu.fft_32_interleaved_with_planner
        if batch != 1:
            return None

        def step(i):
            fft(z[(i % steps) * n:((i % steps) + 1) * n], gpu.Direction.Forward, pl)
        keep = (z,)
    else:
        h1 = n // 2 + 1
        x, a, b = bufs if bufs else (torch.empty(steps * batch * n, dtype=tdt, device="cuda"), torch.empty(steps * batch * h1, dtype=tdt, device="cuda"),
                                     torch.empty(steps * batch * h1, dtype=tdt, device="cuda"))
        x.uniform_(-1, 1); a.uniform_(-1, 1); b.uniform_(-1, 1)

        def step(i):
            j = i % steps
            xs, as_, bs = x[j * batch * n:(j + 1) * batch * n], a[j * batch * h1:(j + 1) * batch * h1], b[j * batch * h1:(j + 1) * batch * h1]
            if kind == "r2c":
                gpu.r2c_fft_batched(xs, as_, bs, pl, batch)
            else:
                gpu.c2r_fft_batched(as_, bs, xs, pl, batch)
        keep = (x, a, b)
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    g, _ = capture_steps(torch, gpu, step, 0, steps, touch=lambda: step(0))
    run = g.replay if g is not None else (lambda: [step(i) for i in range(steps)])
    return run, steps, pl.describe_call(batch, kind_id), (pl, keep, g)


def _time(torch, run, steps, gpu=None):
    """us per call of one replay; with `gpu` (the module) the caches are drained first (bench.py: settle -- what the previous
    replay left dirty in the Infinity Cache would otherwise be written back inside this one)"""
    if gpu is not None:
        import sys

        sys.path.insert(0, ROOT)
        from bench import settle

        settle(torch, gpu)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / steps


def test_builtin_wisdom_is_not_slower_than_the_static_rules_on_this_box(gpu):
    import torch

    sel, keyed = cases()
    report, bad = [], []
    for dt, kind, L, bucket, why in sel:
        on = _bench_call(gpu, torch, dt, kind, L, bucket, True)
        if on is None:
            continue
        if not on[2].startswith("tuned"):        # no built-in plan for this key (the static rule stood when the table was made)
            report.append(f"{dt} {kind} 2^{L} b{bucket} [{why}]: static rule runs ({on[2][:60]})")
            del on
            torch.cuda.empty_cache()
            continue
        off = _bench_call(gpu, torch, dt, kind, L, bucket, False, reuse=on[3])
        assert not off[2].startswith("tuned"), off[2]

        def measure(rounds):
            t_on, t_off = [], []
            for _ in range(rounds):      # interleaved: what drifts (clocks, neighbours) hits both alike
                t_on.append(_time(torch, on[0], on[1], gpu))
                t_off.append(_time(torch, off[0], off[1], gpu))
            return float(np.median(t_on)), float(np.median(t_off))

        m_on, m_off = measure(5)
        if m_on > MARGIN * m_off + SLACK_US:     # a second look with more rounds before calling it slower
            m_on, m_off = measure(11)
        line = f"{dt} {kind} 2^{L} b{bucket} [{why}]: wisdom {m_on:.2f} us ({on[2][:50]}) static {m_off:.2f} us ({off[2][:50]})"
        report.append(line)
        if m_on > MARGIN * m_off + SLACK_US:
            bad.append(line)
        del on, off
        torch.cuda.empty_cache()
    print("\n".join(report))
    path = os.environ.get("PHAST_WISDOM_AB_LOG")
    if path:
        with open(path, "w") as f:
            f.write("\n".join(report) + "\n")
    assert not bad, "built-in wisdom plans slower than the static rules on this box:\n" + "\n".join(bad)


def test_set_plan_beats_builtin_wisdom_and_the_restore_form_brings_it_back(gpu, oracle):
    """`phast_planner_*_set_plan` used to be silently ignored for every (kind, bucket) that had a tuned or
    built-in-wisdom plan -- `choose()` asked the tuned table first.  On a length whose ONE-transform call has a built-in line:
    the planner starts `tuned`, a forced plan runs (`describe_call` says `forced` and names ITS passes, for every batch and kind),
    the result is still right, and the restore form (no passes) puts the tuned plan back."""
    import torch

    from tests import tolerances as tol

    lines = [ln for ln in builtin_lines() if ln[0] == "f64" and ln[1] == "c2c" and ln[3] == 0 and 14 <= ln[2] <= 22]
    if not lines:
        pytest.skip("no built-in line for one f64 C2C transform")
    L = lines[0][2]
    n = 1 << L
    was = gpu.wisdom_builtin(True)
    try:
        pl = gpu.PlannerDit64(n)
    finally:
        gpu.wisdom_builtin(was)
    assert pl.describe_call(1, 0).startswith("tuned"), pl.describe_call(1, 0)
    a, b = L // 2, L - L // 2
    pl.set_plan((b, a), 12 if L <= 20 else 13, 3)
    forced = pl.describe_call(1, 0)
    assert forced.startswith("forced") and f"[{1 << b}x" in forced, forced
    assert pl.describe_call(64, 0).startswith("forced") and pl.describe_call(1, 1).startswith("forced")
    h_re, h_im = oracle.fill(n, np.float64, seed=0x5E7, transform_id=L)
    z = np.fft.fft(h_re + 1j * h_im)
    d_re, d_im = torch.from_numpy(h_re.copy()).cuda(), torch.from_numpy(h_im.copy()).cuda()
    gpu.fft_64_dit_with_planner(d_re, d_im, gpu.Direction.Forward, pl)
    tol.check("forced_over_wisdom", "f64", L, d_re.cpu().numpy(), d_im.cpu().numpy(), z.real, z.imag)
    pl.set_plan()                      # the restore form: the library's own plans, wisdom in force again
    assert pl.describe_call(1, 0).startswith("tuned"), pl.describe_call(1, 0)
