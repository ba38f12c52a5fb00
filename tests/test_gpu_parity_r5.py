"""Round-5 GPU tests (VERDICT r04 "Next round" items 1, 3 and ADVICE r04).

* PlannerMode::Tune is real (planner.rs:18-32): a tuning run is never slower than the static rules (within 3 %) at eight
  (length, batch, kind) points, its plan's results stay within the parity gates, N = 2^20 tunes in under a second, and what it
  finds travels as wisdom text to planners made later.
* The parity gates (tests/tolerances.py) are tight enough to notice ONE twiddle-table entry that is off by 5e-13 (f64) /
  5e-5 (f32) -- the round-4 gates (tolerances.ROUND4_GATES) would have passed both.
* Non-finite and subnormal inputs go through the HIP path as through the oracle (C2C, R2C, C2R at 2^10 and 2^20).
* Stream capture: a capture never takes a workspace another thread's stream is working in; 8192-point
  planners capture without a warm-up call again; graph workspaces can be handed back; replaced plans leave
  no tables behind.
"""
import os
import statistics
import subprocess
import sys
import threading

import numpy as np
import pytest

from tests import tolerances as tol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _forget_wisdom():
    """tuning runs leave wisdom behind for every planner made later in this process: not for the other tests"""
    yield
    import phastft_amd as P

    P.wisdom_forget()


def dev(x):
    import torch

    return torch.from_numpy(x).cuda()


def _ref_c2c(h_re, h_im):
    z = np.fft.fft(h_re.astype(np.float64) + 1j * h_im.astype(np.float64))
    return z.real, z.imag


# ---------------------------------------------------------------- PlannerMode::Tune
TUNE_POINTS = [("f64", 20, 1, "c2c"), ("f64", 18, 16, "c2c"), ("f32", 20, 1, "c2c"), ("f32", 24, 1, "c2c"), ("f32", 19, 32, "c2c"),
               ("f64", 21, 1, "c2ci"), ("f64", 17, 32, "r2c"), ("f32", 18, 16, "r2c"), ("f64", 20, 4, "c2r"), ("f64", 14, 8, "c2r")]


def _make_calls(P, dt, L, batch, kind, planners, ring):
    """[(call(i), check())] per planner for one tune point, all over the SAME cold ring of buffer sets (where a buffer landed is
    worth a few per cent at the large sizes: the planners must not differ in that): `call(i)` runs the point's call on buffer
    set i; check() runs set 0 from known inputs and compares with a float64 reference"""
    import torch

    n = 1 << L
    tdt = torch.float64 if dt == "f64" else torch.float32
    ndt = np.float64 if dt == "f64" else np.float32
    rng = np.random.default_rng(L * 131 + batch)
    out = []
    if kind in ("c2c", "c2ci"):
        h_re = rng.uniform(-1, 1, n * batch).astype(ndt)
        h_im = rng.uniform(-1, 1, n * batch).astype(ndt)
        if kind == "c2c":
            re = torch.empty(ring, n * batch, dtype=tdt, device="cuda")
            im = torch.empty_like(re)
            for planner in planners:
                def call(i, planner=planner):
                    P.fft_dit_batched(re[i], im[i], n, P.Direction.Forward, planner)

                def check(call=call):
                    re[0].copy_(torch.from_numpy(h_re)); im[0].copy_(torch.from_numpy(h_im))
                    call(0)
                    g_re, g_im = re[0].cpu().numpy(), im[0].cpu().numpy()
                    for b in (0, batch - 1):
                        sl = slice(b * n, (b + 1) * n)
                        tol.check(f"tune:{kind}", dt, L, g_re[sl], g_im[sl], *_ref_c2c(h_re[sl], h_im[sl]))
                out.append((call, check))
        else:
            assert batch == 1
            cdt = torch.complex128 if dt == "f64" else torch.complex64
            sig = torch.empty(ring, n, dtype=cdt, device="cuda")
            fn = P.fft_64_interleaved_with_planner if dt == "f64" else P.fft_32_interleaved_with_planner
            for planner in planners:
                def call(i, planner=planner):
                    fn(sig[i], P.Direction.Forward, planner)

                def check(call=call):
                    sig[0].copy_(torch.from_numpy(h_re.astype(np.float64) + 1j * h_im).to(cdt))
                    call(0)
                    g = sig[0].cpu().numpy()
                    tol.check(f"tune:{kind}", dt, L, g.real, g.imag, *_ref_c2c(h_re, h_im))
                out.append((call, check))
        return out
    h = n // 2 + 1
    x = torch.empty(ring, n * batch, dtype=tdt, device="cuda")
    sr = torch.empty(ring, h * batch, dtype=tdt, device="cuda")
    si = torch.empty_like(sr)
    hx = rng.uniform(-1, 1, n * batch).astype(ndt)
    x.uniform_(-1, 1)
    sr.uniform_(-1, 1); si.uniform_(-1, 1)
    if kind == "r2c":
        for planner in planners:
            def call(i, planner=planner):
                P.r2c_fft_batched(x[i], sr[i], si[i], planner, batch)

            def check(call=call):
                x[0].copy_(torch.from_numpy(hx))
                call(0)
                g_re, g_im = sr[0].cpu().numpy(), si[0].cpu().numpy()
                for b in (0, batch - 1):
                    ref = np.fft.rfft(hx[b * n:(b + 1) * n].astype(np.float64))
                    tol.check("tune:r2c", dt, L, g_re[b * h:(b + 1) * h], g_im[b * h:(b + 1) * h], ref.real, ref.imag)
            out.append((call, check))
    else:
        spec = [np.fft.rfft(hx[b * n:(b + 1) * n].astype(np.float64)) for b in range(batch)]
        h_sr = np.concatenate([s_.real for s_ in spec]).astype(ndt)
        h_si = np.concatenate([s_.imag for s_ in spec]).astype(ndt)
        for planner in planners:
            def call(i, planner=planner):
                P.c2r_fft_batched(sr[i], si[i], x[i], planner, batch)

            def check(call=call):
                sr[0].copy_(torch.from_numpy(h_sr)); si[0].copy_(torch.from_numpy(h_si))
                call(0)
                got = x[0].cpu().numpy()
                for b in (0, batch - 1):
                    want = np.fft.irfft(h_sr[b * h:(b + 1) * h].astype(np.float64) + 1j * h_si[b * h:(b + 1) * h], n)
                    tol.check("tune:c2r", dt, L, got[b * n:(b + 1) * n], np.zeros(n), want, np.zeros(n))
                sr[0].uniform_(-1, 1); si[0].uniform_(-1, 1)
            out.append((call, check))
    return out


def _time_alternating(calls, ring, rounds=7):
    """median per-call time (us) of each callable, measured alternately (A B A B ...), each round over the whole ring"""
    import torch

    times = [[] for _ in calls]
    for c in calls:  # warm-up: first launches, scratch
        c(0)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, c in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(ring):
                c(i)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(1e3 * e0.elapsed_time(e1) / ring)
    return [statistics.median(t) for t in times]


@pytest.mark.parametrize("dt,L,batch,kind", TUNE_POINTS)
def test_tune_is_not_slower_and_stays_within_tolerance(gpu, static_rules, dt, L, batch, kind):
    """Heuristic planner against a planner that tuned for exactly this call: interleaved timing over a cold ring, Tune >=
    Heuristic - 3 %; the tuned plan's results within the gates of a float64 reference, first and last transform of the batch."""
    P = gpu
    n = 1 << L
    real = kind in ("r2c", "c2r")
    Pl = (P.PlannerR2c64 if dt == "f64" else P.PlannerR2c32) if real else (P.PlannerDit64 if dt == "f64" else P.PlannerDit32)
    kinds = {"c2c": P.TuneKind.C2C, "c2ci": P.TuneKind.C2CInterleaved, "r2c": P.TuneKind.R2C, "c2r": P.TuneKind.C2R}
    heur, tuned = Pl(n), Pl(n)
    rep = tuned.tune(batch, kinds[kind])
    assert rep["candidates"] >= 8 and rep["us_heuristic"] > 0 and rep["us_best"] <= rep["us_heuristic"] * 1.0001, rep
    es = 8 if dt == "f64" else 4
    ring = max(3, min(32, (1 << 30) // (2 * es * n * batch)))
    (call_h, check_h), (call_t, check_t) = _make_calls(P, dt, L, batch, kind, [heur, tuned], ring)
    check_h()
    check_t()
    which = Pl.describe_call
    if not rep["adopted"]:
        # the static rule's plan stood: both planners run the SAME plan (two timings of it differ by where their scratch
        # landed -- up to 6 % at 2^20 x 1 in round 5's first run -- and say nothing about the tuner)
        assert which(tuned, batch, kinds[kind]) == which(heur, batch, kinds[kind]) and "tuned" not in which(tuned, batch, kinds[kind])
        print(f"\n{dt} 2^{L} x {batch} {kind}: the static rule's plan stood ({rep['candidates']} plans in {rep['seconds']:.2f} s; "
              f"medians {rep['us_heuristic']:.2f} / {rep['us_best']:.2f} us)")
        return
    assert which(tuned, batch, kinds[kind]).startswith("tuned ") and "tuned:" in tuned.describe() and "tuned:" not in heur.describe()
    # Tune >= Heuristic - 3 %.  The two planners work in DIFFERENT scratch allocations (the tuning run compared its candidates in
    # one): from 64 MiB of planes on, where a scratch landed is worth +- 5 % of a transform whatever its plan
    # (profiles/r04_placement_probe.log; f32 2^24 x 1 in round 5: the run's own medians 175 -> 161 us, two other allocations
    # 168 against 175) -- there the margin is the placement noise, 8 %.
    margin = 1.03 if 2 * es * n * batch < (64 << 20) else 1.08
    for attempt in range(3):   # (re-measured on a miss: a few per cent on a shared box)
        us_h, us_t = _time_alternating([call_h, call_t], ring)
        print(f"\n{dt} 2^{L} x {batch} {kind}: heuristic {us_h:.2f} us, tuned {us_t:.2f} us ({rep['plan']}, "
              f"{rep['candidates']} plans in {rep['seconds']:.2f} s; the run's own medians {rep['us_heuristic']:.2f} / {rep['us_best']:.2f})")
        if us_t <= us_h * margin:
            break
    assert us_t <= us_h * margin, (us_h, us_t, margin, rep, tuned.describe())


def test_tune_2p20_takes_under_a_second(gpu, static_rules):
    """'Adds planning overhead proportional to FFT size' (planner.rs:30-31): N = 2^20, one transform -- every plan that exists
    (several hundred) screened and the finalists confirmed in under a second; with_mode(Tune) is that plus the planner."""
    import time

    P = gpu
    pl = P.PlannerDit64(1 << 20)
    rep = pl.tune(1)
    assert rep["candidates"] >= 100 and rep["seconds"] < 1.0, rep
    P.wisdom_forget()
    t0 = time.perf_counter()
    pl2 = P.PlannerDit64.with_mode(1 << 20, P.PlannerMode.Tune)
    dt_tune = time.perf_counter() - t0
    assert dt_tune < 1.5, dt_tune
    # ... and the measurement is remembered: a second Tune planner of this length does not measure again
    t0 = time.perf_counter()
    pl3 = P.PlannerDit64.with_mode(1 << 20, P.PlannerMode.Tune)
    assert time.perf_counter() - t0 < 0.25 * max(dt_tune, 0.2)
    assert pl3.describe() == pl2.describe()
    # lengths served by one kernel have nothing to tune
    rep = P.PlannerDit64(1 << 10).tune(1)
    assert rep["candidates"] == 0 and rep["plan"] == "one pass"
    with pytest.raises(P.PhastPanic):
        pl.tune(1, P.TuneKind.R2C)  # a kind of the other planner family
    with pytest.raises(P.PhastPanic):
        pl.tune(0)


def test_wisdom_travels_to_planners_made_later(gpu, oracle, static_rules):
    """Import a plan as wisdom text -> the next planner of that type and length runs it for the bucket it names (and only for
    that bucket and kind), bit-identical to a planner forced onto the same plan; forget -> planners are static again."""
    import torch

    P = gpu
    n = 1 << 20
    base = P.PlannerDit64(n)
    assert "tuned:" not in base.describe()
    P.wisdom_import("phastft-hip-wisdom 1 cus=%d\nf64 c2c 20 0 7,7,6@12,12,12:p8 fuse=0 us=1.0 heur=2.0\n" % P.device_info()["compute_units"])
    wise = P.PlannerDit64(n)
    assert "tuned:c2c/b0" in wise.describe(), wise.describe()
    forced = P.PlannerDit64(n)
    forced.set_plan((7, 7, 6), 12, 3)
    h_re, h_im = oracle.fill(n, np.float64, seed=5, transform_id=3)
    outs = []
    for pl in (wise, forced, base):
        d_re, d_im = dev(h_re.copy()), dev(h_im.copy())
        P.fft_64_dit_with_planner(d_re, d_im, P.Direction.Forward, pl)
        outs.append((d_re.cpu().numpy(), d_im.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    tol.check("wisdom", "f64", 20, *outs[0], *_ref_c2c(h_re, h_im))
    assert not np.array_equal(outs[0][0], outs[2][0])  # another plan than the static rule's: other last bits
    # a batch of 8 is another bucket: the static rule's plan, bit-identical to the planner without wisdom
    re = torch.from_numpy(np.tile(h_re, 8)).cuda(); im = torch.from_numpy(np.tile(h_im, 8)).cuda()
    re2, im2 = re.clone(), im.clone()
    P.fft_dit_batched(re, im, n, P.Direction.Forward, wise)
    P.fft_dit_batched(re2, im2, n, P.Direction.Forward, base)
    assert torch.equal(re, re2) and torch.equal(im, im2)
    # wisdom measured on a device with another CU count is not applied
    P.wisdom_forget()
    P.wisdom_import("phastft-hip-wisdom 1 cus=7\nf64 c2c 20 0 7,7,6@12,12,12:p8 fuse=0 us=1.0 heur=2.0\n")
    assert "tuned:" not in P.PlannerDit64(n).describe()
    P.wisdom_forget()
    assert "tuned:" not in P.PlannerDit64(n).describe()
    # what a tuning run finds is exported, and imported again it reproduces the tuned planner
    t = P.PlannerR2c64(1 << 17)
    rep = t.tune(32, P.TuneKind.R2C)
    text = P.wisdom_export()
    assert "f64 r2c 17 5 " in text, text
    P.wisdom_forget()
    P.wisdom_import(text)
    again = P.PlannerR2c64(1 << 17)
    assert ("tuned:r2c/b5" in again.describe()) == bool(rep["adopted"]), (rep, again.describe())
    if rep["adopted"]:
        assert again.describe() == t.describe()


# ---------------------------------------------------------------- the gates notice a perturbed twiddle
_PERTURB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import phastft_amd as P
from tests import tolerances as tol
dt, L, mode = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "single")
n = 1 << L
ndt = np.float64 if dt == "f64" else np.float32
rng = np.random.default_rng(11)
h_re, h_im = rng.uniform(-1, 1, n).astype(ndt), rng.uniform(-1, 1, n).astype(ndt)
pl = (P.PlannerDit64 if dt == "f64" else P.PlannerDit32)(n)
fft = P.fft_64_dit_with_planner if dt == "f64" else P.fft_32_dit_with_planner
z = np.fft.fft(h_re.astype(np.float64) + 1j * h_im.astype(np.float64))
if mode == "forced":        # a plan reached only through the plan hook: generic 4096-point tiles, three passes
    pl.set_plan((7, 7, 6), 12, 4)
    assert pl.describe_call().startswith("forced"), pl.describe_call()
if mode == "throughput":    # 2^25 points in flight: the throughput plan; the LAST transform of the batch is the one compared
    batch = (1 << 25) // n
    d_re = torch.from_numpy(np.tile(h_re, batch)).cuda()
    d_im = torch.from_numpy(np.tile(h_im, batch)).cuda()
    P.fft_dit_batched(d_re, d_im, n, P.Direction.Forward, pl)
    assert pl.describe_call(batch).split()[0] in ("throughput", "tuned"), pl.describe_call(batch)
    g_re, g_im = d_re[-n:].cpu().numpy(), d_im[-n:].cpu().numpy()
elif mode == "strided":     # column FFTs of a row-major [n][16] array (phast_fft_*_dit_strided_dev): column 3 is the one compared
    stride = 16
    a_re, a_im = rng.uniform(-1, 1, (n, stride)).astype(ndt), rng.uniform(-1, 1, (n, stride)).astype(ndt)
    a_re[:, 3], a_im[:, 3] = h_re, h_im
    d_re, d_im = torch.from_numpy(a_re.reshape(-1).copy()).cuda(), torch.from_numpy(a_im.reshape(-1).copy()).cuda()
    P.fft_dit_strided(d_re, d_im, n, P.Direction.Forward, pl, batch=stride, stride=stride)
    g_re = np.ascontiguousarray(d_re.cpu().numpy().reshape(n, stride)[:, 3])
    g_im = np.ascontiguousarray(d_im.cpu().numpy().reshape(n, stride)[:, 3])
else:
    d_re, d_im = torch.from_numpy(h_re.copy()).cuda(), torch.from_numpy(h_im.copy()).cuda()
    fft(d_re, d_im, P.Direction.Forward, pl)
    g_re, g_im = d_re.cpu().numpy(), d_im.cpu().numpy()
rel, worst = tol.rel_l2(g_re, g_im, z.real, z.imag), tol.max_bin_err(g_re, g_im, z.real, z.imag)
new_ok = (rel <= tol.f64_rel(L) and worst <= tol.f64_bin(L)) if dt == "f64" else (rel <= tol.f32_rel(L) and worst <= tol.f32_bin(L))
old_ok = rel <= tol.ROUND4_GATES[dt][0] and worst <= tol.ROUND4_GATES[dt][1]
print("RESULT", int(new_ok), int(old_ok), rel, worst)
"""


@pytest.mark.parametrize("dt,L,perturb,mode", [("f64", 20, "5e-13", "single"), ("f32", 20, "5e-5", "single"), ("f64", 24, "1e-9", "single"),
                                               # round 6: plans that until now were only ever compared
                                               # through the old gates -- a forced plan, a throughput plan, a strided batch
                                               ("f64", 20, "5e-13", "forced"), ("f64", 18, "5e-13", "throughput"),
                                               ("f64", 16, "5e-13", "strided")])
def test_gates_notice_a_perturbed_twiddle(gpu, dt, L, perturb, mode, tmp_path):
    """PHAST_TEST_PERTURB_TW3 (a test hook of Planner::table) puts a relative error on ONE entry of every three-level twiddle
    table.  The same script runs clean and perturbed: clean passes the gates of tests/tolerances.py; perturbed fails them --
    and for the small perturbations the round-4 gates (tolerances.ROUND4_GATES) would still have passed."""
    script = tmp_path / "perturb.py"
    script.write_text(_PERTURB_SCRIPT % {"root": ROOT})
    res = {}
    for name, val in (("clean", None), ("perturbed", perturb)):
        env = dict(os.environ)
        env.pop("PHAST_TEST_PERTURB_TW3", None)
        if val:
            env["PHAST_TEST_PERTURB_TW3"] = val
        r = subprocess.run([sys.executable, str(script), dt, str(L), mode], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0].split()
        res[name] = (int(line[1]), int(line[2]), float(line[3]), float(line[4]))
    assert res["clean"][0] == 1 and res["clean"][1] == 1, res
    assert res["perturbed"][0] == 0, res
    if perturb != "1e-9":
        assert res["perturbed"][1] == 1, ("the old gates were expected to miss this", res)


# ---------------------------------------------------------------- non-finite and subnormal inputs
def _run_kind(P, oracle, kind, dt, n, h_a, h_b):
    """(got arrays, oracle arrays) of one transform of kind c2c / r2c / c2r on host inputs (device-resident call)"""
    import torch

    f64 = dt == "f64"
    ndt = np.float64 if f64 else np.float32
    if kind == "c2c":
        d_re, d_im = dev(h_a.copy()), dev(h_b.copy())
        (P.fft_64_dit_with_planner if f64 else P.fft_32_dit_with_planner)(d_re, d_im, P.Direction.Forward, (P.PlannerDit64 if f64 else P.PlannerDit32)(n))
        o_re, o_im = h_a.copy(), h_b.copy()
        (oracle.fft_64_dit if f64 else oracle.fft_32_dit)(o_re, o_im, oracle.FORWARD)
        return (d_re.cpu().numpy(), d_im.cpu().numpy()), (o_re, o_im)
    h = n // 2 + 1
    pl = (P.PlannerR2c64 if f64 else P.PlannerR2c32)(n)
    if kind == "r2c":
        x = dev(h_a.copy())
        sr = torch.empty(h, dtype=x.dtype, device="cuda"); si = torch.empty_like(sr)
        P.r2c_fft_batched(x, sr, si, pl, 1)
        o_re, o_im = np.empty(h, ndt), np.empty(h, ndt)
        (oracle.r2c_fft_f64 if f64 else oracle.r2c_fft_f32)(h_a.copy(), o_re, o_im)
        return (sr.cpu().numpy(), si.cpu().numpy()), (o_re, o_im)
    sr, si = dev(h_a[:h].copy()), dev(h_b[:h].copy())
    out = torch.empty(n, dtype=sr.dtype, device="cuda")
    P.c2r_fft_batched(sr, si, out, pl, 1)
    o = np.empty(n, ndt)
    (oracle.c2r_fft_f64 if f64 else oracle.c2r_fft_f32)(h_a[:h].copy(), h_b[:h].copy(), o)
    return (out.cpu().numpy(),), (o,)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("L", [10, 20])
@pytest.mark.parametrize("kind", ["c2c", "r2c", "c2r"])
def test_nonfinite_and_subnormal_inputs_like_the_oracle(gpu, oracle, kind, L, dt):
    """Every output of an FFT depends on every input: one NaN (or one +Inf: Inf * w and Inf - Inf) in the input leaves NO finite
    output bin -- on the HIP path as in the oracle (which components are NaN, which Inf and which survive depends on where an
    algorithm multiplies by an exact 0 or 1, so only the finiteness of whole bins is compared).  Subnormal inputs are NOT flushed: the outputs agree with the oracle's
    to a few quanta of the subnormal range and are far from zero."""
    n = 1 << L
    ndt = np.float64 if dt == "f64" else np.float32
    rng = np.random.default_rng(L * 7 + len(kind))
    for bad in (np.nan, np.inf):
        a, b = rng.uniform(-1, 1, n).astype(ndt), rng.uniform(-1, 1, n).astype(ndt)
        a[n // 3] = bad
        got, want = _run_kind(gpu, oracle, kind, dt, n, a, b)
        # per BIN (c2r: per sample): a complex bin may keep ONE finite component where an algorithm skips the multiplication by a
        # trivial twiddle (W = +-1, +-i: the GPU's radix butterflies do, the oracle's table-driven stages multiply NaN by 0)
        bad_g = np.zeros(len(got[0]), bool)
        bad_w = np.zeros(len(want[0]), bool)
        for g, w in zip(got, want):
            bad_g |= ~np.isfinite(g)
            bad_w |= ~np.isfinite(w)
        assert bad_g.all() and bad_w.all(), (kind, dt, L, bad, float(bad_g.mean()), float(bad_w.mean()))
    # subnormal inputs: |x| < 2^-1040 (f64: subnormal below 2^-1022, quantum 2^-1074) / 2^-133 (f32: 2^-126, 2^-149).  Every
    # rounding on the way is at least half a quantum ABSOLUTE; through log2 N stages those errors random-walk like the signal,
    # ~ quantum * sqrt(N) (c2r: times its 1/(N/2) scale, plus the final rounding).
    scale, quantum = (2.0 ** -1040, 2.0 ** -1074) if dt == "f64" else (2.0 ** -133, 2.0 ** -149)
    a = (rng.uniform(-1, 1, n) * scale).astype(ndt)
    b = (rng.uniform(-1, 1, n) * scale).astype(ndt)
    assert np.count_nonzero(a) > n // 2 and float(np.max(np.abs(a))) < (2.0 ** -1022 if dt == "f64" else 2.0 ** -126)
    got, want = _run_kind(gpu, oracle, kind, dt, n, a, b)
    bound = 64.0 * np.sqrt(L) * quantum * np.sqrt(n)
    if kind == "c2r":
        bound = bound * 2.0 / n + 2.0 * quantum
    for g, w in zip(got, want):
        g64, w64 = g.astype(np.float64), w.astype(np.float64)
        assert np.isfinite(g64).all()
        peak = float(np.max(np.abs(w64)))
        assert peak > 16 * bound, (kind, dt, L, peak, bound)  # (the test discriminates: a flushed input would be an error of `peak`)
        assert float(np.max(np.abs(g64))) >= 0.25 * peak, (kind, dt, L, "flushed to zero?")
        assert float(np.max(np.abs(g64 - w64))) <= bound, (kind, dt, L, float(np.max(np.abs(g64 - w64))), bound)


# ---------------------------------------------------------------- stream capture
def test_capture_of_8192_points_needs_no_warmup_again(gpu, oracle):
    """8192-point planners route up to 128 transforms to a multi-pass twin, whose scratch cannot be allocated
    under capture -- a capture WITHOUT an eager call first (which the one-pass kernel never needed) failed.  Now such a capture
    runs the one-pass kernel; after an eager call (or reserve_batch) the twin's plan is captured."""
    import torch

    P = gpu
    n = 8192
    h_re, h_im = oracle.fill(n, np.float64, seed=9, transform_id=1)
    ref = _ref_c2c(h_re, h_im)
    for warm in (False, True):
        pl = P.PlannerDit64(n)
        if warm:
            pl.reserve_batch(4)
        g_re, g_im = dev(h_re.copy()), dev(h_im.copy())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                P.fft_64_dit_with_planner(g_re, g_im, P.Direction.Forward, pl)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        tol.check("capture8192", "f64", 13, g_re.cpu().numpy(), g_im.cpu().numpy(), *ref)
        del g


def test_capture_never_takes_another_threads_busy_workspace(gpu, oracle):
    """ADVICE r04 (medium).  Thread B keeps batches of 8 running through the planner on its own stream; the main thread warms
    up ONE transform on a stream of its own, captures it on another and replays while B's work is in flight.  Round 4's rule
    took the LARGEST fitting workspace under capture -- B's, still in use by B's kernels -- and the replay shared scratch with
    them.  Now a capture takes a workspace with nothing in flight, or this stream's, or this thread's own: every replay is
    bit-identical to the eager result."""
    import torch

    P = gpu
    n = 1 << 18
    pl = P.PlannerDit64(n)
    h_re, h_im = oracle.fill(n, np.float64, seed=21, transform_id=2)
    stop = threading.Event()
    started = threading.Event()

    def traffic():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            re = torch.rand(8 * n, dtype=torch.float64, device="cuda"); im = torch.rand_like(re)
            while not stop.is_set():
                for _ in range(16):
                    P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
                    P.fft_dit_batched(re, im, n, P.Direction.Reverse, pl)
                started.set()
            s.synchronize()

    t = threading.Thread(target=traffic)
    t.start()
    try:
        assert started.wait(60)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            a_re, a_im = dev(h_re.copy()), dev(h_im.copy())
            P.fft_64_dit_with_planner(a_re, a_im, P.Direction.Forward, pl)   # the warm-up: this thread's workspace
        s1.synchronize()
        want_re, want_im = a_re.clone(), a_im.clone()
        g_re, g_im = dev(h_re.copy()), dev(h_im.copy())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s2):
            # (thread-local capture mode: in the default global mode ANY thread's event query or allocation -- thread B's,
            # here -- invalidates a capture in progress, whatever library made the call)
            with torch.cuda.graph(g, stream=s2, capture_error_mode="thread_local"):
                P.fft_64_dit_with_planner(g_re, g_im, P.Direction.Forward, pl)
        src_re, src_im = dev(h_re.copy()), dev(h_im.copy())
        for _ in range(40):
            with torch.cuda.stream(s2):
                g_re.copy_(src_re); g_im.copy_(src_im)
                g.replay()
            s2.synchronize()
            assert torch.equal(g_re, want_re) and torch.equal(g_im, want_im)
    finally:
        stop.set()
        t.join(120)
    torch.cuda.synchronize()
    # the graph's workspace is the planner's until it is handed back
    before = pl.device_bytes()
    del g
    freed = pl.release_graph_workspaces()
    assert freed >= 2 * n * 8 and pl.device_bytes() <= before - freed + 4096, (freed, before, pl.device_bytes())
    # ... and eager calls still work
    P.fft_64_dit_with_planner(a_re, a_im, P.Direction.Reverse, pl)
    torch.cuda.synchronize()
    assert float((a_re.cpu() - torch.from_numpy(h_re)).abs().max()) < 1e-12


def test_replaced_plans_leave_no_tables_behind(gpu, static_rules):
    """every set_plan used to park the replaced plan's tables until the planner died and count them twice.
    Tables are shared per planner now: forty plan changes between three plans add three plans' worth of tables, once."""
    P = gpu
    pl = P.PlannerDit64(1 << 20)
    plans = [((7, 7, 6), 12, 3), ((6, 8, 6), 12, 3), ((10, 10), 13, 4)]
    for lrs, tl, lp in plans:
        pl.set_plan(lrs, tl, lp)
    settled = pl.device_bytes()
    for k in range(40):
        lrs, tl, lp = plans[k % 3]
        pl.set_plan(lrs, tl, lp)
    assert pl.device_bytes() == settled, (settled, pl.device_bytes())
    pl.set_plan(())
    assert pl.device_bytes() <= settled + (1 << 20)


# ---------------------------------------------------------------- a torch-free sharded host over the C ABI + RCCL
def test_torch_free_sharded_host_with_rccl_gather(gpu, oracle, tmp_path):
    """tests/cpp/shard_host.cpp: BASELINE configs[4]'s path without Python or torch -- one C++ thread per visible device, a
    planner per device, on-device fill -> phast_fft_64_dit_dev -> phast_digest_f64_dev, ncclCommInitAll + ONE ncclAllGather of
    the 32-byte digests through librccl.so.  Compiled and run here on the visible device(s) (an RCCL communicator of that
    size); every gathered digest is checked: Parseval for all transforms, sampled ones against digests of the CPU oracle's
    output of the same seeded inputs (the oracle is the checker).  Its throughput is that of the Python host's path."""
    import json

    import torch

    from phastft_amd import build

    lib = build.build()
    exe = tmp_path / "shard_host"
    r = subprocess.run([build.hipcc(), "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "shard_host.cpp"),
                        "-I", os.path.join(ROOT, "include"), "-L", os.path.dirname(lib), "-lphastft_hip", "-L", "/opt/rocm/lib", "-lrccl",
                        "-pthread", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    gpus = torch.cuda.device_count()
    shard, n = 256, 1 << 20
    dig = tmp_path / "digests.bin"
    env = {k: v for k, v in os.environ.items() if not k.startswith(("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_"))}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([str(exe), "--shard", str(shard), "--steps", "5", "--warmup", "2", "--digests-out", str(dig)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    total = shard * gpus
    assert out["n_gpus"] == gpus and out["scaling"] == "weak" and out["config"]["digest_ok"] is True
    assert "ncclAllGather" in out["config"]["digest_gather"]
    # round 6: the line validates against the schema both hosts share, and so does bench.py's N > 1 line
    # (two ranks on this box's one GPU through gloo: the same code path as the driver's torch.distributed.run launch)
    from tests.test_bench_helpers import validate_multi_gpu_line

    assert validate_multi_gpu_line(out) == [], out
    rb = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--backend", "gloo", "--shard", "8",
                         "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert rb.returncode == 0, rb.stdout[-2000:] + rb.stderr[-3000:]
    bl = json.loads([ln for ln in rb.stdout.splitlines() if ln.startswith("{")][-1])
    assert validate_multi_gpu_line(bl) == [], bl
    assert bl["n_gpus"] == 2 and bl["config"]["ranks_seen"] == 2 and bl["config"]["backend"] == "gloo" and bl["config"]["shard"] == 8
    d = np.fromfile(dig, dtype=np.float64).reshape(total, 4)
    # Parseval on every transform: sum |X|^2 = N sum |x|^2, inputs regenerated with the same counter-based generator
    ids = sorted({0, 1, shard // 2, shard - 1, total - 1})
    for tid in ids:
        h_re, h_im = oracle.fill(n, np.float64, seed=0xCAFE, transform_id=tid)
        e_in = float(np.sum(h_re ** 2 + h_im ** 2))
        assert abs(d[tid, 2] / (n * e_in) - 1.0) < 1e-12, tid
        oracle.fft_64_dit(h_re, h_im, oracle.FORWARD)
        scale = np.sqrt(n * e_in)
        assert abs(d[tid, 0] - h_re.sum()) <= 1e-10 * scale * np.sqrt(n) and abs(d[tid, 1] - h_im.sum()) <= 1e-10 * scale * np.sqrt(n), tid
        assert abs(d[tid, 2] / float(np.sum(h_re ** 2 + h_im ** 2)) - 1.0) <= 1e-12 and abs(d[tid, 3] - h_re[1]) <= 1e-12 * scale, tid
    assert np.all(np.isfinite(d)) and np.all(d[:, 2] > 0)
    # the same shard through the Python host on this device: the torch-free host is not slower (3 % + noise)
    P = gpu
    pl = P.PlannerDit64(n)
    re = torch.empty(shard * n, dtype=torch.float64, device="cuda"); im = torch.empty_like(re)
    P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)
    for _ in range(2):
        P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
    P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        P.fft_dit_batched(re, im, n, P.Direction.Forward, pl)
    e1.record()
    torch.cuda.synchronize()
    py_value = shard * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    print(f"\nshard_host {out['value']:.1f} GSamples/s over {gpus} device(s) (per device {out['value'] / gpus:.1f}); Python host {py_value:.1f} on one")
    # (one device is what this pool has measured; with several, the threads of one process share a host and the step ends with
    # the slowest device: the bar there is that nothing serialises them)
    assert out["value"] / gpus >= (0.9 if gpus == 1 else 0.6) * py_value, (out["value"], gpus, py_value)


# ---------------------------------------------------------------- the committed error budget still holds
def test_error_budget_holds(gpu):
    """tests/golden/error_budget.json (tests/golden/make_error_budget.py on the MI355X) is what the gates of tests/tolerances.py
    are derived from: every committed value sits at least 2.5 x below its gate, and a sample re-measured here with the
    generator's own code (whatever plans wisdom has put in force since) still sits 2 x below."""
    import importlib.util
    import json

    budget = json.load(open(os.path.join(ROOT, "tests", "golden", "error_budget.json")))["budget"]
    for key, per_len in budget.items():
        dt = "f64" if "f64" in key else "f32"
        for L, e in per_len.items():
            if "roundtrip" in key or "vs_oracle" in key:
                continue  # (two transforms / the reference's f32 twiddles: their own criteria)
            g_rel, g_bin = (tol.f64_rel(int(L)), tol.f64_bin(int(L))) if dt == "f64" else (tol.f32_rel(int(L)), tol.f32_bin(int(L)))
            assert e["rel"] * 2.5 <= g_rel and e["bin"] * 2.5 <= g_bin, (key, L, e, g_rel, g_bin)
    spec = importlib.util.spec_from_file_location("make_error_budget", os.path.join(ROOT, "tests", "golden", "make_error_budget.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for lo, hi in ((10, 10), (16, 16), (20, 20)):
        now = mod.measure(lo, hi, seeds=(1,))
        for key, per_len in now.items():
            dt = "f64" if "f64" in key else "f32"
            for L, e in per_len.items():
                if "roundtrip" in key or "vs_oracle" in key:
                    continue
                g_rel, g_bin = (tol.f64_rel(int(L)), tol.f64_bin(int(L))) if dt == "f64" else (tol.f32_rel(int(L)), tol.f32_bin(int(L)))
                assert e["rel"] * 2.0 <= g_rel and e["bin"] * 2.0 <= g_bin, (key, L, e, budget[key][L], g_rel, g_bin)


# ---------------------------------------------------------------- 4096 points on the multi-pass twin (per call kind)
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_4096_points_run_the_twin_where_it_was_measured_faster(gpu, oracle, dt, static_rules):
    """Round 5 gave 4096 points the multi-pass twin round 4 gave 8192 -- per call kind, as measured
    (profiles/r05_small_twin_4096.log): C2C (8.5 -> 7.7 us in f64) and C2R of 8192 real points (12.5 -> 8.7) run two passes of
    four 64 x 16 tiles, R2C keeps the one-pass kernel (its untangle would become a third kernel).  Whatever the route: the same
    bits from host slices, device pointers and a captured graph (captured WITHOUT a warm-up call too: the fallback to the
    one-pass kernel is another factorisation, so only the tolerance holds there), results within the gates."""
    import torch

    P = gpu
    f64 = dt == "f64"
    npdt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    n = 1 << 12
    pl = (P.PlannerDit64 if f64 else P.PlannerDit32)(n)
    assert pl.describe_call(1).startswith("single [64x16A p8][64x16 p8]"), pl.describe_call(1)
    assert pl.describe_call(500) == "one-pass"
    fft = P.fft_64_dit_with_planner if f64 else P.fft_32_dit_with_planner
    h_re, h_im = oracle.fill(n, npdt, seed=0x412, transform_id=1)
    ref = _ref_c2c(h_re, h_im)
    a_re, a_im = h_re.copy(), h_im.copy()
    fft(a_re, a_im, P.Direction.Forward, pl)                                     # host slices
    tol.check("twin4096", dt, 12, a_re, a_im, *ref)
    d_re, d_im = dev(h_re.copy()), dev(h_im.copy())
    fft(d_re, d_im, P.Direction.Forward, pl)                                     # device pointers: the same bits
    assert np.array_equal(d_re.cpu().numpy(), a_re) and np.array_equal(d_im.cpu().numpy(), a_im)
    g_re, g_im = dev(h_re.copy()), dev(h_im.copy())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(3):   # every call of the capture on the twin, not only the first (this round's capture_ready bug)
                fft(g_re, g_im, P.Direction.Forward, pl)
                fft(g_re, g_im, P.Direction.Reverse, pl)
            fft(g_re, g_im, P.Direction.Forward, pl)
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    tol.check("twin4096:graph", dt, 12, g_re.cpu().numpy(), g_im.cpu().numpy(), *ref)
    fft(d_re, d_im, P.Direction.Reverse, pl)
    assert float((d_re.cpu() - torch.from_numpy(h_re)).abs().max()) <= (tol.f64_bin(12) if f64 else 5 * tol.ROUNDTRIP_ABS["f32"])
    batch = 37                                                                    # a batch below the twin's limit: every row the same bits
    b_re, b_im = dev(np.tile(h_re, batch)), dev(np.tile(h_im, batch))
    P.fft_dit_batched(b_re, b_im, n, P.Direction.Forward, pl)
    rows = b_re.cpu().numpy().reshape(batch, n)
    assert all(np.array_equal(rows[b], a_re) for b in range(batch))
    # real transforms of 8192 points: C2R on the twin, R2C in the one-pass kernel
    m = 2 * n
    rp = (P.PlannerR2c64 if f64 else P.PlannerR2c32)(m)
    assert rp.describe_call(1, P.TuneKind.C2R).startswith("single ") and rp.describe_call(1, P.TuneKind.R2C) == "one-pass"
    x, _ = oracle.fill(m, npdt, seed=0x413, transform_id=2)
    X = np.fft.rfft(x.astype(np.float64))
    ore, oim = np.zeros(m // 2 + 1, npdt), np.zeros(m // 2 + 1, npdt)
    (P.r2c_fft_f64_with_planner if f64 else P.r2c_fft_f32_with_planner)(x, ore, oim, rp)
    tol.check("twin4096:r2c", dt, 13, ore, oim, X.real, X.imag)
    c2r = P.c2r_fft_f64_with_planner if f64 else P.c2r_fft_f32_with_planner
    back = np.zeros(m, npdt)
    c2r(ore, oim, back, rp)                                                       # host slices
    want = np.fft.irfft(ore.astype(np.float64) + 1j * oim.astype(np.float64), m)
    tol.check("twin4096:c2r", dt, 13, back, np.zeros(m), want, np.zeros(m))
    t_out = torch.zeros(m, dtype=tdt, device="cuda")
    c2r(dev(ore), dev(oim), t_out, rp)                                            # device pointers: the same bits
    assert np.array_equal(t_out.cpu().numpy(), back)


def test_wisdom_file_carries_a_tuning_run_to_the_next_process(gpu, tmp_path):
    """PHAST_WISDOM=<path>: process 1 tunes f64 2^18 x 16 (the static rule loses by ~30 % there) -- the library rewrites the file;
    process 2, same variable, makes a planner of that length and starts with the measured plan without measuring anything."""
    path = tmp_path / "wisdom.txt"
    env = dict(os.environ, PHAST_WISDOM=str(path), PHAST_BUILTIN_WISDOM="0")
    code1 = ("import sys; sys.path.insert(0, %r)\nimport phastft_amd as P\npl = P.PlannerDit64(1 << 18)\n"
             "rep = pl.tune(16)\nprint('ADOPTED', int(rep['adopted']), rep['plan'])\nprint('CALL', pl.describe_call(16))" % ROOT)
    r1 = subprocess.run([sys.executable, "-c", code1], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    adopted = "ADOPTED 1" in r1.stdout
    text = path.read_text()
    assert text.startswith("phastft-hip-wisdom 1 cus=") and "f64 c2c 18 4 " in text, text
    assert ("f64 c2c 18 4 heuristic" in text) == (not adopted)
    code2 = ("import sys; sys.path.insert(0, %r)\nimport phastft_amd as P\npl = P.PlannerDit64(1 << 18)\n"
             "print('CALL', pl.describe_call(16))\nprint('OTHER', pl.describe_call(1))" % ROOT)
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-3000:]
    call1 = [ln for ln in r1.stdout.splitlines() if ln.startswith("CALL")][0]
    call2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("CALL")][0]
    assert call1 == call2 and call2.startswith("CALL tuned ") == adopted, (call1, call2)
    assert "tuned" not in [ln for ln in r2.stdout.splitlines() if ln.startswith("OTHER")][0]   # another bucket: the static rules


# ---------------------------------------------------------------- a tuning run beside callers of the same planner
def test_tuning_while_other_threads_transform_with_the_planner(gpu, static_rules):
    """The planners are shared values (planner.rs:38-39: Send + Sync) and a tuning run CHANGES one: it holds the plans shared
    while it measures and exclusively for the moment it installs the winner (planner_plans.hpp: install_built).  Three threads
    keep 16 x 2^18 f64 transforms going through one planner, each on its own stream and buffers, while the main thread tunes
    that very call (f64 2^18 x 16: the static rule's plan loses by 25 % there, so a plan IS swapped under them) and once more
    for another bucket: every result before, during and after the swap is inside the gates of a float64 reference, nothing
    deadlocks, and the threads' later calls run the tuned plan."""
    import torch

    P = gpu
    L, batch = 18, 16
    n = 1 << L
    pl = P.PlannerDit64(n)
    rng = np.random.default_rng(77)
    h_re, h_im = rng.uniform(-1, 1, batch * n), rng.uniform(-1, 1, batch * n)
    z = np.fft.fft((h_re + 1j * h_im).reshape(batch, n), axis=1).reshape(-1)
    src = torch.stack((dev(h_re), dev(h_im)))
    want = torch.stack((dev(z.real.copy()), dev(z.imag.copy())))
    den = float(torch.linalg.vector_norm(want))
    rms = den / np.sqrt(2 * batch * n)
    stop = threading.Event()
    errors, counts = [], [0, 0, 0]

    def caller(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                buf = torch.empty_like(src)
                while not stop.is_set():
                    buf.copy_(src)
                    P.fft_dit_batched(buf[0], buf[1], n, P.Direction.Forward, pl)
                    d = buf - want
                    rel = float(torch.linalg.vector_norm(d)) / den
                    worst = float(d.abs().max()) / rms
                    if not (rel <= tol.f64_rel(L) and worst <= tol.f64_bin(L) * np.sqrt(2.0)):
                        errors.append((k, counts[k], rel, worst))
                        return
                    counts[k] += 1
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=caller, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    try:
        import time

        t0 = time.time()
        while min(counts) < 5 and not errors and time.time() - t0 < 60:
            time.sleep(0.01)
        before = list(counts)
        rep = pl.tune(batch)
        rep2 = pl.tune(2)
        mid = list(counts)
        t0 = time.time()
        while min(c - m for c, m in zip(counts, mid)) < 20 and not errors and time.time() - t0 < 60:
            time.sleep(0.01)
    finally:
        stop.set()
        for t in threads:
            t.join(120)
    torch.cuda.synchronize()
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    assert min(before) >= 5 and min(c - m for c, m in zip(counts, mid)) >= 20, (before, mid, counts)
    assert rep["candidates"] >= 8 and rep2["candidates"] >= 8, (rep, rep2)
    print(f"\ntuned beside 3 callers: {rep['plan']} ({rep['us_heuristic']:.1f} -> {rep['us_best']:.1f} us, adopted={rep['adopted']}), "
          f"x2: {rep2['plan']}; calls per thread before / during / after: {before} / {[m - b for m, b in zip(mid, before)]} / {[c - m for c, m in zip(counts, mid)]}")
    if rep["adopted"]:
        assert pl.describe_call(batch).startswith("tuned "), pl.describe_call(batch)
    P.wisdom_forget()


def test_cpp_tuning_beside_callers_program(gpu, tmp_path):
    """tests/cpp/tune_beside_callers_test.cpp -- the interleaving of the test above as a C++ program over the C ABI (it is what
    runs under ThreadSanitizer / AddressSanitizer, tools/sanitize_host.sh): a tuning run, a second bucket and a PHAST_MODE_TUNE
    planner beside three calling threads; every call succeeds and every transform keeps Parseval whichever plan ran it."""
    import json

    from phastft_amd import build

    lib = build.build()
    libdir = os.path.dirname(lib)
    exe = str(tmp_path / "tune_beside_callers_test")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
           os.path.join(ROOT, "tests", "cpp", "tune_beside_callers_test.cpp"), "-o", exe,
           "-L", libdir, "-lphastft_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = {k: v for k, v in os.environ.items() if not k.startswith("PHAST_")}
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(res)
    assert res["bad_transforms"] == 0 and min(res["calls"]) >= 13 and res["tune"]["candidates"] >= 8, res
    if res["tune"]["adopted"]:
        assert res["call_now"].startswith("tuned "), res


def test_a_graph_captured_before_a_tuning_run_keeps_replaying_its_plan(gpu, static_rules):
    """A captured call holds its kernels' arguments by value -- table pointers (the planner's table cache: released with the
    planner, never with a plan) and the scratch of the workspace that went to the graph.  A tuning run afterwards widens the
    scratch pitch, installs another plan for the very call and drops the candidates' descriptors: the graph's replays must not
    notice (bit-identical to the replay before), eager calls run the tuned plan (inside the gates, not bit-identical), and a
    new capture after one eager call records the tuned plan."""
    import torch

    P = gpu
    L, batch = 18, 16
    n = 1 << L
    pl = P.PlannerDit64(n)
    rng = np.random.default_rng(78)
    h_re, h_im = rng.uniform(-1, 1, batch * n), rng.uniform(-1, 1, batch * n)
    z = np.fft.fft((h_re + 1j * h_im).reshape(batch, n), axis=1).reshape(-1)
    src_re, src_im = dev(h_re), dev(h_im)
    s = torch.cuda.Stream()

    def captured():
        g_re, g_im = src_re.clone(), src_im.clone()
        with torch.cuda.stream(s):
            P.fft_dit_batched(g_re, g_im, n, P.Direction.Forward, pl)   # eager once: the workspace a capture may take
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                P.fft_dit_batched(g_re, g_im, n, P.Direction.Forward, pl)
        return g, g_re, g_im

    def replay(g, g_re, g_im):
        with torch.cuda.stream(s):
            g_re.copy_(src_re); g_im.copy_(src_im)
            g.replay()
        s.synchronize()
        return g_re.clone(), g_im.clone()

    g1 = captured()
    a_re, a_im = replay(*g1)
    tol.check("graph before tune", "f64", L, a_re.cpu().numpy(), a_im.cpu().numpy(), z.real, z.imag)
    plan_before = pl.describe_call(batch)
    rep = pl.tune(batch)
    assert rep["adopted"], rep   # (f64 2^18 x 16: the static rule's plan loses by 20 .. 25 % on every box so far)
    assert pl.describe_call(batch).startswith("tuned ") and pl.describe_call(batch) != plan_before
    for _ in range(3):
        b_re, b_im = replay(*g1)
        assert torch.equal(a_re, b_re) and torch.equal(a_im, b_im)      # the graph still runs the plan it captured
    e_re, e_im = src_re.clone(), src_im.clone()
    P.fft_dit_batched(e_re, e_im, n, P.Direction.Forward, pl)
    torch.cuda.synchronize()
    tol.check("eager after tune", "f64", L, e_re.cpu().numpy(), e_im.cpu().numpy(), z.real, z.imag)
    assert not (torch.equal(e_re, a_re) and torch.equal(e_im, a_im))    # another plan: the last bits differ
    g2 = captured()
    c_re, c_im = replay(*g2)
    assert torch.equal(c_re, e_re) and torch.equal(c_im, e_im)          # the new capture recorded the tuned plan
    b_re, b_im = replay(*g1)
    assert torch.equal(a_re, b_re) and torch.equal(a_im, b_im)
    del g1, g2
    pl.release_graph_workspaces()
    P.wisdom_forget()


# ---------------------------------------------------------------- every plan the library ships as built-in wisdom
def _builtin_entries():
    import re

    path = os.path.join(ROOT, "phastft_amd", "csrc", "builtin_wisdom.inc")
    out = {}
    for m in re.finditer(r'^"(f64|f32) (c2c|c2ci|r2c|c2r) (\d+) (\d+) (\S+) fuse=([01])', open(path).read(), re.M):
        if m.group(5) != "heuristic":
            out.setdefault((m.group(1), "real" if m.group(2) in ("r2c", "c2r") else "cplx", int(m.group(3))), []).append(
                (m.group(2), int(m.group(4)), m.group(5), int(m.group(6))))
    return out


def test_every_builtin_wisdom_plan_runs_its_call_within_the_gates(gpu):
    """csrc/builtin_wisdom.inc holds ~ 425 (type, call kind, length, batch bucket) -> plan lines that a tuning run on an MI355X
    adopted after a coarse result check (digests at two bins).  Here EVERY line is held to the parity gates at the call it
    was measured for: a planner made with the wisdom in force must say it runs that plan for a batch in the bucket (the line
    names a plan this build can make), and its output -- all transforms of the batch, every bin, compared on the device --
    must agree with the static rule's plan of a planner made without the wisdom (parity-tested against the oracle throughout
    tests/test_gpu_parity*.py) within the sum of the two gates; up to 2^20 points the first and the last transform also
    against numpy in float64.  C2C calls also run in reverse through the tuned plan (the round trip returns the input); C2R inputs are
    half-spectra of real signals (made by r2c), so the round trip must return them."""
    import re

    import torch

    P = gpu
    entries = _builtin_entries()
    assert sum(len(v) for v in entries.values()) >= 300
    rng = np.random.default_rng(2025)
    kinds = {"c2c": P.TuneKind.C2C, "c2ci": P.TuneKind.C2CInterleaved, "r2c": P.TuneKind.R2C, "c2r": P.TuneKind.C2R}
    checked = 0

    def gates(dt, L):
        return (tol.f64_rel(L), tol.f64_bin(L)) if dt == "f64" else (tol.f32_rel(L), tol.f32_bin(L))

    def close(tag, dt, L, got, want, n_per, factor=2.0):
        """device tensors [batch * n_per] (real views of whatever the call wrote): rel-L2 over the batch, worst element against
        the rms element, both within `factor` x the gate (two plans, each inside its own gate of the exact transform)"""
        g, w = got.to(torch.float64), want.to(torch.float64)
        den = float(torch.linalg.vector_norm(w))
        rel = float(torch.linalg.vector_norm(g - w)) / den
        worst = float((g - w).abs().max()) / (den / np.sqrt(w.numel()))
        g_rel, g_bin = gates(dt, L)
        # (per-bin gate: rms over re and im together is 1/sqrt(2) of the rms bin magnitude the gate is written for)
        assert rel <= factor * g_rel and worst <= factor * g_bin * np.sqrt(2.0), (tag, rel, factor * g_rel, worst, factor * g_bin)

    for (dt, cls, L), lines in sorted(entries.items()):
        n = 1 << L
        tdt = torch.float64 if dt == "f64" else torch.float32
        Pl = (P.PlannerR2c64 if dt == "f64" else P.PlannerR2c32) if cls == "real" else (P.PlannerDit64 if dt == "f64" else P.PlannerDit32)
        was = P.wisdom_builtin(False)
        try:
            static = Pl(n)
            P.wisdom_builtin(True)   # this test is about the shipped table: on for the planner that runs it
            tuned = Pl(n)
        finally:
            P.wisdom_builtin(was)
        for kind, bucket, plan, fuse in lines:
            batch = 1 << bucket
            tag = f"{dt} {kind} 2^{L} x {batch} {plan}"
            said = tuned.describe_call(batch, kinds[kind])
            assert said.startswith("tuned "), (tag, said)
            assert not static.describe_call(batch, kinds[kind]).startswith("tuned "), tag
            lrs, tls = (list(map(int, part.split(","))) for part in plan.split(":")[0].split("@"))
            shapes = [(int(r), int(c)) for r, c in re.findall(r"\[(\d+)x(\d+)A? ", said)]
            assert shapes == [(1 << lr, 1 << (tl - lr)) for lr, tl in zip(lrs, tls)], (tag, said)
            if kind == "r2c":
                assert ("untangle-fused" in said) == bool(fuse), (tag, said)
            if kind in ("c2c", "c2ci"):
                re0 = torch.empty(batch * n, dtype=tdt, device="cuda").uniform_(-1, 1)
                im0 = torch.empty_like(re0).uniform_(-1, 1)
                outs = []
                for pl in (static, tuned):
                    if kind == "c2c":
                        a, b = re0.clone(), im0.clone()
                        P.fft_dit_batched(a, b, n, P.Direction.Forward, pl)
                        outs.append(torch.stack((a, b)))
                    else:
                        assert batch == 1
                        z = torch.complex(re0, im0)
                        (P.fft_64_interleaved_with_planner if dt == "f64" else P.fft_32_interleaved_with_planner)(z, P.Direction.Forward, pl)
                        outs.append(torch.stack((z.real, z.imag)))
                close(tag, dt, L, outs[1].flatten(), outs[0].flatten(), n)
                # ... and back through the same plan (swap trick, 1/N in the last pass's store -- whatever tile that is)
                if kind == "c2c":
                    a, b = outs[1][0].clone(), outs[1][1].clone()
                    P.fft_dit_batched(a, b, n, P.Direction.Reverse, tuned)
                    back = torch.stack((a, b))
                else:
                    z = torch.complex(outs[1][0], outs[1][1])
                    (P.fft_64_interleaved_with_planner if dt == "f64" else P.fft_32_interleaved_with_planner)(z, P.Direction.Reverse, tuned)
                    back = torch.stack((z.real, z.imag))
                close(tag + " round trip", dt, L, back.flatten(), torch.stack((re0, im0)).flatten(), n, factor=3.0)
                if L <= 20:
                    for b in sorted({0, batch - 1}):
                        sl = slice(b * n, (b + 1) * n)
                        tol.check("builtin:" + kind, dt, L, outs[1][0][sl].cpu().numpy(), outs[1][1][sl].cpu().numpy(),
                                  *_ref_c2c(re0[sl].cpu().numpy(), im0[sl].cpu().numpy()))
            else:
                h = n // 2 + 1
                x = torch.empty(batch * n, dtype=tdt, device="cuda").uniform_(-1, 1)
                sr, si = (torch.empty(batch * h, dtype=tdt, device="cuda") for _ in range(2))
                P.r2c_fft_batched(x, sr, si, static, batch)
                if kind == "r2c":
                    tr, ti = torch.empty_like(sr), torch.empty_like(si)
                    P.r2c_fft_batched(x, tr, ti, tuned, batch)
                    close(tag, dt, L, torch.stack((tr, ti)).flatten(), torch.stack((sr, si)).flatten(), h)
                    if L <= 20:
                        for b in sorted({0, batch - 1}):
                            ref = np.fft.rfft(x[b * n:(b + 1) * n].cpu().numpy().astype(np.float64))
                            tol.check("builtin:r2c", dt, L, tr[b * h:(b + 1) * h].cpu().numpy(), ti[b * h:(b + 1) * h].cpu().numpy(), ref.real, ref.imag)
                else:
                    ys, yt = torch.empty_like(x), torch.empty_like(x)
                    P.c2r_fft_batched(sr.clone(), si.clone(), ys, static, batch)
                    P.c2r_fft_batched(sr, si, yt, tuned, batch)
                    close(tag, dt, L, yt, ys, n)
                    close(tag + " round trip", dt, L, yt, x, n, factor=3.0)  # r2c (static) + c2r (tuned)
            checked += 1
        del static, tuned
        torch.cuda.empty_cache()
    print(f"\n{checked} built-in wisdom lines checked at their own call")
    assert checked >= 300
