#!/usr/bin/env python3
"""Second opinion on every line of phastft_amd/csrc/builtin_wisdom.inc, in the protocol the calls are BENCHED in.

The in-library tuner (csrc/tune.hpp) ranks plans by eager launches over a ring; bench.py and a caller with a HIP graph see
the same kernels back to back without the launch gaps, and a plan that wins by 4 % in the first protocol can lose by 10 % in
the second (round 6: `f32 r2c 22 0` claimed +4.8 %, ran 41.7 us against the static rule's 36.4 us in the size ladder of
profiles/r06_vs_r05_size_ladder.log).  This tool replays, for EVERY line, the interleaved A/B of tests/test_gpu_wisdom.py
(wisdom plan against the static rule, HIP graph of the calls on a cold ring, caches drained before every replay, medians of 7 rounds) and rewrites the table
with the lines that are faster in BOTH protocols by at least --keep (default 0.96: 4 %; --keep-large, 8 %, from 2^25 points in
flight on); a line between 2 % and 4 % stays if the tuner's own margin was 8 % or more.  Lines too large to replay twice in
memory (more than 2^--max-points points in flight) are dropped.

    python tools/verify_builtin_wisdom.py [--inc phastft_amd/csrc/builtin_wisdom.inc] [--out gpurun_out/builtin_wisdom.verified.inc]
                                          [--log gpurun_out/wisdom_verify.log] [--keep 0.98] [--max-points 28]
"""
import argparse
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402
from tests.test_gpu_wisdom import _bench_call, _time  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--inc", default=os.path.join(ROOT, "phastft_amd", "csrc", "builtin_wisdom.inc"))
ap.add_argument("--out", default="gpurun_out/builtin_wisdom.verified.inc")
ap.add_argument("--log", default="gpurun_out/wisdom_verify.log")
ap.add_argument("--keep", type=float, default=0.96, help="a line stays if graph-protocol time(wisdom) <= keep * time(static)")
ap.add_argument("--keep-large", type=float, default=1 / 1.08, help="the same from 2^25 points in flight on, where the placement of a planner's "
                "scratch moves a call by +-5 %% whatever the plan (profiles/r04_placement_probe.log; bench.py read `f32 c2c 26 0`, +5 %% here, as -5 %% and 0 %%)")
ap.add_argument("--max-points", type=int, default=28)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--imported", action="store_true", help="--inc is a fresh table that is NOT compiled into the library yet: its lines are imported "
                "(phast_wisdom_import) for the 'on' planner and forgotten for the 'off' planner, the compiled table stays off throughout")
a = ap.parse_args()
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)

src = open(a.inc).read().splitlines()
LINE = re.compile(r'^"(f64|f32) (c2c|c2ci|r2c|c2r) (\d+) (\d+) (\S+) fuse=(\d) us=([\d.]+) heur=([\d.]+)\\n"$')
log = open(a.log, "w")


def say(s):
    print(s, flush=True)
    log.write(s + "\n")
    log.flush()




class ImportedTable:
    """stands in for the module in _bench_call: `wisdom_builtin(on)` switches the table under test, which here is imported text"""

    def __init__(self, text):
        self.text = text
        P.wisdom_builtin(False)

    def __getattr__(self, name):
        return getattr(P, name)

    def wisdom_builtin(self, on):
        P.wisdom_forget()
        if on:
            P.wisdom_import(self.text)
        return False


GPU = P
if a.imported:
    GPU = ImportedTable("\n".join(re.findall(r'^"(.*)\\n"$', "\n".join(src), flags=re.M)) + "\n")
info = P.device_info()
say(f"# {info['name']} {info['compute_units']} CUs; wisdom plan vs static rule, HIP graph on a cold ring, median of {a.rounds} interleaved rounds")
say(f"# a line stays if on <= {a.keep} * off ({a.keep_large:.3f} from 2^25 points in flight on); input {os.path.relpath(a.inc, ROOT)}")
kept, dropped, t0 = [], 0, time.time()
for ln in src:
    m = LINE.match(ln)
    if not m:
        kept.append(ln)          # comments and the header line
        continue
    dt, kind, L, bucket, plan, fuse, us, heur = m.groups()
    L, bucket, us, heur = int(L), int(bucket), float(us), float(heur)
    points = L + bucket - (1 if kind in ("r2c", "c2r") else 0)
    tag = f"{dt} {kind} {L} {bucket}"
    if points > a.max_points:
        say(f"{tag}: DROP (2^{points} points in flight: not replayed)")
        dropped += 1
        continue
    try:
        on = _bench_call(GPU, torch, dt, kind, L, bucket, True)
        if on is None or not on[2].startswith("tuned"):
            say(f"{tag}: DROP (the planner does not take the line: {None if on is None else on[2][:50]})")
            dropped += 1
            continue
        off = _bench_call(GPU, torch, dt, kind, L, bucket, False, reuse=on[3])
        t_on, t_off = [], []
        for _ in range(a.rounds):
            t_on.append(_time(torch, on[0], on[1], P))
            t_off.append(_time(torch, off[0], off[1], P))
        m_on, m_off = float(np.median(t_on)), float(np.median(t_off))
    except (P.PhastHipError, P.PhastPanic, RuntimeError) as e:
        say(f"{tag}: DROP ({type(e).__name__}: {str(e)[:80]})")
        dropped += 1
        torch.cuda.empty_cache()
        continue
    ok = m_on <= (a.keep_large if points >= 25 else a.keep) * m_off
    # ... or, below 2^25 points: two measurement protocols that agree on the sign, one of them by a wide margin (launch-bound
    # transforms of ~10 us: `f32 r2c 15 0` is +10 % / +13 % for the tuner's eager launches in two rounds and +3 % as a graph)
    ok = ok or (points < 25 and m_on <= 0.98 * m_off and heur >= 1.08 * us)
    say(f"{tag}: {'keep' if ok else 'DROP'}  graph {m_on:.2f} vs static {m_off:.2f} us ({100 * (m_off / m_on - 1):+.1f} %); tuner said {us:.2f} vs {heur:.2f} "
        f"({100 * (heur / us - 1):+.1f} %)  {plan}")
    if ok:
        kept.append(ln)
    else:
        dropped += 1
    del on, off
    torch.cuda.empty_cache()
with open(a.out, "w") as f:
    for ln in kept:
        if ln.startswith("// Plans the in-library tuner"):
            ln = ("// Plans the in-library tuner (tune.hpp) measured to beat the static rules of plan.hpp by more than 3 % AND that "
                  "tools/verify_builtin_wisdom.py measured >= 4 % faster again (8 % from 2^25 points in flight on) as a HIP graph on a cold ring, in the text format of")
        f.write(ln + "\n")
say(f"# kept {sum(1 for ln in kept if LINE.match(ln))} lines, dropped {dropped}, {time.time() - t0:.0f} s -> {a.out}")
