#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X PhastFT path.

    python bench.py [--steps K] [--warmup W]                                (N = 1)
    python bench.py --gpus N ...            (N > 1: launches its own N ranks under torch.distributed.run on 127.0.0.1;
                                             refuses -- rc 2, no JSON line -- what it cannot measure: fewer GPUs than N, a
                                             launcher whose WORLD_SIZE differs from --gpus)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (the driver's form)
    python bench.py --dist-fft L [--gpus N] (ONE transform of 2^L points over the ranks, every stage timed: SURVEY 8 f-3)

Metric (BASELINE.json): GSamples/s of the f64 forward planar FFT at N=2^20 and 2^26 (and % of the HBM roofline),
complex samples transformed per second over the whole job, inputs resident in HBM when the timed region starts.

  * N = 1  -> `value` = configs[1]: "Single f64 forward FFT, N=2^20, 1xMI355X".  One step = one in-place
    `fft_64_dit_with_planner` (the _dev entry point of the C ABI) on one 16 MiB transform.  Every step uses a fresh
    buffer of a pre-filled ring (values never overflow; the ring, > 512 MiB, defeats the 256 MiB Infinity Cache);
    the K steps are captured once into a HIP graph and replayed inside the timed region so that the host launch
    path (Python + ctypes) is not what is measured.
    After the timed region four buffers of that ring are compared, every bin, with the CPU oracle's fft_64_dit of the
    same seeded inputs (`config.result_check`; the oracle is the checker, never the thing measured).
    `config.results` (round 5: the driver's record keeps the top-level keys and `config`) holds the WHOLE metric in
    compact form -- {value, ms_per_step, frac, frac_transform, cpu_value, traffic_over_algorithmic} for N = 2^20, N = 2^26,
    the round trip, R2C / C2R f32 2^24, f32 C2C 2^20 / 2^26 and the 1024-transform shard.
    The same line carries, under "configs", the other single-GPU BASELINE configurations measured in the same run,
    each with its own value / ms_per_step / roofline / cpu_baseline:
      n2p26_forward   (the second half of BASELINE's metric: N=2^26, 1 GiB per transform, three 2 GiB-traffic passes)
      n2p26_roundtrip (configs[2]: forward + inverse on the same buffers, error against the input checked)
      r2c_f32_2p24    (configs[3]: r2c_fft_f32, N=2^24)
      c2r_f32_2p24    (its inverse, c2r_fft_f32 -- SURVEY.md 8f-1)
    and "weak_scaling_reference": one rank's shard of the N > 1 workload on this one GPU; "host_slice_api": the same
    transform called with HOST slices as the reference's callers hold them (H2D + kernels + D2H, PCIe-bound: the drop-in
    cost, never `value`).
  * N > 1  -> configs[4]: 8192 independent N=2^20 transforms per 8 GPUs = 1024 per GPU, fixed per-GPU work
    ("scaling": "weak"); one step = every rank transforms its 1024-transform shard in place.  The path has no
    exchange step, so there is no data-path collective; RCCL (torch.distributed "nccl") only carries the barrier,
    the max-over-ranks time and the trivial digest gather.  After the timed region every rank checks its results:
    Parseval on every transform of the shard and >= 8 sampled transforms against digests computed from the CPU
    oracle's output (the oracle is the checker here, never the thing measured).

"roofline": HIP-event duration of the dominant pass kernel vs the 8 TB/s HBM peak = `frac` =
`frac_dominant_pass`; `frac_transform` = the transform's compulsory bytes / the sum of its kernels' durations / peak (SURVEY.md
8d's figure, <= 1/passes); `traffic` = HBM bytes per launch from the committed PMC profile of the plan that ran; and -- N = 1 --
`stream_probe` = {read, write, copy} GB/s of the library's own hand-written streaming kernels on 1 GiB in the same run
(csrc/probe.hip) with `frac_of_copy` / `pass_frac_of_copy` for every config: the ceiling of THIS box on the line;
"cpu_baseline": the -O3 build of the oracle (a C restatement of the reference's CPU algorithm, bit-identical to the
checker build) timed on this host on a bounded sample; rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
N = 1 << LOG_N
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_SAMPLE = 32          # planar C2C f64 = 4 * sizeof(f64) per complex sample
SHARD = 1024                   # transforms per GPU in the multi-GPU workload (configs[4])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly from Python")
    ap.add_argument("--no-scaling-reference", action="store_true",
                    help="skip the one-GPU run of the sharded workload (profiling runs of the headline kernels)")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip N=2^26, the round trip and R2C")
    ap.add_argument("--shard", type=int, default=SHARD, help="transforms per GPU when --gpus > 1")
    ap.add_argument("--plan", default=None, help="experiment: force a plan, e.g. 6,8,6@12p8 (default: the library's own)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--same-gpu", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--dist-fft", type=int, default=0, metavar="L",
                    help="instead of the batch: ONE f64 transform of 2^L points spread over the --gpus ranks (SURVEY.md 8 f-3, "
                         "phastft_amd/distributed.py), every stage timed with HIP events; --gpus 1 runs it on a one-rank process "
                         "group (the exchanges are self-copies)")
    ap.add_argument("--sharded", action="store_true",
                    help="run the N > 1 code path (process group, sharded batch, digest gather) even with WORLD_SIZE=1: on "
                         "a one-GPU box this is what puts RCCL init and the device collectives on real hardware")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline legs (the oracle's -O3 build on this host; examples/benchmark.rs:19-63 protocol: planner outside the
# timer, input regenerated before every timed call)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s: float = 8.0):
    from oracle import oracle as O

    t1 = O.time_fft_64_dit(N, 3)  # warm + calibrate
    per = max(t1 / 3, 1e-4)
    iters = max(10, min(2000, int(budget_s / per)))
    total = O.time_fft_64_dit(N, iters)
    # the crate's optional `parallel` feature emulated: a short second leg, reported beside
    # the default-features number above, never instead of it
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best = None
    for threads in [c for c in (2, 4, 8, 16, 32, 64) if c <= avail] or [1]:
        O.time_fft_64_dit_parallel(N, 2, threads=threads)
        p_iters = max(5, min(200, int(0.04 * budget_s / per)))
        p_total = O.time_fft_64_dit_parallel(N, p_iters, threads=threads)
        if best is None or p_total / p_iters < best[1] / best[2]:
            best = (threads, p_total, p_iters)
    threads, p_total, p_iters = best
    return {
        "value": iters * N / total / 1e9, "unit": "GSamples/s", "cores": 1, "kind": "port",
        "sample": f"{iters} forward fft_64_dit_with_planner calls at N=2^{LOG_N} "
                  f"({total:.1f} s of CPU work, {1e3 * total / iters:.2f} ms each), oracle/ C restatement built "
                  f"{O.timing_build()} (vectorised butterfly loops, bit-identical to the checker build), 1 thread",
        "parallel_feature": {"value": p_iters * N / p_total / 1e9, "unit": "GSamples/s", "cores": threads,
                             "sample": f"{p_iters} calls, {1e3 * p_total / p_iters:.2f} ms each, best of 2..64 threads "
                                       f"({avail} schedulable); rayon::join emulated with OpenMP tasks (2-way bit "
                                       f"reversal, recursive join while size > 16384, spanning stages serial)"},
    }


def host_slice_api(P, calls: int = 5):
    """The drop-in cost: fft_64_dit_with_planner on HOST slices, as the reference's callers hold them --
    H2D of 16 MiB, the transform, D2H, blocking.  PCIe-bound; reported beside `value`, never as it."""
    import time

    import numpy as np

    rng = np.random.default_rng(0xCAFE)
    re, im = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
    pl = P.PlannerDit64(N)
    for _ in range(2):
        P.fft_64_dit_with_planner(re, im, P.Direction.Forward, pl)
    t0 = time.perf_counter()
    for _ in range(calls):
        P.fft_64_dit_with_planner(re, im, P.Direction.Forward, pl)
    dt = (time.perf_counter() - t0) / calls
    return {"ms_per_call": 1e3 * dt, "value": N / dt / 1e9, "unit": "GSamples/s", "calls": calls,
            "bytes_over_pcie_per_s_GB": 32.0 * N / dt / 1e9,  # 16 MiB in + 16 MiB out, one after the other
            "what": f"fft_64_dit_with_planner(host slices) N=2^{LOG_N}: H2D + 3 kernels + D2H, blocking (pageable numpy arrays)"}


def cpu_leg(kind: str, n: int, iters: int):
    """One short CPU leg beside a "configs" entry: `kind` in forward / roundtrip / r2c_f32."""
    from oracle import oracle as O

    fn = {"forward": O.time_fft_64_dit, "roundtrip": O.time_fft_64_roundtrip, "r2c_f32": O.time_r2c_fft_f32,
          "forward_f32": O.time_fft_32_dit}[kind]
    total = fn(n, iters)
    samples = (2 if kind == "roundtrip" else 1) * n * iters
    return {"value": samples / total / 1e9, "unit": "GSamples/s", "cores": 1, "kind": "port",
            "sample": f"{iters} x {kind} at N=2^{n.bit_length() - 1} ({total:.1f} s of CPU work), oracle/ C restatement "
                      f"built {O.timing_build()}, 1 thread, planner outside the timer"}


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def plan_used(planner, batch: int = 1, kind: int = 0):
    """(which, pass list) of the plan a call with `batch` transforms runs -- the library's own answer
    (phast_planner_*_describe_call -> Planner::choose), e.g. ("single", "[64x16A w16][256x16 q16][64x16 w16]");
    kind: 0 planar C2C, 1 Complex<T> pairs, 2 R2C, 3 C2R (PHAST_TUNE_*)"""
    text = planner.describe_call(batch, kind)
    which, _, rest = text.partition(" ")
    return which, rest


def kernel_tags(plan_list: str, dtype: str = "double"):
    """Substrings of the pass kernels' names of a plan, in launch order -- used to check that a committed PMC profile
    belongs to the plan that ran: 'tile_fft_kernel<double, LR, LC, LP, PRE_TW, TRANSPOSE,' for LDS tiles,
    'wave_fft_kernel<double, PRE_TW, TRANSPOSE>' for wave tiles."""
    import math
    import re

    tags = []
    for i, (rows, cols, kind, pts) in enumerate(re.findall(r"\[(\d+)x(\d+)A? ([pwq])(\d+)", plan_list)):
        flags = "false, true" if i == 0 else "true, false"
        if kind == "w":
            tags.append(f"wave_fft_kernel<{dtype}, {flags}>")
        elif kind == "q":
            tags.append(f"quad_fft_kernel<{dtype}")   # (+ ", true>": one tile per workgroup, ", false>": the persistent form)
        else:
            tags.append(f"tile_fft_kernel<{dtype}, {int(math.log2(int(rows)))}, {int(math.log2(int(cols)))}, "
                        f"{int(math.log2(int(pts)))}, {flags},")
    return tags


def roofline_of(pass_ms, alg_bytes, names=None, plan_used="", traffic_key=None, tags=None, step_ms=None):
    dom = max(range(len(pass_ms)), key=lambda i: pass_ms[i])
    achieved = alg_bytes / (pass_ms[dom] * 1e-3) / 1e9
    total_ms = sum(pass_ms)
    label = names[dom] if names else f"tile_fft pass {dom} of {len(pass_ms)}"
    return {
        # `frac` (the contract's key) = `frac_dominant_pass`: the dominant KERNEL's algorithmic bytes per launch / its duration /
        # peak.  `frac_transform` is SURVEY.md 8(d)'s number: the TRANSFORM's compulsory bytes (every element read once, written
        # once) / the measured time of one step (`ms_per_step`, the K-step region) / peak -- capped at 1/passes by construction.
        # `frac_transform_pass_sum` divides by the sum of the separately timed passes instead (what rounds 1-5 printed as
        # frac_transform: it leaves out the gaps between the kernels and read 3-6 % high).
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "frac_dominant_pass": achieved / HBM_PEAK_GBS,
        "frac_transform": alg_bytes / ((step_ms if step_ms else total_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "frac_transform_pass_sum": alg_bytes / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "passes": len(pass_ms),
        "traffic": None, "kernel": label + (f" ({plan_used})" if plan_used else ""),
        "kernel_ms": pass_ms[dom], "pass_ms": pass_ms, "algorithmic_bytes_per_launch": alg_bytes,
    }, dom


_DRAIN = {}


def settle(torch, P, refill=None):
    """ONE defined state in front of every timed region: inputs as generated, resident in HBM -- and nothing of them, nor of
    the kernel that wrote them, left in the caches.  Whatever ran last (the fill, the untimed replay) leaves up to 256 MiB of
    DIRTY lines in the Infinity Cache; their write-back then competes with the timed steps, and how much of it lands inside
    the region depends on what ran before (tools/replay_variance.py, profiles/r06_replay_variance.log: the same 20-step graph of
    f32 2^20 transforms reads 18.5 / 20.8 / 23.7 / 22.4 us per step in a period of four behind `refill; synchronize`, and 17.9-18.1
    every time behind refill + drain).  The drain READS 768 MiB of a buffer nothing else uses (the digest kernel: three times
    the cache), so the dirty lines are written back before the region starts and the cache holds clean lines of no use to it."""
    if refill is not None:
        refill()
    if os.environ.get("PHAST_BENCH_NO_DRAIN"):   # tools only: the protocol without the drain, for the A/B in profiles/r06_replay_variance.log
        torch.cuda.synchronize()
        return
    dev = torch.cuda.current_device()
    if dev not in _DRAIN:
        re = torch.zeros(48 << 20, dtype=torch.float64, device="cuda")
        _DRAIN[dev] = (re, torch.zeros_like(re))
    P.digest(_DRAIN[dev][0], _DRAIN[dev][1], 1 << 20)
    torch.cuda.synchronize()


def replay_stats(torch, run, steps: int, first_ms: float, before=None, extra: int = 2):
    """"median and min".  `first_ms` is the contract's timed region (EXACTLY K steps, once); `extra` more
    K-step regions are timed after it -> per-step min / median over the 1 + extra regions.  `before()` puts every region into
    the state the first one started from (`settle`: inputs re-generated where the steps work in place, caches drained)."""
    per = [first_ms / steps]
    for _ in range(extra):
        if before is not None:
            before()
        torch.cuda.synchronize()
        per.append(event_ms(torch, run) / steps)
    per.sort()
    return {"ms_per_step_min": per[0], "ms_per_step_median": per[len(per) // 2], "timed_regions": len(per)}


def static_rule_ms(P, torch, used: str, make_planner, make_step, steps: int):
    """where the library ran a `tuned` plan (built-in wisdom, measured on another box), the static rule's
    plan timed beside it in this run -- the same K steps from a HIP graph on the same ring.  None when the static rule ran."""
    if used != "tuned":
        return None
    was = P.wisdom_builtin(False)
    try:
        q = make_planner()
    finally:
        P.wisdom_builtin(was)
    step = make_step(q)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    graph, _ = capture_steps(torch, P, step, 3, steps, touch=lambda: step(0))
    run = graph.replay if graph is not None else (lambda: [step(3 + i) for i in range(steps)])
    per = []
    for _ in range(3):
        settle(torch, P)
        per.append(event_ms(torch, run) / steps)
    per.sort()
    del graph, q
    return per[1]


def attach_traffic(roof, key, tag, scale=1.0):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile `key` (profiles/traffic_latest.json), if the
    profile holds exactly one kernel whose name contains `tag` -- i.e. was taken with the plan that ran"""
    tr = traffic_for(key, [tag], scale=scale) if tag else None
    if tr:
        roof.update(tr)
        roof["traffic_over_algorithmic"] = tr["traffic"] / roof["algorithmic_bytes_per_launch"]
    return tr is not None


def add_copy_frac(roof, sp):
    """quote a config's dominant kernel -- and every pass -- against the hand-written copy kernel's rate (stream_probe)"""
    if not roof or not sp or "pass_ms" not in roof or "achieved" not in roof:
        return
    roof["frac_of_copy"] = roof["achieved"] / sp["copy"]
    per = roof.get("algorithmic_bytes_per_launch")
    if per:
        roof["pass_frac_of_copy"] = [per / (t * 1e-3) / 1e9 / sp["copy"] for t in roof["pass_ms"]]


def hbm_copy_probe(torch, dev, mib: int = 1024, reps: int = 10):
    """Device-to-device copy of ``mib`` MiB (read + write counted): what this box's HBM gives a plain streaming
    kernel, reported beside the 8 TB/s spec peak (SURVEY.md 8d "bounding roofline")."""
    src = torch.empty(mib << 17, dtype=torch.float64, device=dev).fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * src.numel() * 8 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def event_ms(torch, fn):
    """HIP events on torch's current stream (the stream the library launches on) around fn()"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def capture_steps(torch, P, step, first: int, steps: int, touch=None):
    """The K timed steps step(first) .. step(first + K - 1) captured into one HIP graph, instantiated and UPLOADED
    (hipGraphUpload), so that the replay inside the timed region is the launch of a resident executable graph: the
    Python + ctypes launch path is not the product, and neither is the one-time upload of the graph.  Returns
    (graph or None, description)."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        if touch is not None:
            with torch.cuda.stream(side):
                touch()  # first use of the planner on the side stream happens outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(steps):
                step(first + i)
        uploaded = False
        try:
            uploaded = P.graph_upload(g)
        except Exception as e:  # the upload is an optimisation of the first launch only
            print(f"note: hipGraphUpload unavailable ({e})", file=sys.stderr)
        # ... and replayed once, untimed: the first launch of an executable graph still pays one-time work that the
        # upload does not cover (tools/graph_protocol.py, profiles/r03_graph_protocol.log: K = 20 steps 27.8 us per step
        # cold, 27.0 uploaded, 25.9 after one replay; K = 200: 24.3 / 24.3 / 23.2).  The ring is larger than the Infinity
        # Cache, so the timed replay finds every buffer as cold as the first one did; values grow by <= 2^10 per
        # transform, far from overflow.
        g.replay()
        torch.cuda.synchronize()
        return g, ("hipGraph replay of the K steps (graph " + ("uploaded and " if uploaded else "") +
                   "replayed once, untimed, before the timed region)")
    except Exception as e:  # capture is an optimisation of the launch path only
        print(f"note: HIP graph capture unavailable ({e}); launching eagerly", file=sys.stderr)
        torch.cuda.synchronize()
        return None, "eager launches from Python"


def config_f32(P, torch, dev, log_n: int, steps: int, cpu: bool):
    """f32 forward C2C (`fft_32_dit`), one transform per step on a pre-filled ring (> 512 MiB, every
    buffer HBM-cold), the K steps replayed from one HIP graph and timed with HIP events on the launch stream.
    Algorithmic bytes: 16 B per complex sample per pass (read re + im once, write once)."""
    n = 1 << log_n
    pl = P.PlannerDit32(n)
    plan_text = pl.describe()
    ring = max(steps + 3, (640 << 20) // (8 * n) + 1) if log_n < 26 else steps + 1
    re = torch.empty(ring * n, dtype=torch.float32, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0)
    views = [(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n]) for i in range(ring)]

    def step(i):
        r, m = views[i % ring]
        P.fft_32_dit_with_planner(r, m, P.Direction.Forward, pl)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    graph, launch = capture_steps(torch, P, step, 3, steps, touch=lambda: step(0))
    fresh = lambda: settle(torch, P, refill=lambda: P.fill_uniform(re, im, n, seed=0xCAFE, first_id=0))
    fresh()
    run = graph.replay if graph is not None else (lambda: [step(3 + i) for i in range(steps)])
    ms = event_ms(torch, run) / steps
    stats = replay_stats(torch, run, steps, ms * steps, before=fresh)
    fresh()   # (the per-pass event times below start from the same state as the timed regions)
    acc, reps = None, min(ring, 16)
    for i in range(reps):
        t = pl.time_passes(views[i][0], views[i][1], n, reps=1)
        acc = t if acc is None else [a + b for a, b in zip(acc, t)]
    pass_ms = [a / reps for a in acc]
    used, plan_list = plan_used(pl, 1)
    roof, dom = roofline_of(pass_ms, 16 * n, plan_used=f"{used} plan {plan_list}", step_ms=ms)
    tags = kernel_tags(plan_list, "float")
    attach_traffic(roof, f"f32_2p{log_n}", tags[dom] if dom < len(tags) else None)
    out = {"workload": f"single f32 forward FFT N=2^{log_n}, in place, planar (fft_32_dit_with_planner)",
           "value": n / (ms * 1e-3) / 1e9, "unit": "GSamples/s", "steps": steps, "ms_per_step": ms, **stats, "dtype": "f32",
           "plan": plan_text, "plan_ran": f"{used} {plan_list}", "launch": launch, "roofline": roof}
    static_ms = static_rule_ms(P, torch, used, lambda: P.PlannerDit32(n), lambda q: (lambda i: P.fft_32_dit_with_planner(
        views[i % ring][0], views[i % ring][1], P.Direction.Forward, q)), steps)
    if static_ms is not None:
        out["static_ms"] = static_ms
    del re, im, views, pl, graph
    torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = cpu_leg("forward_f32", n, 200 if log_n <= 20 else 2)
    return out


# ------------------------------------------------------------------------------------------------
# the other single-GPU BASELINE configs, on the driver-run line
# ------------------------------------------------------------------------------------------------
def config_n2p26(P, torch, dev, steps: int, cpu: bool):
    """N = 2^26 f64: forward (the second half of BASELINE's metric) and the forward+inverse round trip (configs[2]), each
    timed step on its own pre-filled buffer of a ring (the headline's protocol)."""
    n = 1 << 26
    pl = P.PlannerDit64(n)
    plan_text = pl.describe()
    # the same protocol as the headline: a pre-filled ring of distinct transforms (1 GiB each, so every one is HBM-cold
    # by construction), one timed in-place transform per buffer -- no fill kernel's write-back tail inside a timed step
    ring = steps
    ring_re = torch.empty(ring * n, dtype=torch.float64, device=dev)
    ring_im = torch.empty_like(ring_re)
    re, im = ring_re[:n], ring_im[:n]
    P.fill_uniform(re, im, n, seed=0xCAFE)
    for _ in range(3):  # untimed warm-up: scratch allocation, clocks (the CPU legs above left the GPU idle for seconds)
        P.fft_64_dit_with_planner(re, im, P.Direction.Forward, pl)
    settle(torch, P, refill=lambda: P.fill_uniform(ring_re, ring_im, n, seed=0xCAFE, first_id=0))   # (256 MiB of the fill's
    # lines would otherwise be written back under the first transform's passes: ~4 % of its traffic)

    def forward_all():  # the K timed steps back to back, one event pair around them (as the headline's K-step region)
        for i in range(steps):
            P.fft_64_dit_with_planner(ring_re[i * n:(i + 1) * n], ring_im[i * n:(i + 1) * n], P.Direction.Forward, pl)

    ms = event_ms(torch, forward_all) / steps
    stats = replay_stats(torch, forward_all, steps, ms * steps,
                         before=lambda: settle(torch, P, refill=lambda: P.fill_uniform(ring_re, ring_im, n, seed=0xCAFE, first_id=0)))
    settle(torch, P, refill=lambda: P.fill_uniform(re, im, n, seed=0xCAFE))
    pass_ms = pl.time_passes(re, im, n, reps=3)
    used, plan_list = plan_used(pl, 1)
    roof, dom = roofline_of(pass_ms, BYTES_PER_SAMPLE * n, plan_used=f"{used} plan {plan_list}", step_ms=ms)
    tags = kernel_tags(plan_list)
    attach_traffic(roof, "single_2p26", tags[dom] if dom < len(tags) else None)
    fwd = {"workload": "single f64 forward FFT N=2^26, in place, planar (BASELINE metric, second size)",
           "value": n / (ms * 1e-3) / 1e9, "unit": "GSamples/s", "steps": steps, "ms_per_step": ms, **stats, "dtype": "f64",
           "plan": plan_text, "plan_ran": f"{used} {plan_list}", "roofline": roof}
    # configs[2]: forward then inverse on the same buffers; the error against the regenerated input is part of it
    settle(torch, P, refill=lambda: P.fill_uniform(ring_re, ring_im, n, seed=0xBEEF, first_id=0))

    def roundtrip_all():
        for i in range(steps):
            r_i, m_i = ring_re[i * n:(i + 1) * n], ring_im[i * n:(i + 1) * n]
            P.fft_64_dit_with_planner(r_i, m_i, P.Direction.Forward, pl)
            P.fft_64_dit_with_planner(r_i, m_i, P.Direction.Reverse, pl)

    rt_total = event_ms(torch, roundtrip_all)
    ref_re, ref_im = torch.empty_like(re), torch.empty_like(im)
    P.fill_uniform(ref_re, ref_im, n, seed=0xBEEF, first_id=steps - 1)
    last_re, last_im = ring_re[(steps - 1) * n:steps * n], ring_im[(steps - 1) * n:steps * n]
    err = max(float((last_re - ref_re).abs().max()), float((last_im - ref_im).abs().max()))
    del ref_re, ref_im
    rt_ms = rt_total / steps
    # one launch of a pass moves 32 B/sample; the round trip is 2 x passes launches
    rt = {"workload": "single f64 forward+inverse round trip N=2^26 on the same buffers (BASELINE configs[2])",
          "value": 2 * n / (rt_ms * 1e-3) / 1e9, "unit": "GSamples/s (both directions counted)", "steps": steps,
          "ms_per_step": rt_ms, "dtype": "f64", "max_abs_err_vs_input": err, "err_ok": bool(err < 1e-10),
          # the inverse is the swap trick + 1/N in the last store (algorithms/dit.rs:297-300): the SAME pass kernels on swapped
          # plane pointers -- a step is 2 x passes launches of them, so the per-pass figures are the forward transform's
          "roofline": dict(roof, passes=2 * len(pass_ms), pass_ms=pass_ms + pass_ms,
                           algorithmic_bytes_per_step=2 * BYTES_PER_SAMPLE * n,
                           frac_transform=2 * BYTES_PER_SAMPLE * n / (rt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           note="forward then inverse: the same pass kernels (swap trick + 1/N in the last store); per-pass "
                                "durations are the forward transform's, frac_transform is this config's own step time")}
    del re, im, ring_re, ring_im, last_re, last_im, pl
    torch.cuda.empty_cache()
    if cpu:
        fwd["cpu_baseline"] = cpu_leg("forward", n, 2)
        rt["cpu_baseline"] = cpu_leg("roundtrip", n, 1)
    return fwd, rt


def config_r2c(P, torch, dev, steps: int, cpu: bool):
    """configs[3]: r2c_fft_f32 at N = 2^24 (algorithmic bytes 4N in + 8(N/2+1) out)."""
    n = 1 << 24
    pl = P.PlannerR2c32(n)
    # a ring of distinct input/output sets (64 MiB in + 64 MiB out each; 9 sets = 1.1 GiB) so that no step finds its
    # input or leaves its output in the 256 MiB Infinity Cache -- the headline's protocol (one set would keep the whole
    # working set, 192 MiB with the inner scratch, cache-resident and the HBM roofline fraction would overstate)
    ring = 9
    xs = torch.empty(ring * n, dtype=torch.float32, device=dev)
    P.fill_uniform(xs, None, n, seed=0xCAFE)
    half1 = n // 2 + 1
    pitch = (half1 + 63) // 64 * 64  # every set's outputs start on a 256-byte boundary, as separately allocated slices do
    ores = torch.empty(ring * pitch, dtype=torch.float32, device=dev)
    oims = torch.empty_like(ores)
    sets = [(xs[i * n:(i + 1) * n], ores[i * pitch:i * pitch + half1], oims[i * pitch:i * pitch + half1]) for i in range(ring)]
    x, ore, oim = sets[0]
    P.r2c_fft_f32_with_planner(x, ore, oim, pl)

    def step(i):
        P.r2c_fft_f32_with_planner(*sets[i % ring], pl)   # the input is read-only (r2c.rs:535): no refill needed

    for i in range(ring):
        step(i)
    torch.cuda.synchronize()
    # the headline's protocol: the K steps captured into one HIP graph and replayed inside the timed region (HIP events on
    # the launch stream) -- the Python + ctypes launch path is not the product
    graph, launch = capture_steps(torch, P, step, 0, steps, touch=lambda: step(0))
    run = graph.replay if graph is not None else (lambda: [step(i) for i in range(steps)])
    settle(torch, P)   # (inputs are read-only: nothing to re-generate; the outputs of the untimed replay leave the caches)
    ms = event_ms(torch, run) / steps
    stats = replay_stats(torch, run, steps, ms * steps, before=lambda: settle(torch, P))
    settle(torch, P)
    acc = None
    for i in range(ring):
        t = pl.time_passes(*sets[i], reps=1)
        acc = t if acc is None else [a + b for a, b in zip(acc, t)]
    pass_ms = [a / ring for a in acc]
    r2c_bytes = 4 * n + 8 * (n // 2 + 1)
    plan_text = pl.describe()
    used, r2c_list = plan_used(pl, 1, 2)
    r2c_ran = f"{used} {r2c_list}"
    r2c_list = r2c_list.replace(" untangle-fused", "")
    n_inner = len(kernel_tags(r2c_list, "float"))
    fused = len(pass_ms) == n_inner  # round 3: the last pass takes the untangle with it (r2c_fused.hpp): no sweep of its own
    names = [f"tile_fft pass {i} of the inner 2^23-point transform" for i in range(n_inner)]
    if fused:
        names[-1] += " with the untangle fused in (r2c_last_pass_kernel)"
    else:
        names.append("untangle sweep")
    # per-kernel algorithmic bytes differ (first pass reads the real input, the untangle re-reads and re-writes the
    # half spectrum): the kernel fraction is quoted against the bytes THAT kernel must move
    k_bytes = [8 * (n // 2) + 8 * (n // 2)] * n_inner + ([] if fused else [16 * (n // 2 + 1)])
    fr = [b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS for b, t in zip(k_bytes, pass_ms)]
    dom = max(range(len(pass_ms)), key=lambda i: pass_ms[i])
    roof = {"bound": "hbm", "achieved": k_bytes[dom] / (pass_ms[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": fr[dom], "traffic": None, "kernel": names[dom], "kernel_ms": pass_ms[dom], "pass_ms": pass_ms,
            "algorithmic_bytes_per_launch": k_bytes[dom], "algorithmic_bytes_per_transform": r2c_bytes,
            "frac_dominant_pass": fr[dom], "frac_transform": r2c_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_transform_pass_sum": r2c_bytes / (sum(pass_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, "passes": len(pass_ms)}
    out = {"workload": "r2c_fft_f32 N=2^24, real input -> N/2+1 planar outputs (BASELINE configs[3])",
           "value": n / (ms * 1e-3) / 1e9, "unit": "GSamples/s (real samples)", "steps": steps, "ms_per_step": ms, **stats,
           "dtype": "f32", "plan": plan_text, "plan_ran": r2c_ran, "launch": launch, "roofline": roof}
    static_ms = static_rule_ms(P, torch, used, lambda: P.PlannerR2c32(n),
                               lambda q: (lambda i: P.r2c_fft_f32_with_planner(*sets[i % ring], q)), steps)
    if static_ms is not None:
        out["static_ms"] = static_ms
    tags = kernel_tags(r2c_list, "float")
    if fused:
        tags[-1] = tags[-1].replace("tile_fft_kernel", "r2c_last_pass_kernel").split(", true, false")[0]
    attach_traffic(roof, "r2c_f32_2p24", "untangle_kernel" if (not fused and dom == len(pass_ms) - 1) else (tags[dom] if dom < len(tags) else None))
    del x, ore, oim, sets, xs, ores, oims, pl, graph
    torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = cpu_leg("r2c_f32", n, 8)
    return out


def config_c2r(P, torch, dev, steps: int, cpu: bool):
    """The inverse of configs[3] (the first NEXT row): c2r_fft_f32 at N = 2^24 on a cold ring, per-kernel
    times from the library's event hook.  Algorithmic bytes 8(N/2+1) in + 4N out."""
    n = 1 << 24
    pl = P.PlannerR2c32(n)
    ring = 9
    half1 = n // 2 + 1
    pitch = (half1 + 63) // 64 * 64
    ires = torch.empty(ring * pitch, dtype=torch.float32, device=dev).uniform_(-1, 1)
    iims = torch.empty(ring * pitch, dtype=torch.float32, device=dev).uniform_(-1, 1)
    ys = torch.empty(ring * n, dtype=torch.float32, device=dev)
    sets = [(ires[i * pitch:i * pitch + half1], iims[i * pitch:i * pitch + half1], ys[i * n:(i + 1) * n]) for i in range(ring)]
    P.c2r_fft_f32_with_planner(*sets[0], pl)

    def step(i):
        P.c2r_fft_f32_with_planner(*sets[i % ring], pl)  # the half-spectrum is read-only (r2c.rs:740): no refill needed

    for i in range(ring):
        step(i)
    torch.cuda.synchronize()
    graph, launch = capture_steps(torch, P, step, 0, steps, touch=lambda: step(0))   # the headline's protocol (config_r2c)
    run = graph.replay if graph is not None else (lambda: [step(i) for i in range(steps)])
    settle(torch, P)   # (inputs are read-only: nothing to re-generate; the outputs of the untimed replay leave the caches)
    ms = event_ms(torch, run) / steps
    stats = replay_stats(torch, run, steps, ms * steps, before=lambda: settle(torch, P))
    settle(torch, P)
    acc = None
    for i in range(ring):
        t = pl.time_c2r_passes(*sets[i], reps=1)
        acc = t if acc is None else [a + b for a, b in zip(acc, t)]
    pass_ms = [a / ring for a in acc]
    c2r_bytes = 4 * n + 8 * half1
    used, c2r_list = plan_used(pl, 1, 3)
    c2r_ran = f"{used} {c2r_list}"
    fused = " preprocess-fused" in c2r_list
    c2r_list = c2r_list.replace(" preprocess-fused", "")
    tags = kernel_tags(c2r_list, "float")
    n_inner = len(tags)
    names = [f"tile_fft pass {i} of the inner 2^23-point transform" for i in range(n_inner)]
    if fused:
        names[0] += " forming z from the half-spectrum on load (c2r_first_pass_kernel)"
        tags[0] = tags[0].replace("tile_fft_kernel", "c2r_first_pass_kernel").split(", false, true")[0]
    else:
        names.append("preprocess sweep")
    # every inner pass reads and writes 2^23 complex f32 points once; the fused first pass reads the half-spectrum (each
    # element twice: as itself and as its mirror's partner -- the second read is an L2 hit by the tile order) and writes them
    k_bytes = [8 * (n // 2) + 8 * (n // 2)] * n_inner + ([] if fused else [16 * half1])
    fr = [b_ / (t * 1e-3) / 1e9 / HBM_PEAK_GBS for b_, t in zip(k_bytes, pass_ms)]
    dom = max(range(len(pass_ms)), key=lambda i: pass_ms[i])
    roof = {"bound": "hbm", "achieved": k_bytes[dom] / (pass_ms[dom] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": fr[dom], "frac_dominant_pass": fr[dom],
            "frac_transform": c2r_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_transform_pass_sum": c2r_bytes / (sum(pass_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, "passes": len(pass_ms), "traffic": None,
            "kernel": names[dom], "kernel_ms": pass_ms[dom], "pass_ms": pass_ms, "algorithmic_bytes_per_launch": k_bytes[dom],
            "algorithmic_bytes_per_transform": c2r_bytes}
    attach_traffic(roof, "c2r_f32_2p24", tags[dom] if dom < len(tags) else "c2r_preprocess_kernel")
    out = {"workload": "c2r_fft_f32 N=2^24, N/2+1 planar inputs -> real output (inverse of BASELINE configs[3])",
           "value": n / (ms * 1e-3) / 1e9, "unit": "GSamples/s (real samples)", "steps": steps, "ms_per_step": ms, **stats, "dtype": "f32",
           "plan": pl.describe(), "plan_ran": c2r_ran, "launch": launch, "roofline": roof}
    static_ms = static_rule_ms(P, torch, used, lambda: P.PlannerR2c32(n),
                               lambda q: (lambda i: P.c2r_fft_f32_with_planner(*sets[i % ring], q)), steps)
    if static_ms is not None:
        out["static_ms"] = static_ms
    del sets, ires, iims, ys, pl, graph
    torch.cuda.empty_cache()
    if cpu:
        from oracle import oracle as O

        iters = 8
        total = O.time_c2r_fft_f32(n, iters)
        out["cpu_baseline"] = {"value": n * iters / total / 1e9, "unit": "GSamples/s", "cores": 1, "kind": "port",
                               "sample": f"{iters} x c2r_fft_f32_with_planner_and_scratch at N=2^24 ({total:.1f} s of CPU work), "
                                         f"oracle/ C restatement built {O.timing_build()}, 1 thread, planner outside the timer"}
    return out


def shard_on_one_gpu(P, torch, dev, shard: int, steps: int = 5):
    """The per-GPU workload of the --gpus N > 1 runs (BASELINE configs[4]: `shard` transforms of 2^20 per GPU,
    transformed in place per step) timed on this one GPU with HIP events: the denominator for weak-scaling
    efficiency.  The N = 1 headline is a different workload (ONE transform, configs[1])."""
    pl = P.PlannerDit64(N)
    re = torch.empty(shard * N, dtype=torch.float64, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
    P.fft_dit_batched(re, im, N, P.Direction.Forward, pl)  # warm-up (scratch allocation)
    P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
    torch.cuda.synchronize()

    def all_steps():  # in place step after step, as the --gpus N run does (values grow 2^10-fold per step: 5 steps are safe)
        for _ in range(steps):
            P.fft_dit_batched(re, im, N, P.Direction.Forward, pl)

    ms = event_ms(torch, all_steps) / steps
    P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
    pass_ms = pl.time_passes(re, im, N, reps=2)
    used, plan_list = plan_used(pl, shard)
    roof, dom = roofline_of(pass_ms, BYTES_PER_SAMPLE * N * shard, plan_used=f"{used} plan {plan_list}", step_ms=ms)
    tags = kernel_tags(plan_list)
    if attach_traffic(roof, "batch_2p20", tags[dom] if dom < len(tags) else None, scale=shard / 1024.0):  # profiled per 1024-transform launch
        roof["traffic_note"] = "PMC bytes of one 1024-transform launch of the same kernel (scaled to the shard if it differs)"
    return {"workload": f"{shard} independent f64 forward FFTs N=2^{LOG_N} on 1 GPU, in place (one rank's shard of "
                        f"BASELINE configs[4])", "value": shard * N / (ms * 1e-3) / 1e9, "unit": "GSamples/s",
            "steps": steps, "ms_per_step": ms, "plan_ran": f"{used} {plan_list}", "roofline": roof}


def shard_cpu_leg(shard: int, transforms: int = 8):
    """the CPU leg of configs[4] is `transforms` (8) transforms of 2^20 timed on this host and
    extrapolated x shard / transforms -- the transforms are independent and the reference runs them one after the other on one
    thread (default features), so the extrapolation is exact up to cache warmth."""
    from oracle import oracle as O

    O.time_fft_64_dit(N, 1)
    total = O.time_fft_64_dit(N, transforms)
    return {"value": transforms * N / total / 1e9, "unit": "GSamples/s", "cores": 1, "kind": "port",
            "sample": f"{transforms} forward transforms at N=2^{LOG_N} timed ({total:.2f} s of CPU work), x {shard // transforms} "
                      f"extrapolated to the {shard}-transform shard = {total * shard / transforms:.1f} s per step on one core; oracle/ C "
                      f"restatement built {O.timing_build()}, 1 thread, planner outside the timer",
            "shard_seconds_extrapolated": total * shard / transforms}


def check_shard(P, torch, re, im, refill, step, first: int, shard: int, samples: int = 8):
    """After the timed region: one step on fresh inputs, then (i) Parseval on EVERY transform of the shard from the
    32-byte digests (sum |X|^2 = N sum |x|^2), (ii) `samples` transforms against digests computed from the CPU oracle's
    output of the same seeded input (first, last and evenly spaced ids)."""
    import numpy as np

    from oracle import oracle as O

    refill()
    before = P.digest(re, im, N, probe=1).cpu().numpy()
    step()
    after_t = P.digest(re, im, N, probe=1)
    after = after_t.cpu().numpy()
    ok = bool(np.all(np.isfinite(after))) and after.shape[0] == shard
    parseval = float(np.max(np.abs(after[:, 2] / (N * before[:, 2]) - 1.0)))
    ok = ok and parseval < 1e-12
    ids = sorted({int(round(i * (shard - 1) / max(1, samples - 1))) for i in range(samples)})
    worst = 0.0
    for b in ids:
        r, m = O.fill(N, np.float64, seed=0xCAFE, transform_id=first + b)
        O.fft_64_dit(r, m, O.FORWARD)
        scale = float(np.sqrt(N * before[b, 2]))
        want = np.array([r.sum(), m.sum(), (r * r + m * m).sum(), r[1]])
        dev_ = max(abs(after[b, 0] - want[0]) / (scale * np.sqrt(N)), abs(after[b, 1] - want[1]) / (scale * np.sqrt(N)),
                   abs(after[b, 2] / want[2] - 1.0), abs(after[b, 3] - want[3]) / scale)
        worst = max(worst, float(dev_))
    ok = ok and worst < 1e-10
    return ok, {"parseval_max_rel_dev": parseval, "oracle_digest_max_dev": worst, "oracle_checked_ids": len(ids)}, after_t


def check_headline(P, torch, views, planner, ids, already: int):
    """Buffers `ids` of the headline's ring (transform id = ring index, seed 0xCAFE) against the CPU oracle's fft_64_dit of the
    same inputs: every bin.  Buffers below `already` hold the forward transform of their fill; the others are transformed here."""
    import numpy as np

    from oracle import oracle as O

    worst_rel, worst_bin = 0.0, 0.0
    for i in sorted(set(ids)):
        r, m = views[i]
        if i >= already:
            P.fft_64_dit_with_planner(r, m, P.Direction.Forward, planner)
        o_re, o_im = O.fill(N, np.float64, seed=0xCAFE, transform_id=i)
        O.fft_64_dit(o_re, o_im, O.FORWARD)
        g_re, g_im = r.cpu().numpy(), m.cpu().numpy()
        den = float(np.sqrt(np.sum(o_re ** 2 + o_im ** 2)))
        rel = float(np.sqrt(np.sum((g_re - o_re) ** 2 + (g_im - o_im) ** 2))) / den
        rms = den / np.sqrt(N)
        worst_rel = max(worst_rel, rel)
        worst_bin = max(worst_bin, float(max(np.max(np.abs(g_re - o_re)), np.max(np.abs(g_im - o_im)))) / rms)
    ok = worst_rel <= 8e-16 * LOG_N and worst_bin <= 64 * 2.220446049250313e-16 * LOG_N    # tests/tolerances.py: the f64 gates
    return {"ok": bool(ok), "rel_l2_max": worst_rel, "worst_bin_over_rms_max": worst_bin, "buffers_checked": len(set(ids)),
            "gates": {"rel_l2": 8e-16 * LOG_N, "worst_bin_over_rms": 64 * 2.220446049250313e-16 * LOG_N},
            "what": "ring buffers of the timed workload, every bin, against the CPU oracle's fft_64_dit of the same seeded input"}


def fail(msg: str, rc: int = 2):
    """bench.py never prints a number for a job other than the one it was asked to measure: refuse, loudly, rc != 0"""
    print(f"bench.py: error: {msg}", file=sys.stderr, flush=True)
    sys.exit(rc)


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n: int, argv) -> list:
    """the driver's own N > 1 command line (one rank per GPU on 127.0.0.1), with a free rendezvous port"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(args, torch) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this script under torch.distributed.run with N ranks on
    127.0.0.1 (one per GPU), stream rank 0's JSON line through, return the job's exit code.  Refuses (rc 2) when the
    node does not have N GPUs -- unless --same-gpu (dry run of the N-rank code path on one device over gloo)."""
    import subprocess

    have = torch.cuda.device_count()
    if args.same_gpu:
        if args.backend == "nccl":
            fail("--same-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        if have < 1:
            fail("no GPU visible")
    elif have < args.gpus:
        fail(f"--gpus {args.gpus} but this node has {have} GPU(s) visible; nothing measured "
             f"(use --same-gpu --backend gloo for a one-GPU dry run of the {args.gpus}-rank path)")
    cmd = launch_command(args.gpus, sys.argv[1:])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def dist_fft_mode(args, P, torch, dev, rank, world):
    """bench.py --dist-fft L [--gpus N]: ONE f64 transform of 2^L points over the ranks of the process group
    (phastft_amd/distributed.py: four-step split, three all_to_all_single exchanges per plane over RCCL).  Every stage is
    timed with HIP events on the rank's stream; the step time is the wall clock between barriers, MAX over ranks.  Prints
    one JSON line: per-stage milliseconds (rank 0's, averaged over the steps), the bytes every exchange moves per rank and
    how many of them leave the GPU."""
    import torch.distributed as dist

    from phastft_amd.distributed import gpu_transform
    from phastft_amd.sharding import max_over_ranks

    L = args.dist_fft
    n = 1 << L
    steps = args.steps if args.steps is not None else 5
    warmup = args.warmup if args.warmup is not None else 2
    slab = n // world
    re = torch.empty(slab, dtype=torch.float64, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, slab, seed=0xCAFE + rank)
    t = gpu_transform(n, rank, world, dist, "f64")
    for _ in range(warmup):
        t.run(re, im)
    P.fill_uniform(re, im, slab, seed=0xCAFE + rank)
    before = float((re.double() ** 2 + im.double() ** 2).sum())
    marks = []

    def on_stage(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    t.on_stage = on_stage
    stage_ms = {}
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        marks.clear()
        t.run(re, im, reverse=bool(i & 1))   # forward / inverse alternately: values stay in range
        torch.cuda.synchronize()
        prev = start
        for name, e in marks:
            stage_ms[name] = stage_ms.get(name, 0.0) + prev.elapsed_time(e) / steps
            prev = e
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev)
    if steps % 2 == 0:   # an even number of steps is `steps / 2` round trips: the input must be back
        after = float((re.double() ** 2 + im.double() ** 2).sum())
        ratio = after / before
        energy_ok = abs(ratio - 1.0) < 1e-9
    else:                # ... plus one forward transform: Parseval, sum |X|^2 = N sum |x|^2 (summed over the ranks)
        loc = torch.tensor([before, float((re.double() ** 2 + im.double() ** 2).sum())], dtype=torch.float64,
                           device="cpu" if args.backend == "gloo" else dev)
        dist.all_reduce(loc)
        ratio = float(loc[1]) / (n * float(loc[0]))
        energy_ok = abs(ratio - 1.0) < 1e-9
    if rank == 0:
        per_exchange = 2 * slab * 8                      # both planes of the rank's slab
        ms = 1e3 * elapsed / steps
        local = sum(v for k, v in stage_ms.items() if k.startswith(("fft", "twiddle")))
        exch = sum(v for k, v in stage_ms.items() if k.startswith("exchange"))
        perm = sum(v for k, v in stage_ms.items() if k.startswith(("pack", "unpack", "store")))
        out = {"metric": f"GSamples/s ONE f64 transform of 2^{L} points over {world} GPU(s) (four-step split, SURVEY.md 8 f-3)",
               "value": n / (ms * 1e-3) / 1e9, "unit": "GSamples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic (counter-based uniform [-1,1), generated on device)",
               "config": {"workload": f"one transform of 2^{L} f64 points, slab of 2^{L}/{world} per rank, natural order in and out",
                          "n1_x_n2": [t.n1, t.n2], "backend": args.backend, "energy_ok": bool(energy_ok),
                          "energy_ratio_minus_1": ratio - 1.0},
               "stages_ms": {k: round(v, 4) for k, v in stage_ms.items()},
               "summary_ms": {"local_ffts": round(local, 4), "exchanges": round(exch, 4), "pack_unpack_store": round(perm, 4)},
               "exchange": {"count_per_plane": 3, "bytes_per_exchange_per_rank": per_exchange,
                            "bytes_leaving_the_gpu_per_exchange": per_exchange * (world - 1) // world,
                            "note": ("one rank: every exchange is RCCL's self-copy -- no xGMI byte moves" if world == 1 else
                                     "RCCL all_to_all_single over xGMI: (P-1)/P of the slab leaves the GPU per exchange")}}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    # the host driver of this pool only supports dmabuf IPC: without this RCCL / device-tensor sharing across the ranks
    # fails with `hipIpcGetMemHandle: invalid argument` (already exported on the boxes; kept here for any other launcher)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    if args.gpus < 1:
        fail(f"--gpus {args.gpus}: need at least one GPU")
    launched = "WORLD_SIZE" in os.environ            # started by torch.distributed.run (the driver's N > 1 command)
    if args.gpus > 1 and not launched:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and passes the
        # one JSON line of rank 0 through; it never measures fewer GPUs than it was asked for
        sys.exit(self_launch(args, torch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        fail(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started {world} rank(s); refusing to report a "
             f"{world}-GPU number as a {args.gpus}-GPU one")
    if args.same_gpu and args.backend == "nccl" and world > 1:
        fail("--same-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    have = torch.cuda.device_count()
    need = 1 if args.same_gpu else int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if have < need:
        fail(f"{need} GPU(s) needed on this node, {have} visible"
             + ("" if need == 1 else " (use --same-gpu --backend gloo for a one-GPU dry run)"))
    multi = world > 1 or args.sharded or args.dist_fft > 0
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist

        if args.same_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        n_gpus = dist.get_world_size()                # the number of ranks the process group actually has
        if n_gpus != args.gpus:
            fail(f"process group has {n_gpus} rank(s), --gpus {args.gpus}")
    else:
        torch.cuda.set_device(0)
        n_gpus = 1

    import phastft_amd as P

    dev = torch.device("cuda", local_rank if multi else 0)
    if args.dist_fft:
        dist_fft_mode(args, P, torch, dev, rank, n_gpus)
        return
    planner = P.PlannerDit64(N)
    if args.plan:
        lrs_s, rest = args.plan.split("@")
        tl_s, p_s = rest.split("p")
        tls = tuple(int(x) for x in tl_s.split(","))
        wave = 0x10 if p_s.endswith("w") else 0   # "...p8w": 64 x 16 tiles as wave tiles (wave_fft.hpp)
        planner.set_plan(tuple(int(x) for x in lrs_s.split(",")), tls if len(tls) > 1 else tls[0],
                         {8: 3, 16: 4, 32: 5}[int(p_s.rstrip("w"))] | wave)
    plan_text = planner.describe()
    check_info = None

    if not multi:
        steps = args.steps if args.steps is not None else 200
        warmup = args.warmup if args.warmup is not None else 20
        ring = max(steps + warmup, 40)  # >= 640 MiB of distinct transforms; no buffer is transformed twice
        re = torch.empty(ring * N, dtype=torch.float64, device=dev)
        im = torch.empty_like(re)
        P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
        views = [(re[i * N:(i + 1) * N], im[i * N:(i + 1) * N]) for i in range(ring)]

        def step(i):
            r, m = views[i % ring]
            P.fft_64_dit_with_planner(r, m, P.Direction.Forward, planner)

        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        graph, launch = None, "eager launches from Python"
        if not args.no_graph:
            graph, launch = capture_steps(
                torch, P, step, warmup, steps,
                touch=lambda: P.fft_64_dit_with_planner(*views[0], P.Direction.Forward, planner))
        # every buffer of the ring as generated again (the warm-up steps and the untimed replay transformed theirs), and the caches
        # drained of what those left behind: the K timed steps find their inputs in HBM and nowhere else (`settle`)
        fresh = lambda: settle(torch, P, refill=lambda: P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0))
        if os.environ.get("PHAST_BENCH_NO_DRAIN") == "2":   # tools only: rounds 3-5's protocol (the timed replay right behind the untimed one)
            fresh = torch.cuda.synchronize
        fresh()
        # HIP events around the K timed steps, on the stream they are launched on (torch's current
        # stream: the graph replay / the library's launches go there); the wall clock around the same region --
        # synchronize, K steps, synchronize -- is kept beside it as ms_per_step_wall (it adds the fixed ~50 us of one graph
        # launch + the closing synchronisation to the K steps: profiles/r03_graph_protocol.log)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        if graph is not None:
            graph.replay()
        else:
            for i in range(steps):
                step(warmup + i)
        ev1.record()
        torch.cuda.synchronize()
        elapsed_wall = time.perf_counter() - t0
        elapsed = ev0.elapsed_time(ev1) * 1e-3
        # SURVEY.md 8(d) "median and min": two more K-step regions after the contract's one (ring re-filled before each: the
        # steps work in place), reported beside ms_per_step, never instead of it
        head_stats = replay_stats(torch, graph.replay if graph is not None else (lambda: [step(warmup + i) for i in range(steps)]),
                                  steps, 1e3 * elapsed, before=fresh)
        # ... and, for information only (never `value`): the same K-step region with the graph ALREADY QUEUED behind a running
        # kernel when the start event is reached -- the host's launch of the graph (10-20 us, once per K steps: 0.5-1 us per
        # step at K = 20) then happens while the GPU is busy with that kernel, outside the events
        if graph is not None and torch.cuda.current_device() in _DRAIN:
            fresh()                                                     # the region's usual starting state ...
            q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            P.digest(_DRAIN[torch.cuda.current_device()][0], _DRAIN[torch.cuda.current_device()][1], 1 << 20)   # ... and ~130 us of
            q0.record()                                                 # READ-ONLY work in front (a kernel that writes would leave
            graph.replay()                                              # its write-back to the timed steps: settle())
            q1.record()
            torch.cuda.synchronize()
            head_stats["ms_per_step_queued"] = q0.elapsed_time(q1) / steps
        samples_per_step = N
        workload = f"single f64 forward FFT N=2^{LOG_N}, in place, planar (BASELINE configs[1])"
        # --- roofline of the dominant pass kernel, HIP events bound to the dispatches (fresh buffers) ---
        fresh()
        acc = None
        reps = min(ring, 64)
        for i in range(reps):
            ms = planner.time_passes(views[i][0], views[i][1], N, reps=1)
            acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
        pass_ms = [a / reps for a in acc]
        units = 1
        # ... and the headline is CHECKED (round 5; until now only the shard was): the ring was just re-filled and transformed
        # once by the timing loop above -- buffers 0, reps // 2 and reps - 1 against the oracle's output of the same seeded
        # inputs (every bin: rel-L2 and the worst bin / rms bin), and one that was not touched transformed now.  The oracle is
        # the checker here, never the thing measured.
        check_info = check_headline(P, torch, views, planner, [0, reps // 2, reps - 1, ring - 1], already=reps)
        del re, im, views
        torch.cuda.empty_cache()
    else:
        steps = args.steps if args.steps is not None else 10
        warmup = args.warmup if args.warmup is not None else 2
        import torch.distributed as dist

        from phastft_amd.sharding import ShardedBatch, fabric_info, max_over_ranks, times_over_ranks

        total = args.shard * world
        first, shard = rank * args.shard, args.shard
        re = torch.empty(shard * N, dtype=torch.float64, device=dev)
        im = torch.empty_like(re)

        def refill():
            P.fill_uniform(re, im, N, seed=0xCAFE, first_id=first)

        def transform(first_id, count):  # the rank's contiguous shard, in place, no communication
            P.fft_dit_batched(re, im, N, P.Direction.Forward, planner)

        state = {"digest": None}

        def digest_fn(first_id, count):
            return state["digest"] if state["digest"] is not None else P.digest(re, im, N, probe=1)

        sb = ShardedBatch(total, N, rank, world, transform, digest_fn)
        assert (sb.first, sb.count) == (first, shard)
        refill()
        for i in range(warmup):
            sb.step()
        refill()  # values grow by sqrt(N) per in-place step: K timed steps from fresh inputs stay far from overflow
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sb.step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        elapsed = max_over_ranks(mine, dist, dev)
        rank_s = times_over_ranks(mine, dist, dev)   # every rank's own time: what makes a bad N-GPU point readable
        elapsed_wall = elapsed
        # after the timed region: every rank checks its shard (Parseval on all, sampled ids against the oracle), then
        # the trivial gather (RCCL over xGMI): one 32-byte digest per transform -> (total, 4) on every rank
        ok, check_info, state["digest"] = check_shard(P, torch, re, im, refill, sb.step, first, shard)
        digests = sb.gather_digests(dist)
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cpu" if args.backend == "gloo" else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        digest_ok = bool(flag.item() == 1.0) and digests.shape[0] == total and bool(torch.isfinite(digests).all())
        samples_per_step = sb.samples_per_step()
        workload = (f"{total} independent f64 forward FFTs N=2^{LOG_N}, {shard} per GPU, in place "
                    f"(BASELINE configs[4])")
        launch = "eager batched launches"
        refill()
        torch.cuda.synchronize()
        pass_ms = planner.time_passes(re, im, N, reps=2)
        units = shard

    ms_per_step = 1e3 * elapsed / steps
    value = samples_per_step * steps / elapsed / 1e9

    if rank == 0:
        used, plan_list = plan_used(planner, units)
        if args.plan:
            used = "forced"
        alg_bytes = BYTES_PER_SAMPLE * N * units            # what ONE launch of a pass must read + write
        roofline, dom = roofline_of(pass_ms, alg_bytes, plan_used=f"{used} plan {plan_list}", step_ms=ms_per_step)
        achieved = roofline["achieved"]
        out = {
            "metric": "GSamples/s f64 forward FFT N=2^20" + (" (N=2^26, round trip, R2C: see configs)" if not multi else ""),
            "value": value, "unit": "GSamples/s",
            "n_gpus": n_gpus, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
            "ms_per_step_wall": 1e3 * elapsed_wall / steps,
            "timing": ("HIP events on the launch stream around the K steps (SURVEY.md 8d); ms_per_step_wall = wall clock "
                       "around synchronize + K steps + synchronize" if not multi else
                       "wall clock between barrier + synchronize on both sides, MAX over ranks"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (counter-based uniform [-1,1), seed 0xCAFE, generated on device)",
            # the driver's record keeps the SCALAR keys of `config` (strings cut at 120 characters): `plan` is the plan that
            # ran, every table the planner holds is under top-level "plan_tables"
            "config": {"workload": workload, "n": N, "transforms_per_step": samples_per_step // N,
                       "plan": f"{used} {plan_list}"[:110], "plan_used": used, "launch": launch[:110]},
            "plan_tables": plan_text,
            "roofline": roofline,
        }
        if not multi:
            out.update(head_stats)
        if not multi and check_info is not None:
            out["result_check"] = check_info
        if multi:
            out["config"]["digest_gather"] = f"all_gather of {samples_per_step // N} x 32 B digests over RCCL"
            out["config"]["digest_ok"] = digest_ok
            out["digest_check"] = dict(check_info, what="every rank: Parseval on all transforms of its shard + "
                                       "sampled transforms vs digests of the CPU oracle's output; rank 0's numbers")
            # flat, self-diagnosing keys of the N > 1 line (tests/golden/multi_gpu_line.schema.json; tests/cpp/shard_host.cpp
            # prints the same ones): per-rank times of the timed region, the ranks the communicator really has, the collective
            # library's version and -- where rocm-smi is there -- the XGMI link count
            out["config"]["rank_ms_min"] = round(1e3 * min(rank_s) / steps, 6)
            out["config"]["rank_ms_max"] = round(1e3 * max(rank_s) / steps, 6)
            out["config"]["ranks_seen"] = len(rank_s)
            out["config"]["backend"] = str(dist.get_backend())
            out["config"]["shard"] = shard
            out["config"]["parseval_max_rel_dev"] = float(check_info.get("parseval_max_rel_dev", -1.0))
            out["config"]["oracle_digest_max_dev"] = float(check_info.get("oracle_digest_max_dev", -1.0))
            out["config"].update(fabric_info())
        traffic = load_profiled_traffic(2 if multi else 1, dom, len(pass_ms), kernel_tags(plan_list))
        if traffic is not None:
            if multi:  # profiled per 1024-transform launch; kernel_ms / algorithmic bytes here are per pass over the shard
                traffic["traffic"] *= units / 1024.0
                traffic["traffic_note"] = "PMC bytes of one 1024-transform launch scaled to the shard"
            roofline.update(traffic)
            roofline["traffic_over_algorithmic"] = traffic["traffic"] / alg_bytes
        if not multi:
            probe = hbm_copy_probe(torch, dev)
            roofline["copy_probe_GBps"] = probe          # torch's d2d copy_ on this box, same run (kept for continuity)
            roofline["frac_of_copy_probe"] = achieved / probe
            # the ceilings that matter: the library's own hand-written read / write / copy kernels on 1 GiB, this box, this
            # run (csrc/probe.hip).  A pass reads and writes every byte once: `copy` is what 100 % looks like for it here.
            sp = P.stream_probe(1024, 5)
            roofline["stream_probe"] = sp
            roofline["frac_of_copy"] = achieved / sp["copy"]
            roofline["pass_frac_of_copy"] = [alg_bytes / (t * 1e-3) / 1e9 / sp["copy"] for t in pass_ms]
            torch.cuda.empty_cache()
        cpu = not multi and not args.no_cpu_baseline
        if cpu:
            out["cpu_baseline"] = cpu_baseline()
            out["host_slice_api"] = host_slice_api(P)
        if not multi and not args.no_configs:
            fwd, rt = config_n2p26(P, torch, dev, 5, cpu)
            out["configs"] = {"n2p26_forward": fwd, "n2p26_roundtrip": rt, "r2c_f32_2p24": config_r2c(P, torch, dev, 20, cpu),
                              "c2r_f32_2p24": config_c2r(P, torch, dev, 20, cpu),
                              "f32_2p20": config_f32(P, torch, dev, 20, 20, cpu),
                              "f32_2p26": config_f32(P, torch, dev, 26, 5, cpu)}
            for c in out["configs"].values():   # every config's passes against the copy kernel of this box, this run
                add_copy_frac(c.get("roofline"), sp)
        if not multi and not args.no_scaling_reference:
            out["weak_scaling_reference"] = shard_on_one_gpu(P, torch, dev, args.shard)
            add_copy_frac(out["weak_scaling_reference"]["roofline"], sp)
            if cpu:
                out["weak_scaling_reference"]["cpu_baseline"] = shard_cpu_leg(args.shard)
        # The driver's record keeps the top-level keys and the SCALAR keys of `config` (nested dicts were dropped in rounds 3-5):
        # the WHOLE metric -- N = 2^20 and 2^26, absolute and as a fraction of the HBM roofline -- and every other
        # configuration measured in this run go there as flat scalars.  *_gsps GSamples/s, *_ms ms per step,
        # *_frac_transform = compulsory bytes / ms_per_step / 8 TB/s, *_frac_pass = dominant kernel / 8 TB/s,
        # *_static_ms = the static rule's plan timed beside a `tuned:` one.
        cfg = out["config"]

        def flat(prefix, c, keys=("gsps", "ms", "frac_transform", "frac_pass")):
            r = c.get("roofline", {})
            vals = {"gsps": c["value"], "ms": c["ms_per_step"], "frac_transform": r.get("frac_transform"), "frac_pass": r.get("frac"),
                    "ms_min": c.get("ms_per_step_min"), "ms_median": c.get("ms_per_step_median"), "static_ms": c.get("static_ms"),
                    "ms_queued": c.get("ms_per_step_queued"),
                    "cpu_gsps": c.get("cpu_baseline", {}).get("value"),
                    "traffic_x": (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None}
            for k in keys:
                if vals.get(k) is not None:
                    cfg[f"{prefix}_{k}"] = round(float(vals[k]), 6 if k.endswith("ms") or "ms_" in k else 4)

        head = {"value": value, "ms_per_step": ms_per_step, "roofline": roofline, **(head_stats if not multi else {}),
                "cpu_baseline": out.get("cpu_baseline", {})}
        flat("n2p20", head, ("gsps", "ms", "ms_min", "ms_median", "ms_queued", "frac_transform", "frac_pass", "cpu_gsps", "traffic_x"))
        names = {"n2p26_forward": "n2p26", "n2p26_roundtrip": "rt2p26", "r2c_f32_2p24": "r2c_f32_2p24", "c2r_f32_2p24": "c2r_f32_2p24",
                 "f32_2p20": "f32_2p20", "f32_2p26": "f32_2p26"}
        for name, c in out.get("configs", {}).items():
            flat(names.get(name, name), c, ("gsps", "ms", "ms_min", "ms_median", "frac_transform", "frac_pass", "static_ms", "cpu_gsps",
                                            "traffic_x"))
            if c.get("plan_ran"):
                cfg[f"{names.get(name, name)}_plan"] = c["plan_ran"][:110]
        if "weak_scaling_reference" in out:
            flat(f"shard{args.shard}", out["weak_scaling_reference"], ("gsps", "ms", "frac_transform", "frac_pass", "cpu_gsps", "traffic_x"))
        if "host_slice_api" in out:
            cfg["host_slice_gsps"] = round(out["host_slice_api"]["value"], 4)   # the reference's calling convention: PCIe-bound
            cfg["host_slice_ms"] = round(out["host_slice_api"]["ms_per_call"], 4)
        if not multi and check_info is not None:
            cfg["check_rel_l2"] = float(check_info.get("rel_l2_max", -1.0))
            cfg["check_ok"] = bool(check_info.get("ok", False))
        if "configs" in out and "n2p26_roundtrip" in out["configs"]:
            cfg["rt2p26_err_ok"] = bool(out["configs"]["n2p26_roundtrip"].get("err_ok", False))
        if not multi:
            cfg["copy_probe_GBps"] = round(float(roofline.get("stream_probe", {}).get("copy", 0.0)), 1)
        print(json.dumps(out), flush=True)
    if multi:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def _traffic_file():
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def load_profiled_traffic(n_gpus, dom, n_passes, tags):
    """HBM bytes per launch of the dominant pass kernel from the rocprofv3 PMC runs committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950).  bench.py cannot collect PMC counters itself; the file records which
    command produced the numbers.  None when no profile matches the plan that ran."""
    t = _traffic_file()
    key = "single_2p20" if n_gpus == 1 else "batch_2p20"
    if not t or key not in t:
        return None
    ks = t[key].get("kernels", [])
    # the profile may hold other kernels too: pick, pass by pass, the entries of THIS plan (first / later passes differ
    # in the PRE_TW, TRANSPOSE flags that follow the shape in the kernel's name)
    picked = []
    for tag in tags:
        hits = [k for k in ks if tag in k["kernel"]]
        if len(hits) != 1:
            return None  # the profile was taken with another plan
        picked.append(hits[0])
    if len(picked) != n_passes:
        return None
    return {"traffic": picked[dom]["hbm_bytes_per_launch"], "traffic_source": t[key]["source"],
            "traffic_kernel": picked[dom]["kernel"]}


def traffic_for(key, substrings, scale=1.0):
    """traffic of the profiled kernel whose name contains every one of `substrings` (None if absent / ambiguous)"""
    t = _traffic_file()
    if not t or key not in t:
        return None
    hits = [k for k in t[key].get("kernels", []) if all(x in k["kernel"] for x in substrings)]
    if len(hits) != 1:
        return None
    return {"traffic": hits[0]["hbm_bytes_per_launch"] * scale, "traffic_source": t[key]["source"],
            "traffic_kernel": hits[0]["kernel"]}


if __name__ == "__main__":
    main()
