#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X PhastFT path.

    python bench.py [--gpus N] [--steps K] [--warmup W]                     (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): GSamples/s of the f64 forward planar FFT, complex samples transformed per
second over the whole job, inputs resident in HBM when the timed region starts.

  * N = 1  -> configs[1]: "Single f64 forward FFT, N=2^20, 1xMI355X".  One step = one in-place
    `fft_64_dit_with_planner` (the _dev entry point of the C ABI) on one 16 MiB transform.  Every step
    uses a fresh buffer of a pre-filled ring (so values never overflow and the ring, > 256 MiB, defeats
    the Infinity Cache); the K steps are captured once into a HIP graph and replayed inside the timed
    region so that the host launch path (Python + ctypes) is not what is measured.
  * N > 1  -> configs[4]: 8192 independent N=2^20 transforms per 8 GPUs = 1024 per GPU, fixed per-GPU
    work ("scaling": "weak"); one step = every rank transforms its 1024-transform shard in place.  The
    path has no exchange step, so there is no data-path collective; RCCL (torch.distributed "nccl") only
    carries the barrier, the max-over-ranks time and the trivial digest gather.

At N = 1 the line also carries "weak_scaling_reference": one rank's shard of the N > 1 workload timed on this one
GPU -- the denominator for scaling efficiency (the N = 1 headline is ONE transform, a different workload).

Extra objects on the JSON line: "roofline" (HIP-event duration of the dominant pass kernel vs the
8 TB/s HBM peak) and "cpu_baseline" (the oracle -- a C restatement of the
reference's CPU algorithm -- timed on this host on a bounded sample; rank 0, N = 1 only).
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
    ap.add_argument("--shard", type=int, default=SHARD, help="transforms per GPU when --gpus > 1")
    ap.add_argument("--extra", action="store_true", help="also measure N=2^26 and the batched shard at --gpus 1")
    ap.add_argument("--plan", default=None, help="experiment: force a plan, e.g. 6,8,6@12p8 (default: the library's own)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--same-gpu", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def cpu_baseline(budget_s: float = 12.0):
    """The oracle (C restatement of PhastFT's CPU path) on the host cores of this box -- the checker
    doubling as the reported CPU baseline, examples/benchmark.rs protocol (planner outside the timer,
    input regenerated before every timed call)."""
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
                  f"({total:.1f} s of CPU work, {1e3 * total / iters:.2f} ms each), oracle/ C restatement, 1 thread",
        "parallel_feature": {"value": p_iters * N / p_total / 1e9, "unit": "GSamples/s", "cores": threads,
                             "sample": f"{p_iters} calls, {1e3 * p_total / p_iters:.2f} ms each, best of 2..64 threads "
                                       f"({avail} schedulable); rayon::join emulated with OpenMP tasks (2-way bit "
                                       f"reversal, recursive join while size > 16384, spanning stages serial)"},
    }


def shard_on_one_gpu(P, torch, dev, shard: int, steps: int = 3):
    """The per-GPU workload of the --gpus N > 1 runs (BASELINE configs[4]: `shard` transforms of 2^20 per GPU,
    transformed in place per step) timed on this one GPU: the denominator for weak-scaling efficiency.  The N = 1
    headline above is a different workload (ONE transform, configs[1]) and must not be used for that."""
    pl = P.PlannerDit64(N)
    re = torch.empty(shard * N, dtype=torch.float64, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
    P.fft_dit_batched(re, im, N, P.Direction.Forward, pl)  # warm-up (scratch allocation)
    P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        P.fft_dit_batched(re, im, N, P.Direction.Forward, pl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": f"{shard} independent f64 forward FFTs N=2^{LOG_N} on 1 GPU, in place (one rank's shard of "
                        f"BASELINE configs[4])", "value": shard * N * steps / dt / 1e9, "unit": "GSamples/s",
            "steps": steps, "ms_per_step": 1e3 * dt / steps}


def kernel_tags(plan_text: str, latency: bool):
    """'<double, LR, LC, LP,' template-argument prefixes of the pass kernels of the plan that ran, parsed from
    planner.describe() -- used to check that a committed PMC profile belongs to this plan."""
    import math
    import re

    part = plan_text.split("latency=")[1] if (latency and "latency=" in plan_text) else plan_text.split("latency=")[0]
    tags = []
    for rows, cols, pts in re.findall(r"\[(\d+)x(\d+)A? p(\d+)", part):
        tags.append(f"<double, {int(math.log2(int(rows)))}, {int(math.log2(int(cols)))}, {int(math.log2(int(pts)))},")
    return tags


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


def main():
    args = parse()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1
    if multi:
        import torch.distributed as dist

        if args.same_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and (args.gpus > 1 or world > 1):
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world

    import phastft_amd as P

    dev = torch.device("cuda", local_rank if multi else 0)
    planner = P.PlannerDit64(N)
    if args.plan:
        lrs_s, rest = args.plan.split("@")
        tl_s, p_s = rest.split("p")
        planner.set_plan(tuple(int(x) for x in lrs_s.split(",")), int(tl_s), {8: 3, 16: 4, 32: 5}[int(p_s)])
    plan_text = planner.describe()

    if n_gpus == 1:
        steps = args.steps if args.steps is not None else 200
        warmup = args.warmup if args.warmup is not None else 20
        ring = max(steps + warmup, 32)  # >= 512 MiB of distinct transforms; no buffer is transformed twice
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
        graph = None
        if not args.no_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    P.fft_64_dit_with_planner(*views[0], P.Direction.Forward, planner)  # touch on the side stream
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                P.fill_uniform(views[0][0], views[0][1], N, seed=0xCAFE, first_id=0)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(steps):
                        step(warmup + i)
                graph = g
            except Exception as e:  # capture is an optimisation of the launch path only
                print(f"note: HIP graph capture unavailable ({e}); launching eagerly", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            for i in range(steps):
                step(warmup + i)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        samples_per_step = N
        workload = f"single f64 forward FFT N=2^{LOG_N}, in place, planar (BASELINE configs[1])"
        launch = "hipGraph replay of the K steps" if graph is not None else "eager launches from Python"
        # --- roofline of the dominant pass kernel, HIP events on the launch stream (fresh buffers) ---
        P.fill_uniform(re, im, N, seed=0xCAFE, first_id=0)
        torch.cuda.synchronize()
        acc = None
        reps = min(ring, 64)
        for i in range(reps):
            ms = planner.time_passes(views[i][0], views[i][1], N, reps=1)
            acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
        pass_ms = [a / reps for a in acc]
        units = 1
    else:
        steps = args.steps if args.steps is not None else 10
        warmup = args.warmup if args.warmup is not None else 2
        import torch.distributed as dist

        from phastft_amd.sharding import ShardedBatch, max_over_ranks

        total = args.shard * world
        first, shard = rank * args.shard, args.shard
        re = torch.empty(shard * N, dtype=torch.float64, device=dev)
        im = torch.empty_like(re)

        def refill():
            P.fill_uniform(re, im, N, seed=0xCAFE, first_id=first)

        def transform(first_id, count):  # the rank's contiguous shard, in place, no communication
            P.fft_dit_batched(re, im, N, P.Direction.Forward, planner)

        def digest_fn(first_id, count):
            return P.digest(re, im, N, probe=1)

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
        elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev)
        # the trivial gather (RCCL over xGMI): one 32-byte digest per transform -> (total, 4) on every rank.  Taken
        # from one step on fresh inputs so that it stays finite whatever K was (values grow 2^10-fold per step)
        refill()
        sb.step()
        digests = sb.gather_digests(dist)
        digest_ok = bool(torch.isfinite(digests).all()) and digests.shape[0] == total
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

    out = None
    if rank == 0:
        dom = max(range(len(pass_ms)), key=lambda i: pass_ms[i])
        alg_bytes = BYTES_PER_SAMPLE * N * units            # what ONE launch of a pass must read + write
        achieved = alg_bytes / (pass_ms[dom] * 1e-3) / 1e9  # GB/s of the dominant pass kernel
        total_ms = sum(pass_ms)
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": f"tile_fft pass {dom} of {len(pass_ms)}",
            "kernel_ms": pass_ms[dom], "pass_ms": pass_ms, "algorithmic_bytes_per_launch": alg_bytes,
            # whole transform against the one-pass ideal (32 B/sample once): capped at 1/passes by construction
            "transform_frac": alg_bytes / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "passes": len(pass_ms),
        }
        out = {
            "metric": "GSamples/s f64 forward FFT N=2^20", "value": value, "unit": "GSamples/s",
            "n_gpus": n_gpus, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (counter-based uniform [-1,1), seed 0xCAFE, generated on device)",
            "config": {"workload": workload, "n": N, "transforms_per_step": samples_per_step // N,
                       "plan": plan_text, "launch": launch},
            "roofline": roofline,
        }
        if n_gpus > 1:
            out["config"]["digest_gather"] = f"all_gather of {samples_per_step // N} x 32 B digests over RCCL"
            out["config"]["digest_ok"] = digest_ok
        traffic = load_profiled_traffic(n_gpus, dom, len(pass_ms), kernel_tags(plan_text, n_gpus == 1))
        if traffic is not None:
            roofline.update(traffic)
        if n_gpus == 1:
            probe = hbm_copy_probe(torch, dev)
            roofline["copy_probe_GBps"] = probe          # measured d2d copy rate of this box, same run
            roofline["frac_of_copy_probe"] = achieved / probe
        if n_gpus == 1 and not args.no_scaling_reference:
            out["weak_scaling_reference"] = shard_on_one_gpu(P, torch, dev, args.shard)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if n_gpus == 1 and args.extra:
            out["extra"] = extra_measurements(P, torch, dev)
        print(json.dumps(out), flush=True)
    if multi:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def load_profiled_traffic(n_gpus, dom, n_passes, kernel_tags):
    """HBM bytes per launch of the dominant pass kernel from the rocprofv3 PMC runs committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950).  bench.py cannot collect PMC counters itself; the file records which
    command produced the numbers.  None when no profile matches this workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    key = "single_2p20" if n_gpus == 1 else "batch_2p20"
    if key not in t:
        return None
    ks = t[key].get("kernels", [])
    # the profile may hold other kernels too: pick, pass by pass, the entries of THIS plan (first / later passes differ
    # in the PRE_TW, TRANSPOSE flags that follow the shape in the kernel's name)
    picked = []
    for i, tag in enumerate(kernel_tags):
        flags = " false, true," if i == 0 else " true, false,"
        hits = [k for k in ks if tag + flags in k["kernel"]]
        if len(hits) != 1:
            return None  # the profile was taken with another plan
        picked.append(hits[0])
    if len(picked) != n_passes:
        return None
    return {"traffic": picked[dom]["hbm_bytes_per_launch"], "traffic_source": t[key]["source"],
            "traffic_kernel": picked[dom]["kernel"]}


def extra_measurements(P, torch, dev):
    """The other BASELINE configs on one GPU, reported beside the headline: N=2^26 forward and the forward+inverse
    round trip (configs[2]), f32 R2C at N=2^24 (configs[3]), one GPU's 1024-transform shard (configs[4]), and the
    host-slice (drop-in, PCIe-inclusive) call at N=2^20."""
    import numpy as np

    res = {}

    def wall(fn, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    n26 = 1 << 26
    pl = P.PlannerDit64(n26)
    re = torch.empty(n26, dtype=torch.float64, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, n26)
    P.fft_64_dit_with_planner(re, im, P.Direction.Forward, pl)
    P.fill_uniform(re, im, n26)
    ms = pl.time_passes(re, im, n26, reps=3)
    res["n2p26_single"] = {"plan": pl.describe(), "pass_ms": ms, "gsamples_per_s": n26 / (sum(ms) * 1e-3) / 1e9,
                           "transform_frac": 32 * n26 / (sum(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # configs[2]: forward + inverse round trip on the same buffers, error against the input
    P.fill_uniform(re, im, n26)
    ref_re, ref_im = re.clone(), im.clone()

    def roundtrip():
        P.fft_64_dit_with_planner(re, im, P.Direction.Forward, pl)
        P.fft_64_dit_with_planner(re, im, P.Direction.Reverse, pl)

    roundtrip()
    ms_rt = wall(roundtrip, 3)
    err = max(float((re - ref_re).abs().max()), float((im - ref_im).abs().max()))
    res["n2p26_roundtrip"] = {"ms": ms_rt, "gsamples_per_s": 2 * n26 / (ms_rt * 1e-3) / 1e9,
                              "max_abs_err_after_4_roundtrips": err}
    del re, im, pl, ref_re, ref_im
    # configs[3]: f32 R2C, N=2^24 (algorithmic bytes 4N + 8(N/2+1))
    n24 = 1 << 24
    plr = P.PlannerR2c32(n24)
    x = torch.empty(n24, dtype=torch.float32, device=dev)
    P.fill_uniform(x, None, n24)
    ore = torch.empty(n24 // 2 + 1, dtype=torch.float32, device=dev)
    oim = torch.empty_like(ore)
    P.r2c_fft_f32_with_planner(x, ore, oim, plr)
    ms_r2c = wall(lambda: P.r2c_fft_f32_with_planner(x, ore, oim, plr), 20)
    r2c_bytes = 4 * n24 + 8 * (n24 // 2 + 1)
    res["r2c_f32_2p24"] = {"ms": ms_r2c, "gsamples_per_s": n24 / (ms_r2c * 1e-3) / 1e9,
                           "algorithmic_GBps": r2c_bytes / (ms_r2c * 1e-3) / 1e9,
                           "transform_frac": r2c_bytes / (ms_r2c * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del x, ore, oim, plr
    # the drop-in call on host slices: H2D + transform + D2H, blocking (never the headline value)
    pl = P.PlannerDit64(N)
    h_re = np.random.default_rng(1).uniform(-1, 1, N)
    h_im = np.random.default_rng(2).uniform(-1, 1, N)
    P.fft_64_dit_with_planner(h_re, h_im, P.Direction.Forward, pl)
    t0 = time.perf_counter()
    for _ in range(5):
        P.fft_64_dit_with_planner(h_re, h_im, P.Direction.Forward, pl)
    ms_host = 1e3 * (time.perf_counter() - t0) / 5
    res["n2p20_host_slices"] = {"ms": ms_host, "gsamples_per_s": N / (ms_host * 1e-3) / 1e9,
                                "note": "pageable numpy arrays, PCIe both ways inside the call"}
    re = torch.empty(SHARD * N, dtype=torch.float64, device=dev)
    im = torch.empty_like(re)
    P.fill_uniform(re, im, N)
    P.fft_dit_batched(re, im, N, P.Direction.Forward, pl)
    P.fill_uniform(re, im, N)
    ms = pl.time_passes(re, im, N, reps=2)
    res["n2p20_batch1024"] = {"plan": pl.describe(), "pass_ms": ms,
                              "gsamples_per_s": SHARD * N / (sum(ms) * 1e-3) / 1e9,
                              "transform_frac": 32 * SHARD * N / (sum(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return res


if __name__ == "__main__":
    main()
