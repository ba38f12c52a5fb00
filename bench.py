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
    ap.add_argument("--shard", type=int, default=SHARD, help="transforms per GPU when --gpus > 1")
    ap.add_argument("--extra", action="store_true", help="also measure N=2^26 and the batched shard at --gpus 1")
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
    return {
        "value": iters * N / total / 1e9, "unit": "GSamples/s", "cores": 1, "kind": "port",
        "sample": f"{iters} forward fft_64_dit_with_planner calls at N=2^{LOG_N} "
                  f"({total:.1f} s of CPU work, {1e3 * total / iters:.2f} ms each), oracle/ C restatement, 1 thread",
    }


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
        # the trivial gather (RCCL over xGMI): one 32-byte digest per transform -> (total, 4) on every rank
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
        traffic = load_profiled_traffic(n_gpus)
        if traffic is not None:
            roofline.update(traffic)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if n_gpus == 1 and args.extra:
            out["extra"] = extra_measurements(P, torch, dev)
        print(json.dumps(out), flush=True)
    if multi:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def load_profiled_traffic(n_gpus):
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
    return {"traffic": t[key]["hbm_bytes_per_launch"], "traffic_source": t[key]["source"]}


def extra_measurements(P, torch, dev):
    """N=2^26 single transform and the 1024-transform shard on one GPU (reported beside the headline)."""
    res = {}
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
    del re, im, pl
    pl = P.PlannerDit64(N)
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
