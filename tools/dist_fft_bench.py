#!/usr/bin/env python3
"""One transform over the GPUs of a node (phastft_amd/distributed.py): time per transform.

    python tools/dist_fft_bench.py --log-n 28                       # world = 1: the local stages only
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/dist_fft_bench.py --log-n 30                          # RCCL all-to-all over xGMI
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import phastft_amd as P  # noqa: E402
from phastft_amd.distributed import gpu_transform  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=28)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--backend", default="nccl")
a = ap.parse_args()
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist

    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend)
n = 1 << a.log_n
dt = torch.float64 if a.dtype == "f64" else torch.float32
re = torch.empty(n // world, dtype=dt, device="cuda")
im = torch.empty_like(re)
P.fill_uniform(re, im, n // world, seed=0xCAFE + rank)
t = gpu_transform(n, rank, world, dist, a.dtype)
t.run(re, im)
torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
t0 = time.perf_counter()
for _ in range(a.reps):
    t.run(re, im)
torch.cuda.synchronize()
if dist is not None:
    dist.barrier()
ms = 1e3 * (time.perf_counter() - t0) / a.reps
if rank == 0:
    print(f"N=2^{a.log_n} {a.dtype} over {world} rank(s): {ms:.3f} ms per transform, {n / ms / 1e6:.1f} GSamples/s "
          f"(N1 x N2 = {t.n1} x {t.n2})")
if dist is not None:
    dist.destroy_process_group()
