"""One-shard-per-GPU batching (BASELINE configs[4]).

Independent transforms share nothing (the planner is read-only: planner.rs:38-39), so a batch is cut into
contiguous shards, one per rank, and every rank transforms its shard with zero communication.  The only
collective is the trivial gather of a 32-byte digest per transform (sum re, sum im, energy, one probe bin)
-- gathering full outputs would be xGMI-bound by construction (112 GiB inbound for 8192 x 2^20 f64).
`torch.distributed` supplies the process group: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests of the host logic.
"""
from __future__ import annotations

from typing import Callable, Sequence


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """(first transform id, count) of `rank`'s contiguous shard; the remainder goes to the first ranks."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


class ShardedBatch:
    """`total` independent length-`n` transforms split over the ranks of a process group.

    `transform(first_id, count)` runs the rank's shard in place and `digest(first_id, count)` returns a
    (count, 4) float64 tensor; both are injected so that the GPU path (phastft_amd on device tensors) and the
    CPU tests of this host logic share the code.
    """

    def __init__(self, total: int, n: int, rank: int, world: int, transform: Callable[[int, int], None],
                 digest: Callable[[int, int], "object"]):
        self.total, self.n, self.rank, self.world = total, n, rank, world
        self.first, self.count = shard_bounds(total, world, rank)
        self._transform, self._digest = transform, digest

    def step(self) -> None:
        if self.count:
            self._transform(self.first, self.count)

    def samples_per_step(self) -> int:
        """whole-job samples per step (all ranks)"""
        return self.total * self.n

    def gather_digests(self, dist=None):
        """All-gather of the per-transform digests; returns a (total, 4) tensor ordered by transform id."""
        import torch

        mine = self._digest(self.first, self.count)
        if dist is None:  # no process group: a single process
            return mine
        # with a process group the collective runs even for one rank (a world-size-1 "nccl" group on a one-GPU box
        # is how RCCL init and the device-tensor all_gather are exercised on hardware, tests/test_gpu_parity_r3.py)
        if dist.get_backend() == "gloo":  # CPU collectives (tests, dry runs)
            mine = mine.cpu()
        counts = [shard_bounds(self.total, self.world, r)[1] for r in range(self.world)]
        width = max(counts)
        pad = torch.zeros((width, 4), dtype=mine.dtype, device=mine.device)
        pad[: self.count] = mine
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad)
        return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def times_over_ranks(seconds: float, dist=None, device=None) -> list:
    """Every rank's own time for the timed region (all-gather of one scalar): what one needs to read a bad N-GPU point --
    a straggler shows as one outlier, a slow fabric or a clock-capped box as a uniform shift."""
    if dist is None:
        return [seconds]
    import torch

    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def fabric_info() -> dict:
    """Flat, best-effort description of the collective library and the GPU fabric for the N > 1 bench line: `rccl_version`
    (torch.cuda.nccl.version() -- RCCL on ROCm) and, when `rocm-smi --showtopo` is on PATH, `xgmi_links` = the number of
    GPU pairs it reports as XGMI-linked.  Missing pieces are simply absent; nothing here is needed by the data path."""
    info = {}
    try:
        import torch

        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass
    try:
        import shutil
        import subprocess

        exe = shutil.which("rocm-smi")
        if exe:
            r = subprocess.run([exe, "--showtopotype"], capture_output=True, text=True, timeout=20)
            if r.returncode == 0:
                info["xgmi_links"] = sum(line.count("XGMI") for line in r.stdout.splitlines() if line.startswith("GPU")) // 2
    except Exception:
        pass
    return info


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """The bench contract: a step is as slow as its slowest rank."""
    if dist is None:
        return seconds
    import torch

    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
