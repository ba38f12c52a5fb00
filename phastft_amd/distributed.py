"""One transform spread over the GPUs of a node.

PhastFT's stated niche is "gigabytes of data" (README.md:12) in ONE address space; this module is what comes after
that: a single power-of-two transform of N points whose planar (reals, imags) arrays are cut into P contiguous
slabs, rank r holding ``x[r*N/P : (r+1)*N/P]`` -- natural order in, natural order out, the same contract as
``fft_64_dit`` (lib.rs:180), forward unnormalised and reverse scaled by 1/N (algorithms/dit.rs:297-331).

Four-step split ``N = N1*N2`` (``n = n1*N2 + n2``, ``k = k1 + N1*k2``):

    slab [n1 (mine)][n2]  --all-to-all-->  [n1][n2 (mine)]
        FFT over n1 (local, batched)  ->  [k1][n2 (mine)],  times W_N^(n2*k1)   (TwiddleGrid, twiddle.hip)
    --all-to-all-->  [k1 (mine)][n2]
        FFT over n2 (local, batched)  ->  [k1 (mine)][k2]
    --all-to-all-->  [k2 (mine)][k1]  =  X[k] in natural order, slab of rank ``k2 block``

Three exchanges because both ends are block-distributed in natural order: the first stage of any Cooley-Tukey
split combines the HIGH index bits, which are exactly the bits the block distribution spreads over the ranks.
Each exchange moves (P-1)/P of the data over xGMI, which is what bounds the whole thing (7 links x ~50-150 GB/s
per GPU against ~5 TB/s of HBM): the local stages are the library's batched kernels and take a few per cent.

The exchanges are ``torch.distributed.all_to_all_single`` (backend "nccl" = RCCL on GPUs; "gloo" moves the blocks
through host memory and serves the CPU tests of this logic and single-GPU dry runs).  The local stages are injected
(`local_fft`, `twiddle`) so that the host logic is testable without a GPU (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Callable


def split_factors(log_n: int, world: int) -> tuple[int, int]:
    """(N1, N2) with N1*N2 = 2^log_n, N1 >= N2, both divisible by `world` (a power of two)."""
    if world <= 0 or world & (world - 1):
        raise ValueError("the number of ranks must be a power of two")
    n1 = 1 << ((log_n + 1) // 2)
    n2 = 1 << (log_n // 2)
    if n2 < world:
        raise ValueError(f"N = 2^{log_n} is too small for {world} ranks (needs N >= ranks^2)")
    return n1, n2


class DistributedFft:
    """`n`-point planar transform over the ranks of a process group; every rank calls :meth:`run` with its slab.

    ``local_fft(re, im, length, count)`` transforms `count` contiguous length-`length` transforms in place
    (forward); ``twiddle(re, im, rows, cols, row0)`` multiplies the row-major block by W_n^((row0+r)*c).
    Both act on 1-D tensors of the process group's device type.
    """

    def __init__(self, n: int, rank: int, world: int, local_fft: Callable, twiddle: Callable, dist=None):
        if n <= 0 or n & (n - 1):
            raise ValueError("assertion failed: num_points > 0 && num_points.is_power_of_two()")  # planner.rs:66
        self.n, self.rank, self.world, self.dist = n, rank, world, dist
        self.n1, self.n2 = split_factors(n.bit_length() - 1, world)
        self._fft, self._twiddle = local_fft, twiddle
        if world > 1 and dist is None:
            raise ValueError("more than one rank needs a torch.distributed process group")

    # -- one exchange: `x` viewed as [rows_local][world][cols/world]; rank q receives every rank's q-th column
    #    block; returns [rows_local * world][cols / world] (row index = source rank major) --
    def _exchange(self, x, rows_local: int, cols: int):
        import torch

        w = self.world
        if w == 1:
            return x
        send = x.view(rows_local, w, cols // w).permute(1, 0, 2).contiguous().view(-1)
        recv = torch.empty_like(send)
        if self.dist.get_backend() == "gloo" and send.is_cuda:  # dry run on one GPU: through host memory
            r = torch.empty(send.shape, dtype=send.dtype)
            self.dist.all_to_all_single(r, send.cpu())
            recv.copy_(r)
        else:
            self.dist.all_to_all_single(recv, send)
        return recv

    @staticmethod
    def _transpose(x, rows: int, cols: int):
        return x.view(rows, cols).t().contiguous().view(-1)

    def run(self, reals, imags, reverse: bool = False):
        """In place on the rank's slab (1-D tensors of n/world elements)."""
        w, n1, n2 = self.world, self.n1, self.n2
        if reals.numel() != self.n // w or imags.numel() != self.n // w:
            raise ValueError("assertion `left == right` failed: reals.len() == imags.len()")  # dit.rs:284
        re, im = (imags, reals) if reverse else (reals, imags)  # the swap trick, algorithms/dit.rs:297-300
        r1, c2 = n1 // w, n2 // w
        # [n1 mine][n2] -> [n1][n2 mine] -> [n2 mine][n1]; FFT over n1; twiddle W_N^(n2*k1)
        a_re = self._transpose(self._exchange(re, r1, n2), n1, c2)
        a_im = self._transpose(self._exchange(im, r1, n2), n1, c2)
        self._fft(a_re, a_im, n1, c2)
        self._twiddle(a_re, a_im, c2, n1, self.rank * c2)
        # [n2 mine][k1] -> [n2][k1 mine] -> [k1 mine][n2]; FFT over n2
        b_re = self._transpose(self._exchange(a_re, c2, n1), n2, r1)
        b_im = self._transpose(self._exchange(a_im, c2, n1), n2, r1)
        del a_re, a_im
        self._fft(b_re, b_im, n2, r1)
        # [k1 mine][k2] -> [k1][k2 mine] -> [k2 mine][k1] = natural order
        c_re = self._transpose(self._exchange(b_re, r1, n2), n1, c2)
        c_im = self._transpose(self._exchange(b_im, r1, n2), n1, c2)
        del b_re, b_im
        if reverse:
            scale = 1.0 / self.n  # algorithms/dit.rs:325-331
            c_re *= scale
            c_im *= scale
        re.copy_(c_re)
        im.copy_(c_im)


def gpu_transform(n: int, rank: int, world: int, dist=None, dtype: str = "f64") -> DistributedFft:
    """The GPU instance: local stages = the library's batched kernels and the TwiddleGrid kernel."""
    import phastft_amd as P

    f64 = dtype == "f64"
    n1, n2 = split_factors(n.bit_length() - 1, world)
    planners = {m: (P.PlannerDit64 if f64 else P.PlannerDit32)(m) for m in {n1, n2}}
    grid = (P.TwiddleGrid64 if f64 else P.TwiddleGrid32)(n)

    def local_fft(re, im, length, count):
        P.fft_dit_batched(re, im, length, P.Direction.Forward, planners[length])

    def twiddle(re, im, rows, cols, row0):
        grid.apply(re, im, rows, cols, row0=row0)

    t = DistributedFft(n, rank, world, local_fft, twiddle, dist)
    t._keep = (planners, grid)
    return t
