"""One transform spread over the GPUs of a node.

PhastFT's stated niche is "gigabytes of data" (README.md:12) in ONE address space; this module is what comes after
that: a single power-of-two transform of N points whose planar (reals, imags) arrays are cut into P contiguous
slabs, rank r holding ``x[r*N/P : (r+1)*N/P]`` -- natural order in, natural order out, the same contract as
``fft_64_dit`` (lib.rs:180), forward unnormalised and reverse scaled by 1/N (algorithms/dit.rs:297-331).

Four-step split ``N = N1*N2`` (``n = n1*N2 + n2``, ``k = k1 + N1*k2``), every array row-major:

    slab [n1 (mine)][n2]  --pack, all-to-all-->  [n1][n2 (mine)]
        COLUMN FFTs over n1 (strided batch, in place)  ->  [k1][n2 (mine)]
    --all-to-all (row blocks are contiguous: no pack), unpack-->  [n2][k1 (mine)]
        times W_N^(n2*k1), fused into the first load of the COLUMN FFTs over n2  ->  [k2][k1 (mine)]
        (phast_fft_*_dit_strided_tw_dev; shapes it declines: TwiddleGrid sweep, twiddle.hip)
    --all-to-all (contiguous row blocks), unpack-->  [k2 (mine)][k1]  =  X[k] in natural order, slab of rank ``k2 block``

The local transforms are strided batches ("column FFTs", ``phast_fft_*_dit_strided_dev``): they run on the layout an
exchange delivers, so each exchange costs ONE local permutation (the pack or the unpack a block-distributed global
transpose cannot avoid) instead of a pack plus a transpose -- three sweeps per plane where round 1 made six.

Three exchanges because both ends are block-distributed in natural order: the first stage of any Cooley-Tukey
split combines the HIGH index bits, which are exactly the bits the block distribution spreads over the ranks.
Each exchange moves (P-1)/P of the data over xGMI, which is what bounds the whole thing (7 links x ~50-150 GB/s
per GPU against ~5 TB/s of HBM): the local stages are the library's batched kernels and take a few per cent.

The exchanges are ``torch.distributed.all_to_all_single`` (backend "nccl" = RCCL on GPUs; "gloo" moves the blocks
through host memory and serves the CPU tests of this logic and single-GPU dry runs).  The local stages are injected
(`column_fft`, `twiddle`, optionally `column_fft_tw`) so that the host logic is testable without a GPU (tests/test_distributed_cpu.py).
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

    ``column_fft(re, im, length, count)`` transforms the `count` columns of the row-major ``[length][count]`` array in
    place (forward); ``twiddle(re, im, rows, cols, col0)`` multiplies element (r, c) of the row-major block by
    W_n^(r*(col0+c)).  Both act on 1-D tensors of the process group's device type.  Optional
    ``column_fft_tw(re, im, length, count, col0)``: the same transform with the multiplication by W_n^(j*(col0+c)) fused
    into its first load and returning True -- or False when it cannot, in which case `twiddle` + `column_fft` run.
    """

    def __init__(self, n: int, rank: int, world: int, column_fft: Callable, twiddle: Callable, dist=None,
                 column_fft_tw: Callable | None = None):
        if n <= 0 or n & (n - 1):
            raise ValueError("assertion failed: num_points > 0 && num_points.is_power_of_two()")  # planner.rs:66
        self.n, self.rank, self.world, self.dist = n, rank, world, dist
        self.n1, self.n2 = split_factors(n.bit_length() - 1, world)
        self._fft, self._twiddle, self._fft_tw = column_fft, twiddle, column_fft_tw
        # measurement hook (bench.py --dist-fft): called with the name of every stage of run() as it has been ENQUEUED --
        # "pack1", "exchange1", "fft_n1", "twiddle" (only when not fused), "exchange2", "unpack2", "fft_n2", "exchange3",
        # "unpack3", "store" -- so that the caller can drop an event on its stream behind each
        self.on_stage: Callable[[str], None] | None = None
        if world > 1 and dist is None:
            raise ValueError("more than one rank needs a torch.distributed process group")

    def _twiddle_t(self, re, im, rows, cols, col0):
        """W_n^(r * (col0 + c)) on a row-major [rows][cols] block: the same call as `twiddle` (its roles are symmetric)"""
        self._twiddle(re, im, rows, cols, col0)

    # largest block one collective call hands to one peer; larger blocks go in pieces.  RCCL's all_to_all_single was
    # measured to corrupt the result from 2 GiB per peer on (one f64 transform of 2^28 points on a one-rank group lost 93 %
    # of its energy, bench.py --dist-fft 28, round 4) -- 1 GiB pieces keep every count far below 2^31 bytes
    max_block_bytes = 1 << 30

    def _all_to_all(self, send):
        """rank q receives every rank's q-th equal chunk of `send`; returns the chunks in source-rank order"""
        import torch

        if self.dist is None:  # a single process without a process group
            return send
        # with a process group the exchange is a real collective even for one rank (world-size-1 "nccl" group: RCCL's
        # all_to_all_single on device tensors, the one-GPU hardware test of this path)
        recv = torch.empty_like(send)
        via_host = self.dist.get_backend() == "gloo" and send.is_cuda  # dry run on one GPU: through host memory
        src = send.cpu() if via_host else send
        dst = torch.empty(send.shape, dtype=send.dtype) if via_host else recv
        block = send.numel() // self.world
        pieces = -(-block * send.element_size() // self.max_block_bytes)
        if pieces <= 1:
            self.dist.all_to_all_single(dst, src)
        else:  # piece j of every peer's block per call
            step = -(-block // pieces)
            views = self.dist.get_backend() == "nccl"  # RCCL: lists of views, no staging copy; gloo has no list form
            for lo in range(0, block, step):
                hi = min(block, lo + step)
                if views:
                    self.dist.all_to_all([dst[q * block + lo:q * block + hi] for q in range(self.world)],
                                         [src[q * block + lo:q * block + hi] for q in range(self.world)])
                else:
                    out = torch.empty((self.world, hi - lo), dtype=src.dtype, device=src.device)
                    self.dist.all_to_all_single(out.view(-1), src.view(self.world, block)[:, lo:hi].contiguous().view(-1))
                    dst.view(self.world, block)[:, lo:hi] = out
        if via_host:
            recv.copy_(dst)
        return recv

    def run(self, reals, imags, reverse: bool = False):
        """In place on the rank's slab (1-D tensors of n/world elements)."""
        w, n1, n2 = self.world, self.n1, self.n2
        if reals.numel() != self.n // w or imags.numel() != self.n // w:
            raise ValueError("assertion `left == right` failed: reals.len() == imags.len()")  # dit.rs:284
        re, im = (imags, reals) if reverse else (reals, imags)  # the swap trick, algorithms/dit.rs:297-300
        r1, c2 = n1 // w, n2 // w

        def pack(x):       # [n1 mine][n2] -> [dest q][n1 mine][n2 in q's block]: the send order of exchange 1
            return x.view(r1, w, c2).permute(1, 0, 2).contiguous().view(-1) if w > 1 else x

        def unpack2(x):    # received [src s][k1 mine][n2 in s's block] -> [n2][k1 mine]
            return x.view(w, r1, c2).permute(0, 2, 1).contiguous().view(-1)

        def unpack3(x):    # received [src s][k2 mine][k1 in s's block] -> [k2 mine][k1]
            return x.view(w, c2, r1).permute(1, 0, 2).contiguous().view(-1) if w > 1 else x

        mark = self.on_stage or (lambda name: None)
        # exchange 1 -> [n1][n2 mine] (source-major = natural n1 order): column FFTs over n1
        p_re, p_im = pack(re), pack(im)
        mark("pack1")
        a_re, a_im = self._all_to_all(p_re), self._all_to_all(p_im)
        del p_re, p_im
        if self.dist is None:  # no exchange happened: a_re / a_im still alias the caller's slab
            a_re, a_im = a_re.clone(), a_im.clone()
        mark("exchange1")
        self._fft(a_re, a_im, n1, c2)
        mark("fft_n1")
        # The inter-factor twiddle W_N^(k1 * n2) commutes with the exchange (it is element-wise): it is applied on the
        # other side, fused into the first load of the column FFTs over n2 (rows n2, columns k1 = rank*r1 + c) where
        # the kernels can do that, else as its own sweep over [k1][n2 mine] before the exchange.
        fused = self._fft_tw is not None
        if not fused:
            self._twiddle(a_re, a_im, n1, c2, self.rank * c2)
            mark("twiddle")
        # exchange 2: the block for rank q is the rows k1 of q's range -- contiguous as it stands
        x_re, x_im = self._all_to_all(a_re), self._all_to_all(a_im)
        mark("exchange2")
        b_re, b_im = unpack2(x_re), unpack2(x_im)
        del a_re, a_im, x_re, x_im
        mark("unpack2")
        if not fused or not self._fft_tw(b_re, b_im, n2, r1, self.rank * r1):   # -> [k2][k1 mine]
            if fused:  # the fused form declined this shape: W_N^(n2 * k1) on [n2][k1 mine], then the plain transform
                self._twiddle_t(b_re, b_im, n2, r1, self.rank * r1)
            self._fft(b_re, b_im, n2, r1)
        mark("fft_n2")
        # exchange 3: rows k2 of q's range, contiguous; unpack to natural order
        y_re, y_im = self._all_to_all(b_re), self._all_to_all(b_im)
        mark("exchange3")
        c_re, c_im = unpack3(y_re), unpack3(y_im)
        del b_re, b_im, y_re, y_im
        mark("unpack3")
        if reverse:
            scale = 1.0 / self.n  # algorithms/dit.rs:325-331
            c_re *= scale
            c_im *= scale
        re.copy_(c_re)
        im.copy_(c_im)
        mark("store")


def gpu_transform(n: int, rank: int, world: int, dist=None, dtype: str = "f64") -> DistributedFft:
    """The GPU instance: local stages = the library's strided-batch kernels and the TwiddleGrid kernel."""
    import phastft_amd as P

    f64 = dtype == "f64"
    n1, n2 = split_factors(n.bit_length() - 1, world)
    planners = {m: (P.PlannerDit64 if f64 else P.PlannerDit32)(m) for m in {n1, n2}}
    grid = (P.TwiddleGrid64 if f64 else P.TwiddleGrid32)(n)

    def column_fft(re, im, length, count):
        try:
            P.fft_dit_strided(re, im, length, P.Direction.Forward, planners[length], batch=count, stride=count)
        except P.PhastPanic as e:
            if e.code != P.ERR_INVALID_ARG:  # only "this shape is not covered" falls back; real failures propagate
                raise
            # shapes the strided kernels do not cover (fewer than 64 points, or too few columns for a tile row):
            # transpose, contiguous batch, transpose back
            t_re = re.view(length, count).t().contiguous().view(-1)
            t_im = im.view(length, count).t().contiguous().view(-1)
            P.fft_dit_batched(t_re, t_im, length, P.Direction.Forward, planners[length])
            re.copy_(t_re.view(count, length).t().contiguous().view(-1))
            im.copy_(t_im.view(count, length).t().contiguous().view(-1))

    def twiddle(re, im, rows, cols, col0):
        grid.apply(re, im, rows, cols, row0=0, col0=col0)

    def column_fft_tw(re, im, length, count, col0):
        try:
            P.fft_dit_strided(re, im, length, P.Direction.Forward, planners[length], batch=count, stride=count,
                              twiddle_n=n, twiddle_col0=col0)
            return True
        except P.PhastPanic as e:
            if e.code != P.ERR_INVALID_ARG:
                raise
            return False

    t = DistributedFft(n, rank, world, column_fft, twiddle, dist, column_fft_tw)
    t._keep = (planners, grid)
    return t
