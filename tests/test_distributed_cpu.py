"""world_size-2 gloo tests of the N>1 host path (phastft_amd/sharding.py): shard bounds, per-rank
transform ids, the digest all-gather and the max-over-ranks reduction.  The compute is injected: here
the CPU oracle stands in for the HIP path (tests may use the oracle; the product never does)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from phastft_amd.sharding import ShardedBatch, max_over_ranks, shard_bounds


def test_shard_bounds_partition():
    for total in (0, 1, 7, 8, 1000, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == total
            nxt = 0
            for first, count in spans:
                assert first == nxt
                nxt += count
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_bounds(8192, 8, 3) == (3072, 1024)  # BASELINE configs[4]
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O

    store = {}

    def transform(first, count):
        for t in range(first, first + count):
            re, im = store.get(t) or O.fill(n, np.float64, seed=0xCAFE, transform_id=t)
            O.fft_64_dit(re, im, O.FORWARD)
            store[t] = (re, im)

    def digest(first, count):
        rows = []
        for t in range(first, first + count):
            re, im = store[t]
            rows.append([re.sum(), im.sum(), float(np.sum(re * re + im * im)), re[1]])
        return torch.tensor(rows, dtype=torch.float64).reshape(count, 4)

    sb = ShardedBatch(total, n, rank, world, transform, digest)
    sb.step()
    gathered = sb.gather_digests(dist)
    slow = max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "digests.npy"), gathered.numpy())
        np.save(os.path.join(out_dir, "slow.npy"), np.array([slow, sb.samples_per_step()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_batch_gloo(tmp_path):
    total, n, world = 5, 256, 2  # ragged: 3 + 2
    mp.spawn(_worker, args=(world, _free_port(), total, n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "digests.npy")
    slow, samples = np.load(tmp_path / "slow.npy")
    assert got.shape == (total, 4) and slow == 2.0 and samples == total * n
    from oracle import oracle as O

    for t in range(total):  # every transform id was produced exactly once, in order, by the right rank
        re, im = O.fill(n, np.float64, seed=0xCAFE, transform_id=t)
        e_in = np.sum(re * re + im * im)
        O.fft_64_dit(re, im, O.FORWARD)
        assert np.allclose(got[t], [re.sum(), im.sum(), np.sum(re * re + im * im), re[1]], rtol=1e-12, atol=1e-9)
        assert abs(got[t][2] / (n * e_in) - 1) < 1e-12  # Parseval ties the digest to the input


# ---------------------------------------------------------------- one transform over several ranks (f-3)
def _dist_fft_worker(rank, world, port, log_n, reverse, out_dir, fused=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from phastft_amd.distributed import DistributedFft

    n = 1 << log_n

    def column_fft(re, im, length, count):  # the oracle stands in for the strided-batch HIP kernels
        r, m = re.numpy().reshape(length, count), im.numpy().reshape(length, count)
        for c in range(count):
            x, y = np.ascontiguousarray(r[:, c]), np.ascontiguousarray(m[:, c])
            O.fft_64_dit(x, y, O.FORWARD)
            r[:, c], m[:, c] = x, y

    def twiddle(re, im, rows, cols, col0):
        r = torch.arange(rows, dtype=torch.int64).view(rows, 1)
        c = torch.arange(col0, col0 + cols, dtype=torch.int64).view(1, cols)
        ang = ((r * c) % n).to(torch.float64) * (-2.0 * np.pi / n)
        wr, wi = torch.cos(ang).view(-1), torch.sin(ang).view(-1)
        x, y = re.clone(), im.clone()
        re.copy_(x * wr - y * wi)
        im.copy_(x * wi + y * wr)

    full_re, full_im = O.fill(n, np.float64, seed=0xD157, transform_id=log_n)
    lo, hi = rank * n // world, (rank + 1) * n // world
    re, im = torch.from_numpy(full_re[lo:hi].copy()), torch.from_numpy(full_im[lo:hi].copy())
    def column_fft_tw(re, im, length, count, col0):  # fused form: declines odd-numbered problem sizes (fallback path)
        if log_n % 2:
            return False
        twiddle(re, im, length, count, col0)
        column_fft(re, im, length, count)
        return True

    t = DistributedFft(n, rank, world, column_fft, twiddle, dist, column_fft_tw if fused else None)
    if os.environ.get("PHAST_TEST_SMALL_BLOCKS"):  # force the chunked exchange (blocks above max_block_bytes go in pieces)
        t.max_block_bytes = 96
    t.run(re, im, reverse=reverse)
    np.save(os.path.join(out_dir, f"re{rank}.npy"), re.numpy())
    np.save(os.path.join(out_dir, f"im{rank}.npy"), im.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,reverse,fused", [(2, 10, False, False), (4, 11, False, True), (2, 13, True, False),
                                                       (4, 8, True, True), (2, 12, False, True)])
def test_one_transform_over_ranks_gloo(tmp_path, world, log_n, reverse, fused):
    """Block-distributed natural order in, natural order out, three all-to-alls: the concatenated slabs must be the
    transform of the concatenated input (the oracle run on one rank), forward and reverse."""
    from oracle import oracle as O

    mp.spawn(_dist_fft_worker, args=(world, _free_port(), log_n, reverse, str(tmp_path), fused), nprocs=world, join=True)
    n = 1 << log_n
    got_re = np.concatenate([np.load(tmp_path / f"re{r}.npy") for r in range(world)])
    got_im = np.concatenate([np.load(tmp_path / f"im{r}.npy") for r in range(world)])
    re, im = O.fill(n, np.float64, seed=0xD157, transform_id=log_n)
    O.fft_64_dit(re, im, O.REVERSE if reverse else O.FORWARD)
    err = np.sqrt(np.sum((got_re - re) ** 2 + (got_im - im) ** 2) / np.sum(re ** 2 + im ** 2))
    assert err < 1e-13, err


def test_one_transform_over_ranks_gloo_chunked_exchanges(tmp_path, monkeypatch):
    """Blocks above DistributedFft.max_block_bytes are exchanged in pieces (dist.all_to_all on views: RCCL corrupts
    all_to_all_single from 2 GiB per peer on -- found at 2^28 points on a one-rank group in round 4).  Forced here with a
    96-byte limit: 2^12 points over 2 ranks = 8 KiB per peer and plane in 86 pieces, ragged last piece; same result."""
    from oracle import oracle as O

    monkeypatch.setenv("PHAST_TEST_SMALL_BLOCKS", "1")
    world, log_n = 2, 12
    mp.spawn(_dist_fft_worker, args=(world, _free_port(), log_n, False, str(tmp_path), True), nprocs=world, join=True)
    n = 1 << log_n
    got_re = np.concatenate([np.load(tmp_path / f"re{r}.npy") for r in range(world)])
    got_im = np.concatenate([np.load(tmp_path / f"im{r}.npy") for r in range(world)])
    re, im = O.fill(n, np.float64, seed=0xD157, transform_id=log_n)
    O.fft_64_dit(re, im, O.FORWARD)
    assert np.sqrt(np.sum((got_re - re) ** 2 + (got_im - im) ** 2) / np.sum(re ** 2 + im ** 2)) < 1e-13


def test_split_factors():
    from phastft_amd.distributed import split_factors

    assert split_factors(28, 8) == (1 << 14, 1 << 14)
    assert split_factors(29, 8) == (1 << 15, 1 << 14)
    with pytest.raises(ValueError):
        split_factors(4, 8)   # N < ranks^2
    with pytest.raises(ValueError):
        split_factors(20, 3)
