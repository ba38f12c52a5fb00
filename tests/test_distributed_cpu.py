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
