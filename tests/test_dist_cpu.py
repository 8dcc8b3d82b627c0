"""N > 1 path on CPU: the frame sharding + result-summary gather + max-over-ranks timing that
bench.py runs over RCCL, exercised here with world_size 2 on the gloo backend."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, total_pairs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_pairs(total_pairs, world, rank)
    # the functions bench.main() itself runs over RCCL: per-rank summary {pairs, keypoints, matches} (fake but
    # rank-dependent counts), barrier, max-over-ranks step time
    total = bench.gather_summary(hi - lo, sum(range(lo, hi)) * 2, sum(range(lo, hi)), "cpu", True)
    dist.barrier()
    tmax = bench.max_over_ranks(0.5 + rank, "cpu", True)
    out[rank] = (lo, hi, total, tmax)
    dist.destroy_process_group()


@pytest.mark.parametrize("total_pairs", [512, 7, 1])
def test_shard_and_gather_world2(total_pairs):
    world = 2
    port = 29500 + (os.getpid() + total_pairs) % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, total_pairs, out), nprocs=world, join=True)
    covered = []
    for r in range(world):
        lo, hi, total, tmax = out[r]
        covered += list(range(lo, hi))
        assert total == [total_pairs, sum(range(total_pairs)) * 2, sum(range(total_pairs))]
        assert tmax == 0.5 + world - 1                    # MAX over ranks
    assert covered == list(range(total_pairs))            # every pair exactly once, contiguous blocks


def test_shard_pairs_properties():
    sys.path.insert(0, ROOT)
    import bench
    for total in (0, 1, 5, 64, 512, 513):
        for world in (1, 2, 4, 8):
            parts = [bench.shard_pairs(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
