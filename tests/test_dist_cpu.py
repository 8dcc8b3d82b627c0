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


def test_launcher_command_for_two_ranks():
    """`python bench.py --gpus 2 ...` without a launcher starts exactly this (one rank per GPU, rendezvous on 127.0.0.1)"""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(["--gpus", "2", "--steps", "3"], 2, 29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"]


def test_more_ranks_than_devices_fails_loudly():
    """no GPU in this container: `--gpus 2` must refuse before any rank starts (exit 2, a message naming both numbers)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    ndev = torch.cuda.device_count()
    if ndev >= 2:
        pytest.skip("this box has two devices")
    assert r.returncode == 2 and f"2 ranks requested, {ndev} device" in r.stderr


def test_rank_count_must_match_gpus_flag():
    """started by a launcher with the wrong WORLD_SIZE the rank refuses instead of silently using it"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
