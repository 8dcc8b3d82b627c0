"""N > 1 path on CPU: the frame sharding + result-summary gather + max-over-ranks timing that
bench.py runs over RCCL, exercised here with world_size 2 on the gloo backend."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fake_row(g, corrupt=False):
    """a per-pair summary that depends on the pair's SEED only (as real results do): {pair_id, n1 << 32 | n2, matches, checksum}"""
    import bench
    sd = g % bench.NDIST
    return [g, ((2000 + sd) << 32) | (1900 + sd), 1500 + sd, (0x1234567 * (sd + 1)) ^ (1 if corrupt else 0)]


def _worker(rank, world, port, total_pairs, corrupt_pair, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_pairs(total_pairs, world, rank)
    # the functions bench.main() itself runs over RCCL: the all-gather of the per-pair 32-byte summaries (ranks hold unequal
    # numbers of pairs when world does not divide total_pairs), barrier, max-over-ranks step time
    local = np.array([_fake_row(g, g == corrupt_pair) for g in range(lo, hi)], np.int64).reshape(-1, 4)
    table = bench.gather_pair_summaries(local, "cpu", True)
    dist.barrier()
    tmax = bench.max_over_ranks(0.5 + rank, "cpu", True)
    golden = {str(sd): _fake_row(sd)[1:] for sd in range(bench.NDIST)}
    out[rank] = (lo, hi, table.tolist(), tmax, bench.check_pair_table(table, total_pairs, golden))
    dist.destroy_process_group()


@pytest.mark.parametrize("total_pairs,corrupt_pair", [(512, -1), (7, -1), (1, -1), (37, 20)])
def test_shard_and_gather_world2(total_pairs, corrupt_pair):
    world = 2
    port = 29500 + (os.getpid() + total_pairs) % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, total_pairs, corrupt_pair, out), nprocs=world, join=True)
    covered = []
    for r in range(world):
        lo, hi, table, tmax, chk = out[r]
        covered += list(range(lo, hi))
        # every rank ends up with the table of ALL pairs, sorted by pair id, padding rows dropped
        assert [row[0] for row in table] == list(range(total_pairs))
        assert table == [_fake_row(g, g == corrupt_pair) for g in range(total_pairs)]
        assert tmax == 0.5 + world - 1                    # MAX over ranks
        assert chk["pairs"] == total_pairs and chk["complete"] and chk["bytes_per_pair"] == 32
        assert chk["matches"] == sum(_fake_row(g)[2] for g in range(total_pairs))
        if corrupt_pair < 0:
            assert chk["equal_seed_equal_checksum"] and chk["equals_g1_table"] is True
        else:       # one rank's result for one pair differs from the other pairs of its seed: the gather must say so
            assert not chk["equal_seed_equal_checksum"] and chk["equals_g1_table"] is None
    assert covered == list(range(total_pairs))            # every pair exactly once, contiguous blocks


def test_pair_table_against_the_g1_table():
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    tab = np.array([_fake_row(g) for g in range(16)], np.int64)
    golden = {str(sd): _fake_row(sd)[1:] for sd in range(bench.NDIST)}
    assert bench.check_pair_table(tab, 16, golden)["equals_g1_table"] is True
    golden["3"][2] ^= 1                                    # consistent among ranks, but not what one GPU gives
    chk = bench.check_pair_table(tab, 16, golden)
    assert chk["equal_seed_equal_checksum"] and chk["equals_g1_table"] is False
    assert not bench.check_pair_table(tab[:-1], 16, None)["complete"]
    # the committed table (written by the ORACLE, tests/golden/make_golden.py bench_checksums) covers both bench workloads
    import json
    g = json.load(open(bench.CHECKSUM_FILE))
    assert set(g) == {"1920x1080", "1280x720"} and all(len(v) == bench.NDIST for v in g.values())


def test_pair_digest_covers_every_record_field():
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    import okz
    a = np.zeros(5, okz.POINT_DTYPE)
    b = np.zeros(3, okz.POINT_DTYPE)
    base = bench.pair_digest(a, b)
    for f in bench.KP_FIELDS + bench.MATCH_FIELDS:
        c = a.copy()
        c[f][2] = 1
        assert bench.pair_digest(c, b) != base, f
    for f in bench.KP_FIELDS:
        c = b.copy()
        c[f][1] = 1
        assert bench.pair_digest(a, c) != base, f
    assert bench.pair_digest(a[:4], b) != base and -2**63 <= base < 2**63


def test_device_count_without_hip(monkeypatch):
    """the rank launcher's parent counts devices from sysfs (KFD topology) and honours the *_VISIBLE_DEVICES masks; it must not
    import a HIP runtime for that"""
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpu_count()
    assert n == torch.cuda.device_count()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0


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
    """started by a launcher with the wrong WORLD_SIZE the rank refuses instead of silently using it; WITHOUT --gpus the launcher's
    world size is taken (torchrun --nproc-per-node=N bench.py): here that gets as far as "needs a HIP device" """
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
    if torch.cuda.device_count() == 0:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-configs"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "WORLD_SIZE=3" not in r.stderr and "needs a HIP device" in r.stderr
