"""bench.py as the driver runs it, in a fresh child process on the GPU box: the JSON contract, the self-verification
against the oracle, the per-class roofline table, and the torch.distributed (RCCL) code path with one rank
(HAK_BENCH_FORCE_DIST=1: init_process_group("nccl"), barrier, MAX all-reduce, summary all-gather on the MI355X)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("HAK_HESS_STREAM", "HAK_FUSE_SF", "HAK_BASE_STREAM"):      # the bench runs the library's default kernel selection
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_contract_verified_and_roofline():
    out = run_bench(["--steps", "2", "--warmup", "1", "--pairs", "16", "--no-cpu-baseline", "--no-configs"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "gather"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["scaling"] == "weak" and out["dtype"] == "f32"
    assert out["value"] > 100 and abs(out["value"] - 16 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 0.01
    v = out["verified"]
    # the oracle on all 16 distinct pairs; every slot of both contexts' last timed downloads; the FAST leg the same way
    assert v["images"] == 32 and v["points_equal"] and v["matches_equal"]
    assert v["slots"] == 32 and v["contexts"] == 2 and v["slots_equal"] and v["gathered_pairs"] == 16 and v["gathered_pairs_equal"]
    assert v["fast"]["slots"] == 32 and v["fast"]["slots_equal"]
    g = out["gather"]
    assert g["pairs"] == 16 and g["complete"] and g["equal_seed_equal_checksum"] and g["equals_g1_table"] is True
    # the default N = 1 run measures the upload-inclusive and the FAST rates too
    assert out["upload_inclusive_pairs_per_s"] > 100 and out["fast_path_pairs_per_s"] > 100
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 3000 < r["copy_ceiling_GBs"] < 8000
    # the HBM traffic of the FED family is measured inside the run: two child runs under rocprofv3 --pmc (bench.run_pmc_legs)
    assert r["traffic"] and "THIS run" in r["traffic_source"], r["traffic_source"]
    assert 1.0 <= r["traffic"] / r["bytes_per_launch"] <= 1.3 and 0.2 < r["traffic_frac"] <= 1.0     # (16-pair sequences: more halo per byte than the 256-pair headline)
    assert r["fusion_gain"] > 1.0
    names = [c["class"] for c in r["classes"]]
    assert names[:4] == ["fed", "hessian", "describe", "prologue"]
    for c in r["classes"]:
        assert c["ms"] > 0
        if c["frac_peak"] is not None:
            assert 0.0 < c["frac_peak"] <= 1.0, c
    mm = [c for c in r["classes"] if c["class"] == "match"][0]       # the one matrix-core class: priced against the dense fp4 peak
    assert mm["bound"] == "mfma" and mm["matrix_peak_TOPs"] == 10000.0 and 0.0 < mm["frac_matrix_peak"] < 1.0
    assert abs(mm["matrix_TOPs"] - mm["matrix_ops"] / (mm["ms"] * 1e-3) / 1e12) <= 0.06 * mm["matrix_TOPs"] + 0.1


def test_bench_rccl_path_with_one_rank_weak_and_strong():
    """barrier / all_reduce(MAX) / all_gather over RCCL on the MI355X (world size 1), in both scaling modes"""
    env = {"HAK_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29611"}
    weak = run_bench(["--steps", "2", "--warmup", "1", "--pairs", "16", "--no-cpu-baseline", "--no-configs", "--no-roofline", "--no-fast",
                      "--no-upload"], env)
    assert weak["scaling"] == "weak" and weak["config"]["total_pairs_per_step"] == 16 and weak["verified"]["slots_equal"]
    assert weak["upload_inclusive_pairs_per_s"] is None and weak["fast_path_pairs_per_s"] is None
    env["MASTER_PORT"] = "29612"
    # configs[3]'s shape through its shorthand, with fewer pairs: 40 pairs of 1280x720 = launch sequences of 16 + 16 + 8
    strong = run_bench(["--config", "3", "--steps", "2", "--warmup", "1", "--pairs", "16", "--total-pairs", "40",
                        "--no-cpu-baseline", "--no-configs", "--no-roofline"], env)
    assert strong["scaling"] == "strong" and strong["config"]["total_pairs_per_step"] == 40
    assert strong["metric"].endswith("720p") and strong["config"]["pairs_per_launch_sequence"] == 16
    assert abs(strong["value"] - 40 * 2 / (strong["ms_per_step"] * 2e-3)) / strong["value"] < 0.01
    assert strong["verified"]["slots_equal"] and strong["verified"]["gathered_pairs"] == 40 and strong["verified"]["gathered_pairs_equal"]
    g = strong["gather"]               # all 40 pairs of the step, each once, equal to the committed 720p table of one GPU
    assert g["pairs"] == 40 and g["complete"] and g["equal_seed_equal_checksum"] and g["equals_g1_table"] is True
    assert strong["upload_inclusive_pairs_per_s"] > 0


def test_bench_through_its_own_rank_launcher():
    """`--launch`: bench.py starts torch.distributed.run itself (the N > 1 path of a plain `python bench.py --gpus N`) with one
    rank; the child runs over RCCL and the parent relays its line"""
    out = run_bench(["--gpus", "1", "--launch", "--steps", "2", "--warmup", "1", "--pairs", "16", "--no-cpu-baseline", "--no-configs",
                     "--no-roofline", "--no-fast", "--no-upload"], {"MASTER_PORT": "29613", "HAK_BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["verified"]["slots_equal"]


def test_bench_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2 and f"{n + 1} ranks requested, {n} device" in r.stderr


def _run_two_ranks(extra_env, port):
    """torch.distributed.run with two ranks on the one MI355X of the GPU box (HAK_BENCH_DIST_BACKEND=gloo: both ranks compute on
    cuda:0, the collectives run over gloo): the launcher is started before anything touches the GPU in this child tree"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HAK_HESS_STREAM", "HAK_FUSE_SF", "HAK_BASE_STREAM")}
    env.update({"HAK_BENCH_DIST_BACKEND": "gloo"})
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "3", "--total-pairs", "37", "--pairs", "16", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-configs", "--no-roofline"]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)


def test_two_ranks_end_to_end_on_one_gpu_unequal_shards():
    """world = 2 on real results: 37 pairs of 1280x720 = shards of 19 (rank 0, lo = 0) and 18 (rank 1, lo = 19 -- not a multiple of the
    16 seeds, so seed_of, the slot digests with lo != 0 and the unequal last chunks (16 + 3 and 16 + 2 pairs) all run); rank 0 checks the gathered table against
    the oracle and the committed G = 1 table"""
    r = _run_two_ranks({}, 29621)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "rehearsal" in out and out["rccl_ranks"] == 0
    assert out["config"]["total_pairs_per_step"] == 37 and out["config"]["pairs_per_step_per_gpu"] == 19
    assert abs(out["value"] - 37 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 0.01
    g = out["gather"]
    assert g["pairs"] == 37 and g["complete"] and g["equal_seed_equal_checksum"] and g["equals_g1_table"] is True and g["distinct_seeds"] == 16
    v = out["verified"]
    assert v["slots_equal"] and v["gathered_pairs"] == 37 and v["gathered_pairs_equal"]
    assert v["fast"] is None or v["fast"]["slots_equal"]


def test_two_ranks_a_corrupted_pair_on_rank_1_is_caught():
    """the same run with one descriptor bit of rank 1's first pair flipped after its download: rank 0 must refuse (exit code 3)"""
    r = _run_two_ranks({"HAK_BENCH_CORRUPT_RANK": "1"}, 29622)
    assert r.returncode != 0 and "results differ" in r.stderr, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["gather"]["complete"] and (not out["gather"]["equal_seed_equal_checksum"] or out["gather"]["equals_g1_table"] is False)
    assert out["verified"]["gathered_pairs_equal"] is False and out["verified"]["slots_equal"]      # rank 0's own slots are fine
