#!/usr/bin/env python3
"""bench.py -- detect + describe + match throughput of the HIP AKAZE path on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload = BASELINE.json configs[1]: 1920x1080 grayscale pairs, 4 octaves x 4 sublevels, PM_G2, MLDB,
max 10000 points (main.cpp:156-166).  One "step" = one pass of the hot path over one batch of
--pairs synthetic pairs per GPU: Akazer::detectAndCompute on both images of every pair + cuMatch,
ending with host-visible keypoints / descriptors / matches (the reference's timed region,
main.cpp:199-209, with its D2H copies).  Inputs are resident in HBM before the timer starts.
Frames shard by independent pair across ranks (weak scaling, no data-path collective); RCCL is used
only for the barrier, the max-over-ranks time and the trivial result-summary gather.

Extra objects: "roofline" for the dominant kernel (the FED step, 12 B/px/step algorithmic)
measured with HIP events on the launch stream inside this run; "cpu_baseline" = the CPU oracle
(kind "port") timed on this box's host cores on a bounded sample (rank 0, N == 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _sub in ("cuda-akaze_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, _sub))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def shard_pairs(total_pairs, world, rank):
    """contiguous block partition of independent pairs over ranks"""
    per = (total_pairs + world - 1) // world
    lo = min(rank * per, total_pairs)
    return lo, min(lo + per, total_pairs)


def cpu_baseline(w, h, p, u8_pairs, budget_s=20.0):
    """the oracle (CPU port of the reference algorithm, OpenMP) on the same 1080p pairs"""
    # the GPU box gives one GPU a 16-core CPU share; an OpenMP team per hardware thread (256 here)
    # only oversubscribes it.  Must be set before libgomp initialises (first oracle call).
    cores = min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    import okz
    from akaze_hip import synth
    okz.build()
    imgs = [(synth.to_float(a, p), synth.to_float(b, p)) for a, b in u8_pairs]
    n, t0 = 0, time.time()
    # warm-up pair (page-in, OpenMP pool)
    okz.detect_and_compute(imgs[0][0], w)
    t0 = time.time()
    while True:
        a, b = imgs[n % len(imgs)]
        r1 = okz.detect_and_compute(a, w)
        r2 = okz.detect_and_compute(b, w)
        okz.match(r1.points, r2.points)
        n += 1
        el = time.time() - t0
        if el > budget_s or n >= 40:
            break
    return {"value": round(n / el, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 1920x1080 pairs (detect+describe both images + match), {el:.1f} s, OpenMP oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU per step (batch); halved until the arenas fit the free HBM")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one context: no overlap between consecutive steps")
    ap.add_argument("--pipeline", type=int, default=2, help="contexts used round-robin (batches in flight)")
    ap.add_argument("--upload", action="store_true",
                    help="also measure the upload-inclusive rate: uint8 host images -> H2D -> on-device ingest -> path")
    ap.add_argument("--fast", action="store_true",
                    help="also time the integer FAST path (fastDetectAndCompute, uint8 inputs) on the same pairs")
    ap.add_argument("--upright", action="store_true", help="MLDB-upright (skip the orientation stage; configs[2] of BASELINE.json)")
    ap.add_argument("--serial", action="store_true",
                    help="run the timed region on one stream too (default: octaves on concurrent streams)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    # HAK_BENCH_FORCE_DIST=1 exercises the RCCL code path (barrier / MAX all-reduce / summary all-gather) with one rank too
    use_dist = world > 1 or os.environ.get("HAK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import akaze_hip as ah
    from akaze_hip import synth
    ah.check(ah.lib.hak_set_device(local_rank))

    w, h = args.width, args.height
    p = ah.iAlignUp(w, 128)
    B = args.pairs
    max_pts = 10000
    # arena ~ 19 float planes per octave pyramid + key map + candidate list ~ 31 B/px x 4/3 per image and context
    # (250 MB at 1080p): 128 pairs x 2 contexts = 127 GB of the 288 GB.  Batch size is a batching choice, not part of the workload.
    free_b = torch.cuda.mem_get_info()[0]
    while B > 16 and 2 * B * (w * h * 125 + 2 * max_pts * 104) * (1 if args.no_pipeline else max(1, args.pipeline)) > 0.8 * free_b:
        B //= 2
    nimg = 2 * B

    # ---- synthetic inputs, resident in HBM (2 distinct seeded pairs per rank, cycled over the batch)
    u8_pairs = [synth.pair(w, h, 1 + 2 * rank + i) for i in range(2)]
    host = np.stack([synth.to_float(u8_pairs[(i // 2) % 2][i % 2], p) for i in range(nimg)])
    d_imgs = torch.from_numpy(host).cuda()
    del host

    # two contexts, used alternately: while batch i runs on the GPU, batch i-1 is synchronised and downloaded
    # (every step's work still completes inside the timed bracket)
    NCTX = 1 if args.no_pipeline else max(1, args.pipeline)
    dets, d_pts_l, d_num_l, h_pts_l, h_num_l = [], [], [], [], []
    stream = torch.cuda.current_stream()
    for k in range(NCTX):
        dk = ah.Akazer()
        dk.init((w, h, p), noctaves=args.octaves, max_pts=max_pts, batch=nimg, upright=bool(args.upright))
        if k == 0:
            ah.check(ah.lib.hak_set_stream(dk.ctx, C.c_void_p(stream.cuda_stream)))
        if args.serial:
            ah.check(ah.lib.hak_set_concurrency(dk.ctx, 0))
        dets.append(dk)
        d_pts_l.append(torch.zeros(nimg * max_pts * 104, dtype=torch.uint8, device="cuda"))
        d_num_l.append(torch.zeros(nimg, dtype=torch.int32, device="cuda"))
        hp, hn = C.c_void_p(), C.c_void_p()
        ah.check(ah.lib.hak_host_alloc(C.byref(hp), nimg * max_pts * 104))
        ah.check(ah.lib.hak_host_alloc(C.byref(hn), nimg * 4))
        h_pts_l.append(hp)
        h_num_l.append(hn)
    det, d_pts, d_num, h_pts, h_num = dets[0], d_pts_l[0], d_num_l[0], h_pts_l[0], h_num_l[0]

    def enqueue(k):
        ah.check(ah.lib.hak_detect_and_compute_batch(dets[k].ctx, d_imgs.data_ptr(), h * p, p, nimg,
                                                     d_pts_l[k].data_ptr(), d_num_l[k].data_ptr(), 1))
        ah.check(ah.lib.hak_match_batch(dets[k].ctx, d_pts_l[k].data_ptr(), d_num_l[k].data_ptr(), B))

    def download(k):
        ah.check(ah.lib.hak_download_batch(dets[k].ctx, d_pts_l[k].data_ptr(), d_num_l[k].data_ptr(), nimg,
                                           h_pts_l[k], h_num_l[k]))

    def run_steps(n):
        """n steps = n batches through detect + describe + match + download"""
        for i in range(n):
            enqueue(i % NCTX)
            if i >= NCTX - 1:
                download((i - (NCTX - 1)) % NCTX)
        for i in range(max(0, n - (NCTX - 1)), n):          # drain the batches still in flight
            download(i % NCTX)

    def step():
        enqueue(0)
        download(0)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- optional: PCIe-inclusive rate (never `value`): pinned uint8 batch -> H2D -> hak_ingest_u8 -> same steps
    upload_rate = None
    if args.upload:
        h_u8 = torch.from_numpy(np.stack([u8_pairs[(i // 2) % 2][i % 2] for i in range(nimg)])).pin_memory()
        d_u8 = torch.empty_like(h_u8, device="cuda")

        def up_step(k):
            d_u8.copy_(h_u8, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            ah.check(ah.lib.hak_ingest_u8(dets[k].ctx, d_u8.data_ptr(), h * w, w, d_imgs.data_ptr(), h * p, p, w, h, nimg))
            enqueue(k)
        for i in range(2):
            up_step(i % NCTX); download(i % NCTX)
        fence()
        tu = time.perf_counter()
        for i in range(args.steps):
            up_step(i % NCTX)
            if i >= NCTX - 1:
                download((i - (NCTX - 1)) % NCTX)
        for i in range(max(0, args.steps - (NCTX - 1)), args.steps):
            download(i % NCTX)
        fence()
        upload_rate = world * B * args.steps / (time.perf_counter() - tu)

    # ---- optional: the integer FAST path on the same pairs (secondary figure, never `value`)
    fast_rate = None
    if args.fast:
        d_fu8 = torch.from_numpy(np.stack([np.pad(u8_pairs[(i // 2) % 2][i % 2], ((0, 0), (0, p - w))) for i in range(nimg)])).cuda()

        def fast_enqueue(k):
            ah.check(ah.lib.hak_fast_detect_and_compute_batch(dets[k].ctx, d_fu8.data_ptr(), h * p, p, nimg,
                                                              d_pts_l[k].data_ptr(), d_num_l[k].data_ptr(), 1))
            ah.check(ah.lib.hak_match_batch(dets[k].ctx, d_pts_l[k].data_ptr(), d_num_l[k].data_ptr(), B))
        for i in range(2):
            fast_enqueue(i % NCTX); download(i % NCTX)
        fence()
        tf = time.perf_counter()
        for i in range(args.steps):
            fast_enqueue(i % NCTX)
            if i >= NCTX - 1:
                download((i - (NCTX - 1)) % NCTX)
        for i in range(max(0, args.steps - (NCTX - 1)), args.steps):
            download(i % NCTX)
        fence()
        fast_rate = world * B * args.steps / (time.perf_counter() - tf)
        enqueue(0); download(0)                      # leave the float results in the host buffers for the summary
        fence()

    counts = np.ctypeslib.as_array(C.cast(h_num, C.POINTER(C.c_int)), shape=(nimg,)).copy()
    pts = np.ctypeslib.as_array(C.cast(h_pts, C.POINTER(C.c_uint8)), shape=(nimg * max_pts * 104,)).view(ah.POINT_DTYPE).reshape(nimg, max_pts)
    nmatch = int(sum((pts[2 * k, :counts[2 * k]]["match"] >= 0).sum() for k in range(B)))
    # ---- the trivial result gather: per-rank summary {pairs, keypoints, matches}
    summary = torch.tensor([B, int(counts.sum()), nmatch], dtype=torch.int64, device="cuda")
    if use_dist:
        allsum = [torch.zeros_like(summary) for _ in range(world)]
        dist.all_gather(allsum, summary)
        summary = torch.stack(allsum).sum(0)
    summary = summary.cpu().tolist()

    # ---- roofline leg: same steps with per-launch HIP events on the launch stream
    roof = None
    if not args.no_roofline:
        # per-kernel durations are only meaningful without overlap: the roofline leg runs the same steps
        # strictly serially on one stream (the timed region above overlaps the octaves on separate streams)
        ah.check(ah.lib.hak_set_concurrency(det.ctx, 0))
        ah.check(ah.lib.hak_prof_reset(det.ctx))
        ah.check(ah.lib.hak_prof_enable(det.ctx, 1))
        nprof = max(1, min(args.steps, 3))
        for _ in range(nprof):
            step()
        ms, n = C.c_double(), C.c_int()
        ah.check(ah.lib.hak_prof_read(det.ctx, ah.PROF["fed"], C.byref(ms), C.byref(n)))
        ah.check(ah.lib.hak_prof_enable(det.ctx, 0))
        tr = det.traffic(int(counts.mean()))
        bytes_per_launch = tr.fed_bytes * nimg / tr.fed_launches          # algorithmic bytes (hipakaze.h hak_traffic) x images / launches
        avg_s = ms.value * 1e-3 / n.value
        achieved = bytes_per_launch / avg_s / 1e9
        cls = {}
        for name, k in ah.PROF.items():
            m2, n2 = C.c_double(), C.c_int()
            ah.check(ah.lib.hak_prof_read(det.ctx, k, C.byref(m2), C.byref(n2)))
            cls[name] = round(m2.value / nprof, 4)
        # HBM bytes per FED launch from the PMC passes committed under profiles/ (tools/pmc_traffic.py:
        # separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH x2 correction); null when not measured
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tfile) and (w, h, args.octaves) == (1920, 1080, 4):
            # every FED launch covers the whole batch, so bytes scale with B relative to the batch the passes ran at
            tj = json.load(open(tfile))
            traffic = round(tj["fed_hbm_bytes_per_launch"] * B / float(tj.get("pairs_per_launch_sequence", 16)))
        roof = {"kernel": "k_fed_sf<NS> / k_fed_multi<NS> (FED steps 12 B/px/step; +16 B/px where the sublevel's low-pass and "
                          "conductivity run inside its first FED launch)", "bound": "hbm",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                # the same launches priced by their measured HBM traffic instead of the unfused byte model: frac > 1 above means
                # fusion removed traffic, this one says how close the remaining traffic runs to the HBM peak
                "traffic_GBs": None if traffic is None else round(traffic / avg_s / 1e9, 1),
                "traffic_frac": None if traffic is None else round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_launch": round(bytes_per_launch), "avg_launch_us": round(avg_s * 1e6, 3),
                "launches_per_step": tr.fed_launches, "ms_per_step_by_class": cls,
                # end-to-end achieved = all-stage algorithmic bytes per image x images/s of the timed region
                "end_to_end": {"all_stage_bytes_per_image": round(tr.all_stage_bytes),
                               "achieved_GBs": round(tr.all_stage_bytes * 2.0 * world * B * args.steps / elapsed / 1e9, 1),
                               "frac_of_peak": round(tr.all_stage_bytes * 2.0 * world * B * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
                "mode": "serial leg (one stream); rocprof counterpart: profiles/*_serial_kernel_stats.csv from `bench.py --serial`"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, h, p, u8_pairs)

    if rank == 0:
        total_pairs = world * B * args.steps
        out = {
            "metric": "pairs_per_sec_detect_describe_match_1080p", "value": round(total_pairs / elapsed, 2),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {w}x{h} grayscale pairs, {args.octaves} octaves x 4 sublevels, PM_G2, "
                                   "MLDB-486, max_pts 10000, float path; detect+describe both images + match, D2H included",
                       "pairs_per_step_per_gpu": B, "octave_streams": "serial" if args.serial else "concurrent", "step_pipeline": NCTX, "sharding": "independent pairs per rank, no data-path collective",
                       "keypoints_per_image": round(summary[1] / (2.0 * summary[0]), 1),
                       "matches_per_pair": round(summary[2] / float(summary[0]), 1)},
            "roofline": roof, "cpu_baseline": cpu,
            "upload_inclusive_pairs_per_s": None if upload_rate is None else round(upload_rate, 1),
            "fast_path_pairs_per_s": None if fast_rate is None else round(fast_rate, 1),
        }
        print(json.dumps(out))
    for k in range(NCTX):
        ah.lib.hak_host_free(h_pts_l[k])
        ah.lib.hak_host_free(h_num_l[k])
        dets[k].close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
