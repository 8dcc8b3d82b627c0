#!/usr/bin/env python3
"""bench.py -- detect + describe + match throughput of the HIP AKAZE path on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Started by torch.distributed.run (the driver's way: WORLD_SIZE is set) the process is a rank; started
plainly (`python bench.py --gpus 8`) it launches the N ranks itself as a child process tree and relays rank 0's line and the
exit code; it exits 2 when the node has fewer devices than ranks were asked for.

Workload = BASELINE.json configs[1]: 1920x1080 grayscale pairs, 4 octaves x 4 sublevels, PM_G2, MLDB,
max 10000 points (main.cpp:156-166).  One "step" = one pass of the hot path over one batch of
--pairs synthetic pairs per GPU: Akazer::detectAndCompute on both images of every pair + cuMatch,
ending with host-visible keypoints / descriptors / matches (the reference's timed region,
main.cpp:199-209, with its D2H copies).  Inputs are resident in HBM before the timer starts.
Frames shard by independent pair across ranks (no data-path collective); RCCL is used only for the
barrier, the max-over-ranks time and the trivial result-summary gather.
  default          weak scaling: --pairs per GPU per step
  --total-pairs T  strong scaling: T pairs per step in total, rank r takes shard_pairs(T, N, r)
                   ("pairs/s at G = 1, 2, 4, 8 on the same 512 pairs", BASELINE configs[3])

Extra objects in the line:
  "verified"      the CPU oracle runs on ALL distinct pairs of the batch (16 pairs = 32 images); every slot of the last download of
                  every pipeline context, and every pair of an untimed summary pass, is digested (sha256 over the record fields)
                  and must equal the oracle's digest of its seed; same for the integer FAST leg (exit code 3 on a mismatch)
  "gather"        SURVEY 8e's verification gather: per-pair 32-byte summaries {pair_id, n1, n2, n_matches, 64-bit checksum}
                  all-gathered over RCCL; rank 0 checks equal seeds => equal checksums across ranks and against the G = 1 table
                  committed under tests/golden/bench_pair_checksums.json
  "roofline"      the FED kernel family (dominant class): achieved = the compulsory HBM bytes of its launches AS BUILT
                  (fused: read L [+ g], write L' [+ smooth, g]) / average launch duration from HIP events on the launch
                  stream in a serial leg of this run; "traffic" = PMC bytes per launch from the rocprofv3 passes committed
                  under profiles/ (only when those passes were taken on the same kernel sources, else null);
                  "fusion_gain" = the unfused 12 B/px/step model / the fused bytes; "classes" = the same for every kernel
                  class; "copy_ceiling_GBs" = a float4 HIP copy kernel timed in this run
  "cpu_baseline"  the CPU oracle ("port") on this box's host cores, median over a bounded sample (rank 0, N == 1)
  "configs"       the other BASELINE.json configs measured after the timed region (rank 0, N == 1)
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _sub in ("cuda-akaze_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, _sub))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_FP4_PEAK_TOPS = 10000.0     # dense FP4 / FP6 matrix peak (MI355X_MICROARCH.md: ~10 PF dense; the matcher's operand type)
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")
CHECKSUM_FILE = os.path.join(ROOT, "tests", "golden", "bench_pair_checksums.json")
NDIST = 16                 # distinct seeded pairs; the pair with global id g is drawn from seed 1 + g % NDIST on every rank
# kernel-name substrings and the sources whose hash ties a PMC pass to the kernels it measured (tools/pmc_traffic.py)
CLASS_KERNELS = {
    "fed": ("k_fed_multi", "k_fed_sf", "k_fed_generic", "k_level_tile"),
    "hessian": ("k_hessian_stream", "k_hessian_fused"),
    "prologue": ("k_base_stream", "k_base_a", "kf_base", "k_grad_hist_plane", "k_kcontrast2", "k_lattice_hmax"),
    "describe": ("k_describe", "k_orient", "k_desc_perm"),
    "nms": ("k_nms_cand", "k_row_scan", "k_emit", "k_refine", "k_clear_cand_maps"),
    "match": ("k_match",),
}
CLASS_SOURCES = {
    "fed": ("kernels_fed.hip", "kernels_fedsf.hip", "kernels_level.hip", "fed_common.h"),
    "hessian": ("kernels_hessian_stream.hip", "kernels_hessian.hip", "fed_common.h"),
    "prologue": ("kernels_base_stream.hip", "kernels_base.hip", "fed_common.h"),
    "describe": ("kernels_describe.hip",),
    "nms": ("kernels_detect.hip",),
    "match": ("kernels_match.hip",),
}


# ... plus what every class depends on: the shared header (row-segment rule hak_stream_rows, plane layout, hak_det_at), the launch
# sequence (which kernels run, what is fused) and the compiler flags
COMMON_SOURCES = ("hak_internal.h", "hak_api.hip", "Makefile")


def class_source_hash(klass):
    h = hashlib.sha256()
    for f in CLASS_SOURCES[klass] + COMMON_SOURCES:
        with open(os.path.join(ROOT, "cuda-akaze_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def launcher_command(argv, ngpus, port):
    """the command `bench.py --gpus N` starts when it was not itself started by a launcher: one rank per GPU of this node over
    RCCL, rendezvous on 127.0.0.1 (the container hostname may not resolve)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def visible_gpu_count():
    """devices this node offers, WITHOUT touching the HIP runtime (the parent of a rank launcher must stay GPU-free: a process
    that has initialised HIP may only spawn children, never exec).  KFD topology nodes with simd_count > 0 are GPUs (CPU nodes
    report 0); HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the set.  When sysfs cannot be read a
    CHILD process is asked (torch.cuda.device_count())."""
    import glob
    n = None
    try:
        nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
        if nodes:
            n = 0
            for f in nodes:
                for ln in open(f):
                    k, _, v = ln.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
    except OSError:
        n = None
    if n is None:
        import subprocess
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
        n = int(r.stdout.strip() or 0) if r.returncode == 0 else 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start N ranks as a fresh child process tree and
    relay rank 0's JSON line and the exit code.  This parent never initialises HIP (devices are counted from sysfs), so the
    children own the devices.  Fails loudly when the node has fewer devices than ranks were asked for."""
    import subprocess
    ndev = visible_gpu_count()
    if ndev < args.gpus:
        print(f"bench.py: {args.gpus} ranks requested, {ndev} device{'s' if ndev != 1 else ''} visible on this node", file=sys.stderr)
        sys.exit(2)
    port = int(os.environ.get("MASTER_PORT", "0")) or 29500 + (os.getpid() % 2000)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    proc = subprocess.run(launcher_command(argv, args.gpus, port), env=env)
    sys.exit(proc.returncode)


def shard_pairs(total_pairs, world, rank):
    """contiguous block partition of independent pairs over ranks"""
    per = (total_pairs + world - 1) // world
    lo = min(rank * per, total_pairs)
    return lo, min(lo + per, total_pairs)


def max_over_ranks(elapsed, device, use_dist):
    """the step time of the job is the slowest rank's"""
    if not use_dist:
        return float(elapsed)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


KP_FIELDS = ("x", "y", "octave", "response", "size", "angle", "features")
MATCH_FIELDS = ("match", "distance", "match_x", "match_y")


def pair_digest(p1, p2):
    """64-bit checksum of one pair's result records: sha256 over every field the reference fills (img1: keypoint + match fields,
    img2: keypoint fields), truncated to a signed int64"""
    h = hashlib.sha256()
    for f in KP_FIELDS + MATCH_FIELDS:
        h.update(p1[f].tobytes())
    for f in KP_FIELDS:
        h.update(p2[f].tobytes())
    return int.from_bytes(h.digest()[:8], "little", signed=True)


def summarize_pairs(counts, pts, gids):
    """one 32-byte summary {pair_id, n1 << 32 | n2, n_matches, checksum} per pair of a downloaded batch"""
    out = np.zeros((len(gids), 4), np.int64)
    for k, g in enumerate(gids):
        n1, n2 = int(counts[2 * k]), int(counts[2 * k + 1])
        p1, p2 = pts[2 * k, :n1], pts[2 * k + 1, :n2]
        out[k] = (g, (n1 << 32) | n2, int((p1["match"] >= 0).sum()), pair_digest(p1, p2))
    return out


def gather_pair_summaries(local, device, use_dist):
    """all-gather of the per-pair summaries (the one collective of the path: ceil(P / G) x 32 B per rank); ranks with fewer pairs
    pad with pair_id -1.  Returns the [P, 4] table of all ranks, sorted by pair_id."""
    local = np.asarray(local, np.int64).reshape(-1, 4)
    if not use_dist:
        return local[np.argsort(local[:, 0], kind="stable")]
    n = torch.tensor([len(local)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    buf = torch.full((int(n.item()), 4), -1, dtype=torch.int64, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(local).to(device)
    parts = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, buf)
    tab = torch.cat(parts).cpu().numpy()
    tab = tab[tab[:, 0] >= 0]
    return tab[np.argsort(tab[:, 0], kind="stable")]


def check_pair_table(tab, expected_pairs, golden=None):
    """rank 0's checks on the gathered table: every pair exactly once; equal seeds => equal summaries on every rank; and, when a
    G = 1 table is committed for this configuration, equal to it (so an N-device run equals the 1-device run pair for pair)"""
    ids = tab[:, 0]
    complete = len(tab) == expected_pairs and np.array_equal(ids, np.arange(expected_pairs))
    by_seed = {}
    for row in tab:
        by_seed.setdefault(int(row[0]) % NDIST, set()).add((int(row[1]), int(row[2]), int(row[3])))
    consistent = all(len(v) == 1 for v in by_seed.values())
    res = {"pairs": int(len(tab)), "complete": bool(complete), "bytes_per_pair": 32, "distinct_seeds": len(by_seed),
           "equal_seed_equal_checksum": bool(consistent), "keypoints": int(((tab[:, 1] >> 32) + (tab[:, 1] & 0xFFFFFFFF)).sum()),
           "matches": int(tab[:, 2].sum()), "equals_g1_table": None}
    if golden is not None and consistent:
        res["equals_g1_table"] = all(list(next(iter(v))) == [int(x) for x in golden[str(sd)]] for sd, v in by_seed.items())
    return res


def pin_to_gpu_numa_node(local_rank, world):
    """best effort: run this rank's host threads (launch loop, pinned result buffers, OpenMP oracle) on the cores of the NUMA node its
    GPU hangs off, so that the PCIe traffic of the result download / image upload does not cross sockets (SURVEY 8e names the host
    side as the 8-GPU bottleneck).  Returns the node number or None when the topology cannot be read."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if world > 1 and len(allowed) >= 2:
            os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


def fence(use_dist):
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()


# --------------------------------------------------------------------------- CPU oracle legs (checker / baseline)
def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    # a GPU box gives one GPU a 16-core CPU share; an OpenMP team per hardware thread only oversubscribes it
    return max(1, min(16, n))


def host_topology():
    """what the box has, beside the `cores` (OpenMP threads) the baseline actually used: logical CPUs, CPUs this process may run on,
    physical cores (distinct (package, core id) pairs of /sys) and the model name"""
    info = {"logical_cpus": os.cpu_count()}
    try:
        info["allowed_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        import glob
        cores = set()
        for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology"):
            cores.add((open(d + "/physical_package_id").read().strip(), open(d + "/core_id").read().strip()))
        if cores:
            info["physical_cores"] = len(cores)
            info["sockets"] = len({c[0] for c in cores})
    except OSError:
        pass
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                info["model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def oracle_setup():
    """the oracle for this run: the committed build, or the same sources compiled for this host when a compiler is here.
    OMP_NUM_THREADS must be set before libgomp initialises (first oracle call)."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    import okz
    okz.build()
    tmp = os.environ.get("TMPDIR", "/tmp")
    native = okz.build_native(tmp)
    if native:
        okz.load(native)
    # (the environment variable alone is not enough: torch's import has initialised libgomp long before this point, with one thread
    # per logical CPU -- until round 5 the oracle's loops then ran 256 wide on the GPU boxes and the baseline read 1.0 pairs/s
    # where the same code does 8-10 on 16 threads)
    cores = okz.set_num_threads(cores)
    return okz, cores, ("-O3 -march=native" if native else "-O2 (committed build)")


def oracle_pairs(okz, synth, u8_pairs, w, p, max_pts, npairs):
    """oracle results [(pts1 with match fields, pts2)] of the first npairs pairs + per-pair (detect_s, match_s)"""
    out, times = [], []
    for a, b in u8_pairs[:npairs]:
        t0 = time.perf_counter()
        r1 = okz.detect_and_compute(synth.to_float(a, p), w, max_pts=max_pts)
        r2 = okz.detect_and_compute(synth.to_float(b, p), w, max_pts=max_pts)
        t1 = time.perf_counter()
        okz.match(r1.points, r2.points)
        t2 = time.perf_counter()
        out.append((r1.points, r2.points))
        times.append((t1 - t0, t2 - t1))
    return out, times


def verify_batch(want, pts, counts):
    """field-by-field comparison of a downloaded batch's first len(want) pairs with the oracle (says WHICH bar broke)"""
    pts_ok, m_ok = True, True
    for k, (o1, o2) in enumerate(want):
        for j, o in enumerate((o1, o2)):
            i = 2 * k + j
            g = pts[i, :counts[i]]
            if counts[i] != len(o):
                pts_ok = False
                continue
            for f in KP_FIELDS:
                pts_ok &= g[f].tobytes() == o[f].tobytes()
            if j == 0:
                for f in MATCH_FIELDS:
                    m_ok &= g[f].tobytes() == o[f].tobytes()
    return bool(pts_ok), bool(m_ok)


def oracle_fast_pairs(okz, u8_pairs, max_pts):
    out = []
    for a, b in u8_pairs:
        r1 = okz.fast_detect_and_compute(a, max_pts=max_pts).points
        r2 = okz.fast_detect_and_compute(b, max_pts=max_pts).points
        okz.match(r1, r2)
        out.append((r1, r2))
    return out


def slot_digests(counts, pts, npairs):
    """one 64-bit digest per pair slot of a downloaded batch (read in place from the pinned host buffer)"""
    return np.array([pair_digest(pts[2 * k, :int(counts[2 * k])], pts[2 * k + 1, :int(counts[2 * k + 1])]) for k in range(npairs)], np.int64)


def count_slot_mismatches(digests, want_digests, lo):
    """slot k of a launch sequence holds the pair with global id lo + k (+ a multiple of NDIST), i.e. seed index (lo + k) % NDIST"""
    exp = np.array([want_digests[(lo + k) % NDIST] for k in range(len(digests))], np.int64)
    return int((digests != exp).sum())


def opencv_baseline(u8_pairs, cores):
    """main.cpp:344-399's own CPU leg -- cv::AKAZE::create() defaults + BFMatcher(HAMMING) -- when OpenCV is importable
    (it is not in this image; the oracle port below is what runs)"""
    try:
        import cv2
    except ImportError:
        return None
    cv2.setNumThreads(cores)
    det, bf = cv2.AKAZE_create(), cv2.BFMatcher(cv2.NORM_HAMMING)
    times = []
    for k in range(11):
        a, b = u8_pairs[k % len(u8_pairs)]
        t0 = time.perf_counter()
        _, d1 = det.detectAndCompute(a, None)
        _, d2 = det.detectAndCompute(b, None)
        bf.match(d1, d2)
        if k:
            times.append(time.perf_counter() - t0)
    return {"value": round(1.0 / statistics.median(times), 4), "unit": "pairs/s", "cores": cores,
            "what": "cv2.AKAZE_create() defaults + BFMatcher(NORM_HAMMING), median of 10 pairs"}


def cpu_baseline(okz, synth, cores, flags, u8_pairs, w, p, max_pts, first_times, budget_s=12.0):
    """median pairs/s of the oracle over >= 10 pairs and about `budget_s` seconds of CPU work (the NDIST pairs already run for the
    verification count, minus the first)"""
    times = list(first_times[1:])                 # the very first pair paid page-in and the OpenMP pool start-up
    t_start = time.perf_counter()
    k = 0
    while len(times) < 10 or (time.perf_counter() - t_start < budget_s and len(times) < 96):
        _, t = oracle_pairs(okz, synth, [u8_pairs[k % len(u8_pairs)]], w, p, max_pts, 1)
        times += t
        k += 1
        if time.perf_counter() - t_start > 4 * budget_s:
            break
    det = statistics.median(t[0] for t in times)
    mat = statistics.median(t[1] for t in times)
    tot = statistics.median(t[0] + t[1] for t in times)
    return {"value": round(1.0 / tot, 4), "unit": "pairs/s", "cores": cores, "kind": "port", "host": host_topology(),
            "detect_ms_per_pair": round(det * 1e3, 2), "match_ms_per_pair": round(mat * 1e3, 2),
            "opencv": opencv_baseline(u8_pairs, cores),
            "sample": f"median of {len(times)} synthetic {w}-px-wide pairs (detect+describe both images, then match), "
                      f"OpenMP oracle built {flags}; OpenCV's cv::AKAZE (main.cpp:344-399) is not installed in this image"}


# --------------------------------------------------------------------------- GPU-side helpers
class Pipeline:
    """NCTX contexts used round-robin: while batch i runs on the GPU, batch i-1 is synchronised and downloaded"""

    def __init__(self, ah, w, h, p, nimg, max_pts, nctx, octaves=4, upright=False, serial=False, torch_stream=True):
        # serial = one stream per context instead of one per octave (hak_set_concurrency)
        self.serial = bool(serial)
        self.ah, self.nimg, self.max_pts, self.h, self.p = ah, nimg, max_pts, h, p
        self.dets, self.d_pts, self.d_num, self.h_pts, self.h_num = [], [], [], [], []
        for k in range(nctx):
            dk = ah.Akazer()
            dk.init((w, h, p), noctaves=octaves, max_pts=max_pts, batch=nimg, upright=bool(upright))
            if k == 0 and torch_stream:
                ah.check(ah.lib.hak_set_stream(dk.ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            if serial:
                ah.check(ah.lib.hak_set_concurrency(dk.ctx, 0))
            self.dets.append(dk)
            self.d_pts.append(torch.zeros(nimg * max_pts * 104, dtype=torch.uint8, device="cuda"))
            self.d_num.append(torch.zeros(nimg, dtype=torch.int32, device="cuda"))
            hp, hn = C.c_void_p(), C.c_void_p()
            ah.check(ah.lib.hak_host_alloc(C.byref(hp), nimg * max_pts * 104))
            ah.check(ah.lib.hak_host_alloc(C.byref(hn), nimg * 4))
            self.h_pts.append(hp)
            self.h_num.append(hn)
        self.nctx = nctx
        self.nseq = 0                           # float-path launch sequences enqueued so far (the PMC tools divide by it)
        self.last_pairs = [0] * nctx            # pairs of the last job each context ran (whose download its host buffers hold)
        # off by default: +1.4 % at 1080p in one A/B, nothing in the next, -3 % at 720p
        self.phase_lock = os.environ.get("HAK_BENCH_PHASE_LOCK", "0") != "0"
        self.phase_ev = []
        for k in range(nctx):
            ev = C.c_void_p()
            ah.check(ah.lib.hak_phase_event(self.dets[k].ctx, C.byref(ev)))
            self.phase_ev.append(ev)

    def enqueue(self, k, d_imgs, npairs=None, fast=False):
        """detect + describe on 2*npairs images + match of the npairs pairs (default: the whole batch)"""
        ah = self.ah
        n = self.nimg if npairs is None else 2 * npairs
        fn = ah.lib.hak_fast_detect_and_compute_batch if fast else ah.lib.hak_detect_and_compute_batch
        self.nseq += 0 if fast else 1
        ah.check(fn(self.dets[k].ctx, d_imgs.data_ptr(), self.h * self.p, self.p, n, self.d_pts[k].data_ptr(), self.d_num[k].data_ptr(), 1))
        ah.check(ah.lib.hak_match_batch(self.dets[k].ctx, self.d_pts[k].data_ptr(), self.d_num[k].data_ptr(), n // 2))

    def download(self, k, npairs=None):
        n = self.nimg if npairs is None else 2 * npairs
        self.ah.check(self.ah.lib.hak_download_batch(self.dets[k].ctx, self.d_pts[k].data_ptr(), self.d_num[k].data_ptr(), n,
                                                     self.h_pts[k], self.h_num[k]))

    def run(self, jobs, pre=None):
        """jobs = list of (d_imgs, npairs); every job's results are host-visible when run() returns"""
        n, N = len(jobs), self.nctx
        for i, (imgs, npairs) in enumerate(jobs):
            if pre:
                pre(i % N)
            if self.phase_lock and N == 2:
                # start this context's scale space when the other context's reaches its keypoint stages (hak_phase_event)
                self.ah.check(self.ah.lib.hak_wait_event(self.dets[i % N].ctx, self.phase_ev[(i + 1) % N]))
            self.enqueue(i % N, imgs, npairs)
            self.last_pairs[i % N] = self.nimg // 2 if npairs is None else npairs
            if i >= N - 1:
                j = i - (N - 1)
                self.download(j % N, jobs[j][1])
        for j in range(max(0, n - (N - 1)), n):
            self.download(j % N, jobs[j][1])

    def results(self, k):
        counts = np.ctypeslib.as_array(C.cast(self.h_num[k], C.POINTER(C.c_int)), shape=(self.nimg,)).copy()
        pts = np.ctypeslib.as_array(C.cast(self.h_pts[k], C.POINTER(C.c_uint8)),
                                    shape=(self.nimg * self.max_pts * 104,)).view(self.ah.POINT_DTYPE).reshape(self.nimg, self.max_pts)
        return counts, pts

    def close(self):
        for k in range(self.nctx):
            self.ah.lib.hak_host_free(self.h_pts[k])
            self.ah.lib.hak_host_free(self.h_num[k])
            self.dets[k].close()
        self.d_pts, self.d_num = [], []


def fit_batch(B, w, h, max_pts, nctx, floor=16):
    # arena = 15 float planes per octave pyramid (x 4/3) + key map 8 B/px + candidate list ~ 11 B/px ~ 100 B/px per image and
    # context (207 MB at 1080p): 256 pairs x 2 contexts = 214 GB of the 288 GB.  Batch size is a batching choice, not part of the workload
    # (measured, pairs/s, round 2: 96: 5665, 128: 5654, 144: 5752, 192: 5792; round 3, one box: 128: 6040, 192: 6012, 240: 6123, 256: 6152).
    free_b = torch.cuda.mem_get_info()[0]
    while B > floor and 2 * B * (w * h * 100 + 2 * max_pts * 104) * nctx > 0.8 * free_b:
        B //= 2
    return B


def timed_throughput(pipe, d_imgs, B, steps, warmup):
    jobs = [(d_imgs, B)]
    pipe.run(jobs * warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(jobs * steps)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


def batch_curve(ah, synth, u8_pairs, w, h, p, max_pts):
    """pairs/s against the pairs in flight (contexts x pairs per launch sequence), the small-batch end of the headline: where a
    caller that cannot collect 256 pairs stands.  One context with 1 pair per sequence is the pair-level call (both images of a
    pair in ONE launch sequence + the match, then one synchronisation)."""
    rows = []
    for nctx, ppseq in ((1, 1), (1, 2), (2, 2), (1, 4), (2, 4), (1, 8), (2, 8), (2, 16)):
        nimg = 2 * ppseq
        d = torch.from_numpy(np.stack([synth.to_float(u8_pairs[(i // 2) % NDIST][i % 2], p) for i in range(nimg)])).cuda()
        pipe = Pipeline(ah, w, h, p, nimg, max_pts, nctx, serial=False, torch_stream=False)
        rate = timed_throughput(pipe, d, ppseq, max(20, 320 // (nctx * ppseq)), 5)
        rows.append({"pairs_in_flight": nctx * ppseq, "contexts": nctx, "pairs_per_sequence": ppseq, "pairs_per_s": round(rate, 1),
                     "ms_per_sequence": round(1e3 * nctx * ppseq / rate, 3),
                     "arena_MB": round(nctx * nimg * (w * h * 100 + 2 * max_pts * 104) / 1e6)})
        pipe.close()
        del d, pipe
    return rows


def natural_leg(ah, synth, okz, max_pts):
    """the natural-image 1080p pair (img1 / img2 of BASELINE configs[0], reconstructed from the reference's own result pictures:
    tools/ref_render_check.py) through the same batched path, verified against the oracle"""
    f = os.path.join(ROOT, "tests", "golden", "ref_recon_1080p_u8.npz")
    if not os.path.exists(f):
        return None
    rec = np.load(f)
    a, b = rec["img1"], rec["img2"]
    h, w = a.shape
    p = ah.iAlignUp(w, 128)
    B = 64
    d = torch.from_numpy(np.stack([synth.to_float((a, b)[i % 2], p) for i in range(2 * B)])).cuda()
    pipe = Pipeline(ah, w, h, p, 2 * B, max_pts, 2, serial=True, torch_stream=False)
    rate = timed_throughput(pipe, d, B, 6, 2)
    counts, pts = pipe.results(1)
    out = {"pairs_per_s": round(rate, 1), "pairs_per_sequence": B, "keypoints": [int(counts[0]), int(counts[1])],
           "matches": int((pts[0, :counts[0]]["match"] >= 0).sum()),
           "reference_printed_keypoints": [2205, 2382], "verified": None}
    if okz is not None:
        r1 = okz.detect_and_compute(synth.to_float(a, p), w, max_pts=max_pts).points
        r2 = okz.detect_and_compute(synth.to_float(b, p), w, max_pts=max_pts).points
        okz.match(r1, r2)
        dg = pair_digest(r1, r2)
        bad = sum(pair_digest(pts[2 * k, :counts[2 * k]], pts[2 * k + 1, :counts[2 * k + 1]]) != dg for k in range(B))
        out["verified"] = {"images": 2, "slots": B, "slots_equal": bad == 0}
    pipe.close()
    return out


def other_configs(ah, synth, args, rank, okz=None, u8_pairs=None):
    """BASELINE.json configs[1] as a single pair and as a batch-size curve, configs[0]'s images, configs[2], configs[3]'s shape on
    one GPU, configs[4]; never `value`"""
    out = {}
    max_pts = 10000
    out.update(args.single_pair_result or {})               # configs[1] literally: measured first, in a child process (run_single_pair_leg)
    if u8_pairs is not None:
        out["batch_curve_1080p"] = batch_curve(ah, synth, u8_pairs, 1920, 1080, ah.iAlignUp(1920, 128), max_pts)
        torch.cuda.empty_cache()
    out["natural_1080p"] = natural_leg(ah, synth, okz, max_pts)
    torch.cuda.empty_cache()
    # ---- configs[2]: 3840x2160, 5 octaves, MLDB-upright
    w, h = 3840, 2160
    p = ah.iAlignUp(w, 128)
    B = fit_batch(16, w, h, max_pts, 2, floor=2)
    pr = synth.pair(w, h, 2, nshapes=330)       # ~8 k keypoints per image, as configs[2] says (unclamped: max_pts is 10 000)
    d = torch.from_numpy(np.stack([synth.to_float(pr[i % 2], p) for i in range(2 * B)])).cuda()
    pipe = Pipeline(ah, w, h, p, 2 * B, max_pts, 2, octaves=5, upright=True, torch_stream=False)
    out["pairs_per_s_4k_5oct_upright"] = round(timed_throughput(pipe, d, B, 3, 1), 1)
    counts, _ = pipe.results(0)
    out["keypoints_per_image_4k"] = round(float(counts.mean()), 1)
    pipe.close(); del d, pipe
    # ---- configs[3]'s shape on this GPU: a batch of 64 independent 1280x720 pairs per step
    w, h = 1280, 720
    p = ah.iAlignUp(w, 128)
    prs = [synth.pair(w, h, 3 + i) for i in range(4)]
    d = torch.from_numpy(np.stack([synth.to_float(prs[(i // 2) % 4][i % 2], p) for i in range(128)])).cuda()
    pipe = Pipeline(ah, w, h, p, 128, max_pts, 2, torch_stream=False)
    out["pairs_per_s_720p_batch64"] = round(timed_throughput(pipe, d, 64, 6, 2), 1)
    pipe.close(); del d, pipe
    # ---- configs[4]: 10k x 10k brute-force Hamming (hak_match -> k_match_mfma + k_match_finish), 20 synchronous calls
    n = 10000
    q = synth.random_descriptors(n, 7, ah.POINT_DTYPE)
    t = synth.random_descriptors(n, 8, ah.POINT_DTYPE, planted_from=q, nplanted=4000)
    dq = torch.from_numpy(q.view(np.uint8).copy()).cuda()
    dt = torch.from_numpy(t.view(np.uint8).copy()).cuda()
    for _ in range(3):
        ah.check(ah.lib.hak_match(None, dq.data_ptr(), n, dt.data_ptr(), n, None))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ah.check(ah.lib.hak_match(None, dq.data_ptr(), n, dt.data_ptr(), n, None))     # synchronous (cuMatch contract)
    ms = (time.perf_counter() - t0) * 1e3 / 20
    # SURVEY 8d floor: 1e8 pairs x ~32 VALU lane-ops (8 x 64-bit XOR + 8 x popcount-64) / (256 CU x 64 lanes x 2.4 GHz) = 0.08 ms
    floor_ms = 1e8 * 32 / (256 * 64 * 2.4e9) * 1e3
    out["match_10k_ms"] = round(ms, 4)
    out["match_10k_valu_floor_ms"] = round(floor_ms, 4)
    out["match_10k_frac_of_floor"] = round(floor_ms / ms, 4)
    # the kernel is k_match_mfma: 1e8 / 1024 tiles of 32 x 32 distances x 8 v_mfma_f32_32x32x64_f8f6f4 (fp4 operands, one descriptor
    # bit per k) x 32 cycles on 1024 SIMDs at 2.4 GHz (rounds 2-4: 16 v_mfma_i32_32x32x32_i8 per tile, twice this); the synchronous
    # call also carries the host's launch + wait
    out["match_10k_mfma_floor_ms"] = round(1e8 / 1024 * 8 * 32 / (1024 * 2.4e9) * 1e3, 4)
    # 2-NN ratio test (4/5) + symmetric cross-check + device-side compaction on the same sets (two 10k x 10k searches)
    d_out = torch.zeros(n * ah.MATCH_PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    cnt = C.c_int(0)

    def knn():
        ah.check(ah.lib.hak_match_knn2(None, dq.data_ptr(), n, dt.data_ptr(), n, 4, 5, 1, 0, None, d_out.data_ptr(), C.byref(cnt), None))
    for _ in range(3):
        knn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        knn()
    out["match_knn2_10k_ms"] = round((time.perf_counter() - t0) * 1e3 / 10, 4)
    out["match_knn2_10k_accepted"] = int(cnt.value)
    return out


def _run_leg(flag, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), flag], capture_output=True, text=True, env=env, timeout=600)
    leg = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            leg = json.loads(ln)
    if r.returncode != 0 or leg is None:
        raise RuntimeError(flag + " failed: " + r.stdout[-500:] + r.stderr[-2000:])
    return leg


def run_single_pair_leg():
    """configs[1] literally -- ONE 1080p pair at a time, synchronous calls like the reference demo (main.cpp:199-209) -- in child
    processes, BEFORE this process touches the GPU (two processes with live queues on one device slow each other's short synchronous
    calls: match 0.06 -> 0.16 ms).  Two children, because the two call styles want different numbers of hardware queues and the variable
    is read once, when the HIP runtime initialises (INTEGRATION.md "Hardware queues"): the three-call pattern keeps four
    eagerly launched chains in flight and wants GPU_MAX_HW_QUEUES=8; the pair call replays a captured graph, whose queue placement is
    best with the runtime's default of four (0.57-0.60 ms; 0.83 with eight) -- like the batched legs of the bench process itself."""
    leg = _run_leg("--single-pair-leg", {"GPU_MAX_HW_QUEUES": os.environ.get("HAK_SINGLE_HW_QUEUES", "8")})
    env2 = {}
    if "HAK_PAIR_HW_QUEUES" in os.environ:
        env2["GPU_MAX_HW_QUEUES"] = os.environ["HAK_PAIR_HW_QUEUES"]
    leg.update(_run_leg("--pair-call-leg", env2))
    leg["single_pair_note"] = ("one synchronous pair at a time, two ways: pair_call_latency_ms = the pair-level launch sequence (both images + the match in "
                               "ONE call, hak_detect_and_compute_pair / Akazer::detectAndComputePair); single_pair_latency_ms = the reference demo's "
                               "literal THREE calls (detectAndCompute x 2 + cuMatch, main.cpp:199-209), each synchronous")
    return leg


def _pair_setup():
    import akaze_hip as ah
    from akaze_hip import synth
    assert torch.cuda.is_available(), "needs a HIP device"
    max_pts = 10000
    w, h = 1920, 1080
    p = ah.iAlignUp(w, 128)
    a, b = synth.pair(w, h, 1)
    d1 = torch.from_numpy(synth.to_float(a, p)).cuda()
    d2 = torch.from_numpy(synth.to_float(b, p)).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=max_pts, batch=2)           # as the C++ Akazer: one context for single-image and pair calls
    r1, r2 = ah.AkazeData(), ah.AkazeData()
    ah.initAkazeData(r1, max_pts, True, True, pinned=True)
    ah.initAkazeData(r2, max_pts, True, True, pinned=True)
    return ah, det, d1, d2, r1, r2, (w, h, p)


def single_pair_leg():
    """one 1080p pair at a time through the drop-in class: two synchronous detectAndCompute calls + cuMatch per iteration, h_data pinned as
    the C++ layer's initAkazeData hands it out (host/akaze.cpp); prints one JSON object"""
    ah, det, d1, d2, r1, r2, whp = _pair_setup()
    lat, det_ms, match_ms = [], [], []
    for i in range(45):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        det.detectAndCompute(d1.data_ptr(), r1, whp, True)
        det.detectAndCompute(d2.data_ptr(), r2, whp, True)
        t1 = time.perf_counter()
        ah.cuMatch(r1, r2, det)
        t2 = time.perf_counter()
        if i >= 5:
            lat.append((t2 - t0) * 1e3); det_ms.append((t1 - t0) * 1e3); match_ms.append((t2 - t1) * 1e3)
    print(json.dumps({"single_pair_latency_ms": round(statistics.median(lat), 3),
                      "single_pair_detect_ms": round(statistics.median(det_ms), 3), "single_pair_match_ms": round(statistics.median(match_ms), 3),
                      "single_pair_keypoints": [r1.num_pts, r2.num_pts],
                      "single_pair_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")}))
    ah.freeAkazeData(r1); ah.freeAkazeData(r2); det.close()


def pair_call_leg():
    """the same pair through ONE call (hak_detect_and_compute_pair / Akazer::detectAndComputePair: one launch sequence for both images
    + the match, one wait); prints one JSON object"""
    ah, det, d1, d2, r1, r2, whp = _pair_setup()
    plat = []
    for i in range(65):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        det.detectAndComputePair(d1.data_ptr(), d2.data_ptr(), r1, r2, whp, True, True)
        if i >= 5:
            plat.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"pair_call_latency_ms": round(statistics.median(plat), 3), "pair_call_keypoints": [r1.num_pts, r2.num_pts],
                      "pair_call_matches": int((r1.h_data[:r1.num_pts]["match"] >= 0).sum()),
                      "pair_call_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")}))
    ah.freeAkazeData(r1); ah.freeAkazeData(r2); det.close()


def pmc_class_bytes(fdir, wdir, nseq):
    """HBM bytes per launch sequence and kernel class from one FETCH_SIZE and one WRITE_SIZE pass (rocprofv3 csv): the counters are in
    KiB; FETCH_SIZE is doubled for the classes whose loads are 16 B/lane streams (gfx950 reports half of such a read,
    MI355X_MICROARCH.md) and taken as reported for the gather kernels (profiles/r02_gather_calib.txt); WRITE_SIZE as reported"""
    import csv
    import glob

    def load(d, counter):
        tot = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == counter:
                    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                    tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]) * 1024.0
        return tot
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    if not fetch or not write:
        return None
    out = {}
    for k, names in CLASS_KERNELS.items():
        f = sum(v for n, v in fetch.items() if any(x in n for x in names))
        w = sum(v for n, v in write.items() if any(x in n for x in names))
        out[k] = ((2.0 if k in ("fed", "hessian", "prologue") else 1.0) * f + w) / nseq
    return out


def run_pmc_legs(pairs):
    """`roofline.traffic` measured in THIS run on THIS box: two child runs of the bench's own launch sequence under
    `rocprofv3 --pmc` -- FETCH_SIZE, then WRITE_SIZE, separate passes, no trace option, the python interpreter itself behind `--`
    (MI355X_MICROARCH.md) -- started before this process touches the GPU.  Returns ({class: bytes per launch sequence}, pairs per
    sequence of the profiled run) or None (no rocprofv3, a pass failed): bench.py then falls back to the committed passes."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or os.environ.get("HAK_BENCH_PMC", "1") == "0":
        return None
    tmp = tempfile.mkdtemp(prefix="hak_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=tmp)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    shape = None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-leg", "--pairs", str(pairs)]
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return None
            for ln in r.stdout.splitlines():
                if ln.startswith("{") and '"metric"' in ln:
                    d = json.loads(ln)
                    shape = (int(d["config"]["pairs_per_launch_sequence"]), int(d["config"]["float_sequences_enqueued"]))
        if shape is None:
            return None
        cls = pmc_class_bytes(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), shape[1])
        return None if cls is None else (cls, shape[0])
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# --------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3),
                    help="BASELINE.json configs[]: 1 = 1080p pairs (default); 2 = 3840x2160, 5 octaves, MLDB-upright; "
                         "3 = the same 512 1280x720 pairs sharded over all ranks (--total-pairs 512 --width 1280 --height 720)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=256, help="pairs per GPU per launch sequence (batch); halved until the arenas fit the free HBM")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="strong scaling: this many pairs per step over ALL ranks (rank r takes shard_pairs), in batches of --pairs")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the timed batch")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs measured after the timed region")
    ap.add_argument("--no-pipeline", action="store_true", help="one context: no overlap between consecutive steps")
    ap.add_argument("--pipeline", type=int, default=2, help="contexts used round-robin (batches in flight)")
    ap.add_argument("--upload", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-upload", action="store_true",
                    help="skip the upload-inclusive leg: uint8 host images -> H2D -> on-device ingest -> path")
    ap.add_argument("--fast", action="store_true", help="time the integer FAST path also with N > 1 (default on for N = 1)")
    ap.add_argument("--no-fast", action="store_true",
                    help="skip the integer FAST path leg (fastDetectAndCompute, uint8 inputs, same pairs, verified against its oracle)")
    ap.add_argument("--upright", action="store_true", help="MLDB-upright (skip the orientation stage; configs[2] of BASELINE.json)")
    ap.add_argument("--serial", action="store_true",
                    help="one stream per context in the timed region (the default when two contexts are in flight)")
    ap.add_argument("--concurrent", action="store_true",
                    help="the octaves of a context on concurrent streams in the timed region (the default for --no-pipeline)")
    ap.add_argument("--single-pair-leg", action="store_true", help="internal: run only the one-pair-at-a-time leg of `configs` and print it")
    ap.add_argument("--pair-call-leg", action="store_true", help="internal: run only the one-pair-per-call leg of `configs` and print it")
    ap.add_argument("--pmc-leg", action="store_true",
                    help="internal: two serial launch sequences and the gather pass, nothing else (what run_pmc_legs profiles under rocprofv3 --pmc)")
    ap.add_argument("--launch", action="store_true",
                    help="go through the rank launcher even for --gpus 1 (N > 1 without a launcher always does)")
    args = ap.parse_args()
    if args.single_pair_leg:
        return single_pair_leg()
    if args.pair_call_leg:
        return pair_call_leg()
    if args.pmc_leg:
        args.steps, args.warmup, args.serial, args.no_pipeline = 1, 1, True, True
        args.no_cpu_baseline = args.no_roofline = args.no_configs = args.no_verify = args.no_upload = args.no_fast = True
    if args.config == 2:
        args.width, args.height, args.octaves, args.upright = 3840, 2160, 5, True
        args.pairs = min(args.pairs, 32)
    elif args.config == 3:
        args.width, args.height = 1280, 720
        args.total_pairs = args.total_pairs or 512

    gpus_given = args.gpus is not None
    if not gpus_given:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))  # started by a launcher without --gpus: its world size
    if (args.gpus > 1 or args.launch) and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])                     # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if gpus_given and world != args.gpus:                   # an explicit --gpus that contradicts the launcher is an error
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    args.single_pair_result, live_pmc = None, None
    if rank == 0 and world == 1 and not args.total_pairs and (args.width, args.height, args.octaves) == (1920, 1080, 4) and not args.upright:
        # child processes, before this process initialises HIP
        if not args.no_configs:
            args.single_pair_result = run_single_pair_leg()
        if not args.no_roofline:
            live_pmc = run_pmc_legs(args.pairs)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # HAK_BENCH_DIST_BACKEND=gloo: the REHEARSAL of a multi-rank run on a node with fewer GPUs than ranks (one MI355X): every
    # rank computes on device local_rank % device_count, and the job's three collectives (barrier, MAX of the step time, the
    # all-gather of the 32-byte pair summaries) run over gloo on CPU tensors.  Everything else -- sharding, seeds, per-rank
    # launch sequences, the gathered table and rank 0's checks -- is the code an 8-GPU RCCL run executes.
    backend = os.environ.get("HAK_BENCH_DIST_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        print(f"bench.py: HAK_BENCH_DIST_BACKEND={backend!r} (nccl or gloo)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    dist_dev = "cpu" if backend == "gloo" else "cuda"
    torch.cuda.set_device(dev_index)
    numa_node = pin_to_gpu_numa_node(dev_index, world)
    # HAK_BENCH_FORCE_DIST=1 exercises the RCCL code path (barrier / MAX all-reduce / summary all-gather) with one rank too
    use_dist = world > 1 or os.environ.get("HAK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import akaze_hip as ah
    from akaze_hip import synth
    ah.check(ah.lib.hak_set_device(dev_index))

    w, h = args.width, args.height
    p = ah.iAlignUp(w, 128)
    max_pts = 10000
    NCTX = 1 if args.no_pipeline else max(1, args.pipeline)
    B = fit_batch(args.pairs, w, h, max_pts, NCTX)
    strong = args.total_pairs > 0
    if strong:
        lo, hi = shard_pairs(args.total_pairs, world, rank)
        my_pairs = hi - lo
        B = max(1, min(B, my_pairs)) if my_pairs else 1
        if my_pairs > B and B > NDIST:
            B -= B % NDIST              # every launch sequence of a step starts at a multiple of NDIST: one resident batch serves all
        chunks = [min(B, my_pairs - c) for c in range(0, my_pairs, B)]          # pairs per launch sequence of one step
    else:
        lo, my_pairs, chunks = rank * B, B, [B]
    nimg = 2 * B

    # ---- synthetic inputs, resident in HBM: NDIST distinct seeded pairs, the pair with GLOBAL id g drawn from seed 1 + g % NDIST on
    # whichever rank it lands -- so `--total-pairs T` is the same set of pairs at every G, and equal seeds must give equal results
    # on every rank (the gather below checks exactly that).  Slot k of a launch sequence holds global pair lo + (chunk start) + k.
    u8_pairs = [synth.pair(w, h, 1 + i) for i in range(NDIST)]
    seed_of = lambda k: (lo + k) % NDIST                     # noqa: E731
    assert len(chunks) <= 1 or B % NDIST == 0 or B >= my_pairs
    host = np.stack([synth.to_float(u8_pairs[seed_of(i // 2)][i % 2], p) for i in range(nimg)])
    d_imgs = torch.from_numpy(host).cuda()
    del host

    # One stream per context when two contexts are in flight: the other context's launch sequence is what fills the small octaves'
    # gaps then, and per-context octave streams only add event waits and contention for the runtime's four hardware queues
    # (192 x 1080p pairs, A/B on one box: 5 947 / 5 969 vs 5 855 / 5 826 pairs/s).  A lone context wants its octaves on concurrent
    # streams (5 647 vs 5 397), and so do the smaller launches of the 720p leg (10 850 vs 10 500): those keep the library's default.
    pipe = Pipeline(ah, w, h, p, nimg, max_pts, NCTX, octaves=args.octaves, upright=args.upright,
                    serial=True if args.serial else False if args.concurrent else NCTX >= 2)
    step_jobs = [(d_imgs, c) for c in chunks]

    pipe.run(step_jobs * args.warmup)
    fence(use_dist)
    t0 = time.perf_counter()
    pipe.run(step_jobs * args.steps)
    fence(use_dist)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist_dev, use_dist)
    # the last download of EVERY context of the timed region (checked slot by slot against the oracle below)
    timed_digests, first_slots = [], None
    for k in range(NCTX):
        if pipe.last_pairs[k]:
            ck, pk = pipe.results(k)
            timed_digests.append(slot_digests(ck, pk, pipe.last_pairs[k]))
            if first_slots is None:                         # kept for the field-by-field diagnosis of a mismatch
                nf = min(pipe.last_pairs[k], NDIST)
                first_slots = (ck[:2 * nf].copy(), pk[:2 * nf].copy(), nf)

    # ---- PCIe-inclusive rate (never `value`): pinned uint8 batch -> H2D -> hak_ingest_u8 -> same steps
    upload_rate = None
    if not args.no_upload and my_pairs:
        h_u8 = torch.from_numpy(np.stack([u8_pairs[seed_of(i // 2)][i % 2] for i in range(nimg)])).pin_memory()
        # every pipeline context ingests into its OWN image batch: with two batches in flight the other context may still be
        # reading its images while this one's are rewritten (real frames change from step to step)
        d_u8s = [torch.empty_like(h_u8, device="cuda") for _ in range(NCTX)]
        d_imgs_k = [d_imgs] + [torch.empty_like(d_imgs) for _ in range(NCTX - 1)]

        # H2D on the context's own stream, in front of its ingest and launch sequence: the upload of step i+1 runs beside the other
        # context's kernels of step i and the host waits for nothing.  (A copy stream of the caller's + hak_wait_event does the
        # same with one more stream; measured 8 % slower here, because the HIP runtime multiplexes all streams of the process
        # onto 4 hardware queues and the marker behind a 14 ms copy holds up whichever launch chain shares its queue -- DESIGN 7.)
        # d_u8s[k] is free again when step i-2's results have been downloaded, which Pipeline.run does before pre(k).
        up_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NCTX - 1)]
        for k in range(1, NCTX):
            ah.check(ah.lib.hak_set_stream(pipe.dets[k].ctx, C.c_void_p(up_streams[k].cuda_stream)))

        def up(k):
            with torch.cuda.stream(up_streams[k]):
                d_u8s[k].copy_(h_u8, non_blocking=True)
            ah.check(ah.lib.hak_ingest_u8(pipe.dets[k].ctx, d_u8s[k].data_ptr(), h * w, w, d_imgs_k[k].data_ptr(), h * p, p, w, h, nimg))

        def up_jobs(n):                                      # job i runs on context i % NCTX (Pipeline.run)
            return [(d_imgs_k[i % NCTX], c) for i, c in enumerate(chunks * n)]
        pipe.run(up_jobs(2), pre=up)
        fence(use_dist)
        tu = time.perf_counter()
        pipe.run(up_jobs(args.steps), pre=up)
        fence(use_dist)
        el_u = max_over_ranks(time.perf_counter() - tu, dist_dev, use_dist)
        upload_rate = (args.total_pairs if strong else world * B) * args.steps / el_u
        torch.cuda.synchronize()
        for k in range(1, NCTX):                              # back to the contexts' own streams before the torch streams go away
            ah.check(ah.lib.hak_set_stream(pipe.dets[k].ctx, None))
        del h_u8, d_u8s, d_imgs_k, up_streams

    # ---- the integer FAST path on the same pairs (secondary figure, never `value`; default at N = 1)
    fast_rate, fast_results = None, []          # fast_results: per-context slot digests of the FAST leg's last downloads
    if (args.fast or world == 1) and not args.no_fast and my_pairs:
        d_fu8 = torch.from_numpy(np.stack([np.pad(u8_pairs[seed_of(i // 2)][i % 2], ((0, 0), (0, p - w))) for i in range(nimg)])).cuda()

        def fast_run(n):
            jobs = step_jobs * n
            for i, (_, c) in enumerate(jobs):
                pipe.enqueue(i % NCTX, d_fu8, c, fast=True)
                pipe.last_pairs[i % NCTX] = c
                if i >= NCTX - 1:
                    j = i - (NCTX - 1)
                    pipe.download(j % NCTX, jobs[j][1])
            for j in range(max(0, len(jobs) - (NCTX - 1)), len(jobs)):
                pipe.download(j % NCTX, jobs[j][1])
        fast_run(2)
        fence(use_dist)
        tf = time.perf_counter()
        fast_run(args.steps)
        fence(use_dist)
        fast_rate = (args.total_pairs if strong else world * B) * args.steps / max_over_ranks(time.perf_counter() - tf, dist_dev, use_dist)
        for k in range(NCTX):
            if pipe.last_pairs[k]:
                ck, pk = pipe.results(k)
                fast_results.append(slot_digests(ck, pk, pipe.last_pairs[k]))
        del d_fu8

    # ---- SURVEY 8e's verification gather: an untimed pass over this rank's pairs of one step, one 32-byte summary per pair,
    # all-gathered (the path's only collective); rank 0 checks the table
    local_rows = []
    c0 = 0
    for c in chunks if my_pairs else []:
        pipe.enqueue(0, d_imgs, c)
        pipe.download(0, c)
        ck, pk = pipe.results(0)
        if os.environ.get("HAK_BENCH_CORRUPT_RANK") == str(rank) and c0 == 0 and ck[0] > 0:
            # test hook (tests/test_gpu_bench.py): one descriptor bit of this rank's first pair flipped AFTER the download -- rank 0's
            # table checks must catch it (exit code 3)
            pk = pk.copy()
            pk["features"][0, 0, 0] ^= 1
        local_rows.append(summarize_pairs(ck, pk, [lo + c0 + k for k in range(c)]))
        c0 += c
    table = gather_pair_summaries(np.concatenate(local_rows) if local_rows else np.zeros((0, 4), np.int64), dist_dev, use_dist)
    golden_tab = None
    if os.path.exists(CHECKSUM_FILE) and args.octaves == 4 and not args.upright:
        golden_tab = json.load(open(CHECKSUM_FILE)).get(f"{w}x{h}")
    gather = check_pair_table(table, args.total_pairs if strong else world * B, golden_tab) if rank == 0 else None

    # ---- roofline leg: the same launch sequence strictly serially on one stream with per-launch HIP events
    # (per-kernel durations are only meaningful without overlap; the timed region overlaps the octaves on separate streams)
    roof = None
    if not args.no_roofline and my_pairs and rank == 0:
        det = pipe.dets[0]
        ah.check(ah.lib.hak_set_concurrency(det.ctx, 0))
        ah.check(ah.lib.hak_prof_reset(det.ctx))
        ah.check(ah.lib.hak_prof_enable(det.ctx, 1))
        nprof = max(1, min(args.steps, 3))
        rl_pairs = chunks[0]                # the per-class legs run the launch sequence of the timed region (256 pairs by default)
        # one untimed sequence first, like the timed region's warm-up: the first serial sequence after the switch creates the leg's
        # ~400 events between its launches and reads 0.6-0.8 ms (4 %) more FED time than the ones after it (DESIGN.md 4 lesson 34)
        pipe.enqueue(0, d_imgs, rl_pairs)
        pipe.download(0, rl_pairs)
        m2, n2 = C.c_double(), C.c_int()
        ah.check(ah.lib.hak_prof_read(det.ctx, ah.PROF["fed"], C.byref(m2), C.byref(n2)))
        fed_warm = round(m2.value, 4)
        for k in ah.PROF.values():
            ah.check(ah.lib.hak_prof_read(det.ctx, k, None, None))      # (collects and releases every class's events of that sequence)
        ah.check(ah.lib.hak_prof_reset(det.ctx))
        fed_seq, fed_cum = [], 0.0          # the FED class per profiled sequence (its run-to-run spread is part of the result)
        for _ in range(nprof):
            pipe.enqueue(0, d_imgs, rl_pairs)
            pipe.download(0, rl_pairs)
            m2, n2 = C.c_double(), C.c_int()
            ah.check(ah.lib.hak_prof_read(det.ctx, ah.PROF["fed"], C.byref(m2), C.byref(n2)))
            fed_seq.append(round(m2.value - fed_cum, 4))
            fed_cum = m2.value
        cls_ms, cls_n = {}, {}
        for name, k in ah.PROF.items():
            m2, n2 = C.c_double(), C.c_int()
            ah.check(ah.lib.hak_prof_read(det.ctx, k, C.byref(m2), C.byref(n2)))
            cls_ms[name], cls_n[name] = m2.value / nprof, n2.value // nprof
        ah.check(ah.lib.hak_prof_enable(det.ctx, 0))
        ah.check(ah.lib.hak_set_concurrency(det.ctx, 0 if pipe.serial else 1))
        tr = det.traffic(int(round(gather["keypoints"] / max(1, 2 * gather["pairs"]))))
        nim = 2 * rl_pairs
        gb = C.c_double()
        ah.check(ah.lib.hak_op_copy_probe(2 << 30, 10, C.byref(gb)))           # 2 GiB source + 2 GiB destination, far beyond the caches
        copy_gbs = gb.value
        # PMC passes (tools/pmc_traffic.py): bytes per launch sequence of `pmc_pairs` pairs, valid only for the sources they ran on
        pmc, pmc_src = {}, None
        if live_pmc is not None:
            pmc = {k: v * nim / float(2 * live_pmc[1]) for k, v in live_pmc[0].items()}
            pmc_src = ("two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this launch sequence in child processes of THIS run, on this box "
                       "(bench.run_pmc_legs); durations from this run's HIP events")
        elif os.path.exists(PMC_FILE) and (w, h, args.octaves) == (1920, 1080, 4):
            tj = json.load(open(PMC_FILE))
            for k, v in tj.get("classes", {}).items():
                if v.get("source_sha") == class_source_hash(k):
                    pmc[k] = v["hbm_bytes_per_sequence"] * nim / float(2 * tj["pairs_per_launch_sequence"])
            pmc_src = ("profiles/" + os.path.basename(PMC_FILE) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on the builder's box, "
                       "source-hash-guarded; durations are this run's)")
        algo = {"fed": tr.fed_fused_bytes, "hessian": tr.hessian_bytes, "prologue": tr.prologue_bytes,
                "describe": tr.describe_bytes, "nms": tr.nms_bytes, "match": None}
        prof_of = {"fed": ("fed",), "hessian": ("hessian",), "prologue": ("contrast",), "describe": ("describe",), "nms": ("nms",),
                   "match": ("match",)}
        classes = []
        for k in ("fed", "hessian", "describe", "prologue", "nms", "match"):
            ms = sum(cls_ms[c] for c in prof_of[k])
            ab = None if algo[k] is None else algo[k] * nim
            row = {"class": k, "ms": round(ms, 4), "launches": sum(cls_n[c] for c in prof_of[k]),
                   "algorithmic_bytes": None if ab is None else round(ab), "pmc_bytes": None if k not in pmc else round(pmc[k]),
                   "frac_peak": None if ab is None or ms <= 0 else round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "frac_copy": None if ab is None or ms <= 0 else round(ab / (ms * 1e-3) / 1e9 / copy_gbs, 4),
                   "pmc_frac_peak": None if k not in pmc or ms <= 0 else round(pmc[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if k == "match" and ms > 0 and len(table):
                # the matcher alone runs on the matrix cores: 32 x 32 tiles of distances, 512 fp4 multiply-adds each (8 x
                # v_mfma_f32_32x32x64_f8f6f4); counts of this rank's first rl_pairs pairs from the gathered table (same seeds as the leg)
                n12 = table[:rl_pairs, 1]
                tiles = ((n12 >> 32) + 31) // 32 * (((n12 & 0xFFFFFFFF) + 31) // 32)
                ops = float(tiles.sum()) * 1024 * 512 * 2 * (rl_pairs / float(max(1, len(n12))))
                row.update({"bound": "mfma", "matrix_ops": round(ops), "matrix_TOPs": round(ops / (ms * 1e-3) / 1e12, 1),
                            "matrix_peak_TOPs": MFMA_FP4_PEAK_TOPS, "frac_matrix_peak": round(ops / (ms * 1e-3) / 1e12 / MFMA_FP4_PEAK_TOPS, 4)})
            classes.append(row)
        other_ms = sum(cls_ms[c] for c in ("lowpass", "flow", "down", "extrema"))
        fed_ms, fed_n = cls_ms["fed"], max(1, cls_n["fed"])
        avg_s = fed_ms * 1e-3 / fed_n
        bytes_per_launch = tr.fed_fused_bytes * nim / fed_n
        achieved = bytes_per_launch / avg_s / 1e9
        traffic = None if "fed" not in pmc else pmc["fed"] / fed_n
        unfused_GBs = tr.fed_bytes * nim / fed_n / avg_s / 1e9
        total_pairs_timed = (args.total_pairs if strong else world * B) * args.steps
        roof = {"kernel": "FED family k_fed_sf<NS> / k_fed_multi<NS>: NS <= 4 explicit steps per launch (k_fed_sf also the sublevel's "
                          "low-pass and conductivity); bytes = read L (+ g), write L' (+ smooth, + g) of every launch",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None if traffic is None else round(traffic),
                # where `traffic` comes from: counter passes of this run's own child processes, or (when rocprofv3 is not available /
                # a pass failed) the committed passes of the builder's box, accepted only while the hash of the class's sources matches
                "traffic_source": None if traffic is None else pmc_src,
                "traffic_GBs": None if traffic is None else round(traffic / avg_s / 1e9, 1),
                "traffic_frac": None if traffic is None else round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                "copy_ceiling_GBs": round(copy_gbs, 1), "frac_copy": round(achieved / copy_gbs, 4),
                "bytes_per_launch": round(bytes_per_launch), "avg_launch_us": round(avg_s * 1e6, 3),
                "launches_per_step": fed_n, "fed_ms_by_sequence": fed_seq, "fed_ms_warmup_sequence": fed_warm,
                # the unfused model of SURVEY 8d (12 B/px per STEP + 16 B/px of low-pass and conductivity) priced at the same time:
                # how much HBM traffic temporal fusion removed -- a gain, not a utilisation
                "unfused_model_GBs": round(unfused_GBs, 1), "fusion_gain": round(tr.fed_bytes / tr.fed_fused_bytes, 3),
                "classes": classes, "other_classes_ms": round(other_ms, 4),
                "serial_ms_per_step": round(sum(cls_ms.values()), 3),
                "end_to_end": {"all_stage_bytes_per_image_unfused_model": round(tr.all_stage_bytes),
                               "unfused_model_GBs": round(tr.all_stage_bytes * 2.0 * total_pairs_timed / elapsed / 1e9, 1),
                               # the whole job against the roofline: counter bytes of EVERY class (PMC passes) x the images of the
                               # timed region / its wall time -- what the pipelined run actually pulls from HBM, all ranks together
                               **({} if len(pmc) < 6 else (lambda bpi: {
                                   "pmc_bytes_per_image": round(bpi),
                                   "pmc_traffic_GBs": round(bpi * 2.0 * total_pairs_timed / elapsed / 1e9, 1),
                                   "pmc_frac_peak": round(bpi * 2.0 * total_pairs_timed / elapsed / 1e9 / (HBM_PEAK_GBS * world), 4),
                                   "pmc_frac_copy": round(bpi * 2.0 * total_pairs_timed / elapsed / 1e9 / (copy_gbs * world), 4)})(
                                       sum(pmc.values()) / nim))},
                "mode": "serial leg (one stream, HIP events per launch); rocprof counterpart: profiles/r05_*_serial_kernel_stats.csv"}

    # ---- the oracle legs (rank 0): the oracle on ALL distinct pairs, every slot of every context's last download of the timed
    # region and every pair of the gathered table against it, the FAST leg the same way; then the CPU baseline from the same runs
    verified, cpu, okz = None, None, None
    if rank == 0 and my_pairs and not args.no_verify:
        okz, cores, flags = oracle_setup()
        if args.octaves != 4 or args.upright:
            verified = {"images": 0, "note": "verification covers the default configuration only"}
        else:
            want, first_times = oracle_pairs(okz, synth, u8_pairs, w, p, max_pts, NDIST)
            want_dig = [pair_digest(a, b) for a, b in want]
            nf = first_slots[2]
            pts_ok, m_ok = verify_batch([want[seed_of(k)] for k in range(nf)], first_slots[1], first_slots[0])
            bad = sum(count_slot_mismatches(d, want_dig, lo) for d in timed_digests)
            nslots = sum(len(d) for d in timed_digests)
            tab_bad = int(sum(int(r[3]) != want_dig[int(r[0]) % NDIST] for r in table))
            verified = {"images": 2 * NDIST, "distinct_pairs": NDIST, "points_equal": pts_ok, "matches_equal": m_ok,
                        "slots": nslots, "contexts": len(timed_digests), "slots_equal": bad == 0 and pts_ok and m_ok,
                        "gathered_pairs": int(len(table)), "gathered_pairs_equal": tab_bad == 0,
                        "checker": "CPU oracle (oracle/akaze_oracle.c) on all distinct seeded pairs; every pair slot of each pipeline "
                                   "context's last timed download and every gathered pair summary digested (sha256 of all record fields) "
                                   "against the oracle's records of its seed",
                        "fast": None}
            if fast_results:
                fwant = oracle_fast_pairs(okz, u8_pairs, max_pts)
                fdig = [pair_digest(a, b) for a, b in fwant]
                fbad = sum(count_slot_mismatches(d, fdig, lo) for d in fast_results)
                verified["fast"] = {"images": 2 * NDIST, "slots": sum(len(d) for d in fast_results), "slots_equal": fbad == 0,
                                    "checker": "oracle/akaze_oracle_fast.c"}
            if world == 1 and not args.no_cpu_baseline and (w, h) == (1920, 1080):
                cpu = cpu_baseline(okz, synth, cores, flags, u8_pairs, w, p, max_pts, first_times)

    nseq_total = pipe.nseq
    pipe.close()
    del d_imgs
    extra = None
    if rank == 0 and world == 1 and not args.no_configs and not strong and (w, h, args.octaves) == (1920, 1080, 4):
        torch.cuda.empty_cache()
        extra = other_configs(ah, synth, args, rank, okz, u8_pairs)

    if rank == 0:
        total_pairs = (args.total_pairs if strong else world * B) * args.steps
        out = {
            "metric": f"pairs_per_sec_detect_describe_match_{h}p", "value": round(total_pairs / elapsed, 2),
            "unit": "pairs/s", "n_gpus": world, "rccl_ranks": dist.get_world_size() if use_dist and backend == "nccl" else 0,
            **({"rehearsal": f"HAK_BENCH_DIST_BACKEND=gloo: {world} ranks share {torch.cuda.device_count()} device(s); collectives over gloo"}
               if backend == "gloo" and use_dist else {}),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (+ configs.natural_1080p: img1 / img2 reconstructed from the reference's result pictures)" if extra and extra.get("natural_1080p") else ""),
            "config": {"workload": ("configs[1]: " if (w, h, args.octaves) == (1920, 1080, 4) else "configs[3] shape: " if (w, h) == (1280, 720) else
                                    "configs[2]: " if (w, h, args.octaves) == (3840, 2160, 5) else "") +
                                   f"{w}x{h} grayscale pairs, {args.octaves} octaves x 4 sublevels, PM_G2, "
                                   "MLDB-486, max_pts 10000, float path; detect+describe both images + match, D2H included",
                       "pairs_per_step_per_gpu": my_pairs if strong else B, "pairs_per_launch_sequence": B,
                       "total_pairs_per_step": args.total_pairs if strong else world * B, "distinct_pairs_per_gpu": NDIST,
                       "float_sequences_enqueued": nseq_total,
                       "octave_streams": "one stream per context" if pipe.serial else "concurrent", "step_pipeline": NCTX,
                       "sharding": "independent pairs per rank, no data-path collective", "rank0_gpu_numa_node": numa_node,
                       "keypoints_per_image": round(gather["keypoints"] / max(1.0, 2.0 * gather["pairs"]), 1),
                       "matches_per_pair": round(gather["matches"] / max(1.0, float(gather["pairs"])), 1)},
            "verified": verified, "gather": gather, "roofline": roof, "cpu_baseline": cpu, "configs": extra,
            "upload_inclusive_pairs_per_s": None if upload_rate is None else round(upload_rate, 1),
            "fast_path_pairs_per_s": None if fast_rate is None else round(fast_rate, 1),
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        ok = True
        if verified and verified.get("images"):
            ok = verified["slots_equal"] and verified["gathered_pairs_equal"] and (verified["fast"] is None or verified["fast"]["slots_equal"])
        if gather and not (gather["complete"] and gather["equal_seed_equal_checksum"] and gather["equals_g1_table"] is not False):
            ok = False
        if not ok:
            print("bench.py: results differ from the oracle / between ranks / from the committed G = 1 table", file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
