"""Stress run of the sliced matcher's cross-block hand-off (kernels_match.hip: per-slice summaries -> ticket -> finishing block): the
batched device-count search of hak_match_batch, thousands of launches per shape, every launch compared with the first launch's result
and that with the VALU kernel (which has no hand-off).  `python tests/stress_handoff.py [--iters N]` on the GPU box; a stale summary
shows up as a query block whose matches ignore one slice.  Test infrastructure."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("cuda-akaze_amd", ""):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)


def run(iters, shapes=((1, 2400, 10000), (4, 2000, 2000), (4, 1300, 3000), (8, 530, 3000), (8, 2900, 3000), (12, 700, 1000)), out=sys.stdout):
    import torch
    import akaze_hip as ah
    from akaze_hip import synth
    bad_total = 0
    for npairs, n, mp in shapes:
        det = ah.Akazer()
        det.init((320, 240, 384), max_pts=mp, batch=2 * npairs)
        host = np.zeros((2 * npairs, mp), ah.POINT_DTYPE)
        num = np.zeros(2 * npairs, np.int32)
        for k in range(npairs):
            n1, n2 = n - 13 * k, n - 29 * k
            train = synth.random_descriptors(n2, 700 + k, ah.POINT_DTYPE)
            host[2 * k, :n1] = synth.random_descriptors(n1, 800 + k, ah.POINT_DTYPE, planted_from=train, nplanted=n1 // 3, maxflip=45)
            host[2 * k + 1, :n2] = train
            num[2 * k], num[2 * k + 1] = n1, n2
        src = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).cuda()
        d_num = torch.from_numpy(num).cuda()
        d_pts = src.clone()
        os.environ["HAK_MATCH_VALU"] = "1"
        ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), npairs))
        ah.check(ah.lib.hak_sync(det.ctx))
        ref = d_pts.clone()
        os.environ["HAK_MATCH_VALU"] = "0"
        bad = 0
        for it in range(iters):
            d_pts.copy_(src)
            torch.cuda.synchronize()
            # the summaries of the launch before are the very values this launch writes: a stale read would go unseen without this
            ah.check(ah.lib.hak_debug_fill_match_scratch(det.ctx, (0x00, 0xFF, 0x5A)[it % 3]))
            ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), npairs))
            ah.check(ah.lib.hak_sync(det.ctx))
            if not torch.equal(d_pts, ref):
                bad += 1
                if bad <= 3:
                    g = np.frombuffer(d_pts.cpu().numpy().tobytes(), ah.POINT_DTYPE).reshape(2 * npairs, mp)
                    r = np.frombuffer(ref.cpu().numpy().tobytes(), ah.POINT_DTYPE).reshape(2 * npairs, mp)
                    rows = [(k, np.nonzero(g[2 * k]["match"] != r[2 * k]["match"])[0]) for k in range(npairs)]
                    print("   launch", it, [(k, len(ix), ix[:3].tolist()) for k, ix in rows if len(ix)], file=out, flush=True)
        print(f"{npairs:2d} pairs of ~{n} x {n} (capacity {mp}): {bad} of {iters} launches differ from the VALU kernel's result", file=out, flush=True)
        bad_total += bad
        det.close()
    os.environ.pop("HAK_MATCH_VALU", None)
    return bad_total


def run_fresh(iters, npairs=8, n=530, mp=3000, out=sys.stdout):
    """the FIRST sliced launch of a context: the scratch (tickets zeroed) is allocated in front of the kernel -- a new context per launch"""
    import torch
    import akaze_hip as ah
    from akaze_hip import synth
    host = np.zeros((2 * npairs, mp), ah.POINT_DTYPE)
    num = np.zeros(2 * npairs, np.int32)
    for k in range(npairs):
        n1, n2 = n - 13 * k, n - 29 * k
        train = synth.random_descriptors(n2, 700 + k, ah.POINT_DTYPE)
        host[2 * k, :n1] = synth.random_descriptors(n1, 800 + k, ah.POINT_DTYPE, planted_from=train, nplanted=n1 // 3, maxflip=45)
        host[2 * k + 1, :n2] = train
        num[2 * k], num[2 * k + 1] = n1, n2
    src = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).cuda()
    d_num = torch.from_numpy(num).cuda()
    d_pts = src.clone()
    imgs = torch.rand(2 * npairs, 240, 384, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    scratch_pts = torch.zeros_like(src)
    scratch_num = torch.zeros(2 * npairs, dtype=torch.int32, device="cuda")
    ref, bad = None, 0
    for it in range(iters + 1):
        # small device allocations are carved from recycled fragments that keep their old contents: fill a few hundred of them with
        # garbage and free them, as any other work in the process would have (a scratch recycled from the previous iteration's
        # scratch would hold the zeros its last kernel left behind and hide a clear that is missing or late)
        ptrs = []
        for sz in (2656, 2656, 4096, 10624, 1 << 16):
            for _ in range(40):
                q = C.c_void_p()
                assert hip.hipMalloc(C.byref(q), C.c_size_t(sz)) == 0
                hip.hipMemset(q, 0x7F, C.c_size_t(sz))
                ptrs.append(q)
        hip.hipDeviceSynchronize()
        for q in ptrs:
            hip.hipFree(q)
        det = ah.Akazer()
        det.init((320, 240, 384), max_pts=mp, batch=2 * npairs)
        os.environ["HAK_MATCH_VALU"] = "1" if it == 0 else "0"
        d_pts.copy_(src)
        # a detect sequence first, as a caller would have run: the context's octave streams exist and have been used
        ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, imgs.data_ptr(), 240 * 384, 384, 2 * npairs, scratch_pts.data_ptr(), scratch_num.data_ptr(), 1))
        ah.check(ah.lib.hak_sync(det.ctx))
        torch.cuda.synchronize()
        ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), npairs))
        ah.check(ah.lib.hak_sync(det.ctx))
        if it == 0:
            ref = d_pts.clone()
        elif not torch.equal(d_pts, ref):
            bad += 1
        det.close()
    os.environ.pop("HAK_MATCH_VALU", None)
    print(f"first sliced launch of a fresh context, {npairs} pairs of ~{n} x {n}: {bad} of {iters} launches differ from the VALU kernel's result",
          file=out, flush=True)
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--fresh", type=int, default=1500, help="launches with a new context each")
    a = ap.parse_args()
    sys.exit(1 if run_fresh(a.fresh) + run(a.iters) else 0)
