#!/usr/bin/env python3
"""BASELINE configs[4]: 10k x 10k brute-force Hamming through hak_match (1-NN, reference accept rule) and hak_match_knn2 (2-NN +
ratio test + cross-check + compaction): synchronous-call time (what cuMatch's contract costs) and, under
`rocprofv3 --kernel-trace --stats`, the kernels alone.  GPU box only.
  python tools/match10k.py [--ctx] [--iters N]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import akaze_hip as ah  # noqa: E402
from akaze_hip import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", action="store_true", help="through a context (its scratch) instead of ctx == NULL")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--n", type=int, default=10000)
    args = ap.parse_args()
    n = args.n
    q = synth.random_descriptors(n, 7, ah.POINT_DTYPE)
    t = synth.random_descriptors(n, 8, ah.POINT_DTYPE, planted_from=q, nplanted=4000)
    dq = torch.from_numpy(q.view(np.uint8).copy()).cuda()
    dt = torch.from_numpy(t.view(np.uint8).copy()).cuda()
    ctx = None
    det = None
    if args.ctx:
        det = ah.Akazer()
        det.init((320, 240, 384), max_pts=n)
        ctx = det.ctx
    out = {}

    def timed(fn, name):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        out[name] = round((time.perf_counter() - t0) * 1e3 / args.iters, 4)

    timed(lambda: ah.check(ah.lib.hak_match(ctx, dq.data_ptr(), n, dt.data_ptr(), n, None)), "match_10k_ms")
    d_out = torch.zeros(n * ah.MATCH_PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    cnt = C.c_int(0)
    timed(lambda: ah.check(ah.lib.hak_match_knn2(ctx, dq.data_ptr(), n, dt.data_ptr(), n, 4, 5, 1, 0, None, d_out.data_ptr(), C.byref(cnt), None)),
          "match_knn2_10k_ms")
    out["knn2_accepted"] = int(cnt.value)
    m = np.frombuffer(dq.cpu().numpy().tobytes(), ah.POINT_DTYPE)
    out["accepted"] = int((m["match"] >= 0).sum())
    print(json.dumps(out))
    if det:
        det.close()


if __name__ == "__main__":
    main()
