"""the bench's single-pair leg alone (Python harness, synchronous calls like main.cpp:199-209): median ms per pair, split"""
import os, sys, time, statistics
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "cuda-akaze_amd"))
import torch
import akaze_hip as ah
from akaze_hip import synth
w, h, mp = 1920, 1080, 10000
p = ah.iAlignUp(w, 128)
a, b = synth.pair(w, h, 1)
d1 = torch.from_numpy(synth.to_float(a, p)).cuda(); d2 = torch.from_numpy(synth.to_float(b, p)).cuda()
det = ah.Akazer(); det.init((w, h, p), max_pts=mp)
pinned = os.environ.get("PINNED", "1") == "1"
r1, r2 = ah.AkazeData(), ah.AkazeData()
ah.initAkazeData(r1, mp, True, True, pinned=pinned); ah.initAkazeData(r2, mp, True, True, pinned=pinned)
td, tm = [], []
for i in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    det.detectAndCompute(d1.data_ptr(), r1, (w, h, p), True)
    t1 = time.perf_counter()
    det.detectAndCompute(d2.data_ptr(), r2, (w, h, p), True)
    t2 = time.perf_counter()
    ah.cuMatch(r1, r2, det if os.environ.get("MATCH_CTX", "1") == "1" else None)
    t3 = time.perf_counter()
    if i >= 10:
        td.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3)); tm.append((t3 - t2) * 1e3)
print(f"detect1 {statistics.median(x[0] for x in td):.3f}  detect2 {statistics.median(x[1] for x in td):.3f}  match {statistics.median(tm):.3f} ms  "
      f"pair {statistics.median(x[0] + x[1] + m for x, m in zip(td, tm)):.3f} ms  kp {r1.num_pts}/{r2.num_pts}")
