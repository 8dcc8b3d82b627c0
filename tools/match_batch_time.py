"""match_batch_time.py: hak_match_batch (1-NN, reference accept rule) and hak_match_knn2_batch on a detected batch of 1080p pairs --
the batched matcher alone, ms per 256 pairs (the bench line's `match` class)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import numpy as np, torch
import akaze_hip as ah
from akaze_hip import synth
w, h, mp, B = 1920, 1080, 10000, 64
p = ah.iAlignUp(w, 128)
pairs = [synth.pair(w, h, 1 + i) for i in range(8)]
host = np.stack([synth.to_float(pairs[(i // 2) % 8][i % 2], p) for i in range(2 * B)])
d = torch.from_numpy(host).cuda()
det = ah.Akazer(); det.init((w, h, p), max_pts=mp, batch=2 * B)
pts = torch.zeros(2 * B * mp * 104, dtype=torch.uint8, device="cuda"); num = torch.zeros(2 * B, dtype=torch.int32, device="cuda")
ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, d.data_ptr(), h * p, p, 2 * B, pts.data_ptr(), num.data_ptr(), 1))
out = torch.zeros(B * mp * ah.MATCH_PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda"); cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
ah.check(ah.lib.hak_sync(det.ctx))
def timed(fn, n=20):
    for _ in range(3): fn()
    ah.check(ah.lib.hak_sync(det.ctx)); t0 = time.perf_counter()
    for _ in range(n): fn()
    ah.check(ah.lib.hak_sync(det.ctx)); return (time.perf_counter() - t0) * 1e3 / n * 256 / B
m1 = timed(lambda: ah.check(ah.lib.hak_match_batch(det.ctx, pts.data_ptr(), num.data_ptr(), B)))
m2 = timed(lambda: ah.check(ah.lib.hak_match_knn2_batch(det.ctx, pts.data_ptr(), num.data_ptr(), B, 4, 5, 1, 0, out.data_ptr(), cnt.data_ptr())))
print(f"hak_match_batch {m1:.3f} ms per 256 pairs of ~{int(num.float().mean())} keypoints; hak_match_knn2_batch {m2:.3f} ms per 256 pairs")
