"""phase stamps of k_match_mfma's block (0, HAK_MM_TIMING) for one 10k x 10k call (variant build with -DHAK_MM_TIMING=<slice>):
HAK_LIB=build/ab/libhak_mmtime.so python tools/mm_timing.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import numpy as np, torch
import akaze_hip as ah
from akaze_hip import synth
n = 10000
q = synth.random_descriptors(n, 7, ah.POINT_DTYPE)
t = synth.random_descriptors(n, 8, ah.POINT_DTYPE, planted_from=q, nplanted=4000)
dq = torch.from_numpy(q.view(np.uint8).copy()).cuda(); dt = torch.from_numpy(t.view(np.uint8).copy()).cuda()
det = ah.Akazer(); det.init((320, 240, 384), max_pts=n)
raw = C.CDLL(ah.LIB_PATH)
for _ in range(20):
    ah.check(ah.lib.hak_match(det.ctx, dq.data_ptr(), n, dt.data_ptr(), n, None))
buf = (C.c_ulonglong * 64)()
raw.hak_debug_mm_times(buf)
v = np.array(list(buf), np.uint64).astype(np.int64)
t0 = v[0]
names = {0: "kernel start", 1: "query expanded, first chunks requested", 2: "first chunk staged + barrier", 40: "loop + tail done", 41: "partial stored + barrier", 42: "ticket + barrier"}
prev = t0
for i in range(64):
    if v[i] == 0: continue
    nm = names.get(i, f"chunk {(i - 3) // 2} {'computed' if (i - 3) % 2 == 0 else 'next staged + barrier'}")
    print(f"{i:3d} {nm:45s} {(v[i] - t0) / 100.0:9.2f} us   (+{(v[i] - prev) / 100.0:7.2f})")
    prev = v[i]

blk = (C.c_ulonglong * 2048)()
raw.hak_debug_mm_blocks(blk)
b = np.array(list(blk), np.uint64).astype(np.int64).reshape(2, 1024)
used = b[0] > 0
st, en = b[0][used], b[1][used]
z = st.min()
print(f"{used.sum()} blocks: starts {0:.2f} .. {(st.max() - z) / 100:.2f} us (median {(np.median(st) - z) / 100:.2f}), ends {(en.min() - z) / 100:.2f} .. {(en.max() - z) / 100:.2f} us (median {(np.median(en) - z) / 100:.2f}); block duration median {np.median(en - st) / 100:.2f}, max {(en - st).max() / 100:.2f}")
order = np.argsort(st)
print("start deciles:", [round(float((st[order][int(k * (len(st) - 1) / 10)] - z) / 100), 2) for k in range(11)])
print("end deciles:  ", [round(float((np.sort(en)[int(k * (len(en) - 1) / 10)] - z) / 100), 2) for k in range(11)])
