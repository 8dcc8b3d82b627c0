"""the CPU oracle's rate on this host, outside bench.py: detect + describe of both images of one 1080p pair + match, median of N,
under the OpenMP environment of the caller (OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import numpy as np
import okz
from akaze_hip import synth
okz.build()
nat = okz.build_native(os.environ.get("TMPDIR", "/tmp"))
if nat:
    okz.load(nat)
a, b = synth.pair(1920, 1080, 1)
fa, fb = synth.to_float(a, 1920), synth.to_float(b, 1920)
ts = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    t = time.perf_counter()
    r1 = okz.detect_and_compute(fa, 1920).points
    r2 = okz.detect_and_compute(fb, 1920).points
    okz.match(r1, r2)
    ts.append(time.perf_counter() - t)
print(f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')} PROC_BIND={os.environ.get('OMP_PROC_BIND')} PLACES={os.environ.get('OMP_PLACES')}: "
      f"median {statistics.median(ts[1:]) * 1e3:.1f} ms per pair, min {min(ts) * 1e3:.1f} ({len(r1)} / {len(r2)} keypoints)")
