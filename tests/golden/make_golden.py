"""Regenerates the committed fixtures under tests/golden/ (run in the authoring container only).

  fed_tau.json            FED tau tables produced by the REFERENCE's own fed.cpp (oracle/_ref/libfedref.so,
                          compiled from /root/reference/fed.cpp), as uint32 bit patterns
  left_right_u8.npz       the reference's only bundled input images (data/left.pgm, right.pgm; 1280x960 P5)
                          as uint8 arrays -- data, not source
  left_right_oracle.npz   oracle result on them (demo parameters main.cpp:156-166): points + match fields
  synth_oracle.npz        oracle results on small seeded synthetic scenes (several sizes / parameter sets)
  fast_oracle.npz         integer FAST-path oracle (akaze_oracle_fast.c) results on left/right and two synthetic scenes
  bench_pair_checksums.json  the G = 1 table of bench.py's verification gather: per seed index of the bench's NDIST
                          synthetic pairs {n1 << 32 | n2, n_matches, 64-bit checksum of the records}, from the ORACLE, for the
                          1080p (configs[1]) and 720p (configs[3]) workloads
  ref_recon_1080p_u8.npz, ref_render_report.json   written by tools/ref_render_check.py (not by this script)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "cuda-akaze_amd"))
import okz  # noqa: E402
from akaze_hip import synth  # noqa: E402  (pure numpy module)

OUT = os.path.dirname(os.path.abspath(__file__))


def fed_golden():
    okz.build()
    assert okz.ref_lib() is not None, "needs /root/reference"
    ts = [0.530193, 0.749807, 1.060387, 1.499614, 2.120772, 2.999228, 4.241547, 5.998454, 8.483089, 11.996912,
          16.966187, 23.993816, 33.932358, 47.987648, 67.864746, 95.975266, 135.729431, 191.950592, 271.458984]
    ts += [float(np.float32(x)) for x in np.geomspace(0.01, 400.0, 60)]
    cases = []
    for T in ts:
        for reorder in (0, 1):
            tau = okz.ref_fed_tau(T, 1, 0.25, reorder)
            cases.append(dict(T=float(np.float32(T)), M=1, tau_max=0.25, reordering=reorder,
                              tau_bits=[int(v) for v in tau.view(np.uint32)]))
    json.dump(dict(source="reference fed.cpp via oracle/_ref/libfedref.so", cases=cases),
              open(os.path.join(OUT, "fed_tau.json"), "w"))
    print("fed_tau.json", len(cases), "cases")


def read_pgm(path):
    from PIL import Image
    return np.asarray(Image.open(path)).copy()


def run_oracle(u8, **kw):
    h, w = u8.shape
    p = (w + 127) // 128 * 128
    return okz.detect_and_compute(synth.to_float(u8, p), w, okz.default_params(**kw))


def pgm_golden():
    left = read_pgm("/root/reference/data/left.pgm")
    right = read_pgm("/root/reference/data/right.pgm")
    np.savez_compressed(os.path.join(OUT, "left_right_u8.npz"), left=left, right=right)
    r1, r2 = run_oracle(left), run_oracle(right)
    okz.match(r1.points, r2.points)
    np.savez_compressed(os.path.join(OUT, "left_right_oracle.npz"), pts1=r1.points, pts2=r2.points,
                        kc=np.array([r1.kcontrast, r2.kcontrast], np.float32))
    print("left/right", len(r1.points), len(r2.points), "accepted matches", int((r1.points["match"] >= 0).sum()))


SYNTH_CASES = [
    # name, w, h, seed, params
    ("s320x240", 320, 240, 11, {}),
    ("s211x173", 211, 173, 12, {}),                       # odd sizes, 2 octaves survive the 80 px rule
    ("s640x360_o3", 640, 360, 13, dict(noctaves=3)),
    ("s400x300_upright", 400, 300, 14, dict(upright=1)),
    ("s400x300_charb", 400, 300, 14, dict(diffusivity=3)),
]


def case_scene(w, h, seed):
    """small golden scenes are drawn denser than the bench scenes so they hold enough keypoints"""
    return synth.scene(w, h, seed, nshapes=30 + int(700 * (w * h) / (1920.0 * 1080.0)))


def synth_golden():
    out = {}
    for name, w, h, seed, kw in SYNTH_CASES:
        r = run_oracle(case_scene(w, h, seed), **kw)
        out[name + "_pts"] = r.points
        out[name + "_kc"] = np.array([r.kcontrast], np.float32)
        print(name, len(r.points))
    np.savez_compressed(os.path.join(OUT, "synth_oracle.npz"), **out)


def fast_golden():
    lr = np.load(os.path.join(OUT, "left_right_u8.npz"))
    out = {}
    for name in ("left", "right"):
        r = okz.fast_detect_and_compute(lr[name])
        out[name + "_pts"] = r.points
        out[name + "_kc"] = np.array([r.kcontrast], np.int32)
        print("fast", name, len(r.points), r.kcontrast)
    for name, w, h, seed, kw in SYNTH_CASES[:3]:
        r = okz.fast_detect_and_compute(case_scene(w, h, seed), okz.default_params(**kw))
        out[name + "_pts"] = r.points
        out[name + "_kc"] = np.array([r.kcontrast], np.int32)
        print("fast", name, len(r.points), r.kcontrast)
    np.savez_compressed(os.path.join(OUT, "fast_oracle.npz"), **out)


def bench_checksums():
    sys.path.insert(0, ROOT)
    import bench
    out = {}
    for w, h in ((1920, 1080), (1280, 720)):
        p = (w + 127) // 128 * 128
        rows = {}
        for i in range(bench.NDIST):
            a, b = synth.pair(w, h, 1 + i)
            r1 = okz.detect_and_compute(synth.to_float(a, p), w).points
            r2 = okz.detect_and_compute(synth.to_float(b, p), w).points
            okz.match(r1, r2)
            rows[str(i)] = [(len(r1) << 32) | len(r2), int((r1["match"] >= 0).sum()), bench.pair_digest(r1, r2)]
            print(f"{w}x{h} seed {1 + i}: {len(r1)} / {len(r2)} keypoints, {rows[str(i)][1]} matches")
        out[f"{w}x{h}"] = rows
    json.dump(out, open(os.path.join(OUT, "bench_pair_checksums.json"), "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # e.g. `make_golden.py bench_checksums`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    fed_golden()
    pgm_golden()
    synth_golden()
    fast_golden()
