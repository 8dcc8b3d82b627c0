"""Randomised parity run: the HIP path against the CPU oracle on seeded random shapes, batch sizes, parameters and kernel
selections -- float and integer FAST pipelines, single-image and batch entry points, batched pair matching.

Test infrastructure (it loads the oracle as the checker).  `python tests/fuzz_parity.py --cases 300 --seed 5` on the GPU box prints one
line per case and a summary; tests/test_gpu_fuzz.py runs a short seeded slice of the same generator in the GPU suite.  Every case is a
pure function of (seed, index), so a failure line can be replayed alone with --only INDEX.

What a case draws (reference: main.cpp:156-166 for the parameter set, akaze.cpp:101-150 / 153-201 for the two entry points):
  extents   w in [80, 1500], h in [80, 900], a third of them snapped to the streaming kernels' strip / segment edges and to the
            contrast lattice's blind extents ((n - 1) % 32 == 0)
  batch     1, 2, 3, 5, 8 or 17 images per launch sequence (17 crosses the XCD groups of 8 images of hak_xcd_decode)
  kernels   the library's own size rule, the streaming kernels forced on, or one launch per sublevel (k_level_tile)
  params    octaves 1-5, sublevels 2-5, per, dthreshold, soffset, derivative factor, all four diffusivities, pattern size, upright,
            max_pts small enough to clamp in a quarter of the cases
  content   drawn scenes (tests/golden/make_golden.case_scene), optionally with uniform noise on top
  layout    the caller's images at pitch iAlignUp(w, 128) (main.cpp:174), dense, or at an odd pitch; 0 / 1 / 3 elements into their buffer;
            NaN (0xFF) everywhere outside the images
  legs      float batch -> 1-NN pair matching [-> 2-NN + ratio + cross-check] [-> the batch again on rolled images] [-> the pair call with a
            clamp of its own for image 2, pinned or pageable host arrays] -> the single-image call [-> FAST batch -> FAST single call]
"""
import argparse
import ctypes as C
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("cuda-akaze_amd", "oracle", ""):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)

KNOBS = ("HAK_HESS_STREAM", "HAK_FUSE_SF", "HAK_BASE_STREAM", "HAK_LEVEL_TILE", "HAK_GRAPH", "HAK_SERIAL")
# launch-order variants (read at hak_create, drawn last): the library's rule, every sequence as a replayed graph, never a graph, one stream
ORDERS = {"": {}, "graph always": {"HAK_GRAPH": "2"}, "no graph": {"HAK_GRAPH": "0"}, "one stream": {"HAK_SERIAL": "1"}}
MODES = {"size rule": {}, "streaming": {"HAK_HESS_STREAM": "2", "HAK_FUSE_SF": "2", "HAK_BASE_STREAM": "2", "HAK_LEVEL_TILE": "1"},
         "level tile": {"HAK_LEVEL_TILE": "2"}}
FIELDS = ("x", "y", "octave", "response", "size", "angle", "features")
MFIELDS = ("match", "distance", "match_x", "match_y")


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def draw_case(seed, index, big=False):
    rng = np.random.default_rng([seed, index, int(big)])
    pick = lambda seq: seq[int(rng.integers(len(seq)))]
    w, h = int(rng.integers(80, 1501)), int(rng.integers(80, 901))
    if big:
        # the bench's regime: 720p .. 4K frames, up to 65 images per launch sequence (full-height row segments, every streaming kernel)
        w, h = int(rng.integers(1000, 4201)), int(rng.integers(600, 2401))
        B = pick((1, 2, 4, 9, 33, 65))
        if big == 2:
            # beyond 4K: planes of 10-60 Mpx (the streaming kernels' 32-bit buffer offsets end at 2 GiB per plane group: their launchers
            # must hand such planes to the tile kernels), 5-7 octaves
            w, h = int(rng.integers(4200, 9001)), int(rng.integers(2400, 7001))
            B = pick((1, 1, 2))
            huge_pts = pick((150000, 300000))          # (no clamp: every keypoint of the frame is compared; such sets skip the matching legs)
        while w * h * B > 160e6:
            B = max(1, B // 2)
        kw = dict(noctaves=pick((3, 4, 4, 5)) + (2 if big == 2 else 0), max_scale=pick((3, 4, 4)), per=pick((0.5, 0.7, 0.7, 0.9)), dthreshold=pick((0.0005, 0.001, 0.001)),
                  soffset=pick((1.2, 1.6, 1.6, 2.0)), derivative_factor=pick((1.0, 1.5, 1.5, 2.0)), diffusivity=pick((1, 1, 1, 0, 2, 3)),
                  descriptor_pattern_size=pick((10, 10, 8, 12)), upright=bool(rng.random() < 0.2))
        c = dict(index=index, w=w, h=h, B=B, kw=kw, mode=pick(("size rule", "size rule", "streaming")), max_pts=pick((2000, 10000, 10000)),
                 noise=pick((0, 0, 6)), scene_seed=int(rng.integers(1 << 20)), fast=bool(rng.random() < 0.5), desc=True)
        if big == 2:
            c["max_pts"] = huge_pts
        return draw_more(c, rng, pick)
    if rng.random() < 0.33:
        # strip edges of the streaming kernels (240 / 232 / 248 stored columns per wave), 16-byte rows, lattice-blind extents
        w = max(80, pick((240, 480, 720, 960, 232, 464, 248, 496, 256, 512, 1024, 1280)) + pick((-1, 0, 1, 4, 5)))
    if rng.random() < 0.33:
        h = max(80, pick((96, 128, 135, 256, 270, 512, 540)) + pick((-1, 0, 1, 2)))
    if rng.random() < 0.15:
        w = 32 * int(rng.integers(3, 40)) + 1
    if rng.random() < 0.15:
        h = 32 * int(rng.integers(3, 25)) + 1
    B = pick((1, 1, 2, 2, 3, 5, 8, 17))
    if w * h * B > 12e6:
        B = max(1, int(12e6 // (w * h)))
    kw = dict(noctaves=pick((1, 2, 3, 4, 4, 5)), max_scale=pick((2, 3, 4, 4, 5)), per=pick((0.5, 0.7, 0.7, 0.9)),
              dthreshold=pick((0.0005, 0.001, 0.001, 0.003)), soffset=pick((1.2, 1.6, 1.6, 2.0)),
              derivative_factor=pick((1.0, 1.5, 1.5, 2.0, 2.5)), diffusivity=pick((1, 1, 1, 0, 2, 3)),
              descriptor_pattern_size=pick((10, 10, 6, 8, 12)), upright=bool(rng.random() < 0.2))
    c = dict(index=index, w=w, h=h, B=B, kw=kw, mode=pick(tuple(MODES)), max_pts=pick((150, 3000, 3000, 10000)),
             noise=pick((0, 0, 0, 6, 40)), scene_seed=int(rng.integers(1 << 20)), fast=bool(rng.random() < 0.6),
             desc=bool(rng.random() < 0.9))
    return draw_more(c, rng, pick)


def draw_more(c, rng, pick):
    """the legs added after the first long runs (drawn last, so that the earlier fields of a (seed, index) keep their values):
    knn2 = (ratio, cross-check) of the 2-NN search on the batch's pairs; pair = (clamp of image 2 as a fraction of max_pts, pinned host
    arrays) for the one-launch-sequence pair call; again = the batch a second time on rolled images (same buffers: a replayed graph)"""
    c["knn2"] = (pick(((1, 1), (4, 5), (3, 4))), bool(rng.random() < 0.5)) if rng.random() < 0.5 else None
    c["pair"] = (pick((1.0, 1.0, 0.5, 0.1)), bool(rng.random() < 0.5)) if rng.random() < 0.5 else None
    c["again"] = bool(rng.random() < 0.4)
    # the caller's image layout: pitch = iAlignUp(w, 128) as main.cpp:174 has it, a dense image (pitch = w), or an odd pitch; the first
    # image 0, 1 or 3 elements into its allocation (no 16-byte alignment of rows or base); margins hold NaN / 0xFF
    c["pitch_mode"] = pick((0, 0, 0, 1, 2, 3))
    c["in_offset"] = pick((0, 0, 0, 1, 3))
    # a last call through the same Akazer with OTHER extents (akaze.cpp:109: the arena is rebuilt when the size differs from init)
    c["resize"] = (int(rng.integers(1, 40)), int(rng.integers(1, 40))) if rng.random() < 0.2 else None
    c["order"] = pick(("", "", "", "graph always", "no graph", "one stream"))
    # uint8 images converted on the device (hak_ingest_u8 = main.cpp:149) in front of the float batch; results fetched with
    # hak_download_batch into pinned / pageable host arrays
    c["ingest"] = bool(rng.random() < 0.3)
    c["download"] = pick((None, None, "pinned", "pageable"))
    return c


def describe(c):
    kw = ",".join(f"{k}={v}" for k, v in c["kw"].items())
    return (f"#{c['index']:<4d} {c['w']:4d}x{c['h']:<4d} B={c['B']:<2d} {c['mode']:<10s} max_pts={c['max_pts']:<5d} noise={c['noise']:<2d} "
            f"{'fast ' if c['fast'] else ''}{'' if c['desc'] else 'nodesc '}{'knn2 ' if c['knn2'] else ''}{'again ' if c['again'] else ''}pitch{c['pitch_mode']}+{c['in_offset']} {'resize ' if c['resize'] else ''}{c['order'] + ' ' if c['order'] else ''}{'ingest ' if c['ingest'] else ''}{'dl-' + c['download'] + ' ' if c['download'] else ''}"
            f"{'pair(%.1f%s) ' % (c['pair'][0], ',pinned' if c['pair'][1] else '') if c['pair'] else ''}{kw}")


def diff_points(tag, got, want, fields):
    if len(got) != len(want):
        return [f"{tag}: {len(got)} keypoints, oracle {len(want)}"]
    out = []
    if len(got) == 0:
        return out
    for f in fields:
        a, b = got[f], want[f]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32), b.view(np.uint32)
        bad = np.nonzero((a != b).reshape(len(got), -1).any(axis=1))[0]
        if bad.size:
            out.append(f"{tag}: field {f} differs at {bad.size} of {len(got)} points (first {bad[:4].tolist()})")
    return out


def run_case(ah, okz, torch, synth, mg, c):
    """-> (list of failure lines, keypoints checked, matches checked)"""
    w, h, B, mp, kw = c["w"], c["h"], c["B"], c["max_pts"], c["kw"]
    p = ah.iAlignUp(w, 128)
    nd = min(B, 3)                                              # distinct scenes (the oracle runs once per scene)
    rng = np.random.default_rng(c["scene_seed"])
    u8s = []
    for i in range(nd):
        if w * h > 12e6:        # (huge frames: a 1600 x 1200 scene tiled with alternating mirror images -- the generator draws shape by shape in Python)
            t = mg.case_scene(1600, 1200, (c["scene_seed"] + i) % 9973)
            t = np.concatenate([t, t[:, ::-1]], axis=1)
            t = np.concatenate([t, t[::-1]], axis=0)
            u = np.tile(t, ((h + 2399) // 2400, (w + 3199) // 3200))[:h, :w].astype(np.int32)
        else:
            u = mg.case_scene(max(w, 134), h, (c["scene_seed"] + i) % 9973)[:, :w].astype(np.int32)     # (the scene generator's minimum width)
        if c["noise"]:
            u = u + rng.integers(-c["noise"], c["noise"] + 1, u.shape)
        u8s.append(np.clip(u, 0, 255).astype(np.uint8))
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    fails, npts, nmatch = [], 0, 0
    saved = {k: os.environ.pop(k, None) for k in KNOBS}
    os.environ.update(MODES[c["mode"]])
    os.environ.update(ORDERS[c["order"]])
    det = ah.Akazer()
    try:
        det.init((w, h, p), max_pts=mp, batch=B, **kw)
        d_pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
        d_num = torch.zeros(B, dtype=torch.int32, device="cuda")
        # ---- float path: batch entry point, then pair matching on the device records
        want = [okz.detect_and_compute(synth.to_float(u, p), w, okz.default_params(**okw), max_pts=mp, desc=c["desc"]).points for u in u8s]
        pin = {0: p, 1: w, 2: w + 1, 3: w + 3}[c["pitch_mode"]]
        off = c["in_offset"]

        def upload(order, as_u8):
            """the B images in `order` at pitch pin, the first one `off` elements into the buffer; everything outside the images is NaN / 0xFF"""
            hb = np.full(off + B * h * pin + 4, 0xFF if as_u8 else np.nan, np.uint8 if as_u8 else np.float32)
            for k, i in enumerate(order):
                v = hb[off + k * h * pin: off + (k + 1) * h * pin].reshape(h, pin)
                v[:, :w] = u8s[i % nd] if as_u8 else synth.to_float(u8s[i % nd], p)[:, :w]
            return torch.from_numpy(hb).cuda()

        stack = upload(range(B), False)
        img_ptr = lambda k: stack.data_ptr() + 4 * (off + k * h * pin)
        ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, img_ptr(0), h * pin, pin, B, d_pts.data_ptr(), d_num.data_ptr(), int(c["desc"])))
        ah.check(ah.lib.hak_sync(det.ctx))
        nums = d_num.cpu().numpy()
        allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
        fields = FIELDS if c["desc"] else FIELDS[:-1]
        for i in range(B):
            fails += diff_points(f"float batch image {i}", allp[i, :min(nums[i], mp)], want[i % nd], fields)
            npts += len(want[i % nd])
        small_sets = max(len(x) for x in want) <= 20000      # (the oracle's brute-force matcher is the limit)
        if B >= 2 and c["desc"] and small_sets and not fails:
            ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B // 2))
            ah.check(ah.lib.hak_sync(det.ctx))
            allm = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
            for k in range(B // 2):
                a, b = want[(2 * k) % nd].copy(), want[(2 * k + 1) % nd]
                okz.match(a, b)
                d = diff_points(f"match pair {k}", allm[2 * k, :len(a)], a, MFIELDS)
                if d:
                    g = allm[2 * k, :len(a)]
                    d.insert(0, f"match pair {k}: GPU accepts {int((g['match'] >= 0).sum())} of {len(a)} queries, oracle {int((a['match'] >= 0).sum())}; "
                                f"GPU match values in [{int(g['match'].min())}, {int(g['match'].max())}], distance in [{int(g['distance'].min())}, {int(g['distance'].max())}]")
                fails += d
                nmatch += int((a["match"] >= 0).sum())
            if fails:
                # diagnosis: the same launch again on the same context (the records are only read, the match fields rewritten)
                ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B // 2))
                ah.check(ah.lib.hak_sync(det.ctx))
                again = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
                ok2 = all(not diff_points("", again[2 * k, :len(want[(2 * k) % nd])], okz.match(want[(2 * k) % nd].copy(), want[(2 * k + 1) % nd]), MFIELDS)
                          for k in range(B // 2))
                fails.append(f"the same hak_match_batch launched a second time: {'equal to the oracle' if ok2 else 'still different'}; counts {nums.tolist()}")
        # ---- the batch's results through hak_download_batch (counts + the valid prefix of every image's records)
        if c["download"] and not fails:
            if c["download"] == "pinned":
                pp, pn = C.c_void_p(), C.c_void_p()
                ah.check(ah.lib.hak_host_alloc(C.byref(pp), B * mp * 104))
                ah.check(ah.lib.hak_host_alloc(C.byref(pn), B * 4))
                ah.check(ah.lib.hak_download_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B, pp, pn))
                hn = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_int)), shape=(B,)).copy()
                hp = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint8)), shape=(B * mp * 104,)).view(ah.POINT_DTYPE).reshape(B, mp).copy()
                ah.lib.hak_host_free(pp)
                ah.lib.hak_host_free(pn)
            else:
                hp, hn = np.zeros((B, mp), ah.POINT_DTYPE), np.full(B, -1, np.int32)
                ah.check(ah.lib.hak_download_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B, hp.ctypes.data, hn.ctypes.data))
            cur = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)          # (the records as they are now: the match leg has written its fields)
            for i in range(B):
                if hn[i] != nums[i] or hp[i, :nums[i]].tobytes() != cur[i, :nums[i]].tobytes():
                    fails.append(f"hak_download_batch ({c['download']}): image {i} differs from the device records")
        # ---- the same images as uint8, converted on the device (hak_ingest_u8: dst = (float)(src * (1.0 / 255.0)), main.cpp:149)
        if c["ingest"] and not fails:
            src8 = upload(range(B), True)
            fbuf = torch.full((B * h * p,), float("nan"), dtype=torch.float32, device="cuda")
            ah.check(ah.lib.hak_ingest_u8(det.ctx, src8.data_ptr() + off, h * pin, pin, fbuf.data_ptr(), h * p, p, w, h, B))
            ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, fbuf.data_ptr(), h * p, p, B, d_pts.data_ptr(), d_num.data_ptr(), int(c["desc"])))
            ah.check(ah.lib.hak_sync(det.ctx))
            nums3 = d_num.cpu().numpy()
            allp3 = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
            for i in range(B):
                fails += diff_points(f"device-side uint8 ingest, image {i}", allp3[i, :min(nums3[i], mp)], want[i % nd], fields)
        # ---- 2-NN + ratio + cross-check on the same device records
        if B >= 2 and c["desc"] and c["knn2"] and small_sets and not fails:
            ratio, cross = c["knn2"]
            out = torch.zeros((B // 2) * mp * 32, dtype=torch.uint8, device="cuda")
            cnt = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
            ah.check(ah.lib.hak_match_knn2_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B // 2, ratio[0], ratio[1], int(cross), 0, out.data_ptr(), cnt.data_ptr()))
            ah.check(ah.lib.hak_sync(det.ctx))
            cnts = cnt.cpu().numpy()
            allo = out.cpu().numpy().view(ah.MATCH_PAIR_DTYPE).reshape(B // 2, mp)
            allk = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
            for k in range(B // 2):
                a, b = want[(2 * k) % nd].copy(), want[(2 * k + 1) % nd]
                wl = okz.match_knn2(a, b, ratio, cross)
                if cnts[k] != len(wl):
                    fails.append(f"knn2 pair {k}: {cnts[k]} accepted, oracle {len(wl)}")
                    continue
                for f in ah.MATCH_PAIR_DTYPE.names:
                    if not np.array_equal(allo[k, :cnts[k]][f].view(np.uint32), wl[f].view(np.uint32)):
                        fails.append(f"knn2 pair {k}: list field {f} differs")
                fails += diff_points(f"knn2 pair {k}", allk[2 * k, :len(a)], a, MFIELDS)
                nmatch += len(wl)
        # ---- the same batch again on rolled images: same buffers and arguments (a replayed graph where the path captures one)
        if c["again"] and B >= 2 and not fails:
            stack.copy_(upload([(i - 1) % B for i in range(B)], False))
            torch.cuda.synchronize()
            ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, img_ptr(0), h * pin, pin, B, d_pts.data_ptr(), d_num.data_ptr(), int(c["desc"])))
            ah.check(ah.lib.hak_sync(det.ctx))
            nums2 = d_num.cpu().numpy()
            allp2 = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
            for i in range(B):
                fails += diff_points(f"second batch call, image {i}", allp2[i, :min(nums2[i], mp)], want[((i - 1) % B) % nd], fields)
            stack.copy_(upload(range(B), False))
            torch.cuda.synchronize()
        # ---- both images + the match as ONE launch sequence (detectAndComputePair), image 2 with a clamp of its own
        if c["pair"] and B >= 2 and c["desc"] and small_sets:
            frac, pinned = c["pair"]
            cap2 = max(1, int(mp * frac))
            r1, r2 = ah.AkazeData(), ah.AkazeData()
            ah.initAkazeData(r1, mp, True, True, pinned=pinned)
            ah.initAkazeData(r2, cap2, True, True, pinned=pinned)
            det.detectAndComputePair(img_ptr(0), img_ptr(1 % B), r1, r2, (w, h, pin), True, True)
            a = want[0].copy()
            b = want[1 % nd] if cap2 >= len(want[1 % nd]) else okz.detect_and_compute(synth.to_float(u8s[1 % nd], p), w, okz.default_params(**okw), max_pts=cap2).points
            okz.match(a, b)
            fails += diff_points("pair call image 1", r1.h_data[:r1.num_pts], a, FIELDS + MFIELDS)
            fails += diff_points("pair call image 2", r2.h_data[:r2.num_pts], b, FIELDS)
            npts += len(a) + len(b)
            ah.freeAkazeData(r1)
            ah.freeAkazeData(r2)
        # ---- the single-image entry point (akaze.cpp:101-150) on image 0: other launch shapes than the batch
        data = ah.AkazeData()
        ah.initAkazeData(data, mp, True, True)
        det.detectAndCompute(img_ptr(0), data, (w, h, pin), c["desc"])
        fails += diff_points("float single call", data.h_data[:data.num_pts], want[0], fields)
        del stack
        # ---- integer FAST path (akaze.cpp:153-201): batch + single
        if c["fast"]:
            fwant = [okz.fast_detect_and_compute(u, okz.default_params(**okw), max_pts=mp, desc=c["desc"]).points for u in u8s]
            d8 = upload(range(B), True)
            ah.check(ah.lib.hak_fast_detect_and_compute_batch(det.ctx, d8.data_ptr() + off, h * pin, pin, B, d_pts.data_ptr(), d_num.data_ptr(), int(c["desc"])))
            ah.check(ah.lib.hak_sync(det.ctx))
            nums = d_num.cpu().numpy()
            allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
            for i in range(B):
                fails += diff_points(f"FAST batch image {i}", allp[i, :min(nums[i], mp)], fwant[i % nd], fields)
                npts += len(fwant[i % nd])
            det.fastDetectAndCompute(d8.data_ptr() + off, data, (w, h, pin), c["desc"])
            fails += diff_points("FAST single call", data.h_data[:data.num_pts], fwant[0], fields)
        if c["resize"] and w - c["resize"][0] >= 134 and h - c["resize"][1] >= 80:
            w2, h2 = w - c["resize"][0], h - c["resize"][1]
            p2 = ah.iAlignUp(w2, 128)
            crop = np.ascontiguousarray(synth.to_float(u8s[0], p)[:h2, :p2]) if p2 <= p else None
            if crop is not None:
                crop[:, w2:] = 0
                det.detectAndCompute(torch.from_numpy(crop).cuda().data_ptr(), data, (w2, h2, p2), c["desc"])
                wr = okz.detect_and_compute(crop, w2, okz.default_params(**okw), max_pts=mp, desc=c["desc"]).points
                fails += diff_points(f"call with other extents ({w2} x {h2})", data.h_data[:data.num_pts], wr, fields)
                npts += len(wr)
        ah.freeAkazeData(data)
    finally:
        det.close()
        for k in KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    return fails, npts, nmatch


def run(cases, seed, only=None, verbose=True, out=sys.stdout, big=False):
    import torch
    import akaze_hip as ah
    from akaze_hip import synth
    import okz
    okz.build()
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    okz.set_num_threads(max(1, min(16, ncpu)))
    assert torch.cuda.is_available(), "the parity run needs a HIP device"
    mg = _mg()
    failed, tp, tm, t0 = [], 0, 0, time.time()
    idx = [only] if only is not None else range(cases)
    for i in idx:
        c = draw_case(seed, i, big)
        fails, npts, nm = run_case(ah, okz, torch, synth, mg, c)
        tp += npts
        tm += nm
        if verbose or fails:
            print(f"{'FAIL' if fails else 'ok  '} {describe(c)}  [{npts} keypoints, {nm} matches]", file=out, flush=True)
        for f in fails[:4] + fails[-1:]:
            print("       " + f, file=out, flush=True)
        if fails:
            failed.append(i)
    print(f"== seed {seed}{(' (big)', ' (huge)')[big - 1] if big else ''}: {len(list(idx))} cases, {len(failed)} failed {failed}; {tp} keypoint records and {tm} accepted matches compared "
          f"bit for bit with the oracle in {time.time() - t0:.0f} s", file=out, flush=True)
    return failed


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--big", action="store_true", help="720p .. 4K frames, up to 65 images per launch sequence")
    ap.add_argument("--huge", action="store_true", help="4K .. 9000 x 7000 frames, one or two per launch sequence")
    a = ap.parse_args()
    sys.exit(1 if run(a.cases, a.seed, a.only, not a.quiet, big=2 if a.huge else int(a.big)) else 0)
