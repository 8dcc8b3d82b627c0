#!/usr/bin/env python3
"""Pins the CPU oracle against the reference's OWN CUDA run -- statistically, through the pictures that run left behind.

Build container only (reads /root/reference/data; nothing here is imported by the product or runs on the GPU box).

The reference's demo ran on data/img1.png / img2.png (absent from the checkout, .MISSING_LARGE_BLOBS) and wrote
  akaze_show{1,2}.jpg, fastakaze_show{1,2}.jpg    the gray image + one circle per keypoint at (cvRound(x), cvRound(y)),
                                                  radius (int)max(1, min(5, size)), random colour      (main.cpp:27-40, 221-225)
  akaze_ / fastakaze_ / cvshow_matched.jpg        img1 over img2 + one line per accepted match        (main.cpp:43-84, 223)
and data/timecost.png shows its printed keypoint counts: float 2205 / 2382, FAST 2690 / 2915.

What this script does with them
  (a) reconstructs img1 / img2: the five renderings of an image share one gray base (x/255*255 rounds back to x), every overlay
      pixel is an outlier in luma -> per pixel the largest group of renderings whose luma agrees (ties: lower chroma), its
      median; pixels without consensus are filled from their neighbours.  What is left is the JPEG (q95, 4:2:0) noise.
  (b) runs both oracles on the reconstruction and compares
        - keypoint COUNTS with the screenshot,
        - every oracle keypoint's predicted circle (exact centre pixel, exact radius, OpenCV's midpoint circle) with the
          overlay pixels of the reference's rendering ("ring hit"; control: random positions),
        - circles extracted from the rendering INDEPENDENTLY of the oracle (isolated, one colour, by luma alone) with the
          oracle's keypoints (recall within 1.5 px, radius class),
        - the NMS read-cursor lag (akazed.cu:1581-1593): keypoints that exist only under the literal reading
          vs a mirrored control position,
        - every accepted oracle match's line with the reference's matched rendering (control: perturbed end points), and the
          NUMBER of lines through the overlay pixels of the seam row (every match line crosses it once).
  (c) --calibrate: the same measurement on an EMULATION (left.pgm / right.pgm rendered by this script from the oracle's own
      result, JPEG q95 4:2:0): what the figures look like when the oracle IS the program that drew the pictures.

  python tools/ref_render_check.py                  # report to stdout + tests/golden/ref_render_report.json
  python tools/ref_render_check.py --write-fixture  # also tests/golden/ref_recon_1080p_u8.npz (the reconstructed pair)
"""
import argparse
import io
import json
import os
import sys

import numpy as np
from scipy import ndimage as ndi
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF_DATA = "/root/reference/data"
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_COUNTS = {"float": (2205, 2382), "fast": (2690, 2915)}        # data/timecost.png


# ------------------------------------------------------------------ OpenCV's rasterisers, restated
def cv_circle_offsets(radius):
    """the pixels cv::circle(img, c, radius, colour) sets for thickness 1, LINE_8, shift 0: OpenCV's midpoint circle"""
    pts = set()
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        for p in ((-dx, -dy), (-dx, dy), (dx, -dy), (dx, dy), (-dy, -dx), (-dy, dx), (dy, -dx), (dy, dx)):
            pts.add(p)
        dy += 1
        err += plus
        plus += 2
        mask = (1 if err <= 0 else 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return np.array(sorted(pts), np.int64)


def line_pixels(x0, y0, x1, y1):
    """8-connected Bresenham line (cv::line, thickness 1, LINE_8)"""
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x1 >= x0 else -1), (1 if y1 >= y0 else -1)
    n = max(dx, dy) + 1
    i = np.arange(n)
    if dx >= dy:
        xs = x0 + i * sx
        ys = y0 + sy * ((2 * i * dy + dx) // (2 * dx) if dx else 0 * i)
    else:
        ys = y0 + i * sy
        xs = x0 + sx * ((2 * i * dx + dy) // (2 * dy))
    return xs, ys


def keypoint_circle(pts):
    """main.cpp:35-38: centre cvRound (round-half-even), radius (int)MAX(1, MIN(5, size))"""
    cx = np.rint(pts["x"]).astype(np.int64)
    cy = np.rint(pts["y"]).astype(np.int64)
    rad = np.maximum(1, np.minimum(5, pts["size"])).astype(np.int64)
    return cx, cy, rad


# ------------------------------------------------------------------ renderings
def _ycc(im):
    a = np.asarray(im.convert("YCbCr")).astype(np.float32)
    return a[..., 0], np.abs(a[..., 1] - 128) + np.abs(a[..., 2] - 128)


def load_reference_renderings(data=REF_DATA):
    """per image k in (1, 2): luma stack Y[5, h, w], chroma-magnitude stack C[5, h, w];
    order: akaze_show, fastakaze_show, akaze_matched half, fastakaze_matched half, cvshow_matched half"""
    from PIL import Image
    out = {1: [], 2: []}
    for f in ("akaze_show%d.jpg", "fastakaze_show%d.jpg"):
        for k in (1, 2):
            out[k].append(_ycc(Image.open(os.path.join(data, f % k))))
    for f in ("akaze_show_matched.jpg", "fastakaze_show_matched.jpg", "cvshow_matched.jpg"):
        y, c = _ycc(Image.open(os.path.join(data, f)))
        h = y.shape[0] // 2
        out[1].append((y[:h], c[:h]))
        out[2].append((y[h:], c[h:]))
    return {k: (np.stack([a for a, _ in v]), np.stack([b for _, b in v])) for k, v in out.items()}


def reconstruct(Y, C, tol=3.0):
    """(a) of the module text.  Y, C: [n, h, w].  Returns (uint8 image, info)"""
    agree = np.abs(Y[:, None] - Y[None, :]) <= tol                  # [n, n, h, w]
    support = agree.sum(1)
    gchroma = (agree * C[None]).sum(1) / support
    score = support.astype(np.float32) - np.minimum(gchroma, 50) / 100.0
    best = score.argmax(0)
    sup = np.take_along_axis(support, best[None], 0)[0]
    member = np.take_along_axis(agree, best[None, None], 0)[0]
    with np.errstate(all="ignore"):
        val = np.nanmedian(np.where(member, Y, np.nan), axis=0)
    gc = np.take_along_axis(gchroma, best[None], 0)[0]
    hole = ((sup == 1) & (gc > 2.0)) | ((sup == 2) & (gc > 12.0))
    hole |= (sup <= 2) & (np.abs(val - ndi.median_filter(val, size=3)) > 8)      # isolated speckles of a chance consensus
    v = np.where(hole, 0, val).astype(np.float32)
    wgt = (~hole).astype(np.float32)
    out = val.copy()
    out[hole] = np.nan
    for sig in (0.8, 1.5, 3.0, 6.0, 12.0):
        num, den = ndi.gaussian_filter(v * wgt, sig), ndi.gaussian_filter(wgt, sig)
        fill = ~np.isfinite(out) & (den > 0.05)
        out[fill] = num[fill] / den[fill]
    assert np.isfinite(out).all()
    info = dict(filled_frac=float(hole.mean()), support_hist=[float(x) for x in np.bincount(sup.ravel(), minlength=6) / sup.size])
    return np.clip(np.rint(out), 0, 255).astype(np.uint8), info


# ------------------------------------------------------------------ circle / line evidence
def overlay_mask(Yr, Cr, rec):
    """'fat' evidence of an overlay pixel: chroma (4:2:0, so it bleeds one pixel) or a luma step against the reconstruction"""
    return (Cr > 5) | (np.abs(Yr - rec) > 12)


def ring_scores(ov, cx, cy, rad):
    H, W = ov.shape
    out = np.zeros(len(cx))
    offs = {r: cv_circle_offsets(int(r)) for r in np.unique(rad)}
    for i in range(len(cx)):
        o = offs[rad[i]]
        out[i] = ov[np.clip(cy[i] + o[:, 1], 0, H - 1), np.clip(cx[i] + o[:, 0], 0, W - 1)].mean()
    return out


def thin_ring_scores(Yr, rec, cx, cy, rad, thr=6.0, tol=6.0):
    """fraction of the predicted ring that carries ONE overlay colour: luma within tol of the ring's median and more than
    thr away from the reconstructed gray.  Luma is full resolution, so this does not bleed (random positions score 0)."""
    H, W = Yr.shape
    out = np.zeros(len(cx))
    offs = {r: cv_circle_offsets(int(r)) for r in np.unique(rad)}
    for i in range(len(cx)):
        o = offs[rad[i]]
        yy, xx = np.clip(cy[i] + o[:, 1], 0, H - 1), np.clip(cx[i] + o[:, 0], 0, W - 1)
        v = Yr[yy, xx]
        out[i] = ((np.abs(v - np.median(v)) <= tol) & (np.abs(v - rec[yy, xx]) > thr)).mean()
    return out


def extract_isolated_circles(Yr, rec, radii=(1, 2, 3, 4, 5), thr=8.0, sdmax=5.0):
    """circles found WITHOUT the oracle, by luma alone: every ring pixel differs from the reconstruction by more than thr, the
    ring has one luma (one colour), and no other differing pixel lies in the (2r+3)^2 box.  Rows (x, y, r)."""
    ovY = (np.abs(Yr - rec) > thr).astype(np.float32)
    found = []
    for r in radii:
        R = r + 1
        k = np.zeros((2 * R + 1, 2 * R + 1), np.float32)
        o = cv_circle_offsets(r)
        k[o[:, 1] + R, o[:, 0] + R] = 1
        n = k.sum()
        cnt = ndi.correlate(ovY, k, mode="constant")
        tot = ndi.correlate(ovY, np.ones_like(k), mode="constant")
        s1 = ndi.correlate(Yr, k, mode="constant") / n
        s2 = ndi.correlate(Yr * Yr, k, mode="constant") / n
        sd = np.sqrt(np.maximum(s2 - s1 * s1, 0))
        ys, xs = np.nonzero((cnt >= n - 0.5) & (tot - cnt <= 0) & (sd <= sdmax))
        found += [(x, y, r) for y, x in zip(ys, xs)]
    return np.array(found, np.float32).reshape(-1, 3)


def line_score(Yr, Cr, rec, x0, y0, x1, y1):
    """fraction of the predicted match line drawn in one colour (best of the three 1-px shifts along the minor axis: the
    rasteriser's tie rule is not restated)"""
    xs, ys = line_pixels(x0, y0, x1, y1)
    H, W = Yr.shape
    horiz = abs(x1 - x0) >= abs(y1 - y0)
    best = 0.0
    for s in (-1, 0, 1):
        xx = np.clip(xs + (0 if horiz else s), 0, W - 1)
        yy = np.clip(ys + (s if horiz else 0), 0, H - 1)
        v, g, c = Yr[yy, xx], rec[yy, xx], Cr[yy, xx]
        best = max(best, float(((np.abs(v - np.median(v)) <= 6) & ((np.abs(v - g) > 6) | (c > 5))).mean()))
    return best


# ------------------------------------------------------------------ the oracle under test
def to_float(u8):
    h, w = u8.shape
    p = (w + 127) // 128 * 128
    f = np.zeros((h, p), np.float32)
    f[:, :w] = u8.astype(np.float32) * np.float32(1.0 / 255.0)          # main.cpp:149
    return f


def run_oracle(u8, path, variant=0):
    import okz
    okz.lib().okz_set_reading_variant(int(variant))
    try:
        if path == "float":
            return okz.detect_and_compute(to_float(u8), u8.shape[1]).points
        return okz.fast_detect_and_compute(u8).points
    finally:
        okz.lib().okz_set_reading_variant(0)


# ------------------------------------------------------------------ the measurement
def keypoint_metrics(pts, pts_clean, Yr, Cr, rec, ref_count, rng):
    H, W = Yr.shape
    ov = overlay_mask(Yr, Cr, rec)
    cx, cy, rad = keypoint_circle(pts)
    s = ring_scores(ov, cx, cy, rad)
    s1 = s.copy()
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx or dy:
                s1 = np.maximum(s1, ring_scores(ov, cx + dx, cy + dy, rad))
    rx, ry = rng.integers(30, W - 30, len(pts)), rng.integers(30, H - 30, len(pts))
    sr = ring_scores(ov, rx, ry, rad)
    m = dict(oracle_count=int(len(pts)), reference_count=int(ref_count), count_ratio=round(len(pts) / ref_count, 4),
             ring_hit=round(float((s >= 0.85).mean()), 4), ring_hit_pm1=round(float((s1 >= 0.85).mean()), 4),
             ring_all_set=round(float((s == 1).mean()), 4), ring_hit_random=round(float((sr >= 0.85).mean()), 4))
    # by response quintile: misses should be the weak, noise-sensitive keypoints
    q = np.quantile(pts["response"], np.linspace(0, 1, 6))
    m["ring_hit_pm1_by_response_quintile"] = [round(float((s1[(pts["response"] >= lo) & (pts["response"] <= hi)] >= 0.85).mean()), 4)
                                              for lo, hi in zip(q[:-1], q[1:])]
    m["ring_hit_pm1_by_octave"] = [round(float((s1[pts["octave"] // 4 == o] >= 0.85).mean()), 4) if (pts["octave"] // 4 == o).any() else None
                                   for o in range(4)]
    # independent extraction
    c = extract_isolated_circles(Yr, rec)
    tree = cKDTree(np.c_[pts["x"], pts["y"]])
    d, i = tree.query(c[:, :2]) if len(c) else (np.zeros(0), np.zeros(0, int))
    same = rad[i] == c[:, 2]
    m["isolated_circles"] = dict(n=int(len(c)), by_radius={str(r): int((c[:, 2] == r).sum()) for r in (1, 2, 3, 4, 5)},
                                 recall_1p5px=round(float((d <= 1.5).mean()), 4) if len(c) else None,
                                 recall_1p5px_same_radius=round(float(((d <= 1.5) & same).mean()), 4) if len(c) else None,
                                 exact_pixel_and_radius=round(float(((cx[i] == c[:, 0]) & (cy[i] == c[:, 1]) & same).mean()), 4) if len(c) else None,
                                 # drawn radius class -> radius classes (2, 3, 4 = sublevels {0, 1}, 2, 3) of the oracle keypoints found within 1.5 px
                                 radius_confusion={str(r): [int(((c[:, 2] == r) & (d <= 1.5) & (rad[i] == q)).sum()) for q in (2, 3, 4)]
                                                   for r in (2, 3, 4)} if len(c) else None)
    # NMS cursor lag: keypoints only the literal reading has, against the position mirrored at their stronger right neighbour
    keyc = set(zip(pts_clean["x"].tolist(), pts_clean["y"].tolist(), pts_clean["octave"].tolist()))
    lag = np.array([(x, y, o) not in keyc for x, y, o in zip(pts["x"].tolist(), pts["y"].tolist(), pts["octave"].tolist())])
    st = thin_ring_scores(Yr, rec, cx, cy, rad)
    str_ = thin_ring_scores(Yr, rec, rx, ry, rad)
    P = pts[lag]
    _, nn = tree.query(np.c_[P["x"], P["y"]], k=2)
    Q = pts[nn[:, 1]]
    sm = thin_ring_scores(Yr, rec, np.rint(2 * Q["x"] - P["x"]).astype(np.int64), np.rint(2 * Q["y"] - P["y"]).astype(np.int64), rad[lag])
    m["nms_lag"] = dict(clean_disc_count=int(len(pts_clean)), lag_only=int(lag.sum()),
                        thin_hit_all=round(float((st >= 0.75).mean()), 4), thin_hit_random=round(float((str_ >= 0.75).mean()), 4),
                        thin_hit_lag_only=round(float((st[lag] >= 0.75).mean()), 4),
                        thin_hit_mirrored_control=round(float((sm >= 0.75).mean()), 4))
    return m


def match_metrics(p1, p2, Ym, Cm, recm, h1, rng):
    import okz
    a, b = p1.copy(), p2.copy()
    okz.match(a, b)
    idx = np.nonzero(a["match"] >= 0)[0]
    sc, ctl = [], []
    for i in idx:
        k = a["match"][i]
        x0, y0 = int(np.rint(a["x"][i])), int(np.rint(a["y"][i]))
        x1, y1 = int(np.rint(b["x"][k])), int(np.rint(b["y"][k])) + h1              # main.cpp:72-75 (vertical stack)
        sc.append(line_score(Ym, Cm, recm, x0, y0, x1, y1))
        ang, rr = rng.uniform(0, 2 * np.pi), rng.uniform(6, 40)
        ctl.append(line_score(Ym, Cm, recm, x0, y0, int(x1 + rr * np.cos(ang)), int(y1 + rr * np.sin(ang))))
    sc, ctl = np.array(sc), np.array(ctl)
    # how MANY lines the picture holds: every match line crosses the seam between the two stacked images exactly once, so the
    # overlay pixels of the seam row count the matches (nearly vertical lines: one pixel each; neighbours merge).  Luma only
    # (no bleed); the calibration gives the share of line pixels that differ visibly from the gray underneath.
    seam = np.zeros(Ym.shape[1], bool)
    for i in idx:
        k = a["match"][i]
        xs, ys = line_pixels(int(np.rint(a["x"][i])), int(np.rint(a["y"][i])), int(np.rint(b["x"][k])), int(np.rint(b["y"][k])) + h1)
        seam[np.clip(xs[ys == h1], 0, Ym.shape[1] - 1)] = True
    seam_ref = int((np.abs(Ym[h1] - recm[h1]) > 8).sum())
    return dict(oracle_matches=int(len(idx)), seam_pixels_oracle_lines=int(seam.sum()), seam_pixels_picture=seam_ref,
                seam_ratio=round(seam_ref / max(1, int(seam.sum())), 4), line_hit=round(float((sc >= 0.5).mean()), 4), line_hit_control=round(float((ctl >= 0.5).mean()), 4),
                line_score_median=round(float(np.median(sc)), 4), line_score_median_control=round(float(np.median(ctl)), 4))


def measure(stacks, matched, ref_counts, seed=0):
    """stacks[k] = (Y[5], C[5]) with renderings 0 / 1 = float / FAST keypoint pictures; matched[path] = (Y, C) of the stacked
    match picture.  Returns (report, reconstructions)."""
    rng = np.random.default_rng(seed)
    rep, recs, pts = {"reconstruction": {}, "keypoints": {}, "matches": {}}, {}, {}
    for k in (1, 2):
        recs[k], rep["reconstruction"]["img%d" % k] = reconstruct(*stacks[k])
    for path, ri in (("float", 0), ("fast", 1)):
        for k in (1, 2):
            Y, C = stacks[k]
            p = run_oracle(recs[k], path)
            pc = run_oracle(recs[k], path, variant=1)
            pts[path, k] = p
            rep["keypoints"]["%s_img%d" % (path, k)] = keypoint_metrics(p, pc, Y[ri], C[ri], recs[k].astype(np.float32),
                                                                        ref_counts[path][k - 1], rng)
        Ym, Cm = matched[path]
        recm = np.concatenate([recs[1], recs[2]]).astype(np.float32)
        rep["matches"][path] = match_metrics(pts[path, 1], pts[path, 2], Ym, Cm, recm, recs[1].shape[0], rng)
    return rep, recs


# ------------------------------------------------------------------ calibration: emulate the reference's drawing
def _jpeg(rgb):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, format="JPEG", quality=95, subsampling=2)      # cv::imwrite defaults: q95, 4:2:0
    b.seek(0)
    return Image.open(b)


def _draw_keypoints(u8, pts, rng):
    img = np.repeat(u8[..., None], 3, axis=2).copy()
    H, W = u8.shape
    cx, cy, rad = keypoint_circle(pts)
    for i in range(len(pts)):
        o = cv_circle_offsets(int(rad[i]))
        yy, xx = cy[i] + o[:, 1], cx[i] + o[:, 0]
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        img[yy[ok], xx[ok]] = rng.integers(0, 255, 3)
    return img


def _draw_matches(u1, u2, a, b, rng, extra=0):
    cat = np.concatenate([u1, u2])
    img = np.repeat(cat[..., None], 3, axis=2).copy()
    H, W = cat.shape
    pairs = [(i, a["match"][i]) for i in np.nonzero(a["match"] >= 0)[0]]
    pairs += [(rng.integers(len(a)), rng.integers(len(b))) for _ in range(extra)]
    for i, k in pairs:
        xs, ys = line_pixels(int(np.rint(a["x"][i])), int(np.rint(a["y"][i])), int(np.rint(b["x"][k])), int(np.rint(b["y"][k])) + u1.shape[0])
        ok = (ys >= 0) & (ys < H) & (xs >= 0) & (xs < W)
        img[ys[ok], xs[ok]] = rng.integers(0, 255, 3)
    return img


def calibrate(u, seed=5, draw_variant=0, mutate=None):
    """u = {1: img, 2: img} (uint8): the oracle's own result on them drawn the way main.cpp draws, JPEG-compressed the way
    cv::imwrite does, then measured exactly like the reference's pictures.  'reference counts' = the oracle's counts on u.
    draw_variant / mutate (power table): the DRAWING program reads the reference differently (okz_reading_variant bits) or has its
    descriptors rewritten by mutate(points) before it matches; the measuring oracle is always the default reading."""
    import okz
    rng = np.random.default_rng(seed)
    truth = {(path, k): run_oracle(u[k], path, variant=draw_variant) for path in ("float", "fast") for k in (1, 2)}
    if mutate:
        truth = {key: mutate(v.copy()) for key, v in truth.items()}
    shows = {k: [] for k in (1, 2)}
    matched = {}
    for path in ("float", "fast"):
        for k in (1, 2):
            shows[k].append(_ycc(_jpeg(_draw_keypoints(u[k], truth[path, k], rng))))
    h = u[1].shape[0]
    for j, path in enumerate(("float", "fast", "third")):
        a, b = truth["float" if path == "third" else path, 1].copy(), truth["float" if path == "third" else path, 2].copy()
        okz.match(a, b)
        y, c = _ycc(_jpeg(_draw_matches(u[1], u[2], a, b, rng, extra=400 if path == "third" else 0)))
        if path != "third":
            matched[path] = (y, c)
        shows[1].append((y[:h], c[:h]))
        shows[2].append((y[h:], c[h:]))
    stacks = {k: (np.stack([a for a, _ in v]), np.stack([b for _, b in v])) for k, v in shows.items()}
    counts = {path: (len(truth[path, 1]), len(truth[path, 2])) for path in ("float", "fast")}
    rep, recs = measure(stacks, matched, counts, seed=seed)
    for k in (1, 2):
        rep["reconstruction"]["img%d" % k]["rms_vs_original"] = round(float(np.sqrt(((recs[k].astype(np.float32) - u[k]) ** 2).mean())), 3)
    return rep


# ------------------------------------------------------------------ which reading of hScharrContrast do the pictures prefer?
READINGS = {0: "literal: lattice maximum + threads outside the image count zeros (default since round 5)",
            2: "true maximum of the gradient (rounds 1-4), histogram guard literal",
            4: "lattice maximum, histogram over the w x h pixels only (guard read as ||)",
            6: "true maximum + w x h histogram (rounds 1-4)"}


def score_readings(stacks, matched, recs, ref_counts, seed=0):
    """counts / ring hits / match lines of the reference's pictures under the four readings of akazed.cu:827-877 + 909"""
    out = {}
    for v in READINGS:
        rng = np.random.default_rng(seed)
        row = {"reading": READINGS[v]}
        pts = {}
        for path, ri in (("float", 0), ("fast", 1)):
            for k in (1, 2):
                Y, C = stacks[k]
                p = run_oracle(recs[k], path, variant=v)
                pts[path, k] = p
                rec = recs[k].astype(np.float32)
                ov = overlay_mask(Y[ri], C[ri], rec)
                cx, cy, rad = keypoint_circle(p)
                s1 = ring_scores(ov, cx, cy, rad)
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        if dx or dy:
                            s1 = np.maximum(s1, ring_scores(ov, cx + dx, cy + dy, rad))
                row["%s_img%d" % (path, k)] = dict(count=int(len(p)), reference_count=int(ref_counts[path][k - 1]),
                                                   count_ratio=round(len(p) / ref_counts[path][k - 1], 4),
                                                   ring_hit_pm1=round(float((s1 >= 0.85).mean()), 4),
                                                   ring_hits=int((s1 >= 0.85).sum()))
            Ym, Cm = matched[path]
            recm = np.concatenate([recs[1], recs[2]]).astype(np.float32)
            mm = match_metrics(pts[path, 1], pts[path, 2], Ym, Cm, recm, recs[1].shape[0], rng)
            row["matches_" + path] = dict(oracle_matches=mm["oracle_matches"], line_hit=mm["line_hit"], seam_ratio=mm["seam_ratio"],
                                          line_hits=int(round(mm["line_hit"] * mm["oracle_matches"])))
        out[str(v)] = row
    return out


# ------------------------------------------------------------------ what the pin can and cannot see (power table)
def _permute_descriptor_bits(points, seed=99):
    """a consistent permutation of the 486 descriptor bits (the same for every keypoint of both images): Hamming distances, hence
    every match, are unchanged"""
    perm = np.random.default_rng(seed).permutation(486)
    bits = np.unpackbits(points["features"], axis=1, bitorder="little")
    out = bits.copy()
    out[:, :486] = bits[:, perm]
    points["features"] = np.packbits(out, axis=1, bitorder="little")
    return points


POWER_CASES = [  # name, okz_reading_variant bits of the DRAWER, descriptor mutation
    ("clean disc NMS (cursor advances at the skipped centre)", 1, None),
    ("true-maximum hmax (rounds 1-4) instead of the lattice maximum", 2, None),
    ("histogram over w x h only (guard as ||)", 4, None),
    ("k + 1 histogram bins", 8, None),
    ("sublevel scatter: last writer wins (the D5 race at its extreme)", 16, None),
    ("main orientation + 5 degrees", 32, None),
    ("descriptor bits permuted consistently", 0, _permute_descriptor_bits),
]
POWER_METRICS = [("count_ratio", lambda r, i: r["keypoints"]["float_img%d" % i]["count_ratio"]),
                 ("ring_hit_pm1", lambda r, i: r["keypoints"]["float_img%d" % i]["ring_hit_pm1"]),
                 ("isolated_recall", lambda r, i: r["keypoints"]["float_img%d" % i]["isolated_circles"]["recall_1p5px"]),
                 ("isolated_same_radius", lambda r, i: r["keypoints"]["float_img%d" % i]["isolated_circles"]["recall_1p5px_same_radius"]),
                 ("lag_thin_hit", lambda r, i: r["keypoints"]["float_img%d" % i]["nms_lag"]["thin_hit_lag_only"]),
                 ("line_hit", lambda r, i: r["matches"]["float"]["line_hit"]),
                 ("seam_ratio", lambda r, i: r["matches"]["float"]["seam_ratio"])]


def power_table(recs, null_seeds=(5, 6, 7, 8)):
    """The oracle under reading X draws the pictures (on the reconstructed pair, the reference's density), the DEFAULT oracle is
    measured against them exactly like against the reference's.  A metric 'separates' X from the default reading when its value
    leaves the band the default-vs-default runs span (different colours / control draws: `null_seeds`) by more than that band's
    width again.  Float path, mean over img1 / img2."""
    def vec(rep):
        return {name: float(np.mean([f(rep, i) for i in (1, 2)])) for name, f in POWER_METRICS}
    null = [vec(calibrate(recs, seed=sd)) for sd in null_seeds]
    band = {name: (min(n[name] for n in null), max(n[name] for n in null)) for name, _ in POWER_METRICS}
    rows = {}
    for title, bits, mut in POWER_CASES:
        v = vec(calibrate(recs, seed=null_seeds[0], draw_variant=bits, mutate=mut))
        sep = {}
        for name, _ in POWER_METRICS:
            lo, hi = band[name]
            slack = max(hi - lo, 0.005)
            sep[name] = bool(v[name] < lo - slack or v[name] > hi + slack)
        rows[title] = dict(variant_bits=bits, metrics={k: round(x, 4) for k, x in v.items()}, separates=sep,
                           seen_by=[k for k, b in sep.items() if b])
    return dict(null_band={k: [round(a, 4), round(b, 4)] for k, (a, b) in band.items()}, null_runs=len(null_seeds), cases=rows)


# ------------------------------------------------------------------ why are radius-3 circles found as sublevel 1?  (noise hypothesis)
def _jpeg_gray_generation(u8):
    """one JPEG generation of a gray image the way the demo's pictures went through it (3-channel, q95, 4:2:0), luma back"""
    return np.clip(np.rint(_ycc(_jpeg(np.repeat(u8[..., None], 3, axis=2)))[0]), 0, 255).astype(np.uint8)


def sublevel_class_drift(u8, generations=1):
    """keypoints of the pristine image vs keypoints of the same image after JPEG generation(s), ALL keypoints (no isolation
    filter): for every pristine keypoint of radius class r (2 = sublevels 0 / 1, 3 = sublevel 2, 4 = sublevel 3) the class of
    the keypoint found within 1.5 px afterwards.  If JPEG noise is what turns drawn radius-3 circles into sublevel-1 keypoints,
    3 -> 2 must be common here and 2 -> 3 rare -- with no CUDA race anywhere in the experiment."""
    a = run_oracle(u8, "float")
    v = u8
    for _ in range(generations):
        v = _jpeg_gray_generation(v)
    b = run_oracle(v, "float")
    _, _, ra = keypoint_circle(a)
    _, _, rb = keypoint_circle(b)
    d, i = cKDTree(np.c_[b["x"], b["y"]]).query(np.c_[a["x"], a["y"]])
    conf = {str(r): [int(((ra == r) & (d <= 1.5) & (rb[i] == q)).sum()) for q in (2, 3, 4)] + [int(((ra == r) & (d > 1.5)).sum())] for r in (2, 3, 4)}
    # the coarser octaves' sublevel 1 vs 2 only (same dilation: where the reference's pictures show the effect)
    return dict(pristine=int(len(a)), after=int(len(b)), rms_noise=round(float(np.sqrt(((v.astype(np.float32) - u8) ** 2).mean())), 3),
                class_after_by_class_before=conf, columns=["2", "3", "4", "lost"],
                frac_3_to_2=round(conf["3"][0] / max(1, sum(conf["3"][:3])), 4), frac_2_to_3=round(conf["2"][1] / max(1, sum(conf["2"][:3])), 4))


def summarize(rep, title):
    print("== " + title)
    for k, v in rep["reconstruction"].items():
        print("  %s: %.2f %% of the pixels filled from neighbours%s" % (k, 100 * v["filled_frac"],
              ", rms vs original %.2f" % v["rms_vs_original"] if "rms_vs_original" in v else ""))
    for name, m in rep["keypoints"].items():
        ic, lg = m["isolated_circles"], m["nms_lag"]
        print("  %-11s oracle %d vs %d (%+.1f %%) | ring hit %.3f (+-1 px %.3f, random %.3f), strongest quintile %.3f | isolated circles %d: "
              "recall %.3f (same radius %.3f) | lag-only %d: thin hit %.3f vs mirrored control %.3f (all %.3f)"
              % (name, m["oracle_count"], m["reference_count"], 100 * (m["count_ratio"] - 1), m["ring_hit"], m["ring_hit_pm1"],
                 m["ring_hit_random"], m["ring_hit_pm1_by_response_quintile"][-1], ic["n"], ic["recall_1p5px"] or 0,
                 ic["recall_1p5px_same_radius"] or 0, lg["lag_only"], lg["thin_hit_lag_only"], lg["thin_hit_mirrored_control"], lg["thin_hit_all"]))
    for name, m in rep["matches"].items():
        print("  matches %-6s %d accepted by the oracle | line drawn in the picture: %.3f (control %.3f) | seam row: %d overlay pixels in the "
              "picture / %d pixels of the oracle's lines = %.3f"
              % (name, m["oracle_matches"], m["line_hit"], m["line_hit_control"], m["seam_pixels_picture"], m["seam_pixels_oracle_lines"],
                 m["seam_ratio"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-fixture", action="store_true")
    ap.add_argument("--no-calibrate", action="store_true")
    ap.add_argument("--no-readings", action="store_true")
    ap.add_argument("--no-power", action="store_true")
    ap.add_argument("--out", default=os.path.join(GOLDEN, "ref_render_report.json"))
    args = ap.parse_args()
    from PIL import Image
    stacks = load_reference_renderings()
    matched = {"float": _ycc(Image.open(os.path.join(REF_DATA, "akaze_show_matched.jpg"))),
               "fast": _ycc(Image.open(os.path.join(REF_DATA, "fastakaze_show_matched.jpg")))}
    rep, recs = measure(stacks, matched, REF_COUNTS)
    summarize(rep, "reference renderings (data/*.jpg) vs the oracle on the reconstructed img1 / img2")
    out = {"reference": rep}
    if not args.no_calibrate:
        # A: true originals (first-generation JPEG noise, as in the reference's pictures) but three times the keypoint density
        # of img1 / img2 (3 634 on 1.2 Mpx vs 2 205 on 2.1 Mpx): circles and lines crowd each other far more.
        lr = np.load(os.path.join(GOLDEN, "left_right_u8.npz"))
        out["calibration_left_right"] = calibrate({1: lr["left"], 2: lr["right"]})
        summarize(out["calibration_left_right"], "calibration A: the oracle's own result on left.pgm / right.pgm drawn, JPEG-compressed and "
                  "measured the same way (true originals; 3x the keypoint density)")
        # B: the reconstructed pair itself as the 'original': exactly the reference's density, but its pixels have been through
        # JPEG once already, so the second pass adds less noise than the reference's pictures carry.
        out["calibration_self"] = calibrate(recs)
        summarize(out["calibration_self"], "calibration B: the same on the reconstructed img1 / img2 themselves (same density; second-generation "
                  "JPEG noise only)")
    if not args.no_readings:
        out["readings"] = score_readings(stacks, matched, recs, REF_COUNTS)
        print("== the four readings of hScharrContrast (akazed.cu:827-877, 909) against the reference's pictures")
        for v, row in out["readings"].items():
            print("  variant %s  float %d / %d (ring hits %d / %d), FAST %d / %d (ring hits %d / %d), lines float %d of %d, FAST %d of %d -- %s"
                  % (v, row["float_img1"]["count"], row["float_img2"]["count"], row["float_img1"]["ring_hits"], row["float_img2"]["ring_hits"],
                     row["fast_img1"]["count"], row["fast_img2"]["count"], row["fast_img1"]["ring_hits"], row["fast_img2"]["ring_hits"],
                     row["matches_float"]["line_hits"], row["matches_float"]["oracle_matches"], row["matches_fast"]["line_hits"],
                     row["matches_fast"]["oracle_matches"], row["reading"]))
    if not args.no_power:
        out["power"] = power_table(recs)
        print("== power of the pin: the oracle under reading X draws, the default oracle is measured (float path, mean of img1 / img2)")
        print("  default vs default band:", out["power"]["null_band"])
        for title, row in out["power"]["cases"].items():
            print("  %-62s seen by: %s\n      %s" % (title, ", ".join(row["seen_by"]) or "NOTHING (reading-only)", row["metrics"]))
        lr = np.load(os.path.join(GOLDEN, "left_right_u8.npz"))
        out["sublevel_class_drift"] = {"left_1gen": sublevel_class_drift(lr["left"]), "right_1gen": sublevel_class_drift(lr["right"]),
                                       "recon_img1_1gen": sublevel_class_drift(recs[1]), "recon_img2_1gen": sublevel_class_drift(recs[2])}
        print("== radius class of a keypoint before / after one JPEG generation (all keypoints; rows 2, 3, 4 -> columns 2, 3, 4, lost)")
        for k, v in out["sublevel_class_drift"].items():
            print("  %-16s %s   3->2: %.3f, 2->3: %.3f (noise rms %.2f)" % (k, v["class_after_by_class_before"], v["frac_3_to_2"], v["frac_2_to_3"], v["rms_noise"]))
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)
    if args.write_fixture:
        f = os.path.join(GOLDEN, "ref_recon_1080p_u8.npz")
        np.savez_compressed(f, img1=recs[1], img2=recs[2])
        print("wrote", f, os.path.getsize(f) // 1024, "KB")


if __name__ == "__main__":
    main()
