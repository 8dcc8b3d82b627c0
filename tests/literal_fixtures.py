"""Hand-derived micro-fixtures for the control-flow-heavy stages of the reference.

Every fixture is a tiny hand-made input whose expected output is worked out BY HAND from the cited statements of
/root/reference (file:line in the comment next to each expectation) -- not by running the oracle.  The same fixtures are
run against the CPU oracle (tests/test_reference_literal_cpu.py) and, through the C ABI, against the HIP kernels
(tests/test_gpu_literal.py).  Where the expectation needs float32 arithmetic it is evaluated here with numpy float32 in
the operation order of the cited line; nothing in this file calls the oracle or the HIP library.
"""
import numpy as np

f32 = np.float32

# sizes[j] = esigma * derivative_factor, esigma = soffset * 2^(j/4), demo parameters (akaze.cpp:336, 357-361; main.cpp:156-166)
SIZES = [f32(1.6) * f32(1.5)] + [f32(f32(f32(1.6) * f32(np.float32(2.0) ** f32(j / 4.0))) * f32(1.5)) for j in (1, 2, 3)]
SIGMA = [int(f32(s) + f32(0.5)) for s in SIZES]                 # (int)(sizes + 0.5f)  akaze.cpp:362  -> 2, 3, 3, 4
SMAX = f32(10.0 * np.sqrt(f32(2.0)))                             # akaze.cpp:279
BORDERS = [f32(SMAX * f32(s)) for s in SIGMA]                    # akaze.cpp:363       -> 28.28, 42.43, 42.43, 56.57
PSZ = int(BORDERS[0])                                            # akazed.cu:2572       -> 28
assert SIGMA == [2, 3, 3, 4] and PSZ == 28


# ------------------------------------------------------------------------------------------------ disc NMS
# akazed.cu:1554-1613 (FAST: 3538-3598).  Facts used, per candidate with size fsz:
#   isz = (int)(fsz + 0.5f) (:1570), sqsz = (int)(fsz * fsz) (:1571)  -> layer 0: 2 / 5, layers 1-2: 3 / 8 and 3 / 11, layer 3: 4 / 16
#   the read cursor new_idx starts at column ix - isz on every row (:1578) and is advanced at the END of the j body (:1593);
#   `continue` at the centre (:1581-1584) skips that increment, so on row i == 0 every j > 0 reads column ix + j - 1:
#     j == 1 reads the centre itself (equal, tie rule i <= 0 && j <= 0 false -> no effect),
#     a right-hand neighbour at distance d on the centre row is tested under (d + 1)^2 < sqsz, the one at distance isz never;
#   suppression: in-disc neighbour larger, or equal with i <= 0 && j <= 0 (:1585-1589);
#   only pixels with psz <= ix, ix + psz < width (same for y) are candidates (:1560), but any map pixel can suppress.
NMS_W, NMS_H, NMS_P = 100, 100, 128
# (x, y, layer, response, survives, note)
NMS_CANDIDATES = [
    # size 2.4 (isz 2, sqsz 5): right neighbour at distance 2 is stronger.  j = 2 reads column 39 (empty), column 40 is never read
    (38, 30, 0, 1.0, True, "clean disc would drop it: 0 + 4 < 5"),
    (40, 30, 0, 2.0, True, "left neighbour (j = -2, straight read) is weaker"),
    # a candidate left of the band still suppresses (:1560 only gates the centre)
    (27, 35, 0, 9.0, False, "x < psz: never a centre"),
    (28, 35, 0, 1.0, False, "j = -1 reads column 27: larger"),
    # size 3.394 (isz 3, sqsz 11): distance 3 to the right is read by nobody (j = 3 reads column 32)
    (30, 40, 2, 1.0, True, "clean disc would drop it: 9 < 11"),
    (33, 40, 2, 2.0, True, ""),
    # ... but distance 2 is read at j = 3 under 9 < 11
    (50, 40, 2, 1.0, False, "j = 3 reads column 52: larger, 9 < 11"),
    (52, 40, 2, 2.0, True, ""),
    # size 2.4, distance 1: read at j = 2 under 4 < 5
    (40, 50, 0, 1.0, False, "j = 2 reads column 41: larger"),
    (41, 50, 0, 2.0, True, ""),
    # equal responses on one row, distance 2, size 2.4: the right one loses by the tie rule (j = -2 <= 0), the left one never sees it
    (60, 50, 0, 3.0, True, "j = 2 reads the empty column 61"),
    (62, 50, 0, 3.0, False, "equal at j = -2, i = 0: tie rule"),
    # rows other than the centre row are read straight
    (50, 60, 0, 1.0, False, "i = 1, j = 1 reads (51, 61): larger, 2 < 5"),
    (51, 61, 0, 2.0, True, ""),
    # tie rule covers the up-left quadrant only: equal responses, neighbour up-right -> both stay
    (60, 60, 0, 5.0, True, "equal at i = -1, j = +1: j <= 0 false"),
    (61, 59, 0, 5.0, True, "equal at i = +1, j = -1: i <= 0 false"),
    # ... neighbour up-left -> the lower-right one goes
    (68, 60, 0, 5.0, False, "equal at i = -1, j = -1: tie rule"),
    (67, 59, 0, 5.0, True, "equal at i = +1, j = +1: no"),
    # size 4.036 (isz 4, sqsz 16): distance 3 would be read at j = 4, but 16 < 16 fails
    (40, 70, 3, 1.0, True, "clean disc would drop it: 9 < 16"),
    (43, 70, 3, 2.0, True, ""),
    (55, 70, 3, 1.0, False, "distance 2 is read at j = 3: 9 < 16"),
    (57, 70, 3, 2.0, True, ""),
]
# what a clean disc (no cursor lag) would additionally suppress: the reference keeps these three
NMS_LAG_ONLY = [(38, 30), (30, 40), (40, 70)]


def nms_maps(fast=False):
    """full-resolution maps as akaze.cpp:252-258 leaves them (D1: response -0.0926, layer -1; FAST: 0xC0C0C0C0)"""
    resp = np.full((NMS_H, NMS_P), -1061109568 if fast else f32(-0.0926474631), np.int32 if fast else np.float32)
    size = np.full((NMS_H, NMS_P), f32(-0.0926474631), np.float32)
    layer = np.full((NMS_H, NMS_P), -1, np.int32)
    for x, y, l, r, _, _ in NMS_CANDIDATES:
        resp[y, x] = int(r * 100) if fast else f32(r)
        size[y, x] = SIZES[l]
        layer[y, x] = l
    return resp, size, layer


def nms_expected():
    """survivors in raster order (D6: the build emits raster order; the reference's order is atomic arrival)"""
    keep = [(x, y, l, r) for x, y, l, r, ok, _ in NMS_CANDIDATES if ok]
    return sorted(keep, key=lambda t: (t[1], t[0]))


# ------------------------------------------------------------------------------------------- extrema map
# akazed.cu:1334-1393 gCalcExtremaMap (FAST 3476-3515) at full resolution 200 x 180, octave 1 = 100 x 90.
#   accepted iff (int)(ix - border + 0.5f) - 1 >= 0 and (int)(ix + border + 0.5f) + 1 < width (:1346-1353), same in y:
#     border 28.284 (sigma 2): ix - 27.716 must truncate to >= 1 -> ix >= 29; (int)(ix + 28.784) + 1 = ix + 29 < w -> ix <= w - 30
#     border 42.426 (sigma 3): ix >= 43, ix <= w - 44;   border 56.569 (sigma 4): ix >= 58, ix <= w - 59
#   strict 3 x 3 maximum and strictly above the threshold (:1360-1362); scatter to (ix << octave, iy << octave), the map keeps
#   the larger response, the earlier sublevel on a tie (`response_map[oidx] < *vp`, :1368; sublevels in ascending order, D5)
EXT_W, EXT_H = 200, 180
EXT_THRESHOLD = f32(0.001)                                       # main.cpp:164 dthreshold


def extrema_fixture():
    """-> (dets {octave: (4, h, w) float32}, expected [(x_full, y_full, layer, response)])"""
    d0 = np.zeros((4, EXT_H, EXT_W), np.float32)
    d1 = np.zeros((4, EXT_H // 2, EXT_W // 2), np.float32)
    exp = []

    def put(det, s, x, y, v, ok, octave=0):
        det[s, y, x] = f32(v)
        if ok:
            exp.append((x << octave, y << octave, octave * 4 + s, f32(v)))

    # sublevel 0 (sigma 2): first / last accepted column and row
    put(d0, 0, 28, 90, 1.0, False); put(d0, 0, 29, 100, 1.0, True)
    put(d0, 0, 170, 90, 1.0, True); put(d0, 0, 171, 100, 1.0, False)          # w - 30 = 170
    put(d0, 0, 100, 28, 1.0, False); put(d0, 0, 110, 29, 1.0, True)
    put(d0, 0, 100, 150, 1.0, True); put(d0, 0, 110, 151, 1.0, False)         # h - 30 = 150
    # sublevel 1 (sigma 3)
    put(d0, 1, 42, 60, 1.0, False); put(d0, 1, 43, 70, 1.0, True)
    put(d0, 1, 156, 60, 1.0, True); put(d0, 1, 157, 70, 1.0, False)           # w - 44 = 156
    put(d0, 1, 120, 42, 1.0, False); put(d0, 1, 130, 43, 1.0, True)
    put(d0, 1, 120, 136, 1.0, True); put(d0, 1, 130, 137, 1.0, False)         # h - 44 = 136
    # sublevel 3 (sigma 4)
    put(d0, 3, 57, 80, 1.0, False); put(d0, 3, 58, 90, 1.0, True)
    put(d0, 3, 141, 80, 1.0, True); put(d0, 3, 142, 90, 1.0, False)           # w - 59 = 141
    put(d0, 3, 80, 57, 1.0, False); put(d0, 3, 90, 58, 1.0, True)
    put(d0, 3, 80, 121, 1.0, True); put(d0, 3, 90, 122, 1.0, False)           # h - 59 = 121
    # sublevel 2: threshold and strictness
    put(d0, 2, 50, 50, EXT_THRESHOLD, False)                                   # == threshold: `*vp > threshold` fails
    put(d0, 2, 60, 50, np.nextafter(EXT_THRESHOLD, f32(1)), True)
    put(d0, 2, 70, 50, 1.0, False); put(d0, 2, 71, 50, 1.0, False)            # plateau: neither is strictly larger
    put(d0, 2, 80, 50, 1.0, False); put(d0, 2, 81, 51, 2.0, True)             # diagonal neighbour larger
    # collisions in the shared map: larger response wins ...
    d0[1, 100, 90] = f32(0.5); put(d0, 2, 90, 100, 0.7, True)
    # ... and on a tie the earlier sublevel stays
    put(d0, 1, 100, 100, 0.6, True); d0[3, 100, 100] = f32(0.6)
    # octave 1: (35, 33) -> full resolution (70, 66), layer 4
    put(d1, 0, 35, 33, 1.5, True, octave=1)
    put(d1, 0, 28, 40, 1.5, False, octave=1)
    put(d1, 0, 70, 50, 1.5, True, octave=1); put(d1, 0, 71, 40, 1.5, False, octave=1)   # 100 - 30 = 70
    return {0: d0, 1: d1}, sorted(exp, key=lambda t: (t[1], t[0]))


def blob_plane(w, h, centres, sigma=3.0):
    """sum of isotropic Gaussian blobs: the Hessian determinant of each has its strict 3 x 3 maximum at the (integer) centre"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w), np.float64)
    for cx, cy in centres:
        img += np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2.0 * sigma * sigma))
    return img.astype(np.float32)


def blob_border_fixture():
    """L-planes whose determinant peaks sit on / next to the first and last accepted column of each dilation:
    -> {sublevel: (plane, [accepted (x, y)])} for octave 0 of a 200 x 180 image (column rule of extrema_fixture)"""
    out = {}
    for s, (lo, hi) in {0: (29, 170), 1: (43, 156), 3: (58, 141)}.items():
        ys = (70, 110)
        centres = [(lo - 1, ys[0]), (lo, ys[1]), (hi, ys[0]), (hi + 1, ys[1]), (100, 90)]
        ok = [(lo, ys[1]), (hi, ys[0]), (100, 90)]
        out[s] = (blob_plane(EXT_W, EXT_H, centres), sorted(ok, key=lambda t: (t[1], t[0])))
    return out


# ------------------------------------------------------------------------------------------------ refine
# akazed.cu:1633-1659 gRefine on a hand-made 3 x 3 determinant neighbourhood (c = centre, l r u d, corners ul ur ll lr)
def refine_expected(x, y, o, c, l, r, u, d, ul=0.0, ur=0.0, ll=0.0, lr=0.0):
    c, l, r, u, d, ul, ur, ll, lr = (f32(v) for v in (c, l, r, u, d, ul, ur, ll, lr))
    xi, yi = int(x) >> o, int(y) >> o                           # :1632-1633
    v2 = c + c                                                  # :1636
    dx = f32(0.5) * (r - l)                                     # :1637
    dy = f32(0.5) * (d - u)                                     # :1638
    dxx = r + l - v2                                            # :1639
    dyy = d + u - v2                                            # :1640
    dxy = f32(0.25) * (lr + ul - ur - ll)                       # :1641  det[idx+p+1] + det[idx-p-1] - det[idx-p+1] - det[idx+p-1]
    dd = dxx * dyy - dxy * dxy                                  # :1642
    idd = f32(1.0) / dd if dd != 0 else f32(0.0)                # :1643
    dst0 = idd * (dxy * dy - dyy * dx)                          # :1644
    dst1 = idd * (dxy * dx - dxx * dy)                          # :1645
    if dst0 < -1 or dst0 > 1 or dst1 < -1 or dst1 > 1:          # :1646-1650 weak: position stays integer
        return f32(x), f32(y)
    ratio = f32(1 << o)
    return f32(ratio * (f32(xi) + dst0)), f32(ratio * (f32(yi) + dst1))   # :1657-1658


REFINE_CASES = [
    # name, (x_full, y_full, octave), neighbourhood dict
    ("plain", (40, 40, 0), dict(c=10, l=6, r=8, u=7, d=7)),                         # dst0 = 6/36, dst1 = 0
    ("both axes + dxy", (42, 46, 0), dict(c=10, l=7, r=8, u=6.5, d=8, ul=1, lr=2, ur=0.5, ll=0.25)),
    ("weak: |dst0| > 1", (44, 44, 0), dict(c=10, l=9, r=10.9, u=7, d=7)),           # dst0 = 9.5 -> unchanged
    ("offset exactly 1 is kept", (48, 40, 0), dict(c=10, l=8.5, r=10.5, u=9, d=9)),  # `> 1.f` is strict
    ("dd == 0 -> idd = 0", (52, 40, 0), dict(c=3, l=3, r=3, u=3, d=3, ul=3, ur=3, ll=3, lr=3)),
    ("octave 1: ratio 2", (70, 66, 1), dict(c=10, l=6, r=8, u=9, d=7)),
]


# ---------------------------------------------------------------------------------------------- orientation
def fast_atan2(y, x):
    """dFastAtan2 akazed.cu:173-185 in float32 (fma emulated in float64: products of two floats are exact there)"""
    def fma(a, b, c):
        return f32(np.float64(a) * np.float64(b) + np.float64(c))
    absx, absy = abs(f32(x)), abs(f32(y))
    a = f32(min(absx, absy)) / f32(max(absx, absy))                     # :177  __fdiv_rn
    s = f32(a * a)
    r = fma(fma(fma(f32(-0.0464964749), s, f32(0.15931422)), s, f32(-0.327622764)), f32(s * a), a)   # :179
    r = f32(f32(1.5707963267948966) - r) if absy > absx else r          # :181  H_PI is a float literal (cuda_utils.h:8)
    r = f32(np.float64(np.pi) - np.float64(r)) if x < 0 else r          # :182  M_PI is a double
    r = f32(-r) if y < 0 else r                                         # :183
    return r


def orient_literal(samples, weights):
    """akazed.cu:1691-1734 on 109 (dx, dy) samples given in ascending thread order; per-sample angle bins supplied by the caller
    through `samples` = [(dx_weighted, dy_weighted, bin)]"""
    resx = np.zeros(42, np.float32); resy = np.zeros(42, np.float32)
    for dx, dy, a in samples:                                           # :1703-1704 (D7: ascending thread order)
        resx[a] = f32(resx[a] + f32(dx)); resy[a] = f32(resy[a] + f32(dy))
    re8x = np.zeros(42, np.float32); re8y = np.zeros(42, np.float32)
    for t in range(42):                                                 # :1708-1717
        re8x[t], re8y[t] = resx[t], resy[t]
        for k in range(t + 1, t + 7):
            kk = k if k < 42 else k - 42
            re8x[t] = f32(re8x[t] + resx[kk]); re8y[t] = f32(re8y[t] + resy[kk])
    maxr, maxk = f32(0), 0
    for k in range(42):                                                 # :1722-1731 strict `>`: the first window wins a tie
        r = f32(f32(re8x[k] * re8x[k]) + f32(re8y[k] * re8y[k]))
        if r > maxr:
            maxr, maxk = r, k
    ang = fast_atan2(re8y[maxk], re8x[maxk])
    return (f32(np.float64(ang) + 2.0 * np.pi) if ang < 0 else ang), maxk   # :1734 (2.0f * M_PI is a double)


def orient_disc():
    """(i, j) of the 109 samples in ascending thread order: tix = 0..207, i = (tix & 15) - 6, j = tix / 16 - 6, r2 < 36 (:1691-1694)"""
    out = []
    for tix in range(13 * 16):
        i, j = (tix & 15) - 6, tix // 16 - 6
        if i * i + j * j < 36:
            out.append((i, j))
    assert len(out) == 109
    return out


# ---------------------------------------------------------------------------------------------- MLDB cells
def pair_table():
    """setCompareIndices akazed.cu:65-159: (cell_j, cell_i, channel) for each of the 486 bits, j < i"""
    out = []
    for lo, hi in ((0, 4), (4, 13), (13, 29)):
        for ch in range(3):
            for j in range(lo, hi - 1):
                for i in range(j + 1, hi):
                    out.append((j, i, ch))
    assert len(out) == 486
    return out


def cell_rowcol(cell):
    """(grid, row group, column group) of accumulator cell 0..28: cell = y2*2+x2 | 4+y3*3+x3 | 13+y4*4+x4 (:1931-1953)"""
    if cell < 4:
        return 2, cell // 2, cell % 2
    if cell < 13:
        return 3, (cell - 4) // 3, (cell - 4) % 3
    return 4, (cell - 13) // 4, (cell - 13) % 4


def bits_from_rule(rule):
    """61 descriptor bytes from rule(cell_j, cell_i, channel) -> bool  (bit i of byte b = pair 8b + i, :1987-1999)"""
    out = np.zeros(61, np.uint8)
    for n, (j, i, ch) in enumerate(pair_table()):
        if rule(j, i, ch):
            out[n // 8] |= 1 << (n % 8)
    return out


# ------------------------------------------------------------------------------------------------ matcher
def match_descriptors(dists, n_extra_far=0):
    """query = all zero; train j = the first dists[j] bits set -> Hamming distance dists[j] (61 bytes, D9)"""
    from numpy import packbits
    rows = []
    for d in list(dists) + [300] * n_extra_far:
        bits = np.zeros(488, np.uint8)
        bits[:d] = 1
        rows.append(packbits(bits, bitorder="little")[:61])
    return np.zeros(61, np.uint8), np.stack(rows)


# (name, train distances by index, expected (match, distance))  -- akazed.cu:2144-2241:
#   lane t keeps the strict minimum over j = t (mod 16), first index on a tie (:2180 `dist < distance[tid]`);
#   the swap-reduce (:2190-2204) permutes the 16 lane minima so that distance[0] is the overall minimum;
#   flags[t] = distance[0] < distance[t] (:2207), accepted iff all 15 other lanes are strictly worse and distance < 96 (:2223)
FAR = 200
MATCH_CASES = [
    ("unique minimum", [FAR] * 3 + [5] + [FAR] * 28, (3, 5)),
    ("same minimum in two lanes -> 14 flags", [FAR] * 3 + [5] + [FAR] * 16 + [5] + [FAR] * 11, (-1, -1)),      # indices 3 and 20: lanes 3 and 4
    ("same minimum twice in ONE lane -> first index", [FAR] * 3 + [5] + [FAR] * 15 + [5] + [FAR] * 12, (3, 5)),  # indices 3 and 19: lane 3
    ("95 is accepted", [FAR] * 17 + [95] + [FAR] * 14, (17, 95)),
    ("96 is not (distance < MAX_DIST)", [FAR] * 17 + [96] + [FAR] * 14, (-1, -1)),
    ("minimum in the last lane", [FAR] * 15 + [7] + [FAR] * 16, (15, 7)),
    ("exactly 16 train points", [FAR] * 9 + [1] + [FAR] * 6, (9, 1)),
]


# ----------------------------------------------------------------------------------------- contrast factor
def kcontrast_literal(grad, per):
    """hScharrContrast on a gradient plane grad[h][w], as its kernels and host half literally compute it:
    the maximum (akazed.cu:2413 floor; gFindMaxContrastU4 :827-877 launched :2435): only thread 0 of every 32 x 32-pixel block
    feeds atomicMax, with the largest of ITS four pixels (32 bx + {0,16}, 32 by + {0,16}) -- the "reduction" loop compares against
    absolute pixels of the image's top-left tile, never against the block's other threads -- so what arrives deterministically is the
    maximum over x % 16 == 0 and y % 16 == 0, for the ceil((n / 2) / 16) blocks per axis the grid has;
    the bins (gConstrastHistShared :901-938, grid :2454): 32 x 16 threads per block, `if (ix >= width && iy >= height) return` --
    threads right of the image (rows < h) and below it (columns < w) still count what they read, zeros of the reused arena
    (akaze.cpp:142-149), into bin 0;
    the host half :2450, 2468-2481."""
    g = np.asarray(grad, np.float32)
    h, w = g.shape
    hmax = f32(0.03)                                                # :2413
    for by in range((h // 2 + 15) // 16):                           # :2435 grid1
        for bx in range((w // 2 + 15) // 16):
            for y in (32 * by, 32 * by + 16):                       # thread 0: iy0, iy1 (:832-835); the guards :836, 843, 847, 853
                for x in (32 * bx, 32 * bx + 16):
                    if x < w and y < h:
                        hmax = max(hmax, g[y, x])
    hfactor = f32(300) / f32(hmax)                                  # :2450 NBINS / h_max_contrast
    hist = np.zeros(300, np.int64)
    for iy in range((h + 15) // 16 * 16):                           # :2454 grid2 x block2 (32, 16)
        for ix in range((w + 31) // 32 * 32):
            if ix >= w and iy >= h:                                 # :909
                continue
            v = g[iy, ix] if (ix < w and iy < h) else f32(0)        # outside: pitch padding / the head of the next (zeroed) plane
            hi = int(np.float64(v) * np.float64(hfactor))           # :924 __fmul_rz then float -> int: truncation of the exact product
            hist[min(hi, 299)] += 1                                 # :925-928
    thresh = int(f32(f32(w * h - hist[0]) * f32(per)))              # :2468 (int * float -> float -> int, towards zero)
    cumuv, k = 0, 1
    while k < 300:                                                  # :2472-2480: k ends one past the bin that reached the threshold
        if cumuv >= thresh:
            break
        cumuv += hist[k]
        k += 1
    return f32(f32(k) / hfactor), hmax, hist                        # :2481


def ramp_plane(w, h, slopes):
    """smooth(x) = cumulative sum of slopes[1..w-1], constant along y.  Scharr (akazed.cu:664-666): dy = 0,
    dx = 10 * (s[x+1] - s[x-1]) + 3 * (2 * (s[x+1] - s[x-1])) = 16 * (slopes[x] + slopes[x+1]) for 1 <= x <= w-2, and 0 in the
    first and the last column (reflect-101: both neighbours are the same pixel).  Slopes are dyadic, so everything is exact."""
    s = np.zeros(w, np.float64)
    for x in range(1, w):
        s[x] = s[x - 1] + slopes[x]
    grad_row = np.zeros(w, np.float64)
    for x in range(1, w - 1):
        grad_row[x] = 16.0 * (slopes[x] + slopes[x + 1])
    return np.tile(s.astype(np.float32), (h, 1)), np.tile(grad_row.astype(np.float32), (h, 1))
