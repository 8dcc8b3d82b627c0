"""CPU suite, part 3: the oracle against hand-derived micro-fixtures of the reference's control-flow-heavy stages.

The oracle is pinned to reference CODE only for the FED tau table; every other stage is a restatement, and a restatement
error (the disc-NMS cursor lag, akazed.cu:1579-1593) survived two rounds because every test compared the HIP kernels with
the same restatement.  This file is the second reader: each expectation in tests/literal_fixtures.py is worked out by hand
from the cited reference statements, independently of oracle/*.c.  tests/test_gpu_literal.py runs the same fixtures through
the C ABI on the GPU.
"""
import numpy as np
import pytest

import literal_fixtures as lf

f32 = np.float32


def _pt(okz, x, y, layer, size, angle=0.0):
    p = np.zeros(1, okz.POINT_DTYPE)[0]
    p["x"], p["y"], p["octave"], p["size"], p["angle"] = x, y, layer, size, angle
    return p


def _pitched(dense, p, dtype=np.float32):
    h, w = dense.shape
    out = np.zeros((h, p), dtype)
    out[:, :w] = dense
    return out


# ------------------------------------------------------------------------------------------------ disc NMS
@pytest.mark.parametrize("fast", [False, True])
def test_nms_cursor_lag_and_tie_rule(okz, fast):
    resp, size, layer = lf.nms_maps(fast)
    pts, total = okz.nms(resp, size, layer, lf.NMS_W, lf.PSZ, fast=fast)
    got = [(int(p["x"]), int(p["y"]), int(p["octave"])) for p in pts]
    exp = [(x, y, l) for x, y, l, _ in lf.nms_expected()]
    assert total == len(exp) and got == exp
    for p, (x, y, l, r) in zip(pts, lf.nms_expected()):
        assert p["size"] == lf.SIZES[l]                              # akazed.cu:1609 size_map[idx]
        assert p["response"] == (int(r * 100) if fast else f32(r))   # D8: the map's response
    # the three survivors a clean disc would have dropped are exactly the ones the cursor lag protects
    for xy in lf.NMS_LAG_ONLY:
        assert xy in [(x, y) for x, y, _ in got]


def test_nms_clean_disc_would_differ():
    """documents the quirk: an independent clean-disc reading of akazed.cu:1573-1593 (what SURVEY 9.8 assumed) drops exactly
    the NMS_LAG_ONLY points of the fixture -- i.e. the fixture does tell the two readings apart"""
    resp, size, layer = lf.nms_maps()
    keep = []
    for x, y, l, r, _, _ in lf.NMS_CANDIDATES:
        if not (lf.PSZ <= x and x + lf.PSZ < lf.NMS_W and lf.PSZ <= y and y + lf.PSZ < lf.NMS_H):
            continue
        isz, sq = int(lf.SIZES[l] + f32(0.5)), int(lf.SIZES[l] * lf.SIZES[l])
        sup = False
        for i in range(-isz, isz + 1):
            for j in range(-isz, isz + 1):
                if (i or j) and i * i + j * j < sq:
                    rn = resp[y + i, x + j]
                    sup |= bool(rn > f32(r) or (rn == f32(r) and i <= 0 and j <= 0))
        if not sup:
            keep.append((x, y))
    lit = [(x, y) for x, y, _, _ in lf.nms_expected()]
    assert sorted(set(lit) - set(keep)) == sorted(lf.NMS_LAG_ONLY) and set(keep) <= set(lit)


# ------------------------------------------------------------------------------------------- extrema map
def _run_extrema(okz, dets, fast=False):
    P0 = 256
    rdt = np.int32 if fast else np.float32
    resp = np.full((lf.EXT_H, P0), -1061109568 if fast else f32(-0.0926474631), rdt)
    size = np.full((lf.EXT_H, P0), f32(-0.0926474631), np.float32)
    layer = np.full((lf.EXT_H, P0), -1, np.int32)
    params = np.array(lf.BORDERS + lf.SIZES, np.float32)
    for o in sorted(dets):
        d = dets[o]
        ms, h, w = d.shape
        p = 128 if o else 256
        dp = np.zeros((ms, h, p), rdt)
        dp[:, :, :w] = d
        okz.extrema_map(dp, w, params, o, 65 if fast else lf.EXT_THRESHOLD, (resp, size, layer), P0, fast=fast)
    return resp, size, layer


def test_extrema_border_filter_threshold_and_scatter(okz):
    dets, exp = lf.extrema_fixture()
    resp, size, layer = _run_extrema(okz, dets)
    ys, xs = np.nonzero(layer >= 0)
    got = sorted(((int(x), int(y), int(layer[y, x]), resp[y, x]) for y, x in zip(ys, xs)), key=lambda t: (t[1], t[0]))
    assert [(g[0], g[1], g[2]) for g in got] == [(e[0], e[1], e[2]) for e in exp]
    assert all(g[3] == e[3] for g, e in zip(got, exp))
    for x, y, l, _ in exp:
        assert size[y, x] == lf.SIZES[l % 4]                         # akazed.cu:1371 d_extrema_param[max_scale + curr_scale]


def test_extrema_fast_uses_the_same_rule(okz):
    """fastakaze::gCalcExtremaMap (akazed.cu:3476-3515) is the float kernel with int planes and threshold 65 (akaze.cpp:559)"""
    dets, exp = lf.extrema_fixture()
    idets = {o: np.where(d > 0, np.maximum((d * 1000).astype(np.int32), 0), 0).astype(np.int32) for o, d in dets.items()}
    # float threshold cases do not carry over (0.001 * 1000 = 1 < 65): the integer planes get their own two pixels
    idets[0][2, 50, 50] = 65                                         # == threshold: rejected
    idets[0][2, 50, 60] = 66
    resp, size, layer = _run_extrema(okz, idets, fast=True)
    ys, xs = np.nonzero(layer >= 0)
    got = sorted(((int(x), int(y), int(layer[y, x])) for y, x in zip(ys, xs)), key=lambda t: (t[1], t[0]))
    assert got == [(e[0], e[1], e[2]) for e in exp]


def test_extrema_through_the_hessian_of_blobs(okz):
    """the same column rule, but with determinant planes the oracle computes itself from blob images"""
    for s, (plane, ok) in lf.blob_border_fixture().items():
        lp = _pitched(plane, 256)
        _, _, det = okz.hessian(lp, lf.EXT_W, lf.SIGMA[s])
        dets = np.zeros((4, lf.EXT_H, 256), np.float32)
        dets[s] = det
        resp = np.full((lf.EXT_H, 256), f32(-0.0926474631), np.float32)
        size = resp.copy()
        layer = np.full((lf.EXT_H, 256), -1, np.int32)
        okz.extrema_map(dets, lf.EXT_W, np.array(lf.BORDERS + lf.SIZES, np.float32), 0, lf.EXT_THRESHOLD, (resp, size, layer), 256)
        ys, xs = np.nonzero(layer >= 0)
        assert sorted(zip(xs.tolist(), ys.tolist()), key=lambda t: (t[1], t[0])) == ok, s


# ------------------------------------------------------------------------------------------------ refine
@pytest.mark.parametrize("name,pos,nb", lf.REFINE_CASES, ids=[c[0] for c in lf.REFINE_CASES])
def test_refine_newton_step(okz, name, pos, nb):
    x, y, o = pos
    w, h, p = (200 >> o), (180 >> o), 256
    det = np.zeros((h, p), np.float32)
    xi, yi = x >> o, y >> o
    n = dict(ul=0, ur=0, ll=0, lr=0); n.update(nb)
    det[yi, xi] = n["c"]; det[yi, xi - 1] = n["l"]; det[yi, xi + 1] = n["r"]; det[yi - 1, xi] = n["u"]; det[yi + 1, xi] = n["d"]
    det[yi - 1, xi - 1] = n["ul"]; det[yi - 1, xi + 1] = n["ur"]; det[yi + 1, xi - 1] = n["ll"]; det[yi + 1, xi + 1] = n["lr"]
    got = okz.refine_point(_pt(okz, x, y, o * 4, lf.SIZES[0]), det, o)
    ex, ey = lf.refine_expected(x, y, o, **nb)
    assert (got["x"], got["y"]) == (ex, ey), (name, got["x"], got["y"], ex, ey)


def test_refine_expectations_are_what_the_comments_say():
    assert lf.refine_expected(40, 40, 0, c=10, l=6, r=8, u=7, d=7) == (f32(40) + f32(f32(1) / f32(36)) * f32(6), f32(40))
    assert lf.refine_expected(44, 44, 0, c=10, l=9, r=10.9, u=7, d=7) == (f32(44), f32(44))
    assert lf.refine_expected(48, 40, 0, c=10, l=8.5, r=10.5, u=9, d=9) == (f32(49), f32(40))
    assert lf.refine_expected(52, 40, 0, c=3, l=3, r=3, u=3, d=3) == (f32(52), f32(40))


# ---------------------------------------------------------------------------------------------- orientation
def _orient_case(okz, field):
    """field(i, j) -> (Lx, Ly) at disc offset (i, j); returns (oracle angle, literal angle, literal window)"""
    w, h, p = 200, 180, 256
    cx, cy, step = 100, 90, 2                                        # size 2.4 -> step = (int)(2.4 + 0.5f) = 2 (:1687)
    lx = np.zeros((h, p), np.float32); ly = np.zeros((h, p), np.float32)
    wt = okz.orient_weights()
    samples = []
    for i, j in lf.orient_disc():
        a, b = field(i, j)
        lx[cy + step * j, cx + step * i] = a; ly[cy + step * j, cx + step * i] = b      # :1698 pos
        dx, dy = f32(wt[i * i + j * j] * f32(a)), f32(wt[i * i + j * j] * f32(b))     # :1699-1700
        samples.append((dx, dy))
    return lx, ly, samples, (cx, cy)


@pytest.mark.parametrize("vec,angle_bin,exp", [
    ((1.0, 0.0), 21, f32(0.0)),                                      # atan2 = 0 -> (int)(0) + 21
    ((0.0, 1.0), 31, f32(1.5707963267948966)),                       # pi/2 * 21/pi = 10.5 -> 10 + 21; dFastAtan2: a = 0 -> r = H_PI
    ((-1.0, 0.0), 41, f32(np.pi)),                                   # (float)pi * 21/pi = 21.0000006 -> 42 -> min(.., 41); r = M_PI - 0
    ((0.0, -1.0), 11, f32(np.float64(-f32(1.5707963267948966)) + 2.0 * np.pi)),   # -10.5 -> -10 + 21; angle < 0 -> + 2 pi (double)
])
def test_orient_constant_field(okz, vec, angle_bin, exp):
    lx, ly, samples, (cx, cy) = _orient_case(okz, lambda i, j: vec)
    lit, win = lf.orient_literal([(dx, dy, angle_bin) for dx, dy in samples], None)
    got = okz.orient_point(_pt(okz, cx, cy, 0, lf.SIZES[0]), lx, ly, 0, 200)
    # all samples share one bin: every window holding it sums to the same value, the first one in scan order wins (:1726 strict >)
    assert win == max(angle_bin - 6, 0)
    assert lit == exp and got["angle"] == exp


def test_orient_window_wraps_around_bin_41(okz):
    """rows j < 0 point just below +pi (bin 41), rows j >= 0 just above -pi (bin 1): only the wrapping windows 37..41 hold both
    (k < 42 ? k : k - 42, akazed.cu:1714-1715); the first of them wins"""
    up, dn = (-1.0, 0.0625), (-1.0, -0.0625)
    lx, ly, samples, (cx, cy) = _orient_case(okz, lambda i, j: up if j < 0 else dn)
    disc = lf.orient_disc()
    binned = [(dx, dy, 41 if j < 0 else 1) for (dx, dy), (i, j) in zip(samples, disc)]
    # bins: atan2(0.0625, -1) = pi - 0.0624 -> (int)(20.58) + 21 = 41; atan2(-0.0625, -1) = -(pi - 0.0624) -> -20 + 21 = 1  (:1702)
    lit, win = lf.orient_literal(binned, None)
    assert win == 37
    got = okz.orient_point(_pt(okz, cx, cy, 0, lf.SIZES[0]), lx, ly, 0, 200)
    assert got["angle"] == lit
    assert abs(float(lit) - np.pi) < 0.02 and lit > f32(np.pi)       # sum_y < 0 (the j = 0 row sides with `dn`) -> just past pi


# ---------------------------------------------------------------------------------------------- MLDB cells
def _describe(okz, lt, lx, ly, x, y, layer, size, angle, o=0, w=200):
    pt = _pt(okz, x, y, layer, size, angle)
    return okz.describe_point(pt, lt, lx, ly, o, w)["features"]


def _planes(fn_lt=None, fn_lx=None, fn_ly=None, w=200, h=180, p=256):
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for fn in (fn_lt, fn_lx, fn_ly):
        pl = np.zeros((h, p), np.float32)
        if fn is not None:
            pl[:, :w] = fn(xx, yy)
        out.append(pl)
    return out


# With angle 0 (co = 1, si = 0): xp = xf + scale * k, yp = yf + scale * l, k = window ROW - size2, l = window COLUMN - size2
# (akazed.cu:1919-1922) -- the window row walks along image x.  Integer-valued planes keep every cell sum exact.
def test_mldb_intensity_follows_window_rows_along_image_x(okz):
    lt, lx, ly = _planes(fn_lt=lambda x, y: 1000 - x)
    got = _describe(okz, lt, lx, ly, 100, 90, 0, f32(1.0), 0.0)      # size 1.0 -> scale = (int)(1.5) = 1
    # Lt falls with image x = window row: a cell in a lower row group holds the larger sum; acc[j] > acc[i] (:1996) iff row(j) < row(i)
    exp = lf.bits_from_rule(lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[1] < lf.cell_rowcol(i)[1])
    assert exp[0] == 0x1E                                            # 2x2 pairs (0,1)(0,2)(0,3)(1,2)(1,3)(2,3) -> 0 1 1 1 1 0
    assert np.array_equal(got, exp)


def test_mldb_intensity_follows_window_columns_along_image_y(okz):
    lt, lx, ly = _planes(fn_lt=lambda x, y: 1000 - y)
    got = _describe(okz, lt, lx, ly, 100, 90, 0, f32(1.0), 0.0)
    exp = lf.bits_from_rule(lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[2] < lf.cell_rowcol(i)[2])
    assert exp[0] == 0x25                                            # -> 1 0 1 0 0 1
    assert np.array_equal(got, exp)


def test_mldb_derivative_channels_rx_ry(okz):
    """rx = -dx * si + dy * co, ry = dx * co + dy * si (akazed.cu:1927-1928): at angle 0 channel 1 sums Ly and channel 2 sums Lx"""
    lt, lx, ly = _planes(fn_lx=lambda x, y: 1000 - y, fn_ly=lambda x, y: 1000 - x)
    got = _describe(okz, lt, lx, ly, 100, 90, 0, f32(1.0), 0.0)
    exp = lf.bits_from_rule(lambda j, i, ch: (ch == 1 and lf.cell_rowcol(j)[1] < lf.cell_rowcol(i)[1]) or
                                             (ch == 2 and lf.cell_rowcol(j)[2] < lf.cell_rowcol(i)[2]))
    assert np.array_equal(got, exp)


def test_mldb_scale_and_octave_ratio(okz):
    """layer 5 = octave 1: xf = pt.x * (1 / (1 << o)) (:1882-1885), samples every scale = (int)(size + 0.5f) = 3 px; monotone planes
    give the same bits as at scale 1"""
    lt, lx, ly = _planes(fn_lt=lambda x, y: 1000 - x, w=100, h=90, p=128)
    got = _describe(okz, lt, lx, ly, 100, 90, 5, lf.SIZES[1], 0.0, o=1, w=100)
    exp = lf.bits_from_rule(lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[1] < lf.cell_rowcol(i)[1])
    assert np.array_equal(got, exp)


def test_mldb_grid_extents(okz):
    """window row 20 (k = +10) lies outside the 2x2 grid (m < 2 * size2 = 20, :1930) and the 4x4 grid (m < 4 * size4 = 20, :1948) but
    inside the 3x3 grid (m < 3 * size3 = 21, :1939): darkening the image column it samples lowers the row-group-2 cells of the
    3x3 grid only"""
    lt, lx, ly = _planes(fn_lt=lambda x, y: np.where(x == 110, -50.0, 0.0))
    got = _describe(okz, lt, lx, ly, 100, 90, 0, f32(1.0), 0.0)
    exp = lf.bits_from_rule(lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[0] == 3 and lf.cell_rowcol(i)[1] == 2 and lf.cell_rowcol(j)[1] != 2)
    assert np.array_equal(got, exp)
    # one column earlier (k = +9) it reaches all three grids
    lt, lx, ly = _planes(fn_lt=lambda x, y: np.where(x == 109, -50.0, 0.0))
    got = _describe(okz, lt, lx, ly, 100, 90, 0, f32(1.0), 0.0)
    last = {2: 1, 3: 2, 4: 3}
    exp = lf.bits_from_rule(lambda j, i, ch: ch == 0 and lf.cell_rowcol(i)[1] == last[lf.cell_rowcol(i)[0]] and
                            lf.cell_rowcol(j)[1] != last[lf.cell_rowcol(j)[0]])
    assert np.array_equal(got, exp)


def test_mldb_rotation_by_a_quarter_turn(okz):
    """angle = (float)(pi/2): si = 1, co = -4.37e-8: xp = (int)(xf - l + 0.5f + tiny), yp = yf + k -- the window COLUMN now walks
    against image x, so Lt = 1000 - x grows with the column group: acc[j] > acc[i] iff col(j) > col(i)"""
    lt, lx, ly = _planes(fn_lt=lambda x, y: 1000 - x)
    got = _describe(okz, lt, lx, ly, 100, 90, 0, f32(1.0), f32(np.pi / 2))
    exp = lf.bits_from_rule(lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[2] > lf.cell_rowcol(i)[2])
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------------------------------------ matcher
@pytest.mark.parametrize("name,dists,exp", lf.MATCH_CASES, ids=[c[0] for c in lf.MATCH_CASES])
def test_match_swap_reduce_and_flags(okz, name, dists, exp):
    q, train = lf.match_descriptors(dists)
    p1 = np.zeros(1, okz.POINT_DTYPE); p2 = np.zeros(len(train), okz.POINT_DTYPE)
    p1["features"][0] = q
    p2["features"] = train
    p2["x"] = np.arange(len(train)) + 0.5; p2["y"] = np.arange(len(train)) + 100.25
    okz.match(p1, p2)
    assert (int(p1["match"][0]), int(p1["distance"][0])) == exp
    if exp[0] >= 0:
        assert p1["match_x"][0] == f32(exp[0] + 0.5) and p1["match_y"][0] == f32(exp[0] + 100.25)     # :2228-2229
    else:
        assert p1["match_x"][0] == -1 and p1["match_y"][0] == -1                                      # :2235-2236


# ----------------------------------------------------------------------------------------- contrast factor
def _slopes():
    s = np.zeros(129, np.float64)
    s[48:88] = 2.0 ** -10
    s[88:118] = 2.0 ** -4
    s[118:128] = 2.0 ** -3
    return s


@pytest.mark.parametrize("per", [0.5, 0.7, 0.25, 0.9])
def test_kcontrast_threshold_loop(okz, per):
    """one row of hand-made gradient values repeated: the LATTICE maximum (the true maximum 4.0 sits off the 16-px lattice), bins,
    threshold and the `k` that is one past the stopping bin (:827-877, 2468-2481)"""
    smooth, grad = lf.ramp_plane(128, 32, _slopes())
    # the ramp's gradient is what the comment of ramp_plane says (Scharr, akazed.cu:664-666)
    g = okz.scharr_grad(_pitched(smooth, 128), 128)
    assert np.array_equal(g, grad)
    kc, hmax, hist = okz.kcontrast(g, 128, per)
    ekc, ehmax, ehist = lf.kcontrast_literal(grad, per)
    # worked example: the row is 47 zeros, 2^-6 at x = 47, 2^-5 at 48..86, 1.015625 at 87, 2.0 at 88..116, 3.0 at 117, 4.0 at 118..126,
    # 0 at 127.  Thread 0 of the blocks sees columns 0, 16, .., 112 only: 0, 0, 0, 2^-5, 2^-5, 2^-5, 2.0, 2.0 -> hmax = 2, hfactor = 150
    # (a true maximum would give 4 and 75).  Bins: 2^-6 -> 2, 2^-5 -> 4, 1.015625 -> 152, everything from 2.0 up -> 300 -> 299.
    # 128 = 4 * 32 and 32 = 2 * 16: no thread outside the image, no extra zeros.
    assert hmax == ehmax == f32(2.0) and np.array_equal(hist, ehist) and kc == ekc
    row = ehist // 32
    assert row[0] == 48 and row[2] == 1 and row[4] == 39 and row[152] == 1 and row[299] == 39 and row.sum() == 128
    if per == 0.5:
        # thresh = (int)(80 * 32 * 0.5f) = 1280 = bins 2 + 4 exactly: the loop adds bin 4, k becomes 5, then breaks: 5 / 150
        assert kc == f32(5) / f32(150)


def _slopes_40():
    s = np.zeros(41, np.float64)
    s[10:18] = 2.0 ** -5
    s[18:26] = 2.0 ** -4
    s[26:36] = 2.0 ** -7
    return s


@pytest.mark.parametrize("per", [0.7, 0.25])
def test_kcontrast_threads_outside_the_image_count_zeros(okz, per):
    """40 x 24: the histogram blocks are 32 x 16 threads and return only when BOTH coordinates are outside (akazed.cu:909), so
    (64 - 40) * 24 + (32 - 24) * 40 = 896 threads beside and below the image add zeros to bin 0 -- more than the 648 non-zero pixels,
    `thresh` (:2468) goes negative and the loop stops at once: k = 1"""
    smooth, grad = lf.ramp_plane(40, 24, _slopes_40())
    g = okz.scharr_grad(_pitched(smooth, 128), 40)
    assert np.array_equal(g[:, :40], grad)
    kc, hmax, hist = okz.kcontrast(g, 40, per)
    ekc, ehmax, ehist = lf.kcontrast_literal(grad, per)
    # the row: 9 zeros, 0.5 at x = 9, 1.0 at 10..16, 1.5 at 17, 2.0 at 18..24, 1.125 at 25, 0.25 at 26..34, 0.125 at 35, 4 zeros.
    # lattice columns 0, 16, 32 hold 0, 1.0, 0.25: hmax = 1 (true maximum 2), hfactor = 300; bins 0.125 -> 37, 0.25 -> 75, 0.5 -> 150,
    # 1.0 and above -> 299
    assert hmax == ehmax == f32(1.0) and np.array_equal(hist, ehist) and kc == ekc
    assert ehist[0] == 13 * 24 + 896 and ehist[37] == 24 and ehist[75] == 216 and ehist[150] == 24 and ehist[299] == 384
    assert ehist.sum() == 40 * 24 + 896
    # thresh = (int)((960 - 1208) * per) < 0 -> cumuv = 0 >= thresh at k = 1 (with the guard read as `||`: (int)(648 * 0.7f) = 453,
    # reached only in bin 299 -> k = 300, kcontrast 1.0 instead of 1 / 300)
    assert kc == f32(1) / f32(300)


def test_kcontrast_lattice_stops_where_the_grid_stops(okz):
    """grid1 = ceil((w / 2) / 16) blocks of 32 columns (akazed.cu:2435): w = 97 -> (48 + 15) / 16 = 3 blocks = columns 0..95; the
    lattice column x = 96 belongs to no block, although it is inside the image"""
    g = np.zeros((32, 97), np.float32)
    g[16, 96] = 5.0                                                  # on the lattice, outside the grid
    g[16, 80] = 0.5                                                  # on the lattice, inside
    g[3, 7] = 9.0                                                    # off the lattice
    kc, hmax, hist = okz.kcontrast(_pitched(g, 128), 97, 0.7)
    ekc, ehmax, ehist = lf.kcontrast_literal(g, 0.7)
    assert hmax == ehmax == f32(0.5) and np.array_equal(hist, ehist) and kc == ekc


def test_kcontrast_floor(okz):
    g = np.full((16, 64), f32(0.01), np.float32)
    kc, hmax, hist = okz.kcontrast(g, 64, 0.7)
    assert hmax == f32(0.03)                                         # akazed.cu:2413: h_max_contrast starts at 0.03f
    # hfactor = 300 / 0.03f = 10000.0002.. -> 10000.0f; 0.01f = 0.0099999998 -> product 99.999998, __fmul_rz + truncation -> bin 99 (:924);
    # thresh = (int)(1024 * 0.7f) = 716; bins 1..98 are empty, bin 99 brings cumuv to 1024 and k to 100 (:2472-2480)
    assert hist[99] == 16 * 64
    assert kc == f32(f32(100) / (f32(300) / f32(0.03)))


# ----------------------------------------------------------------------------------------- down + smooth
def test_down_smooth_mirror_on_source_extents(okz):
    """gDownWithSmooth akazed.cu:449-511: taps at source distance 2 and 4, mirrored with borderAdd on the SOURCE height (:490) -- so
    the tap one decimated row below the last row lands on the last row itself (sh -> sh - 2), not on the row before it"""
    k = okz.gauss_taps(1.0, 2)
    sw, sh, sp, dw, dh, dp = 32, 24, 128, 16, 12, 128
    src = np.zeros((sh, sp), np.float32)
    src[22, 10] = 1.0                                                # last sampled row (2 * 11), sampled column (2 * 5)
    dst, sm = okz.down_smooth(src, sw, dw, dh, dp)
    assert dst[11, 5] == 1.0 and dst.sum() == 1.0                    # :506 dst = src[2y][2x]
    r = k[0]                                                         # row pass at dx = 5: k0 * s[10]  (:469-471)
    # column pass at dy = 11 (:507-509): y1 = 20, y3 = borderAdd(22, 2, 24) = 22, y0 = 18, y4 = borderAdd(22, 4, 24) = 20
    assert sm[11, 5] == f32(f32(k[0] * r) + f32(k[1] * f32(0 + r)))
    # dy = 10 (siy 20): y3 = 22, y4 = borderAdd(20, 4, 24) = 22: both the k1 and the k2 tap see the impulse row
    assert sm[10, 5] == f32(f32(f32(k[0] * 0) + f32(k[1] * f32(0 + r))) + f32(k[2] * f32(0 + r)))
    # dy = 9 (siy 18): only the k2 tap (y4 = 22)
    assert sm[9, 5] == f32(k[2] * r)
    assert sm[8, 5] == 0
    # an impulse on an odd row or column is never sampled at all
    src[:] = 0; src[21, 10] = 1.0; src[22, 11] = 1.0
    dst, sm = okz.down_smooth(src, sw, dw, dh, dp)
    assert dst.sum() == 0 and sm.sum() == 0


def test_down_smooth_right_edge_and_odd_height(okz):
    k = okz.gauss_taps(1.0, 2)
    sw, sh, sp, dw, dh, dp = 32, 13, 128, 16, 6, 128
    src = np.zeros((sh, sp), np.float32)
    src[12, 30] = 1.0                                                # row 12 = 2 * 6 is never a centre (dh = 6) but is a tap of rows 5 and 4
    dst, sm = okz.down_smooth(src, sw, dw, dh, dp)
    assert dst.sum() == 0
    # row pass at dx = 15 (six 30): x3 = borderAdd(30, 2, 32) = 30 -> k0 + k1 * (s[28] + s[30]) + k2 * (s[26] + s[28]) = k0 + k1
    r15 = f32(f32(k[0] * 1) + f32(k[1] * f32(0 + 1)))
    # dy = 5 (siy 10): y3 = 12, y4 = borderAdd(10, 4, 13) = 26 - 2 - 14 = 10 -> only the k1 tap
    assert sm[5, 15] == f32(k[1] * f32(0 + r15))
    # dy = 4 (siy 8): y4 = 12 -> the k2 tap
    assert sm[4, 15] == f32(k[2] * f32(0 + r15))
    # dx = 14 (six 28): x3 = 30, x4 = borderAdd(28, 4, 32) = 30 -> k1 + k2
    r14 = f32(f32(f32(k[0] * 0) + f32(k[1] * f32(0 + 1))) + f32(k[2] * f32(0 + 1)))
    assert sm[5, 14] == f32(k[1] * f32(0 + r14))


# ------------------------------------------------------------------------------------------------ low-pass
def test_lowpass_reflect_101_impulses(okz):
    """gConv2d<R> akazed.cu:204-290: reflect-101 doubles a tap that mirrors onto the same pixel"""
    k = okz.gauss_taps(2.56, 4)
    w, h, p = 40, 33, 128
    src = np.zeros((h, p), np.float32)
    src[0, 1] = 1.0
    dst = okz.lowpass(src, w, 2.56, 4)
    # row pass of row 0 (:227-237), impulse at x = 1, reflect-101 on the left (abs(ix - i)):
    #   x = 0: k1 * (s[|0-1|] + s[1]) = k1 * 2          x = 1: k0 + k2 * (s[|1-2|] + s[3]) = k0 + k2
    #   x = 2: k1 * s[1] + k3 * s[|2-3|] = k1 + k3      x = 3: k2 + k4 (s[|3-4|])      x = 4: k3      x = 5: k4      x = 6: 0
    r0 = [f32(k[1] * f32(2)), f32(k[0] + k[2]), f32(k[1] + k[3]), f32(k[2] + k[4]), k[3], k[4], f32(0)]
    # column pass at y = 0: R[0] * k0 + k_i * (R[|0-i|] + R[i]) with only row 0 non-zero -> R[0] * k0 (:281-286)
    for x in range(7):
        assert dst[0, x] == f32(r0[x] * k[0]), x
        assert dst[2, x] == f32(k[2] * f32(r0[x] + 0)), x            # y = 2: k2 * (R[|2-2|] + R[4]) = k2 * R[0]
    # bottom edge: impulse in the last row is seen twice by the row above the last one? no: borderAdd(h-2, 1, h) = h-1 once, and
    # borderAdd(h-1, 1, h) = h-2: the LAST row's own k1 taps both read row h-2
    src[:] = 0; src[h - 2, 20] = 1.0
    dst = okz.lowpass(src, w, 2.56, 4)
    assert dst[h - 1, 20] == f32(k[1] * f32(k[0] + k[0]))            # :284 k1 * (sdata[toy-1] + sdata[toy+1]) with both = row h-2
