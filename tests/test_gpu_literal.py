"""GPU twin of tests/test_reference_literal_cpu.py: the hand-derived micro-fixtures of tests/literal_fixtures.py, run
through the C ABI against the HIP kernels of the launch sequence (hak_op_tail_* / hak_op_orient_describe / hak_match /
hak_op_*).  The expectations come from the fixture file (worked out by hand from the cited reference lines); the oracle
appears only where a fixture needs float planes nobody can derive by hand (blob determinants)."""
import ctypes as C

import numpy as np
import pytest

import literal_fixtures as lf

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _akazer(ah, w, h, **kw):
    det = ah.Akazer()
    det.init((w, h, ah.iAlignUp(w, 128)), max_pts=1000, **kw)
    return det


# ------------------------------------------------------------------------------------------------ disc NMS
@pytest.mark.parametrize("fast", [False, True])
def test_nms_cursor_lag_and_tie_rule(ah, fast):
    det = _akazer(ah, lf.NMS_W, lf.NMS_H)
    assert [f32(v) for v in det.schedule()["sizes"]] == lf.SIZES and det.schedule()["noct"] == 1
    resp, _, layer = lf.nms_maps(fast)
    for _ in range(2):                                               # twice: the sequence leaves the key map clean behind it
        det.tail_begin()
        det.tail_seed(resp[:, :lf.NMS_W], layer[:, :lf.NMS_W])
        pts, total = det.tail_finish(refine=False, fast=fast)
        exp = lf.nms_expected()
        assert total == len(exp)
        assert [(int(p["x"]), int(p["y"]), int(p["octave"])) for p in pts] == [(x, y, l) for x, y, l, _ in exp]
        for p, (x, y, l, r) in zip(pts, exp):
            assert p["size"] == lf.SIZES[l] and p["response"] == (int(r * 100) if fast else f32(r))
    det.close()


# ------------------------------------------------------------------------------------------- extrema map
def test_extrema_border_filter_threshold_and_scatter(ah):
    """the stand-alone extrema kernel (dilation > 4 fallback) on hand-made determinant planes, then NMS (all peaks are isolated)"""
    det = _akazer(ah, lf.EXT_W, lf.EXT_H, noctaves=2)
    assert det.schedule()["noct"] == 2 and [f32(v) for v in det.schedule()["borders"][:4]] == lf.BORDERS
    dets, exp = lf.extrema_fixture()
    det.tail_begin()
    for o in (0, 1):
        for s in range(4):
            det.tail_det_level(o, s, dets[o][s])
    pts, total = det.tail_finish(refine=False)
    assert total == len(exp)
    got = [(int(p["x"]), int(p["y"]), int(p["octave"]), p["response"]) for p in pts]
    assert [(g[0], g[1], g[2]) for g in got] == [(e[0], e[1], e[2]) for e in exp]
    assert all(g[3] == e[3] for g, e in zip(got, exp))
    assert all(p["size"] == lf.SIZES[int(p["octave"]) % 4] for p in pts)
    det.close()


@pytest.mark.parametrize("stream", ["2", "0"], ids=["streaming kernel", "tile kernel"])
def test_extrema_border_filter_in_the_fused_hessian_kernels(ah, monkeypatch, stream):
    """the border rule inside k_hessian_stream / k_hessian_fused: blob images whose determinant peaks sit on and next to the
    first and last accepted column of each dilation"""
    monkeypatch.setenv("HAK_HESS_STREAM", stream)
    det = _akazer(ah, lf.EXT_W, lf.EXT_H, noctaves=1)
    for s, (plane, ok) in lf.blob_border_fixture().items():
        det.tail_begin()
        det.tail_level(0, s, plane)
        pts, total = det.tail_finish(refine=False)
        assert [(int(p["x"]), int(p["y"])) for p in pts] == ok and total == len(ok), s
        assert all(int(p["octave"]) == s for p in pts)
    det.close()


@pytest.mark.parametrize("stream", ["2", "0"], ids=["streaming kernel", "tile kernel"])
def test_refine_on_hand_made_blobs_matches_the_oracle(ah, okz, monkeypatch, stream):
    """gRefine (akazed.cu:1615-1662): the HIP kernel re-evaluates the 3 x 3 determinants from the derivative plane; elliptical,
    off-centre blobs give non-zero offsets in both axes.  (The Newton step itself is pinned by hand on the CPU side.)"""
    monkeypatch.setenv("HAK_HESS_STREAM", stream)
    yy, xx = np.mgrid[0:lf.EXT_H, 0:lf.EXT_W].astype(np.float64)
    img = np.zeros((lf.EXT_H, lf.EXT_W))
    for cx, cy, sx, sy in ((60.3, 70.4, 3.0, 4.0), (120.7, 60.2, 4.0, 3.0), (100.4, 115.3, 3.5, 3.5), (135.1, 100.9, 3.0, 3.2)):
        img += np.exp(-((xx - cx) ** 2 / (2 * sx * sx) + (yy - cy) ** 2 / (2 * sy * sy)))
    img = img.astype(np.float32)
    det = _akazer(ah, lf.EXT_W, lf.EXT_H, noctaves=1)
    for s in (0, 2, 3):
        det.tail_begin()
        det.tail_level(0, s, img)
        pts, total = det.tail_finish(refine=True)
        lp = np.zeros((lf.EXT_H, 256), np.float32); lp[:, :lf.EXT_W] = img
        _, _, odet = okz.hessian(lp, lf.EXT_W, lf.SIGMA[s])
        dets = np.zeros((4, lf.EXT_H, 256), np.float32); dets[s] = odet
        resp = np.full((lf.EXT_H, 256), f32(-0.0926474631), np.float32); size = resp.copy()
        layer = np.full((lf.EXT_H, 256), -1, np.int32)
        okz.extrema_map(dets, lf.EXT_W, np.array(lf.BORDERS + lf.SIZES, np.float32), 0, lf.EXT_THRESHOLD, (resp, size, layer), 256)
        opts, ototal = okz.nms(resp, size, layer, lf.EXT_W, lf.PSZ)
        assert total == ototal == 4
        for g, o in zip(pts, opts):
            r = okz.refine_point(o, odet, 0)
            assert (g["x"], g["y"]) == (r["x"], r["y"]) and g["response"] == o["response"]
            assert g["x"] != np.floor(g["x"]) and g["y"] != np.floor(g["y"])      # really refined
    det.close()


# ---------------------------------------------------------------------------------- orientation + MLDB
def _record(ah, x, y, layer, size, angle=0.0):
    p = np.zeros(1, ah.POINT_DTYPE)
    p["x"], p["y"], p["octave"], p["size"], p["angle"] = x, y, layer, size, angle
    return p


def _dense(fn, w=200, h=180):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.zeros((h, w), np.float32) if fn is None else np.asarray(fn(xx, yy), np.float32) + np.zeros((h, w), np.float32)


def _set_level(ah, det, o, s, lt=None, lx=None, ly=None):
    w, h, _ = det.geometry()[o]
    det.set_plane(0, o, s, _dense(lt, w, h))                         # HAK_PLANE_LT
    det.set_plane(2, o, s, _dense(lx, w, h))                         # HAK_PLANE_LX
    det.set_plane(3, o, s, _dense(ly, w, h))                         # HAK_PLANE_LY


@pytest.mark.parametrize("vec,exp", [
    ((1.0, 0.0), f32(0.0)),
    ((0.0, 1.0), f32(1.5707963267948966)),
    ((-1.0, 0.0), f32(np.pi)),
    ((0.0, -1.0), f32(np.float64(-f32(1.5707963267948966)) + 2.0 * np.pi)),
])
def test_orient_constant_field(ah, vec, exp):
    det = _akazer(ah, 200, 180, noctaves=2)
    _set_level(ah, det, 0, 0, lx=lambda x, y: vec[0], ly=lambda x, y: vec[1])
    out = det.orient_describe(_record(ah, 100, 90, 0, lf.SIZES[0]), desc=1)
    assert out["angle"][0] == exp
    det.close()


def test_orient_window_wraps_around_bin_41(ah, okz):
    det = _akazer(ah, 200, 180, noctaves=2)
    cx, cy, step = 100, 90, 2
    # rows above the keypoint point just below +pi, the keypoint's row and the rows below it just above -pi (fixture of the CPU twin)
    _set_level(ah, det, 0, 0, lx=lambda x, y: -1.0, ly=lambda x, y: np.where(y < cy, 0.0625, -0.0625))
    out = det.orient_describe(_record(ah, cx, cy, 0, lf.SIZES[0]), desc=1)
    # literal reading of akazed.cu:1703-1734 (tests/literal_fixtures.py) with the exp(-0.08 r2) table, which is pinned elsewhere
    # (test_host_tables_match_oracle): here only the window logic matters
    samples = []
    w36 = okz.orient_weights()
    for i, j in lf.orient_disc():
        b = 0.0625 if j < 0 else -0.0625
        samples.append((f32(w36[i * i + j * j] * f32(-1.0)), f32(w36[i * i + j * j] * f32(b)), 41 if j < 0 else 1))
    lit, win = lf.orient_literal(samples, None)
    assert win == 37 and out["angle"][0] == lit
    det.close()


MLDB_CASES = [
    # name, level (o, s), planes, record (x, y, layer, size, angle), rule
    ("rows along image x", (0, 0), dict(lt=lambda x, y: 1000 - x), (100, 90, 0, 1.0, 0.0),
     lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[1] < lf.cell_rowcol(i)[1]),
    ("columns along image y", (0, 0), dict(lt=lambda x, y: 1000 - y), (100, 90, 0, 1.0, 0.0),
     lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[2] < lf.cell_rowcol(i)[2]),
    ("rx sums Ly, ry sums Lx", (0, 0), dict(lx=lambda x, y: 1000 - y, ly=lambda x, y: 1000 - x), (100, 90, 0, 1.0, 0.0),
     lambda j, i, ch: (ch == 1 and lf.cell_rowcol(j)[1] < lf.cell_rowcol(i)[1]) or (ch == 2 and lf.cell_rowcol(j)[2] < lf.cell_rowcol(i)[2])),
    ("octave 1, scale 3", (1, 1), dict(lt=lambda x, y: 1000 - x), (100, 90, 5, lf.SIZES[1], 0.0),
     lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[1] < lf.cell_rowcol(i)[1]),
    ("window row 20 is in the 3x3 grid only", (0, 0), dict(lt=lambda x, y: np.where(x == 110, -50.0, 0.0)), (100, 90, 0, 1.0, 0.0),
     lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[0] == 3 and lf.cell_rowcol(i)[1] == 2 and lf.cell_rowcol(j)[1] != 2),
    ("quarter turn", (0, 0), dict(lt=lambda x, y: 1000 - x), (100, 90, 0, 1.0, float(f32(np.pi / 2))),
     lambda j, i, ch: ch == 0 and lf.cell_rowcol(j)[2] > lf.cell_rowcol(i)[2]),
]


@pytest.mark.parametrize("plan", ["1", "0"], ids=["k_describe_runs", "k_describe"])
@pytest.mark.parametrize("case", MLDB_CASES, ids=[c[0] for c in MLDB_CASES])
def test_mldb_cell_assignment(ah, monkeypatch, plan, case):
    name, (o, s), planes, rec, rule = case
    monkeypatch.setenv("HAK_DESC_PLAN", plan)
    det = _akazer(ah, 200, 180, noctaves=2)
    _set_level(ah, det, o, s, **planes)
    out = det.orient_describe(_record(ah, *rec), desc=2)             # descriptor alone, rotated by the record's angle
    assert np.array_equal(out["features"][0], lf.bits_from_rule(rule)), name
    det.close()


# ------------------------------------------------------------------------------------------------ matcher
@pytest.mark.parametrize("name,dists,exp", lf.MATCH_CASES, ids=[c[0] for c in lf.MATCH_CASES])
def test_match_swap_reduce_and_flags(ah, name, dists, exp):
    q, train = lf.match_descriptors(dists)
    d1, d2 = ah.AkazeData(), ah.AkazeData()
    ah.initAkazeData(d1, 4, True, True)
    ah.initAkazeData(d2, len(train), True, True)
    d1.h_data["features"][0] = q
    d2.h_data["features"] = train
    d2.h_data["x"] = np.arange(len(train)) + 0.5
    d2.h_data["y"] = np.arange(len(train)) + 100.25
    ah.check(ah.lib.hak_memcpy_h2d(d1.d_data, d1.h_data.ctypes.data, d1.h_data.nbytes))
    ah.check(ah.lib.hak_memcpy_h2d(d2.d_data, d2.h_data.ctypes.data, d2.h_data.nbytes))
    d1.num_pts, d2.num_pts = 1, len(train)
    ah.cuMatch(d1, d2)
    p = d1.h_data[0]
    assert (int(p["match"]), int(p["distance"])) == exp
    if exp[0] >= 0:
        assert p["match_x"] == f32(exp[0] + 0.5) and p["match_y"] == f32(exp[0] + 100.25)
    else:
        assert p["match_x"] == -1 and p["match_y"] == -1
    ah.freeAkazeData(d1); ah.freeAkazeData(d2)


# ----------------------------------------------------------------------------------------- contrast factor
def _kcontrast_op(ah, torch, smooth, w, per):
    h = smooth.shape[0]
    sm = np.zeros((h, 128), np.float32); sm[:, :w] = smooth
    kc, hmax = C.c_float(), C.c_float()
    hist = np.zeros(300, np.int32)
    d_sm = torch.from_numpy(sm).cuda()
    ah.check(ah.lib.hak_op_kcontrast(d_sm.data_ptr(), w, h, 128, per, C.byref(kc), C.byref(hmax), hist.ctypes.data_as(C.POINTER(C.c_int))))
    return f32(kc.value), f32(hmax.value), hist


@pytest.mark.parametrize("per", [0.5, 0.7, 0.25, 0.9])
def test_kcontrast_threshold_loop(ah, torch, per):
    """the maximum is the LATTICE maximum (2.0; the true maximum 4.0 lies between lattice columns), akazed.cu:827-877"""
    s = np.zeros(129)
    s[48:88] = 2.0 ** -10; s[88:118] = 2.0 ** -4; s[118:128] = 2.0 ** -3
    smooth, grad = lf.ramp_plane(128, 32, s)
    kc, hmax, hist = _kcontrast_op(ah, torch, smooth, 128, per)
    ekc, ehmax, ehist = lf.kcontrast_literal(grad, per)
    assert ehmax == f32(2.0)
    assert hmax == ehmax and np.array_equal(hist, ehist) and kc == ekc


@pytest.mark.parametrize("per", [0.7, 0.25])
def test_kcontrast_threads_outside_the_image_count_zeros(ah, torch, per):
    """40 x 24: 896 threads of the 32 x 16 histogram blocks lie beside / below the image and count zeros (akazed.cu:909): thresh < 0, k = 1"""
    s = np.zeros(41)
    s[10:18] = 2.0 ** -5; s[18:26] = 2.0 ** -4; s[26:36] = 2.0 ** -7
    smooth, grad = lf.ramp_plane(40, 24, s)
    kc, hmax, hist = _kcontrast_op(ah, torch, smooth, 40, per)
    ekc, ehmax, ehist = lf.kcontrast_literal(grad, per)
    assert ehmax == f32(1.0) and ehist[0] == 13 * 24 + 896 and ekc == f32(1) / f32(300)
    assert hmax == ehmax and np.array_equal(hist, ehist) and kc == ekc


def test_kcontrast_lattice_stops_where_the_grid_stops(ah, torch):
    """w = 97: gFindMaxContrastU4's grid (akazed.cu:2435) has 3 blocks of 32 columns; lattice column 96 belongs to none"""
    rng = np.random.default_rng(5)
    smooth = rng.random((32, 97)).astype(np.float32)
    smooth[17:, 90:] += 50.0                                         # a step under row 16 in the last columns: |grad(96, 16)| ~ 800
    kc, hmax, hist = _kcontrast_op(ah, torch, smooth, 97, 0.7)
    # the gradient plane by the literal Scharr of the fixtures' own reflect-101 indexing (akazed.cu:664-666, 162-170)
    pad = np.pad(smooth.astype(np.float32), 1, mode="reflect")
    dx = f32(10) * (pad[1:-1, 2:] - pad[1:-1, :-2]) + f32(3) * (pad[:-2, 2:] + pad[2:, 2:] - pad[:-2, :-2] - pad[2:, :-2])
    dy = f32(10) * (pad[2:, 1:-1] - pad[:-2, 1:-1]) + f32(3) * (pad[2:, :-2] + pad[2:, 2:] - pad[:-2, :-2] - pad[:-2, 2:])
    grad = np.sqrt((dx * dx + dy * dy).astype(np.float32)).astype(np.float32)
    ekc, ehmax, ehist = lf.kcontrast_literal(grad, 0.7)
    assert ehmax == grad[0:32:16, 0:96:16].max() and grad[16, 96] > 100 * ehmax                 # column 96 is not consulted
    assert hmax == ehmax and np.array_equal(hist, ehist) and kc == ekc


# ----------------------------------------------------------------------------------------- down + smooth
def test_down_smooth_mirror_on_source_extents(ah, torch):
    k = ah.gauss_taps(1.0, 2)
    sw, sh, sp, dw, dh, dp = 32, 24, 128, 16, 12, 128
    src = np.zeros((sh, sp), np.float32); src[22, 10] = 1.0
    d_dst = torch.zeros((dh, dp), dtype=torch.float32, device="cuda"); d_sm = torch.zeros_like(d_dst)
    d_src = torch.from_numpy(src).cuda()
    ah.check(ah.lib.hak_op_down_smooth(d_src.data_ptr(), d_dst.data_ptr(), d_sm.data_ptr(), sw, sh, sp, dw, dh, dp))
    dst, sm = d_dst.cpu().numpy(), d_sm.cpu().numpy()
    r = k[0]
    assert dst[11, 5] == 1.0 and dst[:, :dw].sum() == 1.0
    assert sm[11, 5] == f32(f32(k[0] * r) + f32(k[1] * f32(0 + r)))          # akazed.cu:490: borderAdd(22, 2, 24) = 22
    assert sm[10, 5] == f32(f32(f32(k[0] * 0) + f32(k[1] * f32(0 + r))) + f32(k[2] * f32(0 + r)))
    assert sm[9, 5] == f32(k[2] * r) and sm[8, 5] == 0


def test_lowpass_reflect_101_impulses(ah, torch):
    k = ah.gauss_taps(2.56, 4)
    w, h, p = 40, 33, 128
    src = np.zeros((h, p), np.float32); src[0, 1] = 1.0
    d_dst = torch.zeros((h, p), dtype=torch.float32, device="cuda")
    d_src = torch.from_numpy(src).cuda()
    ah.check(ah.lib.hak_op_lowpass(d_src.data_ptr(), d_dst.data_ptr(), w, h, p, 2.56, 4))
    dst = d_dst.cpu().numpy()
    r0 = [f32(k[1] * f32(2)), f32(k[0] + k[2]), f32(k[1] + k[3]), f32(k[2] + k[4]), k[3], k[4], f32(0)]   # akazed.cu:227-237
    for x in range(7):
        assert dst[0, x] == f32(r0[x] * k[0]) and dst[2, x] == f32(k[2] * f32(r0[x] + 0)), x
    src[:] = 0; src[h - 2, 20] = 1.0
    d_src = torch.from_numpy(src).cuda()
    ah.check(ah.lib.hak_op_lowpass(d_src.data_ptr(), d_dst.data_ptr(), w, h, p, 2.56, 4))
    assert d_dst.cpu().numpy()[h - 1, 20] == f32(k[1] * f32(k[0] + k[0]))   # akazed.cu:284
