"""GPU parity, whole path: Akazer.detectAndCompute x2 + cuMatch through the C ABI against the CPU
oracle and the committed golden fixtures.  Bar (BASELINE.json north_star): keypoint set / count and
MLDB descriptor bits bit-exact; Hessian responses within 1e-4 relative (we require bit-equality and
assert the tolerance separately so a failure says which bar broke)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_points_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def synth():
    from akaze_hip import synth
    return synth


def gpu_detect(ah, torch, synth, u8, max_pts=10000, desc=True, keep=False, **kw):
    """the reference demo's call sequence (main.cpp:172-205) on one image"""
    h, w = u8.shape
    p = ah.iAlignUp(w, 128)
    img = torch.from_numpy(synth.to_float(u8, p)).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=max_pts, **kw)
    data = ah.AkazeData()
    ah.initAkazeData(data, max_pts, True, True)
    det.detectAndCompute(img.data_ptr(), data, (w, h, p), desc)
    pts = data.h_data[:data.num_pts].copy()
    if keep:
        return pts, det, data
    ah.freeAkazeData(data)
    det.close()
    return pts


def okz_params(okz, **kw):
    m = dict(kw)
    return okz.default_params(**m)


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def load_cases():
    return _mg().SYNTH_CASES


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c[0])
def test_synth_golden(ah, torch, synth, golden, case):
    name, w, h, seed, kw = case
    pts = gpu_detect(ah, torch, synth, _mg().case_scene(w, h, seed), **kw)
    g = golden.synth[name + "_pts"]
    assert_points_equal(pts, g)
    rel = np.abs(pts["response"] - g["response"]) / np.abs(g["response"])
    assert rel.max() <= 1e-4


@pytest.mark.parametrize("level_tile", ["1", "2"], ids=["streaming kernels", "k_level_tile"])
def test_planes_small(ah, okz, torch, synth, monkeypatch, level_tile):
    """every persistent plane of every level, bit for bit (localises a failing stage)"""
    monkeypatch.setenv("HAK_LEVEL_TILE", level_tile)
    w, h = 320, 240
    u8 = _mg().case_scene(w, h, 11)
    pts, det, data = gpu_detect(ah, torch, synth, u8, keep=True)
    r = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(w, 128)), w, keep_arena=True)
    assert np.float32(det.kcontrast()).view(np.uint32) == r.kcontrast.view(np.uint32)
    assert len(det.geometry()) == r.noct
    for o in range(r.noct):
        for s in range(4):
            for kind, nm in ((0, "Lt"), (2, "Lx"), (3, "Ly"), (1, "det")):
                a, b = det.plane(kind, o, s), okz.plane(r, kind, o, s)
                bad = (a.view(np.uint32) != b.view(np.uint32)).sum()
                assert bad == 0, f"{nm}({o},{s}): {bad} px differ, max abs {np.abs(a - b).max()}"
    assert_points_equal(pts, r.points)
    ah.freeAkazeData(data)
    det.close()


def test_reference_images_and_match(ah, torch, synth, golden):
    """data/left.pgm vs right.pgm (the reference's bundled pair, main.cpp:141-142): detect + describe + match"""
    res = []
    for name in ("left", "right"):
        pts, det, data = gpu_detect(ah, torch, synth, golden.lr_u8[name], keep=True)
        res.append((pts, det, data))
    assert_points_equal(res[0][0], golden.lr["pts1"])
    assert_points_equal(res[1][0], golden.lr["pts2"])
    ah.cuMatch(res[0][2], res[1][2])
    m = res[0][2].h_data[:res[0][2].num_pts]
    for f in ("match", "distance", "match_x", "match_y"):
        assert np.array_equal(m[f], golden.lr["pts1"][f]), f
    assert (m["match"] >= 0).sum() == 2464
    for _, det, data in res:
        ah.freeAkazeData(data)
        det.close()


@pytest.mark.parametrize("w,h,kw,nshapes", [(1920, 1080, {}, None), (1280, 720, {}, None),
                                            (3840, 2160, dict(noctaves=5, upright=True), None),     # dense scene: the 10 000-point clamp
                                            (3840, 2160, dict(noctaves=5, upright=True), 330)],     # configs[2]'s ~8 k keypoints, unclamped
                         ids=["1080p", "720p", "4k_5oct_upright_clamped", "4k_5oct_upright_8k"])
def test_full_size_vs_oracle(ah, okz, torch, synth, w, h, kw, nshapes):
    """BASELINE.json configs 2-4 at full size, against the oracle run live on the same seeded scene"""
    u8 = synth.scene(w, h, 2 if nshapes else 1, nshapes)
    pts = gpu_detect(ah, torch, synth, u8, **kw)
    okw = {k: int(v) for k, v in kw.items()}
    r = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(w, 128)), w, okz.default_params(**okw))
    assert len(r.points) > 500
    if nshapes:
        assert 7000 < len(r.points) < 9500                     # (bench.py's 4K scene: 8 194 keypoints)
    assert_points_equal(pts, r.points)


def test_natural_1080p_pair_vs_oracle(ah, okz, torch, synth, golden):
    """the natural-image 1080p fixture: img1 / img2 of BASELINE configs[0], reconstructed from the reference's own result
    pictures (tools/ref_render_check.py; the PNGs themselves are missing from the checkout).  Float path + match and FAST
    path against the oracle run live; the oracle's counts on it are the ones the render check compares with the
    reference's screenshot (2156 / 2295 vs 2205 / 2382; FAST 2664 / 2844 vs 2690 / 2915)."""
    rec = np.load(os.path.join(golden.dir, "ref_recon_1080p_u8.npz"))
    res, ora = [], []
    for name, n in (("img1", 2156), ("img2", 2295)):
        u8 = rec[name]
        pts, det, data = gpu_detect(ah, torch, synth, u8, keep=True)
        r = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(u8.shape[1], 128)), u8.shape[1])
        assert len(r.points) == n
        assert_points_equal(pts, r.points)
        res.append((det, data))
        ora.append(r.points)
    ah.cuMatch(res[0][1], res[1][1])
    okz.match(ora[0], ora[1])
    m = res[0][1].h_data[:res[0][1].num_pts]
    for f in ("match", "distance", "match_x", "match_y"):
        assert np.array_equal(m[f], ora[0][f]), f
    assert (m["match"] >= 0).sum() == 1353
    for det, data in res:
        ah.freeAkazeData(data)
        det.close()
    for name, n in (("img1", 2664), ("img2", 2844)):
        r = okz.fast_detect_and_compute(rec[name])
        assert len(r.points) == n
        assert_points_equal(gpu_fast_detect(ah, torch, rec[name]), r.points)


@pytest.mark.parametrize("pinned,side", [(True, "0"), (False, "0"), (True, "1")],
                         ids=["pinned h_data", "pageable h_data", "pinned h_data, octave-0 Hessians on a side stream"])
def test_pair_call_equals_the_three_calls(ah, okz, torch, synth, golden, monkeypatch, pinned, side):
    """hak_detect_and_compute_pair / Akazer.detectAndComputePair: both images + cuMatch as ONE launch sequence -- byte for byte what
    detectAndCompute x 2 + cuMatch leave in the two AkazeData (device and host arrays), on the reference's bundled pair; with
    unequal capacities every image keeps ITS OWN clamp (setMaxNumPoints(result.max_pts), akaze.cpp:246, 451) -- that call comes
    FIRST, on a fresh context, so that no earlier full-capacity call can have left the expected records in the pair buffer --
    and without the match"""
    monkeypatch.setenv("HAK_HESS_SIDE", side)           # (read by hak_create: the launch order HAK_HESS_SIDE=1 selects is off by default)
    a, b = golden.lr_u8["left"], golden.lr_u8["right"]
    h, w = a.shape
    p = ah.iAlignUp(w, 128)
    imgs = [torch.from_numpy(synth.to_float(u, p)).cuda() for u in (a, b)]
    g1, g2 = golden.lr["pts1"], golden.lr["pts2"]
    d = [ah.AkazeData() for _ in range(6)]
    for k, cap in enumerate((10000, 10000, 10000, 2500, 1200, 10000)):
        ah.initAkazeData(d[k], cap, True, True, pinned=pinned)
    # ---- unequal capacities first, fresh context: image 1 unclamped (3631 of 10000), image 2 clamped to its raster-order prefix
    det = ah.Akazer()
    det.init((w, h, p), batch=2)
    det.detectAndComputePair(imgs[0].data_ptr(), imgs[1].data_ptr(), d[2], d[3], (w, h, p), True, True)
    assert d[2].num_pts == len(g1) and d[3].num_pts == 2500
    want = okz.match(g1.copy(), g2[:2500].copy())           # what cuMatch leaves when the train set is the clamped AkazeData
    assert_points_equal(d[2].h_data[:len(g1)], want, fields=("x", "y", "octave", "response", "size", "angle", "features", "match", "distance",
                                                             "match_x", "match_y"))
    assert_points_equal(d[3].h_data[:2500], g2[:2500])
    # ... the other way round, without the match: the fields stay at -1
    det.detectAndComputePair(imgs[0].data_ptr(), imgs[1].data_ptr(), d[4], d[5], (w, h, p), True, False)
    assert d[4].num_pts == 1200 and d[5].num_pts == len(g2)
    assert_points_equal(d[4].h_data[:1200], g1[:1200])
    assert_points_equal(d[5].h_data[:len(g2)], g2)
    assert (d[4].h_data[:1200]["match"] == -1).all()
    det.close()
    # ---- equal capacities: the golden pair with its match fields
    det = ah.Akazer()
    det.init((w, h, p), batch=2)
    det.detectAndComputePair(imgs[0].data_ptr(), imgs[1].data_ptr(), d[0], d[1], (w, h, p), True, True)
    assert d[0].num_pts == len(g1) and d[1].num_pts == len(g2)
    assert_points_equal(d[0].h_data[:d[0].num_pts], g1, fields=("x", "y", "octave", "response", "size", "angle", "features", "match", "distance",
                                                                 "match_x", "match_y"))
    assert_points_equal(d[1].h_data[:d[1].num_pts], g2)
    # the device arrays hold the same records (what a later cuMatch / cuMatchKnn of the caller reads)
    dev = np.zeros(d[0].num_pts, ah.POINT_DTYPE)
    ah.check(ah.lib.hak_memcpy_d2h(dev.ctypes.data, d[0].d_data, dev.nbytes))
    assert dev.tobytes() == d[0].h_data[:d[0].num_pts].tobytes()
    # the call repeats (graph replay of the captured sequence) and a one-image context refuses it
    det.detectAndComputePair(imgs[0].data_ptr(), imgs[1].data_ptr(), d[0], d[1], (w, h, p), True, True)
    assert dev.tobytes() == d[0].h_data[:d[0].num_pts].tobytes()
    one = ah.Akazer()
    one.init((w, h, p))
    with pytest.raises(ah.HakError, match="batch >= 2"):
        one.detectAndComputePair(imgs[0].data_ptr(), imgs[1].data_ptr(), d[0], d[1], (w, h, p))
    one.close()
    for x in d:
        ah.freeAkazeData(x)
    det.close()


def test_max_pts_clamp_is_raster_prefix(ah, okz, torch, synth):
    u8 = _mg().case_scene(640, 480, 21)
    full = okz.detect_and_compute(synth.to_float(u8, 640), 640).points
    assert len(full) > 60
    pts = gpu_detect(ah, torch, synth, u8, max_pts=50)
    assert len(pts) == 50
    assert_points_equal(pts, full[:50])


def test_no_descriptors(ah, okz, torch, synth):
    u8 = _mg().case_scene(320, 240, 11)
    pts = gpu_detect(ah, torch, synth, u8, desc=False)
    r = okz.detect_and_compute(synth.to_float(u8, 384), 320, desc=False)
    assert_points_equal(pts, r.points)
    assert (pts["features"] == 0).all() and (pts["angle"] == 0).all()


@pytest.mark.parametrize("B,level_tile", [(5, "1"), (11, "1"), (16, "1"), (3, "2"), (9, "2")],
                         ids=["5", "11", "16", "3 k_level_tile", "9 k_level_tile"])
def test_batch_equals_single(ah, torch, synth, monkeypatch, B, level_tile):
    """batching: B different images in one launch sequence == B single calls (B below / across / exactly on the XCD groups of 8
    images of hak_xcd_decode; also through the one-launch-per-sublevel kernel with its fused Hessian and the spine order)"""
    monkeypatch.setenv("HAK_LEVEL_TILE", level_tile)
    w, h, mp = 400, 300, 2000
    p = ah.iAlignUp(w, 128)
    imgs = [_mg().case_scene(w, h, 100 + i) for i in range(B)]
    singles = [gpu_detect(ah, torch, synth, u, max_pts=mp) for u in imgs]
    stack = torch.from_numpy(np.stack([synth.to_float(u, p) for u in imgs])).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp, batch=B)
    d_pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
    d_num = torch.zeros(B, dtype=torch.int32, device="cuda")
    ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, stack.data_ptr(), h * p, p, B, d_pts.data_ptr(), d_num.data_ptr(), 1))
    ah.check(ah.lib.hak_sync(det.ctx))
    nums = d_num.cpu().numpy()
    allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    for i in range(B):
        assert nums[i] == len(singles[i]) and nums[i] > 20
        assert_points_equal(allp[i, :nums[i]], singles[i])
    # batched pair matching == single matching
    ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B // 2))
    ah.check(ah.lib.hak_sync(det.ctx))
    allm = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    import okz
    for k in range(B // 2):
        a, b = singles[2 * k].copy(), singles[2 * k + 1]
        okz.match(a, b)
        for f in ("match", "distance", "match_x", "match_y"):
            assert np.array_equal(allm[2 * k, :len(a)][f], a[f]), (k, f)
    det.close()


def test_large_batch_launch_shapes_equal_single(ah, okz, torch, synth):
    """the benchmark's regime: a batch big enough that every streaming kernel runs its full-height row segments (hak_stream_rows:
    four segments of 270 rows per strip at 1080p; fewer images make the launchers cut shorter segments until the grid fills the
    chip: 8 strips x 4 segments x images >= 4096 needs 128 images) -- both paths, 128 x 1080p, against the oracle run live on
    the same two scenes (and against the single-image GPU results)"""
    w, h, mp, B = 1920, 1080, 10000, 128
    p = ah.iAlignUp(w, 128)
    u8s = [synth.scene(w, h, 40 + i) for i in range(2)]
    singles = [gpu_detect(ah, torch, synth, u, max_pts=mp) for u in u8s]
    for u, s1 in zip(u8s, singles):
        assert_points_equal(s1, okz.detect_and_compute(synth.to_float(u, p), w, max_pts=mp).points)
    pad = [np.zeros((h, p), np.uint8) for _ in u8s]
    for q, u in zip(pad, u8s):
        q[:, :w] = u
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp, batch=B)
    fsingles = []
    data = ah.AkazeData()
    ah.initAkazeData(data, mp, True, True)
    for q in pad:
        d8 = torch.from_numpy(q).cuda()
        det.fastDetectAndCompute(d8.data_ptr(), data, (w, h, p), True)
        fsingles.append(data.h_data[:data.num_pts].copy())
    ah.freeAkazeData(data)
    for u, s1 in zip(u8s, fsingles):
        assert_points_equal(s1, okz.fast_detect_and_compute(u, max_pts=mp).points)
    d_pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
    d_num = torch.zeros(B, dtype=torch.int32, device="cuda")
    stack = torch.from_numpy(np.stack([synth.to_float(u8s[i % 2], p) for i in range(B)])).cuda()
    ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, stack.data_ptr(), h * p, p, B, d_pts.data_ptr(), d_num.data_ptr(), 1))
    ah.check(ah.lib.hak_sync(det.ctx))
    nums = d_num.cpu().numpy()
    allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    for i in range(B):
        assert nums[i] == len(singles[i % 2]) > 500, i
        assert_points_equal(allp[i, :nums[i]], singles[i % 2])
    del stack
    stack8 = torch.from_numpy(np.stack([pad[i % 2] for i in range(B)])).cuda()
    ah.check(ah.lib.hak_fast_detect_and_compute_batch(det.ctx, stack8.data_ptr(), h * p, p, B, d_pts.data_ptr(), d_num.data_ptr(), 1))
    ah.check(ah.lib.hak_sync(det.ctx))
    nums = d_num.cpu().numpy()
    allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    for i in range(B):
        assert nums[i] == len(fsingles[i % 2]) > 500, i
        assert_points_equal(allp[i, :nums[i]], fsingles[i % 2])
    det.close()


class default_knobs:
    """run with the kernel-selection environment knobs UNSET (conftest forces the streaming kernels on everywhere), so
    that hak_stream_pays() picks tile or streaming kernels per launch by itself, as in production"""
    KNOBS = ("HAK_HESS_STREAM", "HAK_FUSE_SF", "HAK_BASE_STREAM")

    def __init__(self, ah):
        self.ah = ah

    def __enter__(self):
        self.saved = {k: os.environ.pop(k, None) for k in self.KNOBS}

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            if v is not None:
                os.environ[k] = v


def test_720p_batch64_default_kernel_selection_vs_oracle(ah, okz, torch, synth):
    """BASELINE.json configs[3]'s shape (a batch of independent 1280x720 frames) with the library's own kernel selection:
    64 images make octaves 0-1 take the streaming kernels and octaves 2-3 the tile kernels (hak_stream_pays); every
    image's keypoints, descriptors and every pair's matches against the oracle run live on the four distinct scenes"""
    w, h, mp, B = 1280, 720, 10000, 64
    p = ah.iAlignUp(w, 128)
    pairs = [synth.pair(w, h, 50 + i) for i in range(2)]
    u8s = [pairs[0][0], pairs[0][1], pairs[1][0], pairs[1][1]]
    want = [okz.detect_and_compute(synth.to_float(u, p), w, max_pts=mp).points for u in u8s]
    for k in range(2):
        okz.match(want[2 * k], want[2 * k + 1])
    stack = torch.from_numpy(np.stack([synth.to_float(u8s[i % 4], p) for i in range(B)])).cuda()
    d_pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
    d_num = torch.zeros(B, dtype=torch.int32, device="cuda")
    with default_knobs(ah):
        det = ah.Akazer()
        det.init((w, h, p), max_pts=mp, batch=B)
        ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, stack.data_ptr(), h * p, p, B, d_pts.data_ptr(), d_num.data_ptr(), 1))
        ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), B // 2))
        ah.check(ah.lib.hak_sync(det.ctx))
        det.close()
    nums = d_num.cpu().numpy()
    allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    for i in range(B):
        assert nums[i] == len(want[i % 4]) > 300, i
        fields = ("x", "y", "octave", "response", "size", "angle", "features")
        if i % 2 == 0:
            fields += ("match", "distance", "match_x", "match_y")
        assert_points_equal(allp[i, :nums[i]], want[i % 4], fields=fields)


def test_1080p_default_kernel_selection_vs_oracle(ah, okz, torch, synth):
    """single-image call and a 16-image batch at 1080p with the knobs unset: the tile kernels (single image) and the mix
    the size rule picks at 16 images, both against the oracle"""
    w, h, mp = 1920, 1080, 10000
    p = ah.iAlignUp(w, 128)
    u8 = synth.scene(w, h, 44)
    want = okz.detect_and_compute(synth.to_float(u8, p), w, max_pts=mp).points
    with default_knobs(ah):
        got = gpu_detect(ah, torch, synth, u8, max_pts=mp)
        B = 16
        stack = torch.from_numpy(np.stack([synth.to_float(u8, p)] * B)).cuda()
        d_pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
        d_num = torch.zeros(B, dtype=torch.int32, device="cuda")
        det = ah.Akazer()
        det.init((w, h, p), max_pts=mp, batch=B)
        ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, stack.data_ptr(), h * p, p, B, d_pts.data_ptr(), d_num.data_ptr(), 1))
        ah.check(ah.lib.hak_sync(det.ctx))
        det.close()
    assert_points_equal(got, want)
    nums = d_num.cpu().numpy()
    allp = d_pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    for i in range(B):
        assert nums[i] == len(want)
        assert_points_equal(allp[i, :nums[i]], want)


def test_repeat_calls_are_deterministic(ah, torch, synth):
    u8 = _mg().case_scene(640, 360, 13)
    a = gpu_detect(ah, torch, synth, u8)
    b = gpu_detect(ah, torch, synth, u8)
    assert a.tobytes() == b.tobytes()


# ----------------------------------------------------------------- matcher
def gpu_match(ah, torch, p1, p2):
    d1 = torch.from_numpy(p1.view(np.uint8).copy()).cuda()
    d2 = torch.from_numpy(p2.view(np.uint8).copy()).cuda() if len(p2) else torch.zeros(104, dtype=torch.uint8, device="cuda")
    out = p1.copy()
    ah.check(ah.lib.hak_match(None, d1.data_ptr(), len(p1), d2.data_ptr(), len(p2), out.ctypes.data))
    return out


@pytest.fixture(params=["0", "qt2", "1"], ids=["k_match_mfma", "k_match_mfma 64-query waves", "k_match (VALU)"])
def match_kernel(request, monkeypatch):
    """the matcher kernels: the matrix-core one (default: 32-query waves; HAK_MATCH_QT=2: 64-query waves, one wave per SIMD) and the
    vector-pipe one (HAK_MATCH_VALU=1); both variables are read per call"""
    monkeypatch.setenv("HAK_MATCH_VALU", "1" if request.param == "1" else "0")
    monkeypatch.setenv("HAK_MATCH_QT", "2" if request.param == "qt2" else "1")
    return request.param


@pytest.mark.parametrize("n1,n2", [(1000, 1000), (37, 5), (5, 37), (16, 16), (300, 0), (1, 1), (2205, 2382),
                                   (300, 5000), (3000, 7777), (40, 1023), (40, 1025), (5000, 600),     # sliced / unsliced train sets
                                   # around the 32 x 32 tiles, the 128-query blocks and the 64-row rounds of k_match_mfma
                                   (33, 31), (129, 33), (31, 64), (65, 65), (200, 96), (128, 2047), (127, 97), (256, 32)])
def test_match_vs_oracle(ah, okz, torch, synth, match_kernel, n1, n2):
    base = synth.random_descriptors(max(n2, 1), 7, ah.POINT_DTYPE)[:n2]
    q = synth.random_descriptors(n1, 8, ah.POINT_DTYPE, planted_from=base if n2 else None,
                                 nplanted=min(n1, n2) // 2, maxflip=60)
    q["_pad"] = 0xAB                                 # garbage in the struct padding must not matter (D9)
    base["_pad"] = 0xCD
    got = gpu_match(ah, torch, q, base)
    want = okz.match(q.copy(), base)
    for f in ("match", "distance", "match_x", "match_y"):
        assert np.array_equal(got[f], want[f]), f
    if n2 >= 2 and n1 >= 100:
        assert (got["match"] >= 0).sum() > 0


@pytest.mark.parametrize("shapes", [[(37, 5), (0, 10), (300, 0), (128, 2047)],                       # <= 12 pairs: sliced, device-side counts
                                    [(64, 4096)], [(2205, 2382), (1, 1)],
                                    [(50 + 7 * k, 900 - 31 * k) for k in range(16)]],              # 16 pairs: the unsliced launch
                         ids=["ragged4", "one", "two", "sixteen"])
def test_match_batch_ragged_sets_vs_oracle(ah, okz, torch, synth, match_kernel, shapes):
    """hak_match_batch on hand-made pairs with ragged / empty / tiny sets (n2 < 16 and n2 == 0: D10; n1 == 0): batches of at most
    12 pairs take the sliced search with the counts on the DEVICE (the slice bounds are cut in the kernel), larger ones the plain
    launch; repeated train descriptors put equal distances into different slices"""
    mp = 4100
    det = ah.Akazer()
    det.init((320, 240, 384), max_pts=mp, batch=2 * len(shapes))
    host = np.zeros((2 * len(shapes), mp), ah.POINT_DTYPE)
    num = np.zeros(2 * len(shapes), np.int32)
    rng = np.random.default_rng(3)
    for k, (n1, n2) in enumerate(shapes):
        base = synth.random_descriptors(max(n2, 1), 20 + k, ah.POINT_DTYPE)[:n2]
        if n2 > 64:                                     # duplicates: ties across tiles, classes and slices
            base[n2 // 2:] = base[rng.integers(0, n2 // 2, n2 - n2 // 2)]
            base["x"] = np.arange(n2, dtype=np.float32)
        q = synth.random_descriptors(n1, 40 + k, ah.POINT_DTYPE, planted_from=base if n2 else None, nplanted=min(n1, n2) // 2, maxflip=50)
        host[2 * k, :n1], host[2 * k + 1, :n2] = q, base
        num[2 * k], num[2 * k + 1] = n1, n2
    d_pts = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).cuda()
    d_num = torch.from_numpy(num).cuda()
    for _ in range(2):                                  # twice: the scratch must be back in its initial state after a call
        ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), len(shapes)))
        ah.check(ah.lib.hak_sync(det.ctx))
        got = np.frombuffer(d_pts.cpu().numpy().tobytes(), ah.POINT_DTYPE).reshape(2 * len(shapes), mp)
        for k, (n1, n2) in enumerate(shapes):
            want = okz.match(host[2 * k, :n1].copy(), host[2 * k + 1, :n2].copy())
            for f in ("match", "distance", "match_x", "match_y"):
                assert np.array_equal(got[2 * k, :n1][f], want[f]), (k, n1, n2, f)
    det.close()


def test_match_ties_across_tiles_and_slices(ah, okz, torch, synth, match_kernel):
    """the accept rule counts the residue classes that attain the minimum (akazed.cu:2206) and every class keeps its FIRST minimum:
    a train set made of repeated descriptors puts equal distances into different LDS tiles, different classes and -- for the big
    pair -- different train slices, whose packed keys are merged with atomicMin"""
    rng = np.random.default_rng(5)
    proto = synth.random_descriptors(48, 11, ah.POINT_DTYPE)
    for n1, n2 in ((64, 4096), (4000, 6000)):
        train = proto[rng.integers(0, len(proto), n2)].copy()
        train["x"] = np.arange(n2, dtype=np.float32)
        # every 97th train descriptor is unique (a single class attains its minimum: accepted), the rest are repeats
        uniq = synth.random_descriptors(n2 // 97 + 1, 12, ah.POINT_DTYPE)
        train["features"][::97] = uniq["features"][:len(train[::97])]
        pick = rng.integers(0, n2, n1)
        pick[::2] = rng.choice(np.arange(0, n2, 97), size=len(pick[::2]))     # every other query sits next to a unique train point
        query = train[pick].copy()
        flip = rng.integers(0, 486, n1)
        query["features"][np.arange(n1), flip >> 3] ^= (1 << (flip & 7)).astype(np.uint8)
        query["features"][:, 60] &= 0x3F
        got = gpu_match(ah, torch, query, train)
        want = okz.match(query.copy(), train)
        for f in ("match", "distance", "match_x", "match_y"):
            assert np.array_equal(got[f], want[f]), (n1, n2, f)
        assert 0 < (got["match"] >= 0).sum() < n1               # both outcomes of the rule occur


def test_match_10k_x_10k(ah, okz, torch, synth, match_kernel):
    """BASELINE.json config 5, full size: oracle equality + size-independent properties"""
    train = synth.random_descriptors(10000, 7, ah.POINT_DTYPE)
    query = synth.random_descriptors(10000, 9, ah.POINT_DTYPE, planted_from=train, nplanted=2000, maxflip=40)
    got = gpu_match(ah, torch, query, train)
    want = okz.match(query.copy(), train)
    for f in ("match", "distance", "match_x", "match_y"):
        assert np.array_equal(got[f], want[f]), f
    acc = got["match"] >= 0
    assert acc.sum() >= 1900
    # accepted matches: distance is the true Hamming distance to the matched train descriptor and < 96
    d = np.unpackbits(query["features"][acc] ^ train["features"][got["match"][acc]], axis=1).sum(axis=1)
    assert np.array_equal(d, got["distance"][acc]) and (d < 96).all()
    assert np.array_equal(got["match_x"][acc], train["x"][got["match"][acc]])
    # idempotence: matching again does not change anything
    again = gpu_match(ah, torch, got, train)
    assert again.tobytes() == got.tobytes()


def test_match_batch_capacity_beyond_the_default_grid(ah, okz, torch, synth):
    """device-side counts + few pairs take the sliced search, whose tickets / partial rows exist per 128-query block: the grid must
    follow the context's capacity (a query set above 83 * 128 = 10 624 used to run past both)"""
    mp, n1, n2 = 12000, 11500, 3000
    det = ah.Akazer()
    det.init((320, 240, 384), max_pts=mp, batch=2)
    base = synth.random_descriptors(n2, 61, ah.POINT_DTYPE)
    q = synth.random_descriptors(n1, 62, ah.POINT_DTYPE, planted_from=base, nplanted=2500, maxflip=50)
    host = np.zeros((2, mp), ah.POINT_DTYPE)
    host[0, :n1], host[1, :n2] = q, base
    d_pts = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).cuda()
    d_num = torch.from_numpy(np.array([n1, n2], np.int32)).cuda()
    want = okz.match(q.copy(), base)
    for _ in range(2):
        ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), 1))
        ah.check(ah.lib.hak_sync(det.ctx))
        got = np.frombuffer(d_pts.cpu().numpy().tobytes(), ah.POINT_DTYPE).reshape(2, mp)[0, :n1]
        for f in ("match", "distance", "match_x", "match_y"):
            assert np.array_equal(got[f], want[f]), f
    assert (want["match"][-800:] >= 0).any() or (want["match"] >= 0).sum() > 2000       # the tail beyond query 10 624 is populated
    det.close()


def test_match_sliced_handoff_stress(ah, torch, synth, monkeypatch):
    """the sliced matcher hands its per-slice summaries to the finishing block through agent-scope atomic stores / loads ordered by
    a barrier and a ticket, not by an agent-scope fence (kernels_match.hip: a hardware assumption about gfx950's sc1 accesses, not
    the HIP memory model).  Repeated big pairs and the batched device-count path against the VALU kernel, which has no such
    hand-off: a compiler or cache-policy change that breaks the ordering shows up as a stale partial row"""
    for rep in range(12):
        n1, n2 = 10000 - 37 * rep, 10000 - 101 * rep
        train = synth.random_descriptors(n2, 300 + rep, ah.POINT_DTYPE)
        query = synth.random_descriptors(n1, 400 + rep, ah.POINT_DTYPE, planted_from=train, nplanted=1500, maxflip=45)
        monkeypatch.setenv("HAK_MATCH_VALU", "1")
        ref = gpu_match(ah, torch, query, train)
        monkeypatch.setenv("HAK_MATCH_VALU", "0")
        for _ in range(3):
            got = gpu_match(ah, torch, query, train)
            assert got.tobytes() == ref.tobytes(), rep
    # the pair-call shape: device-side counts, one pair, eight slices
    mp = 10000
    det = ah.Akazer()
    det.init((320, 240, 384), max_pts=mp, batch=2)
    for rep in range(8):
        n1, n2 = 2300 + 211 * rep, 2400 + 173 * rep
        train = synth.random_descriptors(n2, 500 + rep, ah.POINT_DTYPE)
        query = synth.random_descriptors(n1, 600 + rep, ah.POINT_DTYPE, planted_from=train, nplanted=1000, maxflip=45)
        monkeypatch.setenv("HAK_MATCH_VALU", "1")
        ref = gpu_match(ah, torch, query, train)
        monkeypatch.setenv("HAK_MATCH_VALU", "0")
        host = np.zeros((2, mp), ah.POINT_DTYPE)
        host[0, :n1], host[1, :n2] = query, train
        d_pts = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).cuda()
        d_num = torch.from_numpy(np.array([n1, n2], np.int32)).cuda()
        for _ in range(3):
            ah.check(ah.lib.hak_match_batch(det.ctx, d_pts.data_ptr(), d_num.data_ptr(), 1))
            ah.check(ah.lib.hak_sync(det.ctx))
            got = np.frombuffer(d_pts.cpu().numpy().tobytes(), ah.POINT_DTYPE).reshape(2, mp)[0, :n1]
            for f in ("match", "distance", "match_x", "match_y"):
                assert np.array_equal(got[f], ref[f]), (rep, f)
    det.close()


# ----------------------------------------------------------------- parameter space
PARAM_CASES = [
    dict(noctaves=2, max_scale=3),
    dict(noctaves=3, max_scale=5),
    dict(soffset=1.2),                      # base Gaussian radius 3 (ksz 7)
    dict(soffset=2.0, derivative_factor=1.0),
    dict(reordering=False),
    dict(per=0.5, dthreshold=0.0005),
    dict(diffusivity=0),                    # PM_G1   (deterministic exp shared with the oracle)
    dict(diffusivity=2),                    # WEICKERT
    dict(diffusivity=3),                    # CHARBONNIER
    dict(descriptor_pattern_size=12),       # 24x24 sample grid: > 7 samples per lane (tail loop)
    dict(descriptor_pattern_size=6),
    dict(derivative_factor=2.5),            # sigma_size up to 6-7: dilation > 4 -> unfused derivative / extrema fallback
]


@pytest.mark.parametrize("kw", PARAM_CASES, ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_parameter_space_vs_oracle(ah, okz, torch, synth, kw):
    w, h = 512, 384
    u8 = _mg().case_scene(w, h, 31)
    pts = gpu_detect(ah, torch, synth, u8, **kw)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    r = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(w, 128)), w, okz.default_params(**okw))
    assert len(r.points) > 10
    assert_points_equal(pts, r.points)


LEVEL_TILE_CASES = [
    # w, h, seed, parameters: odd extents (no 16-byte rows), every diffusivity, a 5-octave pyramid whose last octave runs FED cycles
    # of 34 / 40 / 48 / 57 steps (more than one launch per sublevel: the continuation variant), upright, other pattern size
    (211, 173, 12, {}),
    (333, 251, 13, dict(diffusivity=0)),
    (320, 240, 14, dict(diffusivity=2)),
    (400, 300, 15, dict(diffusivity=3, upright=True)),
    (1280, 1296, 16, dict(noctaves=5)),
    (640, 360, 17, dict(noctaves=3, descriptor_pattern_size=8)),
]


@pytest.mark.parametrize("w,h,seed,kw", LEVEL_TILE_CASES, ids=lambda v: str(v))
def test_level_tile_vs_oracle(ah, okz, torch, synth, monkeypatch, w, h, seed, kw):
    """kernels_level.hip (one launch per sublevel out of LDS tiles: what single-image calls use) against the oracle, both pipelines"""
    monkeypatch.setenv("HAK_LEVEL_TILE", "2")
    u8 = _mg().case_scene(w, h, seed)
    pts = gpu_detect(ah, torch, synth, u8, **kw)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    r = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(w, 128)), w, okz.default_params(**okw))
    assert len(r.points) > 10
    assert_points_equal(pts, r.points)
    # the integer FAST path through the same kernel template
    p = ah.iAlignUp(w, 128)
    pad = np.zeros((h, p), np.uint8)
    pad[:, :w] = u8
    img = torch.from_numpy(pad).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=10000, **kw)
    data = ah.AkazeData()
    ah.initAkazeData(data, 10000, True, True)
    det.fastDetectAndCompute(img.data_ptr(), data, (w, h, p), True)
    rf = okz.fast_detect_and_compute(u8, okz.default_params(**okw))
    assert_points_equal(data.h_data[:data.num_pts], rf.points)
    ah.freeAkazeData(data)
    det.close()


@pytest.mark.parametrize("w,h,seed,kw", [(640, 480, 31, {}), (324, 200, 32, {}), (1280, 720, 33, dict(noctaves=3)),
                                         (512, 96, 34, dict(noctaves=3)), (960, 540, 35, dict(derivative_factor=1.0, soffset=1.2)),
                                         (256, 256, 36, dict(noctaves=2, upright=True))], ids=lambda v: str(v))
def test_hessian_lp_vs_oracle(ah, okz, torch, synth, monkeypatch, w, h, seed, kw):
    """HAK_HESS_LP=1: k_hessian_stream<., ., LP> low-passes Lt(o,s-1) on the way in and k_fed_sf leaves `smooth` unwritten --
    keypoints and descriptors against the oracle (image edges, short segments, dilations 1..4 through the parameter sets)"""
    monkeypatch.setenv("HAK_HESS_LP", "1")
    u8 = _mg().case_scene(w, h, seed)
    pts = gpu_detect(ah, torch, synth, u8, **kw)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    r = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(w, 128)), w, okz.default_params(**okw))
    assert len(r.points) > 10
    assert_points_equal(pts, r.points)


def test_pinned_results_equal_pageable(ah, torch, synth):
    """h_data in pinned host memory (what the C++ layer's initAkazeData hands out): records and count are written by the launch
    sequence itself; they must equal the pageable route's, call after call, for both images of a pair and for an empty image"""
    w, h, mp = 960, 540, 3000
    p = ah.iAlignUp(w, 128)
    a, b = synth.pair(w, h, 9)
    imgs = [torch.from_numpy(synth.to_float(u, p)).cuda() for u in (a, b)] + [torch.full((h, p), 0.25, dtype=torch.float32, device="cuda")]
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp)
    pin, pag = ah.AkazeData(), ah.AkazeData()
    ah.initAkazeData(pin, mp, True, True, pinned=True)
    ah.initAkazeData(pag, mp, True, True)
    for rep in range(3):
        for i, img in enumerate(imgs):
            det.detectAndCompute(img.data_ptr(), pin, (w, h, p), True)
            det.detectAndCompute(img.data_ptr(), pag, (w, h, p), True)
            assert pin.num_pts == pag.num_pts and (pin.num_pts > 100 or i == 2)
            assert pin.h_data[:pin.num_pts].tobytes() == pag.h_data[:pag.num_pts].tobytes()
    assert pin.num_pts == 0                                    # the flat image came last
    ah.freeAkazeData(pin); ah.freeAkazeData(pag)
    det.close()


def test_serial_equals_concurrent_streams(ah, torch, synth):
    w, h, mp = 640, 480, 4000
    p = ah.iAlignUp(w, 128)
    img = torch.from_numpy(synth.to_float(_mg().case_scene(w, h, 77), p)).cuda()
    outs = []
    for conc in (1, 0):
        det = ah.Akazer()
        det.init((w, h, p), max_pts=mp)
        ah.check(ah.lib.hak_set_concurrency(det.ctx, conc))
        data = ah.AkazeData()
        ah.initAkazeData(data, mp, True, True)
        for _ in range(3):                  # repeated calls reuse the streams / events
            det.detectAndCompute(img.data_ptr(), data, (w, h, p), True)
        outs.append(data.h_data[:data.num_pts].copy())
        ah.freeAkazeData(data)
        det.close()
    assert len(outs[0]) > 50 and outs[0].tobytes() == outs[1].tobytes()


def test_ingest_u8_matches_host_conversion(ah, torch, synth):
    """on-device uint8 -> float32 equals main.cpp:149's host conversion bit for bit"""
    for (w, h, sp, dp) in ((640, 33, 640, 640), (211, 17, 211, 256), (1920, 8, 2048, 1920), (650, 9, 656, 768), (30, 5, 32, 32)):
        rng = np.random.default_rng(w)
        u8 = rng.integers(0, 256, size=(3, h, sp), dtype=np.uint8)
        d_src = torch.from_numpy(u8).cuda()
        d_dst = torch.zeros((3, h, dp), dtype=torch.float32, device="cuda")
        ah.check(ah.lib.hak_ingest_u8(None, d_src.data_ptr(), h * sp, sp, d_dst.data_ptr(), h * dp, dp, w, h, 3))
        torch.cuda.synchronize()
        got = d_dst.cpu().numpy()[:, :, :w]
        want = np.stack([synth.to_float(u8[i, :, :w]) for i in range(3)])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ----------------------------------------------------------------- integer FAST path
def gpu_fast_detect(ah, torch, u8, max_pts=10000, desc=True, **kw):
    h, w = u8.shape
    p = ah.iAlignUp(w, 128)
    pad = np.zeros((h, p), np.uint8)
    pad[:, :w] = u8
    img = torch.from_numpy(pad).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=max_pts, **kw)
    data = ah.AkazeData()
    ah.initAkazeData(data, max_pts, True, True)
    det.fastDetectAndCompute(img.data_ptr(), data, (w, h, p), desc)
    pts = data.h_data[:data.num_pts].copy()
    ah.freeAkazeData(data)
    det.close()
    return pts


@pytest.mark.parametrize("name", ["left", "right"])
def test_fast_path_reference_images(ah, okz, torch, golden, name):
    u8 = golden.lr_u8[name]
    pts = gpu_fast_detect(ah, torch, u8)
    r = okz.fast_detect_and_compute(u8)
    assert len(r.points) > 3000
    assert_points_equal(pts, r.points)


@pytest.mark.parametrize("kw", [dict(), dict(noctaves=3, max_scale=3), dict(soffset=1.2), dict(diffusivity=3), dict(upright=True),
                                dict(descriptor_pattern_size=8), dict(diffusivity=0), dict(diffusivity=2),
                                dict(derivative_factor=2.5),           # dilation > 4: unfused derivative / extrema kernels
                                dict(soffset=2.0, derivative_factor=0.6), dict(reordering=False)],
                         ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()) or "default")
def test_fast_path_parameter_space(ah, okz, torch, kw):
    u8 = _mg().case_scene(512, 384, 41)
    pts = gpu_fast_detect(ah, torch, u8, **kw)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    r = okz.fast_detect_and_compute(u8, okz.default_params(**okw))
    assert len(r.points) > 10
    assert_points_equal(pts, r.points)


def test_fast_path_1080p(ah, okz, torch, synth):
    u8 = synth.scene(1920, 1080, 1)
    pts = gpu_fast_detect(ah, torch, u8)
    r = okz.fast_detect_and_compute(u8)
    assert len(r.points) > 1000
    assert_points_equal(pts, r.points)


def test_fast_path_committed_goldens(ah, torch, golden):
    g = np.load(os.path.join(golden.dir, "fast_oracle.npz"))
    mg = _mg()
    for name, w, h, seed, kw in mg.SYNTH_CASES[:3]:
        pts = gpu_fast_detect(ah, torch, mg.case_scene(w, h, seed), **kw)
        assert_points_equal(pts, g[name + "_pts"])


def test_fast_path_clamp_nodesc_and_batch(ah, okz, torch):
    u8 = _mg().case_scene(640, 480, 43)
    full = okz.fast_detect_and_compute(u8)
    assert len(full.points) > 300
    # clamp: max_pts smaller than the number of NMS survivors keeps the first max_pts in raster order
    pts = gpu_fast_detect(ah, torch, u8, max_pts=200)
    assert_points_equal(pts, okz.fast_detect_and_compute(u8, max_pts=200).points)
    # desc=False: refined keypoints only, angle 0 and features zero
    nd = gpu_fast_detect(ah, torch, u8, desc=False)
    want = okz.fast_detect_and_compute(u8, desc=False).points
    assert_points_equal(nd, want)
    assert not nd["features"].any()
    # batch entry point: 3 different images in one call
    imgs = [u8, _mg().case_scene(640, 480, 44), np.full((480, 640), 9, np.uint8)]
    p = ah.iAlignUp(640, 128)
    stack = np.zeros((3, 480, p), np.uint8)
    for i, im in enumerate(imgs):
        stack[i, :, :640] = im
    d = torch.from_numpy(stack).cuda()
    det = ah.Akazer()
    det.init((640, 480, p), max_pts=2000, batch=3)
    out = torch.zeros(3 * 2000 * ah.POINT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    num = torch.zeros(3, dtype=torch.int32, device="cuda")
    ah.check(ah.lib.hak_fast_detect_and_compute_batch(det.ctx, d.data_ptr(), 480 * p, p, 3, out.data_ptr(), num.data_ptr(), 1))
    ah.check(ah.lib.hak_sync(det.ctx))
    nums = num.cpu().numpy()
    host = out.cpu().numpy().view(ah.POINT_DTYPE).reshape(3, 2000)
    for i, im in enumerate(imgs):
        r = okz.fast_detect_and_compute(im, max_pts=2000)
        assert nums[i] == len(r.points)
        assert_points_equal(host[i, :nums[i]], r.points) if nums[i] else None
    assert nums[2] == 0
    det.close()


# ----------------------------------------------------------------- match post-processing
def _upload_points(ah, torch, pts):
    t = torch.from_numpy(pts.view(np.uint8).reshape(-1).copy()).cuda() if len(pts) else torch.zeros(104, dtype=torch.uint8, device="cuda")
    return t


@pytest.mark.parametrize("n1,n2,ratio,cross", [(260, 300, (1, 1), False), (260, 300, (1, 1), True), (1000, 1700, (4, 5), True),
                                               (1700, 1000, (3, 5), False), (5, 1, (1, 1), True), (5, 0, (1, 1), True),
                                               (10000, 10000, (4, 5), True),
                                               # around the 32 x 32 tiles / 128-query blocks of the matrix-core kernel, 1 and 2 train points
                                               (33, 31, (1, 1), True), (129, 65, (4, 5), True), (64, 2, (1, 1), False), (200, 97, (3, 5), True)])
def test_knn2_matches_oracle(ah, okz, torch, synth, match_kernel, n1, n2, ratio, cross):
    p2 = synth.random_descriptors(max(n2, 1), 3, ah.POINT_DTYPE)[:n2]
    p1 = synth.random_descriptors(n1, 4, ah.POINT_DTYPE, planted_from=p2 if n2 else None, nplanted=min(n1, n2) // 2, maxflip=60)
    if n1 > 8 and n2 > 12:
        p1[7]["features"] = p1[3]["features"]
        p2[11]["features"] = p2[5]["features"]
    want_pts = p1.copy()
    want = okz.match_knn2(want_pts, p2, ratio, cross)
    d1, d2 = _upload_points(ah, torch, p1), _upload_points(ah, torch, p2)
    d_out = torch.zeros(max(n1, 1) * 32, dtype=torch.uint8, device="cuda")
    h_out = np.zeros(max(n1, 1), ah.MATCH_PAIR_DTYPE)
    h_pts = p1.copy()
    cnt = C.c_int(-1)
    ah.check(ah.lib.hak_match_knn2(None, d1.data_ptr(), n1, d2.data_ptr(), n2, ratio[0], ratio[1], int(cross), 0,
                                   h_pts.ctypes.data, d_out.data_ptr(), C.byref(cnt), h_out.ctypes.data))
    assert cnt.value == len(want)
    got = h_out[:cnt.value]
    for f in ah.MATCH_PAIR_DTYPE.names:
        assert np.array_equal(got[f], want[f]), f
    for f in ("match", "distance", "match_x", "match_y"):
        assert np.array_equal(h_pts[f], want_pts[f]), f


@pytest.mark.parametrize("n2", [40, 300, 5000])
def test_knn2_extreme_distances(ah, okz, torch, synth, match_kernel, n2):
    """distances at both ends of the range: all-zero, all-one (61 bytes = 488 bits, D9), alternating and single-bit descriptors on
    both sides -- the matrix-core kernel forms a distance as 2^23 + |b| + sum of signed fp4 products, so |b| = 488 with 488
    negative products (distance 0) and |b| = 0 with 488 positive ones (distance 488) are its corner cases; max_dist 600, ratio
    1/1 and no cross-check, so every query with d1 < d2 reports its two nearest distances unfiltered"""
    rng = np.random.default_rng(n2)
    p2 = synth.random_descriptors(n2, 31, ah.POINT_DTYPE)
    pats = np.zeros((8, 61), np.uint8)
    pats[1] = 0xFF
    pats[2] = 0x55
    pats[3] = 0xAA
    pats[4, 0] = 0x01                                   # one bit at either end of the descriptor
    pats[5, 60] = 0x80
    pats[6] = 0xFF
    pats[6, 30] = 0x7F                                  # all ones but one
    pats[7, :31] = 0xFF                                 # the first lane half's dwords set, the second's clear (bar byte 31's dword)
    where = rng.choice(n2, size=min(n2 // 2, 24), replace=False)
    p2["features"][where] = pats[rng.integers(0, len(pats), len(where))]
    p1 = synth.random_descriptors(64, 32, ah.POINT_DTYPE)
    p1["features"][:32] = pats[np.arange(32) % len(pats)]
    p1["features"][32:40] ^= 0xFF                       # complements of random descriptors: distances near 488 - 243
    want_pts = p1.copy()
    want = okz.match_knn2(want_pts, p2, (1, 1), False, max_dist=600)
    d1, d2 = _upload_points(ah, torch, p1), _upload_points(ah, torch, p2)
    d_out = torch.zeros(len(p1) * 32, dtype=torch.uint8, device="cuda")
    h_out = np.zeros(len(p1), ah.MATCH_PAIR_DTYPE)
    h_pts = p1.copy()
    cnt = C.c_int(-1)
    ah.check(ah.lib.hak_match_knn2(None, d1.data_ptr(), len(p1), d2.data_ptr(), n2, 1, 1, 0, 600,
                                   h_pts.ctypes.data, d_out.data_ptr(), C.byref(cnt), h_out.ctypes.data))
    assert cnt.value == len(want)
    for f in ah.MATCH_PAIR_DTYPE.names:
        assert np.array_equal(h_out[:cnt.value][f], want[f]), f
    assert want["distance"].min() == 0 and want["distance"].max() > 100 and want["second"].max() > 200   # exact hits and far-away neighbours
    # two train points only (all ones, all ones but one): the neighbours of the all-zero query are 487 and 488 bits away
    t3 = synth.random_descriptors(2, 33, ah.POINT_DTYPE)
    t3["features"] = pats[[1, 6]]
    q8 = synth.random_descriptors(8, 34, ah.POINT_DTYPE)
    q8["features"] = pats
    w8 = q8.copy()
    want8 = okz.match_knn2(w8, t3, (1, 1), False, max_dist=600)
    assert want8["distance"].max() == 487 and want8["second"].max() == 488
    dq, dt = _upload_points(ah, torch, q8), _upload_points(ah, torch, t3)
    h8 = np.zeros(8, ah.MATCH_PAIR_DTYPE)
    ah.check(ah.lib.hak_match_knn2(None, dq.data_ptr(), 8, dt.data_ptr(), 2, 1, 1, 0, 600, None, d_out.data_ptr(), C.byref(cnt), h8.ctypes.data))
    assert cnt.value == len(want8)
    for f in ah.MATCH_PAIR_DTYPE.names:
        assert np.array_equal(h8[:cnt.value][f], want8[f]), f
    # the 1-NN entry point on the same sets (accepts only distances < 96)
    got = gpu_match(ah, torch, p1, p2)
    ref = okz.match(p1.copy(), p2)
    for f in ("match", "distance", "match_x", "match_y"):
        assert np.array_equal(got[f], ref[f]), f


def test_knn2_batch_on_detected_pairs(ah, okz, torch, synth, match_kernel):
    w, h = 640, 480
    p = ah.iAlignUp(w, 128)
    pairs = [synth.pair(w, h, 5), synth.pair(w, h, 6)]
    host = np.stack([synth.to_float(pairs[i // 2][i % 2], p) for i in range(4)])
    d = torch.from_numpy(host).cuda()
    mp = 3000
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp, batch=4)
    pts = torch.zeros(4 * mp * 104, dtype=torch.uint8, device="cuda")
    num = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = torch.zeros(2 * mp * 32, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, d.data_ptr(), h * p, p, 4, pts.data_ptr(), num.data_ptr(), 1))
    ah.check(ah.lib.hak_match_knn2_batch(det.ctx, pts.data_ptr(), num.data_ptr(), 2, 4, 5, 1, 0, out.data_ptr(), cnt.data_ptr()))
    ah.check(ah.lib.hak_sync(det.ctx))
    nums, cnts = num.cpu().numpy(), cnt.cpu().numpy()
    allp = pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(4, mp)
    allo = out.cpu().numpy().view(ah.MATCH_PAIR_DTYPE).reshape(2, mp)
    for k in range(2):
        a, b = allp[2 * k, :nums[2 * k]].copy(), allp[2 * k + 1, :nums[2 * k + 1]].copy()
        got_fields = {f: a[f].copy() for f in ("match", "distance", "match_x", "match_y")}
        want = okz.match_knn2(a, b, (4, 5), True)
        assert cnts[k] == len(want) and len(want) > 50
        for f in ah.MATCH_PAIR_DTYPE.names:
            assert np.array_equal(allo[k, :cnts[k]][f], want[f]), f
        for f, v in got_fields.items():
            assert np.array_equal(v, a[f]), f
    det.close()


@pytest.mark.parametrize("level_tile", ["1", "2"], ids=["streaming kernels", "k_level_tile"])
def test_fast_planes_small(ah, okz, torch, monkeypatch, level_tile):
    """every persistent int32 plane of every level of the FAST path, bit for bit (localises a failing stage)"""
    monkeypatch.setenv("HAK_LEVEL_TILE", level_tile)
    w, h = 320, 240
    u8 = _mg().case_scene(w, h, 11)
    p = ah.iAlignUp(w, 128)
    pad = np.zeros((h, p), np.uint8)
    pad[:, :w] = u8
    img = torch.from_numpy(pad).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=5000)
    data = ah.AkazeData()
    ah.initAkazeData(data, 5000, True, True)
    det.fastDetectAndCompute(img.data_ptr(), data, (w, h, p), True)
    r = okz.fast_detect_and_compute(u8, max_pts=5000, keep_arena=True)
    assert len(det.geometry()) == r.noct
    for o in range(r.noct):
        for s in range(4):
            for kind, nm in ((0, "Lt"), (2, "Lx"), (3, "Ly"), (1, "det")):
                a = det.plane(kind, o, s).view(np.int32)
                b = okz.plane(r, kind, o, s)
                bad = (a != b).sum()
                assert bad == 0, f"FAST {nm}({o},{s}): {bad} px differ"
    assert_points_equal(data.h_data[:data.num_pts], r.points)
    ah.freeAkazeData(data)
    det.close()
