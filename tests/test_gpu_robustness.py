"""GPU robustness of the C ABI: error returns instead of faults, degenerate inputs, graph replay with more
argument sets than cache slots, independent contexts interleaved on separate streams."""
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


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def _detector(ah, w, h, **kw):
    det = ah.Akazer()
    det.init((w, h, ah.iAlignUp(w, 128)), **kw)
    det._make_ctx(w, h)
    return det


def test_bad_arguments_return_errors(ah, torch):
    w, h = 256, 192
    p = ah.iAlignUp(w, 128)
    det = _detector(ah, w, h, max_pts=500, batch=2)
    img = torch.zeros((2, h, p), dtype=torch.float32, device="cuda")
    pts = torch.zeros(2 * 500 * 104, dtype=torch.uint8, device="cuda")
    num = torch.zeros(2, dtype=torch.int32, device="cuda")
    n = C.c_int(0)
    lib = ah.lib
    bad = [
        lambda: lib.hak_detect_and_compute_batch(det.ctx, None, h * p, p, 2, pts.data_ptr(), num.data_ptr(), 1),
        lambda: lib.hak_detect_and_compute_batch(det.ctx, img.data_ptr(), h * p, p, 3, pts.data_ptr(), num.data_ptr(), 1),
        lambda: lib.hak_detect_and_compute_batch(det.ctx, img.data_ptr(), h * p, w - 1, 2, pts.data_ptr(), num.data_ptr(), 1),
        lambda: lib.hak_detect_and_compute_batch(None, img.data_ptr(), h * p, p, 2, pts.data_ptr(), num.data_ptr(), 1),
        lambda: lib.hak_detect_and_compute(det.ctx, img.data_ptr(), p, pts.data_ptr(), 0, C.byref(n), None, 1),
        lambda: lib.hak_fast_detect_and_compute_batch(det.ctx, img.data_ptr(), h * p, p, 0, pts.data_ptr(), num.data_ptr(), 1),
        lambda: lib.hak_fast_detect_and_compute(det.ctx, None, p, pts.data_ptr(), 500, C.byref(n), None, 1),
        lambda: lib.hak_match_batch(det.ctx, pts.data_ptr(), None, 1),
        lambda: lib.hak_match_knn2(None, pts.data_ptr(), 10, pts.data_ptr(), 10, 0, 1, 1, 0, None, None, C.byref(n), None),
        lambda: lib.hak_match_knn2_batch(det.ctx, pts.data_ptr(), num.data_ptr(), 2, 4, 5, 1, 0, None, num.data_ptr()),
    ]
    for i, f in enumerate(bad):
        assert f() != 0, f"case {i} was accepted"
        assert lib.hak_last_error(), f"case {i} left no message"
    # the context is still usable afterwards
    ah.check(lib.hak_detect_and_compute_batch(det.ctx, img.data_ptr(), h * p, p, 2, pts.data_ptr(), num.data_ptr(), 1))
    ah.check(lib.hak_sync(det.ctx))
    assert (num.cpu().numpy() == 0).all()
    cfg = ah.hak_config()
    lib.hak_default_config(C.byref(cfg))
    ctx = C.c_void_p()
    for (cw, ch, field, val) in ((0, 100, None, None), (100, -1, None, None), (79, 200, None, None), (256, 192, "noctaves", 0),
                                 (256, 192, "max_scale", 0), (256, 192, "noctaves", 99), (65536, 96, None, None), (96, 70000, None, None)):
        c2 = ah.hak_config.from_buffer_copy(cfg)
        if field:
            setattr(c2, field, val)
        assert lib.hak_create(C.byref(c2), cw, ch, C.byref(ctx)) != 0, (cw, ch, field)
    det.close()


@pytest.mark.parametrize("w,h", [(96, 96), (80, 80), (128, 88), (161, 163)])
def test_degenerate_images_both_paths(ah, okz, torch, synth, w, h):
    """flat / tiny images: no keypoints, no faults; a small textured one still matches the oracle"""
    p = ah.iAlignUp(w, 128)
    det = ah.Akazer()
    det.init((w, h, p), max_pts=300)
    data = ah.AkazeData()
    ah.initAkazeData(data, 300, True, True)
    flat = torch.full((h, p), 0.5, dtype=torch.float32, device="cuda")
    det.detectAndCompute(flat.data_ptr(), data, (w, h, p), True)
    assert data.num_pts == 0
    flat8 = torch.full((h, p), 200, dtype=torch.uint8, device="cuda")
    det.fastDetectAndCompute(flat8.data_ptr(), data, (w, h, p), True)
    assert data.num_pts == 0
    u8 = np.ascontiguousarray(_mg().case_scene(256, 256, 5)[:h, :w])        # the generator needs >= 134 px: crop
    img = torch.from_numpy(synth.to_float(u8, p)).cuda()
    det.detectAndCompute(img.data_ptr(), data, (w, h, p), True)
    r = okz.detect_and_compute(synth.to_float(u8, p), w, max_pts=300)
    assert data.num_pts == len(r.points)
    if data.num_pts:
        assert_points_equal(data.h_data[:data.num_pts], r.points)
    pad = np.zeros((h, p), np.uint8)
    pad[:, :w] = u8
    det.fastDetectAndCompute(torch.from_numpy(pad).cuda().data_ptr(), data, (w, h, p), True)
    rf = okz.fast_detect_and_compute(u8, max_pts=300)
    assert data.num_pts == len(rf.points)
    if data.num_pts:
        assert_points_equal(data.h_data[:data.num_pts], rf.points)
    ah.freeAkazeData(data)
    det.close()


def test_graph_replay_with_more_argument_sets_than_cache_slots(ah, torch, synth):
    """six input buffers cycled three times through one context (the graph cache holds four): every call must equal
    the first result for that buffer, and equal an eager (HAK_GRAPH=0) context's result"""
    w, h, mp = 480, 360, 1500
    p = ah.iAlignUp(w, 128)
    imgs = [torch.from_numpy(synth.to_float(_mg().case_scene(w, h, 100 + i), p)).cuda() for i in range(6)]

    def run(det):
        data = ah.AkazeData()
        ah.initAkazeData(data, mp, True, True)
        outs = [[] for _ in imgs]
        for rep in range(3):
            for i, im in enumerate(imgs):
                det.detectAndCompute(im.data_ptr(), data, (w, h, p), True)
                outs[i].append(data.h_data[:data.num_pts].copy().tobytes())
        ah.freeAkazeData(data)
        return outs

    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp)
    graph = run(det)
    det.close()
    os.environ["HAK_GRAPH"] = "0"
    try:
        det2 = ah.Akazer()
        det2.init((w, h, p), max_pts=mp)
        eager = run(det2)
        det2.close()
    finally:
        del os.environ["HAK_GRAPH"]
    for i in range(len(imgs)):
        assert len(graph[i][0]) > 104 * 20
        assert graph[i][0] == graph[i][1] == graph[i][2] == eager[i][0] == eager[i][2]
    assert len({g[0] for g in graph}) == len(imgs)


def test_two_contexts_interleaved_on_their_own_streams(ah, torch, synth):
    """the bench's pipelining pattern with a parity check: two contexts, batches in flight on both, results equal a
    context used alone"""
    w, h, mp, B = 640, 360, 2000, 4
    p = ah.iAlignUp(w, 128)
    batches = [np.stack([synth.to_float(_mg().case_scene(w, h, 200 + 10 * k + i), p) for i in range(B)]) for k in range(2)]
    d_in = [torch.from_numpy(b).cuda() for b in batches]

    def alloc():
        return (torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"))

    ref = []
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp, batch=B)
    for k in range(2):
        pts, num = alloc()
        ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, d_in[k].data_ptr(), h * p, p, B, pts.data_ptr(), num.data_ptr(), 1))
        ah.check(ah.lib.hak_match_batch(det.ctx, pts.data_ptr(), num.data_ptr(), B // 2))
        ah.check(ah.lib.hak_sync(det.ctx))
        ref.append((pts.cpu().numpy().copy(), num.cpu().numpy().copy()))
    det.close()
    dets = []
    outs = []
    for k in range(2):
        d = ah.Akazer()
        d.init((w, h, p), max_pts=mp, batch=B)
        dets.append(d)
        outs.append(alloc())
    for rep in range(3):
        for k in range(2):                                  # both enqueued before either is synchronised
            ah.check(ah.lib.hak_detect_and_compute_batch(dets[k].ctx, d_in[k].data_ptr(), h * p, p, B, outs[k][0].data_ptr(),
                                                         outs[k][1].data_ptr(), 1))
            ah.check(ah.lib.hak_match_batch(dets[k].ctx, outs[k][0].data_ptr(), outs[k][1].data_ptr(), B // 2))
        for k in range(2):
            ah.check(ah.lib.hak_sync(dets[k].ctx))
            num = outs[k][1].cpu().numpy()
            assert np.array_equal(num, ref[k][1]) and num.min() > 20
            got = outs[k][0].cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
            want = ref[k][0].view(ah.POINT_DTYPE).reshape(B, mp)
            for i in range(B):
                assert got[i, :num[i]].tobytes() == want[i, :num[i]].tobytes()
    for d in dets:
        d.close()


def test_knn2_with_context_scratch_and_repeat(ah, okz, torch, synth):
    p2 = synth.random_descriptors(900, 3, ah.POINT_DTYPE)
    p1 = synth.random_descriptors(700, 4, ah.POINT_DTYPE, planted_from=p2, nplanted=300, maxflip=60)
    want_pts = p1.copy()
    want = okz.match_knn2(want_pts, p2, (4, 5), True)
    det = _detector(ah, 256, 192, max_pts=1000, batch=2)
    d1 = torch.from_numpy(p1.view(np.uint8).reshape(-1).copy()).cuda()
    d2 = torch.from_numpy(p2.view(np.uint8).reshape(-1).copy()).cuda()
    d_out = torch.zeros(700 * 32, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        h_out = np.zeros(700, ah.MATCH_PAIR_DTYPE)
        cnt = C.c_int(-1)
        ah.check(ah.lib.hak_match_knn2(det.ctx, d1.data_ptr(), 700, d2.data_ptr(), 900, 4, 5, 1, 0, None, d_out.data_ptr(),
                                       C.byref(cnt), h_out.ctypes.data))
        assert cnt.value == len(want) > 100
        for f in ah.MATCH_PAIR_DTYPE.names:
            assert np.array_equal(h_out[:cnt.value][f], want[f]), f
    det.close()


_ALT_ORACLE = {}


def _alt_oracle(okz, synth, ah, u8, w, h, mp):
    """oracle bytes (float path records + FAST path records) of the scene the alternatives test runs, computed once"""
    if "v" not in _ALT_ORACLE:
        a = okz.detect_and_compute(synth.to_float(u8, ah.iAlignUp(w, 128)), w, max_pts=mp).points
        b = okz.fast_detect_and_compute(u8, max_pts=mp).points
        _ALT_ORACLE["v"] = (a, b)
    return _ALT_ORACLE["v"]


@pytest.mark.parametrize("env", [{"HAK_FUSE_SF": "0"}, {"HAK_HESS_STREAM": "0"}, {"HAK_FED_MAX_FUSE": "1"}, {"HAK_GRAPH": "0", "HAK_SERIAL": "1"},
                                 {"HAK_FUSE_SF": "1", "HAK_HESS_STREAM": "1"},          # the default size rule
                                 {"HAK_FUSE_HEAD": "0"}, {"HAK_BASE_STREAM": "0"}, {"HAK_BASE_STREAM": "1"},
                                 # the streaming prologue with the lattice maximum found first and the histogram inside the pass, instead of
                                 # its two-pass form (gradient plane + histogram pass)
                                 # (off by default: measured slower)
                                 {"HAK_BASE_HIST": "1"}, {"HAK_BASE_STREAM": "2", "HAK_BASE_HIST": "1"}, {"HAK_BASE_STREAM": "2", "HAK_BASE_HIST": "0"},
                                 {"HAK_FUSE_SF": "0", "HAK_HESS_STREAM": "0", "HAK_BASE_STREAM": "0", "HAK_FED_MAX_FUSE": "2"},
                                 # tile Hessian with a 4-entry candidate staging buffer: nearly every row with a candidate takes the
                                 # overflow path of the reservation (direct global slots) next to staged ones
                                 {"HAK_HESS_STREAM": "0", "HAK_HESS_CBUF": "4"}, {"HAK_HESS_STREAM": "0", "HAK_HESS_CBUF": "1"},
                                 # MLDB: the generic kernel instead of the planned one; block orders of the keypoint kernels
                                 {"HAK_DESC_PLAN": "0"}, {"HAK_DESC_ORDER": "0"}, {"HAK_DESC_ORDER": "3", "HAK_DESC_PLAN": "0"},
                                 # ... visiting the keypoints level by level (k_desc_perm; the default only in batches of 8 and more)
                                 {"HAK_DESC_SORT": "2"}, {"HAK_DESC_SORT": "2", "HAK_DESC_ORDER": "0"}, {"HAK_DESC_SORT": "0"},
                                 # one launch per sublevel out of LDS tiles (kernels_level.hip): what a single-image call uses by default
                                 {"HAK_LEVEL_TILE": "2"}, {"HAK_LEVEL_TILE": "2", "HAK_HESS_STREAM": "0", "HAK_BASE_STREAM": "0"},
                                 {"HAK_LEVEL_TILE": "0", "HAK_FUSE_SF": "1"},
                                 # ... with the level's Hessian as a launch of its own instead of inside k_level_tile
                                 {"HAK_LEVEL_TILE": "2", "HAK_LEVEL_HESS": "0"}, {"HAK_LEVEL_HESS": "0", "HAK_FUSE_SF": "1"},
                                 # the streaming Hessian low-passes Lt(o,s-1) itself (LP variant) and k_fed_sf does not store `smooth`
                                 {"HAK_HESS_LP": "1"}, {"HAK_HESS_LP": "1", "HAK_FED_MAX_FUSE": "2"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_kernel_alternatives_are_bit_identical(ah, okz, torch, synth, env):
    """every kernel-selection knob read by hak_create must give byte-identical keypoints, descriptors and
    persistent planes: the fused / streaming kernels and the tile kernels they replace are interchangeable -- and equal
    to the oracle (both paths), so this is not a self-comparison"""
    w, h, mp = 960, 540, 4000
    p = ah.iAlignUp(w, 128)
    u8 = synth.scene(w, h, 21)
    img = torch.from_numpy(synth.to_float(u8, p)).cuda()
    pad = np.zeros((h, p), np.uint8)
    pad[:, :w] = u8
    img8 = torch.from_numpy(pad).cuda()

    def run():
        det = ah.Akazer()
        det.init((w, h, p), max_pts=mp)
        data = ah.AkazeData()
        ah.initAkazeData(data, mp, True, True)
        det.detectAndCompute(img.data_ptr(), data, (w, h, p), True)
        both = [data.h_data[:data.num_pts].copy()]
        pts = both[0].tobytes()
        planes = [det.plane(kind, o, s).tobytes() for o in range(len(det.geometry())) for s in range(4) for kind in (0, 1, 2, 3)]
        det.fastDetectAndCompute(img8.data_ptr(), data, (w, h, p), True)          # the integer path shares the knobs
        assert data.num_pts > 100
        both.append(data.h_data[:data.num_pts].copy())
        pts += both[1].tobytes()
        planes += [det.plane(kind, o, s).tobytes() for o in range(len(det.geometry())) for s in range(4) for kind in (0, 1, 2, 3)]
        ah.freeAkazeData(data)
        det.close()
        return pts, planes, both

    ref_pts, ref_planes, _ = run()
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        alt_pts, alt_planes, alt_both = run()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert len(ref_pts) > 104 * 100
    assert alt_pts == ref_pts
    assert alt_planes == ref_planes
    want = _alt_oracle(okz, synth, ah, u8, w, h, mp)
    assert_points_equal(alt_both[0], want[0])
    assert_points_equal(alt_both[1], want[1])


def test_two_contexts_keep_their_own_knobs(ah, torch, synth, monkeypatch):
    """the kernel-selection knobs are read by hak_create INTO the context (round 2 kept them in process globals, so the second
    context silently changed the first): two contexts built under different environments, used alternately, agree byte for byte"""
    w, h, mp = 640, 480, 3000
    p = ah.iAlignUp(w, 128)
    img = torch.from_numpy(synth.to_float(synth.scene(w, h, 77), p)).cuda()
    dets = []
    for env in ({"HAK_HESS_STREAM": "0", "HAK_FUSE_SF": "0", "HAK_BASE_STREAM": "0", "HAK_DESC_PLAN": "0", "HAK_LEVEL_TILE": "0"},
                {"HAK_HESS_STREAM": "2", "HAK_FUSE_SF": "2", "HAK_BASE_STREAM": "2", "HAK_DESC_PLAN": "1", "HAK_LEVEL_TILE": "0"},
                {"HAK_LEVEL_TILE": "2", "HAK_HESS_CBUF": "2"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d = ah.Akazer()
        d.init((w, h, p), max_pts=mp)
        dets.append(d)
    data = ah.AkazeData()
    ah.initAkazeData(data, mp, True, True)
    out = []
    for rep in range(2):
        for d in dets:                                              # alternately: a global would now hold the LAST context's modes
            d.detectAndCompute(img.data_ptr(), data, (w, h, p), True)
            out.append(data.h_data[:data.num_pts].tobytes())
    assert len(out[0]) > 104 * 100
    assert all(o == out[0] for o in out)
    ah.freeAkazeData(data)
    for d in dets:
        d.close()


def test_wait_event_orders_context_behind_copy_stream(ah, torch, synth):
    """hak_wait_event: an upload on a copy stream of the caller's, its event handed to the context -- the detection that follows
    sees the uploaded image (no host synchronisation in between), and a null event is refused"""
    w, h, mp = 640, 480, 3000
    p = ah.iAlignUp(w, 128)
    host_a = torch.from_numpy(synth.to_float(synth.scene(w, h, 5), p)).pin_memory()
    host_b = torch.from_numpy(synth.to_float(synth.scene(w, h, 6), p)).pin_memory()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp)
    data = ah.AkazeData()
    ah.initAkazeData(data, mp, True, True)
    want = []
    for hst in (host_a, host_b):
        det.detectAndCompute(hst.cuda().data_ptr(), data, (w, h, p), True)
        want.append(data.h_data[:data.num_pts].tobytes())
    assert want[0] != want[1]
    d_img = host_a.cuda()
    big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    big_h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    cs = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(cs):
        big.copy_(big_h, non_blocking=True)                         # something for the image copy to queue behind
        d_img.copy_(host_b, non_blocking=True)
        ev = cs.record_event()
    ah.check(ah.lib.hak_wait_event(det.ctx, C.c_void_p(ev.cuda_event)))
    det.detectAndCompute(d_img.data_ptr(), data, (w, h, p), True)
    assert data.h_data[:data.num_pts].tobytes() == want[1]
    assert ah.lib.hak_wait_event(det.ctx, None) != 0
    ah.freeAkazeData(data)
    det.close()


def test_phase_event_interlocks_two_contexts(ah, torch, synth):
    """hak_phase_event: each of two batch contexts starts its sequence behind the OTHER one's phase event (what bench.py's
    pipeline does): replayed graphs record the event as a node of their own, results equal the un-interlocked run's"""
    w, h, mp, B = 480, 360, 1500, 4
    p = ah.iAlignUp(w, 128)
    host = np.stack([synth.to_float(_mg().case_scene(w, h, 700 + i), p) for i in range(B)])
    d_in = torch.from_numpy(host).cuda()
    dets, evs, pts, num = [], [], [], []
    for k in range(2):
        d = ah.Akazer()
        d.init((w, h, p), max_pts=mp, batch=B)
        ev = C.c_void_p()
        ah.check(ah.lib.hak_phase_event(d.ctx, C.byref(ev)))
        assert ev.value
        dets.append(d)
        evs.append(ev)
        pts.append(torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda"))
        num.append(torch.zeros(B, dtype=torch.int32, device="cuda"))
    outs = []
    for lock in (False, True, True):
        for i in range(6):                                          # (the second and later rounds replay captured graphs)
            k = i % 2
            if lock:
                ah.check(ah.lib.hak_wait_event(dets[k].ctx, evs[1 - k]))
            ah.check(ah.lib.hak_detect_and_compute_batch(dets[k].ctx, d_in.data_ptr(), h * p, p, B, pts[k].data_ptr(), num[k].data_ptr(), 1))
        for k in range(2):
            ah.check(ah.lib.hak_sync(dets[k].ctx))
            outs.append((num[k].cpu().numpy().copy(), pts[k].cpu().numpy().copy()))
    assert outs[0][0].min() > 20
    for n, q in outs[1:]:
        assert np.array_equal(n, outs[0][0]) and np.array_equal(q, outs[0][1])
    for d in dets:
        d.close()


def test_download_batch_pinned_and_pageable(ah, torch, synth):
    """hak_download_batch: pinned destinations take the one-kernel zero-copy path, pageable ones the per-image copies;
    both must deliver every image's count and the valid prefix of its records"""
    w, h, mp, B = 480, 360, 1200, 5
    p = ah.iAlignUp(w, 128)
    host = np.stack([synth.to_float(_mg().case_scene(w, h, 300 + i), p) for i in range(B)])
    host[3] = 0.25                                          # an image without keypoints
    d_in = torch.from_numpy(host).cuda()
    det = ah.Akazer()
    det.init((w, h, p), max_pts=mp, batch=B)
    pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
    num = torch.zeros(B, dtype=torch.int32, device="cuda")
    ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, d_in.data_ptr(), h * p, p, B, pts.data_ptr(), num.data_ptr(), 1))
    ah.check(ah.lib.hak_sync(det.ctx))
    want_n = num.cpu().numpy()
    want = pts.cpu().numpy().view(ah.POINT_DTYPE).reshape(B, mp)
    assert want_n[3] == 0 and want_n.max() > 50
    # pageable numpy buffers
    hp = np.zeros((B, mp), ah.POINT_DTYPE)
    hn = np.full(B, -1, np.int32)
    ah.check(ah.lib.hak_download_batch(det.ctx, pts.data_ptr(), num.data_ptr(), B, hp.ctypes.data, hn.ctypes.data))
    # pinned buffers from the library's allocator
    pp, pn = C.c_void_p(), C.c_void_p()
    ah.check(ah.lib.hak_host_alloc(C.byref(pp), B * mp * 104))
    ah.check(ah.lib.hak_host_alloc(C.byref(pn), B * 4))
    C.memset(pp, 0, B * mp * 104)
    ah.check(ah.lib.hak_download_batch(det.ctx, pts.data_ptr(), num.data_ptr(), B, pp, pn))
    gn = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_int)), shape=(B,)).copy()
    gp = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint8)), shape=(B * mp * 104,)).view(ah.POINT_DTYPE).reshape(B, mp).copy()
    assert np.array_equal(hn, want_n) and np.array_equal(gn, want_n)
    for i in range(B):
        n = want_n[i]
        assert hp[i, :n].tobytes() == want[i, :n].tobytes()
        assert gp[i, :n].tobytes() == want[i, :n].tobytes()
        assert not gp[i, n:].view(np.uint8).any()           # nothing beyond the valid prefix is touched
    ah.lib.hak_host_free(pp)
    ah.lib.hak_host_free(pn)
    det.close()


# widths chosen around the streaming kernels' strip geometry (a wave stores 240 / 232 / 248 columns for dilation <= 3 / 4 / 1,
# with 8 / 12 / 4 margin columns): exact multiples, one float4 over, last strip narrower than the margin, single strip; heights
# around the 8..128-row segments and the 2S+1 / NS+4 warm-up rows
SWEEP = [(240, 131), (244, 96), (248, 203), (252, 117), (232, 88), (236, 129), (480, 135), (484, 97), (472, 160), (720, 81),
         (964, 92), (1204, 83), (196, 259), (300, 300), (1000, 130),
         # odd extents with (n - 1) % 32 == 0: gFindMaxContrastU4's grid leaves the last lattice column / row to no block (hak_lattice_cov),
         # and the histogram's 32 x 16 blocks hang 31 columns / 15 rows over the image (hak_hist_extra0); the tile kernels' path
         (129, 97), (161, 225), (193, 100)]


@pytest.mark.parametrize("w,h", SWEEP, ids=lambda v: str(v))
def test_strip_geometry_sweep_both_paths(ah, okz, torch, synth, w, h):
    """float and FAST pipelines vs their oracles on shapes that stress strip / segment boundaries of the streaming kernels"""
    seed = 7 * w + h
    big = _mg().case_scene(1280, 320, seed % 97)
    u8 = np.ascontiguousarray(big[:h, :w]) if h <= 320 else _mg().case_scene(max(w, 134), h, seed % 97)[:, :w].copy()
    p = ah.iAlignUp(w, 128)
    noct = 2 + seed % 3
    kw = dict(noctaves=noct, max_scale=3 + seed % 2, derivative_factor=[1.0, 1.5, 2.0][seed % 3])
    det = ah.Akazer()
    det.init((w, h, p), max_pts=3000, **kw)
    data = ah.AkazeData()
    ah.initAkazeData(data, 3000, True, True)
    img = torch.from_numpy(synth.to_float(u8, p)).cuda()
    det.detectAndCompute(img.data_ptr(), data, (w, h, p), True)
    r = okz.detect_and_compute(synth.to_float(u8, p), w, okz.default_params(**kw), max_pts=3000, keep_arena=True)

    def check_planes(ref, as_int):
        # keypoints stay clear of the image border, so the planes themselves are compared: every pixel of every level
        assert len(det.geometry()) == ref.noct
        for o in range(ref.noct):
            for s in range(kw["max_scale"]):
                for kind, nm in ((0, "Lt"), (2, "Lx"), (3, "Ly"), (1, "det")):
                    got = det.plane(kind, o, s)
                    want = okz.plane(ref, kind, o, s)
                    got = got.view(np.int32) if as_int else got.view(np.uint32)
                    want = want if as_int else want.view(np.uint32)
                    bad = np.argwhere(got != want)
                    assert len(bad) == 0, f"{'FAST ' if as_int else ''}{nm}({o},{s}): {len(bad)} px differ, first at (y, x) = {tuple(bad[0])}"

    check_planes(r, False)
    assert data.num_pts == len(r.points), (data.num_pts, len(r.points))
    if data.num_pts:
        assert_points_equal(data.h_data[:data.num_pts], r.points)
    pad = np.zeros((h, p), np.uint8)
    pad[:, :w] = u8
    det.fastDetectAndCompute(torch.from_numpy(pad).cuda().data_ptr(), data, (w, h, p), True)
    rf = okz.fast_detect_and_compute(u8, okz.default_params(**kw), max_pts=3000, keep_arena=True)
    check_planes(rf, True)
    assert data.num_pts == len(rf.points), (data.num_pts, len(rf.points))
    if data.num_pts:
        assert_points_equal(data.h_data[:data.num_pts], rf.points)
    ah.freeAkazeData(data)
    det.close()


def test_constant_image_vs_oracle_both_paths(ah, okz, torch, synth):
    """a constant image: hmax = 0, hfactor = inf, every histogram index 0 * inf = NaN -> bin 0 by the device cast (akazed.cu:924): no
    keypoints and the same contrast factor as the oracle, float and FAST; then a real image through the same context"""
    w, h = 320, 240
    p = ah.iAlignUp(w, 128)
    det = ah.Akazer()
    det.init((w, h, p), max_pts=2000)
    data = ah.AkazeData()
    ah.initAkazeData(data, 2000, True, True)
    flat = torch.full((h, p), 0.25, dtype=torch.float32, device="cuda")
    det.detectAndCompute(flat.data_ptr(), data, (w, h, p), True)
    r = okz.detect_and_compute(np.full((h, p), 0.25, np.float32), w)
    assert data.num_pts == 0 == len(r.points)
    kc = C.c_float()
    ah.check(ah.lib.hak_debug_kcontrast(det.ctx, 0, C.byref(kc)))
    assert np.float32(kc.value).tobytes() == np.float32(r.kcontrast).tobytes()
    flat8 = torch.full((h, p), 64, dtype=torch.uint8, device="cuda")
    det.fastDetectAndCompute(flat8.data_ptr(), data, (w, h, p), True)
    assert data.num_pts == 0 == len(okz.fast_detect_and_compute(np.full((h, w), 64, np.uint8)).points)
    u8 = _mg().case_scene(w, h, 5)
    det.detectAndCompute(torch.from_numpy(synth.to_float(u8, p)).cuda().data_ptr(), data, (w, h, p), True)
    want = okz.detect_and_compute(synth.to_float(u8, p), w, max_pts=2000).points
    assert data.num_pts == len(want) > 20
    assert_points_equal(data.h_data[:data.num_pts], want)
    ah.freeAkazeData(data)
    det.close()


def test_four_host_threads_four_contexts(ah, torch, synth):
    """contexts are independent (the reference's detector is a process-wide singleton: global device symbols, a function-static
    Gaussian cache): four host threads drive four contexts of different extents at once -- batch detect (graph capture
    and replay per thread), batched match, the pool-backed hak_match without a context -- and every round must equal what the same
    context delivered alone"""
    import threading
    shapes = [(400, 300, 4), (512, 384, 3), (333, 251, 2), (640, 360, 4)]
    mp = 2500
    jobs = []
    for t, (w, h, B) in enumerate(shapes):
        p = ah.iAlignUp(w, 128)
        stack = torch.from_numpy(np.stack([synth.to_float(_mg().case_scene(w, h, 700 + 10 * t + i), p) for i in range(B)])).cuda()
        det = ah.Akazer()
        det.init((w, h, p), max_pts=mp, batch=B)
        pts = torch.zeros(B * mp * 104, dtype=torch.uint8, device="cuda")
        num = torch.zeros(B, dtype=torch.int32, device="cuda")
        jobs.append(dict(w=w, h=h, p=p, B=B, stack=stack, det=det, pts=pts, num=num))
    torch.cuda.synchronize()

    def one_round(j):
        ah.check(ah.lib.hak_detect_and_compute_batch(j["det"].ctx, j["stack"].data_ptr(), j["h"] * j["p"], j["p"], j["B"], j["pts"].data_ptr(), j["num"].data_ptr(), 1))
        ah.check(ah.lib.hak_match_batch(j["det"].ctx, j["pts"].data_ptr(), j["num"].data_ptr(), j["B"] // 2))
        ah.check(ah.lib.hak_sync(j["det"].ctx))
        n = j["num"].cpu().numpy().copy()
        # cuMatch without a context (scratch from the per-device pool): image 0 against image 1, on the caller's copies
        a = j["pts"][:mp * 104].clone()
        ah.check(ah.lib.hak_match(None, a.data_ptr(), int(n[0]), j["pts"][mp * 104:].data_ptr(), int(n[1]), None))
        return n, j["pts"].cpu().numpy().copy(), a.cpu().numpy().copy()

    alone = [one_round(j) for j in jobs]
    for n, _, _ in alone:
        assert n.min() > 30
    errors, results = [], [[] for _ in jobs]

    def worker(k):
        try:
            for _ in range(12):
                results[k].append(one_round(jobs[k]))
        except Exception as e:                            # noqa: BLE001 -- reported below, in the test's thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, rounds in enumerate(results):
        assert len(rounds) == 12
        for n, pts, a in rounds:
            assert np.array_equal(n, alone[k][0]) and np.array_equal(pts, alone[k][1]) and np.array_equal(a, alone[k][2]), k
    # the batched match of pair 0 and the context-free hak_match of the same pair agree (both are cuMatch, akaze.cpp:55-64)
    for k, j in enumerate(jobs):
        n0 = int(alone[k][0][0])
        b = alone[k][1].view(ah.POINT_DTYPE).reshape(j["B"], mp)[0, :n0]
        c = alone[k][2].view(ah.POINT_DTYPE)[:n0]
        for f in ("match", "distance", "match_x", "match_y"):
            assert np.array_equal(b[f], c[f]), (k, f)
        j["det"].close()


def test_calls_are_ordered_behind_the_callers_null_stream_work(ah, okz, torch, synth):
    """the reference runs on the default stream, so what its caller enqueued there before a call (an asynchronous copy into the image, a
    hipMemset of an output array) is done when the call starts (akaze.cpp:101-150); a context's streams are non-blocking, and every
    entry point makes its stream wait for the NULL stream's work first (hak_set_null_order, default on).  Here: a 1 GiB fill, then a
    device-to-device copy of ANOTHER image into the input buffer, both only enqueued on the NULL stream, then the call"""
    w, h = 640, 480
    p = ah.iAlignUp(w, 128)
    a, b = (synth.to_float(_mg().case_scene(w, h, 900 + k), p) for k in range(2))
    want_b = okz.detect_and_compute(b, w, max_pts=4000).points
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    work = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    det = ah.Akazer()
    det.init((w, h, p), max_pts=4000, batch=2)
    data, data2 = ah.AkazeData(), ah.AkazeData()
    ah.initAkazeData(data, 4000, True, True)
    ah.initAkazeData(data2, 4000, True, True)
    for entry in ("single", "pair", "batch"):
        buf = da.clone()
        det.detectAndCompute(buf.data_ptr(), data, (w, h, p), True)        # (the context has seen image a in this buffer)
        torch.cuda.synchronize()
        work.fill_(1.0)                                                    # ~0.3 ms of NULL-stream work in front of the copy
        buf.copy_(db)
        if entry == "single":
            det.detectAndCompute(buf.data_ptr(), data, (w, h, p), True)
            got = data.h_data[:data.num_pts]
        elif entry == "pair":
            det.detectAndComputePair(buf.data_ptr(), da.data_ptr(), data, data2, (w, h, p), True, True)
            got = data.h_data[:data.num_pts]
        else:
            d_pts = torch.empty(4000 * 104, dtype=torch.uint8, device="cuda")
            d_num = torch.empty(1, dtype=torch.int32, device="cuda")
            d_pts.zero_()                                                  # an output cleared on the NULL stream, not waited for either
            ah.check(ah.lib.hak_detect_and_compute_batch(det.ctx, buf.data_ptr(), h * p, p, 1, d_pts.data_ptr(), d_num.data_ptr(), 1))
            ah.check(ah.lib.hak_sync(det.ctx))
            n = int(d_num.cpu()[0])
            got = d_pts.cpu().numpy().view(ah.POINT_DTYPE)[:n]
        assert len(got) == len(want_b) > 100, entry
        assert_points_equal(got, want_b)
    ah.freeAkazeData(data)
    ah.freeAkazeData(data2)
    det.close()
