"""okz -- ctypes loader for the CPU parity oracle (oracle/akaze_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libfedref.so")

POINT_DTYPE = np.dtype([
    ("x", "<f4"), ("y", "<f4"), ("octave", "<i4"), ("response", "<f4"), ("size", "<f4"), ("angle", "<f4"),
    ("features", "u1", (61,)), ("_pad", "u1", (3,)),
    ("match", "<i4"), ("distance", "<i4"), ("match_x", "<f4"), ("match_y", "<f4"),
])


class Params(C.Structure):
    _fields_ = [("noctaves", C.c_int), ("max_scale", C.c_int), ("per", C.c_float), ("kcontrast", C.c_float),
                ("soffset", C.c_float), ("reordering", C.c_int), ("derivative_factor", C.c_float),
                ("dthreshold", C.c_float), ("diffusivity", C.c_int), ("descriptor_pattern_size", C.c_int),
                ("upright", C.c_int)]


def default_params(**kw):
    p = Params(4, 4, 0.7, 0.03, 1.6, 1, 1.5, 0.001, 1, 10, 0)      # main.cpp:156-166
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("akaze_oracle.c", "akaze_oracle_fast.c", "okz_math.h")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(REF_LIB)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def build_native(outdir):
    """bench.py's cpu_baseline leg only: the same sources compiled for the host the bench runs on
    (-O3 -march=native; still -ffp-contract=off -fno-fast-math, so results stay bit-identical -- bench.py checks that by
    verifying the GPU results against THIS build).  Returns the library path, or None when the compiler is unavailable."""
    out = os.path.join(outdir, "liboracle_native.so")
    cmd = ["gcc", "-O3", "-march=native", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-Wno-unknown-pragmas",
           "-shared", "-o", out, os.path.join(_HERE, "akaze_oracle.c"), os.path.join(_HERE, "akaze_oracle_fast.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return out


def load(path):
    """switch the module to another build of the oracle library (before or after the first call)"""
    global _lib, LIB
    LIB = path
    _lib = None
    return lib()


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        if LIB == os.path.join(_HERE, "liboracle.so"):
            build()
        _lib = C.CDLL(LIB)
        _lib.okz_arena_floats.restype = C.c_long
        _lib.okz_kcontrast.restype = C.c_float
        _lib.okz_fed_tau.argtypes = [C.c_float, C.c_int, C.c_float, C.c_int, _fp, C.c_int]
        _lib.okz_gauss_taps.argtypes = [C.c_float, C.c_int, _fp]
        _lib.okz_flow.argtypes = [_fp, _fp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib.okz_nld_step.argtypes = [_fp, _fp, _fp, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib.okz_kcontrast.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _ip]
    return _lib


def set_num_threads(n):
    """team size of the oracle's OpenMP loops (omp_set_num_threads: effective whatever OMP_NUM_THREADS said when libgomp started)"""
    lib().okz_set_num_threads(int(n))
    return lib().okz_get_max_threads()


def ref_lib():
    """the reference's own fed.cpp, compiled from /root/reference (None when unavailable)"""
    if not os.path.exists(REF_LIB):
        return None
    r = C.CDLL(REF_LIB)
    r.fedref_tau_by_process_time.argtypes = [C.c_float, C.c_int, C.c_float, C.c_int, _fp, C.c_int]
    return r


def _f(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_fp)


def fed_tau(T, M=1, tau_max=0.25, reordering=True):
    buf = np.zeros(4096, np.float32)
    n = lib().okz_fed_tau(T, M, tau_max, int(reordering), _f(buf), 4096)
    return buf[:n].copy()


def ref_fed_tau(T, M=1, tau_max=0.25, reordering=True):
    buf = np.zeros(4096, np.float32)
    n = ref_lib().fedref_tau_by_process_time(T, M, tau_max, int(reordering), _f(buf), 4096)
    return buf[:n].copy()


def gauss_taps(var, radius):
    k = np.zeros(8, np.float32)
    lib().okz_gauss_taps(var, radius, _f(k))
    return k[:radius + 1].copy()


def deriv_factors():
    a, b = C.c_float(), C.c_float()
    lib().okz_deriv_factors(C.byref(a), C.byref(b))
    return np.float32(a.value), np.float32(b.value)


def compare_indices():
    a = np.zeros(488, np.int32)
    b = np.zeros(488, np.int32)
    lib().okz_compare_indices(a.ctypes.data_as(_ip), b.ctypes.data_as(_ip))
    return a, b


def orient_weights():
    t = np.zeros(36, np.float32)
    lib().okz_orient_weights(_f(t))
    return t


# ---- single stages on pitched planes (2-D float32 arrays, shape (h, p), valid width w)
def lowpass(src, w, var, radius):
    h, p = src.shape
    dst = np.zeros_like(src)
    lib().okz_lowpass(_f(src), _f(dst), w, h, p, _f(gauss_taps(var, radius)), radius)
    return dst


def down_smooth(src, sw, dw, dh, dp):
    sh, sp = src.shape
    dst = np.zeros((dh, dp), np.float32)
    sm = np.zeros((dh, dp), np.float32)
    lib().okz_down_smooth(_f(src), _f(dst), _f(sm), sw, sh, sp, dw, dh, dp, _f(gauss_taps(1.0, 2)))
    return dst, sm


def scharr_grad(src, w):
    h, p = src.shape
    g = np.zeros_like(src)
    lib().okz_scharr_grad(_f(src), _f(g), w, h, p)
    return g


def kcontrast(grad, w, per):
    h, p = grad.shape
    hmax = C.c_float()
    hist = np.zeros(300, np.int32)
    kc = lib().okz_kcontrast(_f(grad), w, h, p, per, C.byref(hmax), hist.ctypes.data_as(_ip))
    return np.float32(kc), np.float32(hmax.value), hist


def flow(src, w, diffusivity, kc):
    h, p = src.shape
    d = np.zeros_like(src)
    lib().okz_flow(_f(src), _f(d), diffusivity, kc, w, h, p)
    return d


def nld_steps(src, g, w, taus):
    h, p = src.shape
    cur = src.copy()
    for t in taus:
        nxt = np.zeros_like(cur)
        lib().okz_nld_step(_f(cur), _f(g), _f(nxt), float(t), w, h, p)
        cur = nxt
    return cur


def hessian(src, w, step):
    h, p = src.shape
    lx, ly, det = np.zeros_like(src), np.zeros_like(src), np.zeros_like(src)
    lib().okz_derivate(_f(src), _f(lx), _f(ly), step, w, h, p)
    lib().okz_hessian(_f(lx), _f(ly), _f(det), step, w, h, p)
    return lx, ly, det


# ---- detector tail and descriptor stages on hand-made inputs (tests/test_reference_literal_cpu.py)
def _pitched(a, dtype=np.float32):
    a = np.ascontiguousarray(a, dtype)
    assert a.ndim == 2
    return a


def extrema_map(dets, w, params, octave, threshold, maps, opitch, fast=False):
    """okz_extrema_map / fkz_extrema (gCalcExtremaMap akazed.cu:1334 / 3476).  dets: (ms, h, p) planes of one octave;
    params = borders[ms] + sizes[ms]; maps = (response (H, P), size (H, P), layer (H, P)) updated in place."""
    ms, h, p = dets.shape
    resp, size, layer = maps
    prm = np.ascontiguousarray(params, np.float32)
    if fast:
        assert dets.dtype == np.int32 and resp.dtype == np.int32
        lib().fkz_extrema(dets.ctypes.data_as(C.c_void_p), resp.ctypes.data_as(C.c_void_p), _f(size), layer.ctypes.data_as(_ip), _f(prm),
                          C.c_int(octave), C.c_int(ms), C.c_int(int(threshold)), C.c_int(w), C.c_int(h), C.c_int(p), C.c_int(opitch))
    else:
        assert dets.dtype == np.float32 and resp.dtype == np.float32
        lib().okz_extrema_map(_f(dets.reshape(-1)), _f(resp.reshape(-1)), _f(size.reshape(-1)), layer.ctypes.data_as(_ip), _f(prm),
                              C.c_int(octave), C.c_int(ms), C.c_float(threshold), C.c_int(w), C.c_int(h), C.c_int(p), C.c_int(opitch))


def nms(resp, size, layer, w, psz, max_pts=10000, fast=False):
    """okz_nms / fkz_nms (gNmsRNaive akazed.cu:1554 / 3538) on full-resolution maps (H, P); returns (points, total)"""
    h, p = resp.shape
    pts = np.zeros(max_pts, POINT_DTYPE)
    fn = lib().fkz_nms if fast else lib().okz_nms
    assert resp.dtype == (np.int32 if fast else np.float32) and size.dtype == np.float32 and layer.dtype == np.int32
    n = fn(pts.ctypes.data_as(C.c_void_p), C.c_int(max_pts), resp.ctypes.data_as(C.c_void_p), _f(size.reshape(-1)),
           layer.ctypes.data_as(_ip), C.c_int(psz), C.c_int(w), C.c_int(h), C.c_int(p))
    return pts[:min(n, max_pts)].copy(), n


def refine_point(pt, det, o, fast=False):
    """okz_refine_point / fkz_refine (gRefine akazed.cu:1615 / 3600) on one record; det: (h, p) plane of its level"""
    rec = np.array([pt], POINT_DTYPE)
    h, p = det.shape
    fn = lib().fkz_refine if fast else lib().okz_refine_point
    fn(rec.ctypes.data_as(C.c_void_p), det.ctypes.data_as(C.c_void_p), C.c_int(o), C.c_int(p))
    return rec[0]


def orient_point(pt, lx, ly, o, w, fast=False):
    """okz_orient_point / fkz_orient (gCalcOrient akazed.cu:1665 / 3649); lx, ly: (h, p) planes of the point's level"""
    rec = np.array([pt], POINT_DTYPE)
    h, p = lx.shape
    fn = lib().fkz_orient if fast else lib().okz_orient_point
    fn(rec.ctypes.data_as(C.c_void_p), lx.ctypes.data_as(C.c_void_p), ly.ctypes.data_as(C.c_void_p), C.c_int(o), C.c_int(w),
       C.c_int(h), C.c_int(p), _f(orient_weights()))
    return rec[0]


def describe_point(pt, lt, lx, ly, o, w, patsize=10, fast=False):
    """okz_describe_point / fkz_describe (gDescribe2 akazed.cu:1869 / 3723)"""
    rec = np.array([pt], POINT_DTYPE)
    h, p = lt.shape
    i1, i2 = compare_indices()
    fn = lib().fkz_describe if fast else lib().okz_describe_point
    fn(rec.ctypes.data_as(C.c_void_p), lt.ctypes.data_as(C.c_void_p), lx.ctypes.data_as(C.c_void_p), ly.ctypes.data_as(C.c_void_p),
       C.c_int(o), C.c_int(w), C.c_int(h), C.c_int(p), C.c_int(patsize), i1.ctypes.data_as(_ip), i2.ctypes.data_as(_ip))
    return rec[0]


def match(pts1, pts2):
    """in-place on pts1 (structured arrays)"""
    lib().okz_match(pts1.ctypes.data_as(C.c_void_p), len(pts1), pts2.ctypes.data_as(C.c_void_p), len(pts2))
    return pts1


MATCH_PAIR_DTYPE = np.dtype([("query", "<i4"), ("train", "<i4"), ("distance", "<i4"), ("second", "<i4"),
                             ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4")])


def match_knn2(pts1, pts2, ratio=(1, 1), cross=True, max_dist=96):
    """2-NN ratio test + cross-check + compaction; pts1 updated in place, returns the accepted-match list"""
    out = np.zeros(max(len(pts1), 1), MATCH_PAIR_DTYPE)
    n = lib().okz_match_knn2(pts1.ctypes.data_as(C.c_void_p), len(pts1), pts2.ctypes.data_as(C.c_void_p), len(pts2),
                             int(ratio[0]), int(ratio[1]), int(cross), int(max_dist), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()


class Result:
    pass


def detect_and_compute(img, w, params=None, max_pts=10000, desc=True, keep_arena=False):
    """img: float32 (h, p) pitched plane with valid width w.  Returns Result(points, kcontrast, layout, arena)."""
    params = params or default_params()
    h, p = img.shape
    L = lib()
    n = L.okz_arena_floats(w, h, p, params.noctaves, params.max_scale)
    arena = np.zeros(n, np.float32)
    pts = np.zeros(max_pts, POINT_DTYPE)
    kc = C.c_float()
    num = L.okz_detect_and_compute(_f(img), w, h, p, C.byref(params), pts.ctypes.data_as(C.c_void_p), max_pts,
                                   int(desc), _f(arena), C.byref(kc))
    r = Result()
    r.points = pts[:num].copy()
    r.kcontrast = np.float32(kc.value)
    owhps = np.zeros(24, np.int32); osizes = np.zeros(8, np.int32); offsets = np.zeros(9, np.int32)
    r.noct = L.okz_layout(w, h, p, params.noctaves, params.max_scale, owhps.ctypes.data_as(_ip),
                          osizes.ctypes.data_as(_ip), offsets.ctypes.data_as(_ip))
    r.owhps, r.osizes, r.offsets, r.ms = owhps, osizes, offsets, params.max_scale
    r.arena = arena if keep_arena else None
    return r


def plane(r, kind, o, s):
    """kind: 0 Lt, 1 det, 2 Lx, 3 Ly (SURVEY 9.1 arena layout); returns the dense (h, w) view"""
    w, h, p = (int(v) for v in r.owhps[3 * o:3 * o + 3])
    base = int(r.offsets[o]) + (kind * r.ms + s) * int(r.osizes[o])
    return r.arena[base:base + h * p].reshape(h, p)[:, :w]


# ------------------------------------------------------------------ integer FAST path (akaze_oracle_fast.c)
def fast_detect_and_compute(u8, params=None, max_pts=10000, desc=True, keep_arena=False):
    """u8: uint8 (h, w) image (dense).  Returns Result(points, kcontrast[, arena of int32])."""
    params = params or default_params()
    u8 = np.ascontiguousarray(u8)
    h, w = u8.shape
    p = (w + 127) // 128 * 128
    L = lib()
    L.fkz_arena_ints.restype = C.c_long
    n = L.fkz_arena_ints(w, h, p, params.noctaves, params.max_scale)
    arena = np.zeros(n, np.int32)
    pts = np.zeros(max_pts, POINT_DTYPE)
    kc = C.c_int()
    num = L.fkz_detect_and_compute(u8.ctypes.data_as(C.c_void_p), w, h, w, p, C.byref(params), pts.ctypes.data_as(C.c_void_p),
                                   max_pts, int(desc), arena.ctypes.data_as(C.c_void_p), C.byref(kc))
    r = Result()
    r.points = pts[:num].copy()
    r.kcontrast = kc.value
    owhps = np.zeros(24, np.int32); osizes = np.zeros(8, np.int32); offsets = np.zeros(9, np.int32)
    r.noct = L.okz_layout(w, h, p, params.noctaves, params.max_scale, owhps.ctypes.data_as(_ip),
                          osizes.ctypes.data_as(_ip), offsets.ctypes.data_as(_ip))
    r.owhps, r.osizes, r.offsets, r.ms = owhps, osizes, offsets, params.max_scale
    r.arena = arena if keep_arena else None
    return r
