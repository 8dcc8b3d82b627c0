"""akaze_hip -- Python host mirror of the CUDA-AKAZE interface over libhipakaze's C ABI.

This is harness code (tests, bench): the product is the C-ABI shared library
``cuda-akaze_amd/libhipakaze.so`` and the C++ header ``include/akaze.h``.  The
names below follow the reference's API (akaze.h:10-30, akaze_structures.h:19-59)
so the parity tests read like the reference's demo (main.cpp:128-233):

    initAkazeData / freeAkazeData / cuMatch / Akazer.init / Akazer.detectAndCompute

There is NO CPU fallback: importing works anywhere (the library loads without a
GPU so symbol tests can run), but every compute call raises ``HakError`` when no
HIP device is usable, and a missing ``libhipakaze.so`` raises at import.
"""
import ctypes as C
import os

import numpy as np

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7, and a process that
# loads /opt/rocm's copy first makes torch see "No HIP GPUs" (and vice versa: measured on the GPU
# box, tools/probe_runtime.py).  Importing torch first lets libhipakaze.so bind to the runtime
# torch already loaded (same soname); without torch the library uses /opt/rocm's.
try:
    import torch as _torch  # noqa: F401
except ImportError:          # pure C-ABI use without PyTorch
    _torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# HAK_LIB selects another build of the same library (A/B measurements of compiler flags / kernel variants); never a fallback
LIB_PATH = os.environ.get("HAK_LIB") or os.path.join(os.path.dirname(_HERE), "libhipakaze.so")

FLEN = 61            # akaze_structures.h:29
MAX_DIST = 96        # akazed.cu:11
PM_G1, PM_G2, WEICKERT, CHARBONNIER = 0, 1, 2, 3   # akaze_structures.h:53-59

# akaze_structures.h:19-40 (104 bytes; numpy adds the 3 padding bytes explicitly)
POINT_DTYPE = np.dtype([
    ("x", "<f4"), ("y", "<f4"), ("octave", "<i4"), ("response", "<f4"), ("size", "<f4"), ("angle", "<f4"),
    ("features", "u1", (FLEN,)), ("_pad", "u1", (3,)),
    ("match", "<i4"), ("distance", "<i4"), ("match_x", "<f4"), ("match_y", "<f4"),
])
assert POINT_DTYPE.itemsize == 104
MATCH_PAIR_DTYPE = np.dtype([("query", "<i4"), ("train", "<i4"), ("distance", "<i4"), ("second", "<i4"),
                             ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4")])      # hak_match_pair
assert MATCH_PAIR_DTYPE.itemsize == 32


class HakError(RuntimeError):
    pass


class hak_config(C.Structure):
    _fields_ = [
        ("noctaves", C.c_int), ("max_scale", C.c_int), ("per", C.c_float), ("kcontrast", C.c_float),
        ("soffset", C.c_float), ("reordering", C.c_int), ("derivative_factor", C.c_float),
        ("dthreshold", C.c_float), ("diffusivity", C.c_int), ("descriptor_pattern_size", C.c_int),
        ("max_pts", C.c_int), ("upright", C.c_int), ("batch", C.c_int),
    ]


class hak_traffic(C.Structure):
    _fields_ = [("fed_px_steps", C.c_double), ("fed_bytes", C.c_double), ("all_stage_bytes", C.c_double),
                ("fed_launches", C.c_int), ("fed_fused_bytes", C.c_double), ("hessian_bytes", C.c_double),
                ("prologue_bytes", C.c_double), ("describe_bytes", C.c_double), ("nms_bytes", C.c_double)]


PROF = dict(fed=0, lowpass=1, flow=2, hessian=3, contrast=4, down=5, extrema=6, nms=7, describe=8, match=9)

# every symbol include/hipakaze.h declares: name -> (restype, argtypes)
_vp, _fp, _ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
SYMBOLS = {
    "hak_device_count": (C.c_int, []),
    "hak_set_device": (C.c_int, [C.c_int]),
    "hak_last_error": (C.c_char_p, []),
    "hak_default_config": (None, [C.POINTER(hak_config)]),
    "hak_create": (C.c_int, [C.POINTER(hak_config), C.c_int, C.c_int, C.POINTER(_vp)]),
    "hak_destroy": (None, [_vp]),
    "hak_set_stream": (C.c_int, [_vp, _vp]),
    "hak_sync": (C.c_int, [_vp]),
    "hak_wait_event": (C.c_int, [_vp, _vp]),
    "hak_phase_event": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "hak_set_concurrency": (C.c_int, [_vp, C.c_int]),
    "hak_set_null_order": (C.c_int, [_vp, C.c_int]),
    "hak_detect_and_compute": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _ip, _vp, C.c_int]),
    "hak_detect_and_compute_pair": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _ip, _ip, _vp, _vp, C.c_int, C.c_int]),
    "hak_detect_and_compute_batch": (C.c_int, [_vp, _vp, C.c_long, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "hak_fast_detect_and_compute": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _ip, _vp, C.c_int]),
    "hak_fast_detect_and_compute_batch": (C.c_int, [_vp, _vp, C.c_long, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "hak_match": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "hak_match_batch": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "hak_match_knn2": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _ip, _vp]),
    "hak_match_knn2_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "hak_points_alloc": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "hak_points_free": (C.c_int, [_vp]),
    "hak_image_alloc": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _ip]),
    "hak_image_upload": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int]),
    "hak_image_free": (C.c_int, [_vp]),
    "hak_ingest_u8": (C.c_int, [_vp, _vp, C.c_long, C.c_int, _vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hak_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_long]),
    "hak_host_free": (C.c_int, [_vp]),
    "hak_download_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "hak_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_long]),
    "hak_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_long]),
    "hak_fed_tau": (C.c_int, [C.c_float, C.c_int, C.c_float, C.c_int, _fp, C.c_int]),
    "hak_gauss_taps": (None, [C.c_float, C.c_int, _fp]),
    "hak_compare_indices": (None, [_ip, _ip]),
    "hak_describe_plan_query": (C.c_int, [C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "hak_query_schedule": (C.c_int, [_vp, _ip, _ip, _fp, _fp]),
    "hak_query_geometry": (C.c_int, [_vp, _ip]),
    "hak_query_traffic": (C.c_int, [_vp, C.c_int, C.POINTER(hak_traffic)]),
    "hak_prof_enable": (C.c_int, [_vp, C.c_int]),
    "hak_prof_read": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), _ip]),
    "hak_prof_reset": (C.c_int, [_vp]),
}
# the TEST ABI (include/hipakaze_test.h -> libhipakaze_test.so): stage operators, plane introspection, probes
TEST_SYMBOLS = {
    "hak_debug_plane": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "hak_debug_kcontrast": (C.c_int, [_vp, C.c_int, _fp]),
    "hak_op_lowpass": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "hak_op_down_smooth": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hak_op_kcontrast": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _ip]),
    "hak_op_flow": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "hak_op_nld_steps": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _fp, C.c_int]),
    "hak_op_rcp_check": (C.c_int, [C.c_uint, C.c_uint, C.POINTER(C.c_ulonglong)]),
    "hak_op_smooth_flow": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "hak_op_hessian": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hak_debug_set_plane": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "hak_op_tail_begin": (C.c_int, [_vp]),
    "hak_op_tail_level": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "hak_op_tail_det_level": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "hak_op_tail_seed": (C.c_int, [_vp, _vp, _vp]),
    "hak_op_tail_finish": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _ip]),
    "hak_op_orient_describe": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "hak_op_copy_probe": (C.c_int, [C.c_long, C.c_int, C.POINTER(C.c_double)]),
    "hak_op_copy_probe_shapes": (C.c_int, [C.c_long, C.c_int, C.POINTER(C.c_double), C.c_int]),
    "hak_op_stream_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hak_op_hess_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hak_debug_fill_match_scratch": (C.c_int, [_vp, C.c_int]),
    "hak_op_gather_probe": (C.c_int, [C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the HIP path)")
TEST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libhipakaze_test.so")


class _Libs:
    """`lib.<symbol>`: the product ABI from libhipakaze.so; hak_op_* / hak_debug_* from libhipakaze_test.so, which is loaded on
    first use (tests and bench only) and links against the product library next to it"""

    def __init__(self):
        self.product = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        self.test = None
        for name, (res, args) in SYMBOLS.items():
            try:
                f = getattr(self.product, name)      # AttributeError here = ABI drift between header and library
            except AttributeError:
                if os.environ.get("HAK_LIB"):        # an older build loaded for an A/B run may lack the newest entry points
                    continue
                raise
            f.restype, f.argtypes = res, args
            setattr(self, name, f)

    def load_test(self):
        if self.test is None:
            if not os.path.exists(TEST_LIB_PATH):
                raise ImportError(f"{TEST_LIB_PATH} is missing: build it with `make -C cuda-akaze_amd/csrc`")
            self.test = C.CDLL(TEST_LIB_PATH)
            for name, (res, args) in TEST_SYMBOLS.items():
                f = getattr(self.test, name)
                f.restype, f.argtypes = res, args
                setattr(self, name, f)
        return self.test

    def __getattr__(self, name):                     # only reached for names not bound yet
        if name in TEST_SYMBOLS:
            self.load_test()
            return self.__dict__[name]
        raise AttributeError(name)


lib = _Libs()


def check(status):
    if status != 0:
        raise HakError(lib.hak_last_error().decode() or f"libhipakaze status {status}")


def iAlignUp(a, b):
    """cuda_utils.h:160"""
    return a - a % b + b if a % b else a


def device_count():
    return lib.hak_device_count()


def default_config(**kw):
    cfg = hak_config()
    lib.hak_default_config(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


# ----------------------------------------------------------------- host helpers
def fed_tau(T, M=1, tau_max=0.25, reordering=True):
    buf = np.zeros(4096, np.float32)
    n = lib.hak_fed_tau(T, M, tau_max, int(reordering), buf.ctypes.data_as(_fp), 4096)
    if n < 0:
        raise HakError("FED cycle too long")
    return buf[:n].copy()


def gauss_taps(var, radius):
    buf = np.zeros(8, np.float32)
    lib.hak_gauss_taps(var, radius, buf.ctypes.data_as(_fp))
    return buf[:radius + 1].copy()


def compare_indices():
    a = np.zeros(488, np.int32)
    b = np.zeros(488, np.int32)
    lib.hak_compare_indices(a.ctypes.data_as(_ip), b.ctypes.data_as(_ip))
    return a, b


def describe_plan(pattern_size):
    """(planned, pos[7][64], cell[7][64]) of hak_describe_plan_query"""
    pos = np.zeros((7, 64), np.uint32)
    cell = np.zeros((7, 64), np.uint32)
    ok = lib.hak_describe_plan_query(pattern_size, pos.ctypes.data_as(C.POINTER(C.c_uint)), cell.ctypes.data_as(C.POINTER(C.c_uint)))
    return bool(ok), pos, cell


# ------------------------------------------------------ reference-shaped API
class AkazeData:
    """akaze_structures.h:44-50 {num_pts, max_pts, h_data, d_data}"""

    def __init__(self):
        self.num_pts = 0
        self.max_pts = 0
        self.h_data = None      # numpy structured array (POINT_DTYPE) or None
        self.d_data = None      # device pointer (int) or None
        self.h_pinned = None    # address of h_data when it lives in pinned host memory


def initAkazeData(data, max_pts, host, dev, pinned=False):
    """akaze.cpp:26-40.  pinned=True gives h_data in device-visible host memory, as the C++ layer's initAkazeData does
    (host/akaze.cpp): detectAndCompute then delivers records and count inside its launch sequence."""
    data.num_pts = 0
    data.max_pts = max_pts
    data.h_pinned = None
    if host and pinned:
        hp = _vp()
        check(lib.hak_host_alloc(C.byref(hp), max_pts * POINT_DTYPE.itemsize))
        data.h_pinned = hp.value
        buf = (C.c_uint8 * (max_pts * POINT_DTYPE.itemsize)).from_address(hp.value)
        data.h_data = np.frombuffer(buf, dtype=POINT_DTYPE)
        data.h_data[:] = np.zeros(1, POINT_DTYPE)[0]
    else:
        data.h_data = np.zeros(max_pts, POINT_DTYPE) if host else None
    data.d_data = None
    if dev:
        p = _vp()
        check(lib.hak_points_alloc(C.byref(p), max_pts))
        data.d_data = p.value


def freeAkazeData(data):
    """akaze.cpp:43-52"""
    if data.d_data is not None:
        check(lib.hak_points_free(data.d_data))
    data.d_data = None
    data.h_data = None
    if getattr(data, "h_pinned", None):
        check(lib.hak_host_free(data.h_pinned))
        data.h_pinned = None
    data.num_pts = 0
    data.max_pts = 0


class Akazer:
    """akaze.h:18-67.  ``whp0`` is (width, height, pitch) like the reference's int3."""

    def __init__(self):
        self._ctx = None
        self._cfg = default_config()
        self.whp = (0, 0, 0)

    def init(self, whp0, noctaves=4, max_scale=4, per=0.7, kcontrast=0.03, soffset=1.6, reordering=True,
             derivative_factor=1.5, dthreshold=0.001, diffusivity=PM_G2, descriptor_pattern_size=10,
             max_pts=10000, upright=False, batch=1):
        self.whp = tuple(whp0)
        self._cfg = default_config(
            noctaves=noctaves, max_scale=max_scale, per=per, kcontrast=kcontrast, soffset=soffset,
            reordering=int(reordering), derivative_factor=derivative_factor, dthreshold=dthreshold,
            diffusivity=int(diffusivity), descriptor_pattern_size=descriptor_pattern_size,
            max_pts=max_pts, upright=int(upright), batch=batch)
        self._make_ctx(self.whp[0], self.whp[1])

    def _make_ctx(self, w, h):
        self.close()
        ctx = _vp()
        check(lib.hak_create(C.byref(self._cfg), w, h, C.byref(ctx)))
        self._ctx = ctx
        self._ctx_wh = (w, h)

    @property
    def ctx(self):
        if self._ctx is None:
            raise HakError("Akazer.init() has not been called")
        return self._ctx

    def detectAndCompute(self, image, result, whp0, desc=True):
        """akaze.cpp:101-150.  ``image`` = device pointer (int) to float32 [0,1], pitch whp0[2]."""
        w, h, p = whp0
        if self._ctx is None or self._ctx_wh != (w, h):       # akaze.cpp:109: size differs from init -> new arena
            self._make_ctx(w, h)
        n = C.c_int(0)
        hptr = result.h_data.ctypes.data if result.h_data is not None else None
        check(lib.hak_detect_and_compute(self.ctx, image, p, result.d_data, result.max_pts, C.byref(n), hptr, int(desc)))
        result.num_pts = n.value

    def detectAndComputePair(self, image1, image2, result1, result2, whp0, desc=True, match=True):
        """build-side addition (akaze.h): both images + cuMatch(result1, result2) as one launch sequence and one synchronisation.
        The context must have been init()-ed with batch >= 2."""
        w, h, p = whp0
        if self._ctx is None or self._ctx_wh != (w, h):
            self._make_ctx(w, h)
        n1, n2 = C.c_int(0), C.c_int(0)
        h1 = result1.h_data.ctypes.data if result1.h_data is not None else None
        h2 = result2.h_data.ctypes.data if result2.h_data is not None else None
        check(lib.hak_detect_and_compute_pair(self.ctx, image1, image2, p, result1.d_data, result2.d_data, result1.max_pts, result2.max_pts,
                                              C.byref(n1), C.byref(n2), h1, h2, int(desc), int(match)))
        result1.num_pts, result2.num_pts = n1.value, n2.value

    def fastDetectAndCompute(self, image, result, whp0, desc=True):
        """akaze.h:30, akaze.cpp:153-201 -- integer FAST path.  ``image`` = device pointer to uint8, pitch whp0[2] bytes."""
        w, h, p = whp0
        if self._ctx is None or self._ctx_wh != (w, h):
            self._make_ctx(w, h)
        n = C.c_int(0)
        hptr = result.h_data.ctypes.data if result.h_data is not None else None
        check(lib.hak_fast_detect_and_compute(self.ctx, image, p, result.d_data, result.max_pts, C.byref(n), hptr, int(desc)))
        result.num_pts = n.value

    # -- introspection used by tests
    def plane(self, kind, octave, sublevel, img=0):
        whp = self.geometry()[octave]
        out = np.zeros((whp[1], whp[0]), np.float32)
        check(lib.hak_debug_plane(self.ctx, img, kind, octave, sublevel, out.ctypes.data))
        return out

    def set_plane(self, kind, octave, sublevel, plane, img=0):
        """hak_debug_set_plane: dense (h, w) float32 -> plane (0 Lt, 2 Lx, 3 Ly) of the level"""
        plane = np.ascontiguousarray(plane, np.float32)
        w, h, _ = self.geometry()[octave]
        assert plane.shape == (h, w), (plane.shape, (h, w))
        check(lib.hak_debug_set_plane(self.ctx, img, kind, octave, sublevel, plane.ctypes.data))

    # -- detector tail / descriptors on hand-made inputs (hipakaze.h hak_op_tail_*; tests/test_gpu_literal.py)
    def tail_begin(self):
        check(lib.hak_op_tail_begin(self.ctx))

    def tail_level(self, octave, sublevel, lplane):
        lplane = np.ascontiguousarray(lplane, np.float32)
        check(lib.hak_op_tail_level(self.ctx, octave, sublevel, lplane.ctypes.data))

    def tail_det_level(self, octave, sublevel, det):
        det = np.ascontiguousarray(det, np.float32)
        check(lib.hak_op_tail_det_level(self.ctx, octave, sublevel, det.ctypes.data))

    def tail_seed(self, response, layer):
        """response: (h, w) float32 (or int32 for the FAST path's map), layer: (h, w) int32, < 0 = no candidate"""
        response = np.ascontiguousarray(response)
        assert response.dtype.itemsize == 4
        layer = np.ascontiguousarray(layer, np.int32)
        check(lib.hak_op_tail_seed(self.ctx, response.ctypes.data, layer.ctypes.data))

    def tail_finish(self, max_pts=1000, refine=False, fast=False):
        """NMS + emit (+ refine) -> host records in raster order"""
        data = AkazeData()
        initAkazeData(data, max_pts, True, True)
        try:
            n = C.c_int(0)
            check(lib.hak_op_tail_finish(self.ctx, data.d_data, max_pts, int(refine), int(fast), C.byref(n)))
            k = min(n.value, max_pts)
            if k:
                check(lib.hak_memcpy_d2h(data.h_data.ctypes.data, data.d_data, k * POINT_DTYPE.itemsize))
            return data.h_data[:k].copy(), n.value
        finally:
            freeAkazeData(data)

    def orient_describe(self, points, desc=1):
        """orientation + MLDB (desc=1) or MLDB with the records' own angles (desc=2) on host records; returns them updated"""
        points = np.ascontiguousarray(points)
        assert points.dtype == POINT_DTYPE and len(points) >= 1
        data = AkazeData()
        initAkazeData(data, len(points), False, True)
        try:
            check(lib.hak_memcpy_h2d(data.d_data, points.ctypes.data, points.nbytes))
            check(lib.hak_op_orient_describe(self.ctx, data.d_data, len(points), int(desc)))
            out = np.zeros(len(points), POINT_DTYPE)
            check(lib.hak_memcpy_d2h(out.ctypes.data, data.d_data, out.nbytes))
            return out
        finally:
            freeAkazeData(data)

    def kcontrast(self, img=0):
        v = C.c_float()
        check(lib.hak_debug_kcontrast(self.ctx, img, C.byref(v)))
        return v.value

    def geometry(self):
        buf = np.zeros(3 * 8, np.int32)
        n = lib.hak_query_geometry(self.ctx, buf.ctypes.data_as(_ip))
        return [tuple(int(v) for v in buf[3 * o:3 * o + 3]) for o in range(n)]

    def schedule(self):
        ns = np.zeros(40, np.int32); ss = np.zeros(40, np.int32)
        sz = np.zeros(40, np.float32); bd = np.zeros(40, np.float32)
        n = lib.hak_query_schedule(self.ctx, ns.ctypes.data_as(_ip), ss.ctypes.data_as(_ip),
                                   sz.ctypes.data_as(_fp), bd.ctypes.data_as(_fp))
        m = n * self._cfg.max_scale
        return dict(noct=n, nsteps=ns[:m].copy(), sigma_size=ss[:m].copy(), sizes=sz[:m].copy(), borders=bd[:m].copy())

    def traffic(self, npts_hint=0):
        t = hak_traffic()
        check(lib.hak_query_traffic(self.ctx, npts_hint, C.byref(t)))
        return t

    def close(self):
        if self._ctx is not None:
            lib.hak_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cuMatch(result1, result2, akazer=None):
    """akaze.h:14, akaze.cpp:55-64: fills match/distance/match_x/match_y of result1."""
    ctx = akazer.ctx if akazer is not None else None
    hptr = result1.h_data.ctypes.data if result1.h_data is not None else None
    check(lib.hak_match(ctx, result1.d_data, result1.num_pts, result2.d_data, result2.num_pts, hptr))


def cuMatchKnn(result1, result2, ratio=(1, 1), cross_check=True, max_dist=0, akazer=None):
    """Match post-processing (hipakaze.h hak_match_knn2): 2-NN ratio test of the reference's unused
    gMatch (akazed.cu:2028-2122) + symmetric cross-check + device-side compaction.  Updates result1 like
    cuMatch and returns the accepted matches (MATCH_PAIR_DTYPE, ascending query index)."""
    import torch
    ctx = akazer.ctx if akazer is not None else None
    n1 = result1.num_pts
    d_out = torch.zeros(max(n1, 1) * MATCH_PAIR_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    h_out = np.zeros(max(n1, 1), MATCH_PAIR_DTYPE)
    cnt = C.c_int(0)
    hptr = result1.h_data.ctypes.data if result1.h_data is not None else None
    check(lib.hak_match_knn2(ctx, result1.d_data, n1, result2.d_data, result2.num_pts, int(ratio[0]), int(ratio[1]),
                             int(cross_check), int(max_dist), hptr, d_out.data_ptr(), C.byref(cnt), h_out.ctypes.data))
    return h_out[:cnt.value].copy()
