import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("cuda-akaze_amd", "oracle", ""):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The library picks the register-streaming kernels (k_hessian_stream, k_fed_sf, k_base_stream) only when a launch is large enough to fill
# the chip; the parity tests run small images and batches, so they force those kernels on wherever they apply (mode 2).  The
# tile kernels and the size rule itself are covered by test_kernel_alternatives_are_bit_identical and by the odd-size cases.
os.environ.setdefault("HAK_HESS_STREAM", "2")
os.environ.setdefault("HAK_FUSE_SF", "2")
os.environ.setdefault("HAK_BASE_STREAM", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def okz():
    """the CPU parity oracle (oracle/akaze_oracle.c) -- the checker, never the thing under test on GPU"""
    import okz as _okz
    _okz.build()
    # (a process that has imported torch has initialised libgomp with one thread per logical CPU: 256 on the GPU boxes, where the
    # oracle's short loops then run ten times slower than on 16)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    _okz.set_num_threads(max(1, min(16, ncpu)))
    return _okz


@pytest.fixture(scope="session")
def ah():
    """the product's Python harness over the C ABI (libhipakaze.so must exist: no fallback)"""
    import akaze_hip
    return akaze_hip


@pytest.fixture(scope="session")
def golden():
    class G:
        pass
    g = G()
    g.dir = GOLDEN
    g.lr_u8 = np.load(os.path.join(GOLDEN, "left_right_u8.npz"))
    g.lr = np.load(os.path.join(GOLDEN, "left_right_oracle.npz"))
    g.synth = np.load(os.path.join(GOLDEN, "synth_oracle.npz"))
    return g


def canon(pts):
    """canonical keypoint order: sort by (layer, y, x) of the refined coordinates' bits"""
    order = np.lexsort((pts["x"].view(np.uint32), pts["y"].view(np.uint32), pts["octave"]))
    return pts[order]


def assert_points_equal(a, b, fields=("x", "y", "octave", "response", "size", "angle", "features")):
    assert len(a) == len(b), f"keypoint count {len(a)} != {len(b)}"
    for f in fields:
        av, bv = a[f], b[f]
        if av.dtype.kind == "f":
            av, bv = av.view(np.uint32), bv.view(np.uint32)
        bad = np.nonzero((av != bv).reshape(len(a), -1).any(axis=1))[0]
        assert bad.size == 0, f"field {f}: {bad.size} of {len(a)} points differ (first at {bad[:5]})"
