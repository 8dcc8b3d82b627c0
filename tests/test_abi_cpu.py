"""CPU suite, part 2: the C-ABI library loads, exports every symbol include/hipakaze.h declares,
its host-side schedule agrees with the reference-pinned golden, and it FAILS LOUDLY without a GPU
(no CPU fallback; the oracle is never reachable from the product)."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_header_symbols_are_exported(ah):
    hdr = open(os.path.join(ROOT, "include", "hipakaze.h")).read()
    declared = set(re.findall(r"\b(hak_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    assert declared == set(ah.SYMBOLS), declared ^ set(ah.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", ah.LIB_PATH], text=True)
    exported = set(re.findall(r" T (hak_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_product_never_links_the_oracle(ah):
    out = subprocess.check_output(["nm", "-D", ah.LIB_PATH], text=True)
    assert "okz_" not in out
    ldd = subprocess.check_output(["ldd", ah.LIB_PATH], text=True)
    assert "oracle" not in ldd
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cuda-akaze_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "import okz" not in txt and "oracle/" not in txt.replace("oracle/okz_math.h", ""), f
    for f in ("akaze.h", "akaze_structures.h", "hip_utils.h", "hipakaze.h"):
        path = os.path.join(ROOT, "include", f)
        if os.path.exists(path):
            assert "liboracle" not in open(path).read()


def test_struct_sizes(ah):
    assert ah.POINT_DTYPE.itemsize == 104
    assert C.sizeof(ah.hak_config) == 13 * 4


def test_host_fed_tau_matches_reference_golden(ah):
    cases = json.load(open(os.path.join(GOLDEN, "fed_tau.json")))["cases"]
    for c in cases:
        tau = ah.fed_tau(c["T"], c["M"], c["tau_max"], c["reordering"])
        assert [int(v) for v in tau.view(np.uint32)] == c["tau_bits"], c["T"]


def test_host_tables_match_oracle(ah, okz):
    for var, r in ((1.0, 2), (2.56, 4), (float(np.float32(1.6) ** 2), 4), (0.7, 3), (4.0, 5)):
        assert np.array_equal(ah.gauss_taps(var, r).view(np.uint32), okz.gauss_taps(var, r).view(np.uint32))
    a, b = ah.compare_indices()
    oa, ob = okz.compare_indices()
    assert np.array_equal(a, oa) and np.array_equal(b, ob)


def test_no_gpu_means_loud_failure(ah):
    if ah.device_count() > 0:
        pytest.skip("a GPU is present")
    a = ah.Akazer()
    with pytest.raises(ah.HakError, match="no HIP device"):
        a.init((640, 480, 640))
    d = ah.AkazeData()
    with pytest.raises(ah.HakError):
        ah.initAkazeData(d, 16, True, True)


def test_reference_shaped_api_surface(ah):
    for name in ("initAkazeData", "freeAkazeData", "cuMatch", "Akazer", "AkazeData"):
        assert hasattr(ah, name)
    for m in ("init", "detectAndCompute", "fastDetectAndCompute"):
        assert hasattr(ah.Akazer, m)
    assert (ah.PM_G1, ah.PM_G2, ah.WEICKERT, ah.CHARBONNIER) == (0, 1, 2, 3)
    assert ah.iAlignUp(1920, 128) == 1920 and ah.iAlignUp(1281, 128) == 1408
