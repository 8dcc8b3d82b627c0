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
    # the product ABI carries no test scaffolding: stage operators, plane introspection and probes are a library of their own
    assert not [n for n in declared if n.startswith(("hak_op_", "hak_debug_"))]
    assert not [n for n in exported if n.startswith(("hak_op_", "hak_debug_"))], "test entry points linked into libhipakaze.so"


def test_test_abi_is_a_separate_library(ah):
    """include/hipakaze_test.h -> libhipakaze_test.so: every declared symbol exported there, the library links against the product
    library (it drives the product's launchers, it does not carry kernels of the launch sequence) and loads without a GPU"""
    hdr = open(os.path.join(ROOT, "include", "hipakaze_test.h")).read()
    declared = set(re.findall(r"\b(hak_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(ah.TEST_SYMBOLS), declared ^ set(ah.TEST_SYMBOLS)
    assert all(n.startswith(("hak_op_", "hak_debug_")) for n in declared) and len(declared) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", ah.TEST_LIB_PATH], text=True)
    exported = set(re.findall(r" T (hak_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert "libhipakaze.so" in subprocess.check_output(["ldd", ah.TEST_LIB_PATH], text=True)
    assert "okz_" not in subprocess.check_output(["nm", "-D", ah.TEST_LIB_PATH], text=True)
    ah.lib.load_test()
    assert callable(ah.lib.hak_op_lowpass) and callable(ah.lib.hak_debug_plane)


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
    for f in ("akaze.h", "akaze_structures.h", "hip_utils.h", "hipakaze.h", "hipakaze_test.h"):
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


def _mldb_rows(patsize):
    """accumulator row of every window sample in the three grids, straight from the reference's loops (akazed.cu:1905-1955):
    rows 0..11 = 2x2 cells, 12..38 = 3x3, 39..86 = 4x4, three rows (value, dx', dy') per cell"""
    s2, s3, s4 = patsize, -(-2 * patsize // 3), -(-patsize // 2)
    win = max(3 * s3, 4 * s4)
    rows = {}
    for i in range(win * win):
        y, x = divmod(i, win)
        m = max(x, y)
        r2 = 3 * ((0 if y < s2 else 2) + (0 if x < s2 else 1)) if m < 2 * s2 else None
        r3 = 3 * (4 + min(y // s3, 2) * 3 + min(x // s3, 2)) if m < 3 * s3 else None
        r4 = 39 + 3 * (min(y // s4, 3) * 4 + min(x // s4, 3)) if m < 4 * s4 else None
        rows[i] = (x - s2, y - s2, (r2, r3, r4))
    return win, rows


@pytest.mark.parametrize("patsize", [4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_describe_plan_restates_the_reference_loops(ah, patsize):
    """hak_describe_plan_query (the table k_describe_runs works from; host code, no GPU): every sample lands in the lane and
    turn the reference's `i = tx; i += 64` loop gives it, with the reference's window offset and grid rows; a lane's samples of
    one row are consecutive turns (that is what lets the kernel carry the row's partial sum in registers); sizes without
    that property, or with more than 7 samples per lane, report 0 and take the generic kernel"""
    ok, pos, cell = ah.describe_plan(patsize)
    win, rows = _mldb_rows(patsize)
    nsmp = win * win
    fits = nsmp <= 7 * 64
    revisits = False
    for lane in range(64):
        for g in range(3):
            seq = [rows[lane + 64 * n][2][g] if lane + 64 * n < nsmp else None for n in range(7)]
            seen, prev = set(), None
            for r in seq:
                if r is not None and r != prev and r in seen:
                    revisits = True
                if r is not None:
                    seen.add(r)
                prev = r
    assert ok == (fits and not revisits)
    assert ok == (patsize in (4, 5, 6, 7, 9, 10))                           # the default size 10 is planned
    if not ok:                              # (the tables are only filled in for a size the planned kernel serves)
        return
    for n in range(7):
        for lane in range(64):
            i = lane + 64 * n
            pw, cw = int(pos[n, lane]), int(cell[n, lane])
            if i >= nsmp:
                assert pw == 0 and (cw & 0xFFFFFF) == 0x7F7F7F and (cw >> 24) == 0
                continue
            l, k, rr = rows[i]
            assert (pw >> 16) & 1
            sb = lambda v: v - 256 if v > 127 else v
            assert sb(pw & 0xFF) == l and sb((pw >> 8) & 0xFF) == k
            for g in range(3):
                b = (cw >> (8 * g)) & 0xFF
                assert (b & 0x7F) == (0x7F if rr[g] is None else rr[g])
                prev = rows[i - 64][2][g] if n > 0 else None
                nxt = rows[i + 64][2][g] if n < 6 and i + 64 < nsmp else None
                assert bool(b & 0x80) == (rr[g] is not None and prev == rr[g])
                assert bool((cw >> (24 + g)) & 1) == (rr[g] is not None and nxt != rr[g])
