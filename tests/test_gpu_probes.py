"""Stand-alone device probes the kernels' design rests on (tools/probes/*.hip), built with hipcc on the GPU box and run as child
processes."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def test_fp4_hamming_probe(tmp_path):
    """k_match_mfma's arithmetic in isolation: Hamming distances of 32 x 32 descriptors of 488 bits as fp4 (E2M1) dot products on
    v_mfma_f32_32x32x64_f8f6f4, accumulator started at 2^23 + |b| -- bit patterns 0x4B000000 + popcount(a ^ b) for every pair, both
    with explicit unit scales and in the unscaled form the kernel uses (all-zero / all-one rows included)"""
    cc = _hipcc()
    if cc is None:
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "fp4_probe")
    b = subprocess.run([cc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "probes", "fp4_hamming_probe.hip")],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    # ... and round 5's form: the accumulator as the finished key, 2^10 + 2 d + P / 4 + q 2^-13 (first and last chunk number)
    assert r.stdout.count(": 0 mismatches of 1024") == 4, r.stdout


def test_access_shape_probes_run(ah):
    """the floor probes of DESIGN.md 4 lesson 31 at small sizes (they are measurements, not checks: this only guards their launch
    geometry against faults): the FED family's stream probe, the Hessian's (every dilation) and the random-sector probe"""
    import ctypes as C
    ms, gbs = C.c_double(), C.c_double()
    for step in (1, 2, 3, 4):
        ah.check(ah.lib.hak_op_hess_probe(480, 270, 8, step, 2, C.byref(ms), C.byref(gbs)))
        assert ms.value > 0 and gbs.value > 10
    ah.check(ah.lib.hak_op_hess_probe(244, 97, 3, 3, 1, C.byref(ms), C.byref(gbs)))
    ah.check(ah.lib.hak_op_stream_probe(480, 270, 8, 2, 7, 2, C.byref(ms), C.byref(gbs)))
    assert gbs.value > 10
    ah.check(ah.lib.hak_op_gather_probe(1 << 28, 512, 16, 2, C.byref(ms)))
    assert ms.value > 0


def test_stream_ordering_probe(tmp_path):
    """what the library relies on, and what it must not rely on, when a caller's blocking runtime calls meet its non-blocking streams
    (tools/probes/memset_order_probe.hip): a blocking hipMemcpy / hipMemcpy2D from host memory IS complete when it returns (the
    reference's upload pattern, main.cpp:181-188, needs no extra synchronisation) -- while hipMemset returns before its fill has run and
    nothing orders that fill in front of a kernel on a non-blocking stream (round 5: the matcher's ticket array was cleared that way;
    1 launch in 600 saw the old contents)"""
    cc = _hipcc()
    if cc is None:
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "memset_probe")
    b = subprocess.run([cc, "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "tools", "probes", "memset_order_probe.hip")],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "non-blocking stream: 0 stale words in 10 x 128" in r.stdout, r.stdout       # the blocking copies
