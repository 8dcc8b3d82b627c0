"""SURVEY.md 5 "sanitizers on the CPU build": the oracle and the C++ drop-in layer under AddressSanitizer + UBSan.
(GPU sanitizers are not available on this pool; the HIP kernels are covered by the parity tests instead.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc + make")


def _make(directory):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


def test_oracle_under_asan_ubsan():
    out = _make("oracle")
    assert "asan_main: oracle clean" in out
    assert out.count(" desc ") == 10                      # float + FAST + both matchers on ten scenes / parameter sets (two of them blow the integer pipeline up)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="needs the HIP headers")
def test_cpp_host_layer_under_asan_ubsan():
    """host/akaze.cpp against a host-memory stub of the C ABI: registry of pinned buffers, context re-creation, per-call clamp"""
    assert "asan_main: host layer clean" in _make(os.path.join("cuda-akaze_amd", "host"))
