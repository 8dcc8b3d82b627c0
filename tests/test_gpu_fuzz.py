"""A short seeded slice of the randomised parity run (tests/fuzz_parity.py: random extents, batch sizes, parameters and kernel
selections, both pipelines and the batched matcher against the oracle).  HAK_FUZZ_CASES / HAK_FUZZ_SEED widen it; the long runs are
committed under profiles/ (r05_fuzz_parity.txt)."""
import io
import os

import pytest

import fuzz_match
import fuzz_parity

pytestmark = pytest.mark.gpu


def test_fuzz_slice_vs_oracle():
    buf = io.StringIO()
    failed = fuzz_parity.run(int(os.environ.get("HAK_FUZZ_CASES", "16")), int(os.environ.get("HAK_FUZZ_SEED", "5")), verbose=False, out=buf)
    assert not failed, buf.getvalue()


def test_cases_are_a_pure_function_of_seed_and_index():
    a, b = fuzz_parity.draw_case(5, 7), fuzz_parity.draw_case(5, 7)
    assert a == b and fuzz_parity.draw_case(5, 8) != a and fuzz_parity.draw_case(6, 7) != a


@pytest.mark.parametrize("seed,index,big", [(5, 1504, 0), (21, 24, 0), (5, 84, 1), (9, 36, 2)], ids=lambda v: str(v))
def test_cases_the_long_runs_found(seed, index, big):
    """round 5's long runs: the integer pipeline's FED cycle blows a coarse level up (31 steps, tau up to 50, 16-bit truncations), the wrapped
    sum of squares of gFlowNaive turns negative and the Charbonnier / PM_G1 conductivity is sqrt(negative) / exp(huge): the device cast gives
    0 / INT_MAX where the oracle's C cast gave INT_MIN (akazed.cu:3427-3443; the oracle was wrong, oracle/akaze_oracle_fast.c f2i_sat).
    (9, 36, huge): a 27 Mpx frame with six octaves -- gRefine leaves NaN coordinates behind at octave 5 (a determinant of inf - inf), and
    the orientation's `(int)(x + 0.5f) >> o` of a NaN is 0 on the device (akazed.cu:1665-1736), INT_MIN in the old oracle"""
    import torch
    import akaze_hip as ah
    from akaze_hip import synth
    import okz
    okz.build()
    c = fuzz_parity.draw_case(seed, index, big)
    assert big == 2 or (c["fast"] and c["kw"]["diffusivity"] in (0, 3))
    fails, npts, _ = fuzz_parity.run_case(ah, okz, torch, synth, fuzz_parity._mg(), c)
    assert not fails and npts > 1000, fails


def test_match_fuzz_slice_vs_oracle():
    """tests/fuzz_match.py: random set sizes and descriptor populations (planted copies, exact duplicates, all-zero / all-one rows)
    through hak_match and hak_match_knn2, matrix-core (both wave shapes) and VALU kernels, with and without a context"""
    buf = io.StringIO()
    failed = fuzz_match.run(int(os.environ.get("HAK_FUZZ_MATCH_CASES", "12")), int(os.environ.get("HAK_FUZZ_SEED", "5")), verbose=False, out=buf)
    assert not failed, buf.getvalue()
