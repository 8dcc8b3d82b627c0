"""GPU parity, per kernel: every HIP stage against the oracle's stage function on small odd-sized
planes, bit-exact (uint32 views), called through the C ABI (hak_op_*)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(211, 173), (128, 96), (83, 81), (300, 17 * 5), (517, 130)]


def plane(rng, w, h, lo=0.0, hi=1.0):
    p = (w + 63) // 64 * 64
    a = np.zeros((h, p), np.float32)
    a[:, :w] = rng.uniform(lo, hi, (h, w)).astype(np.float32)
    return a, p


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def eq(gpu_t, ref, w):
    g = gpu_t.cpu().numpy()
    return np.array_equal(g[:, :w].view(np.uint32), ref[:, :w].view(np.uint32))


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("var,R", [(1.0, 2), (2.56, 4), (1.3, 3), (3.2, 5)])
def test_lowpass(ah, okz, torch, w, h, var, R):
    rng = np.random.default_rng(w * 7 + R)
    a, p = plane(rng, w, h)
    d_src, d_dst = dev(torch, a), torch.zeros((h, p), dtype=torch.float32, device="cuda")
    ah.check(ah.lib.hak_op_lowpass(d_src.data_ptr(), d_dst.data_ptr(), w, h, p, var, R))
    assert eq(d_dst, okz.lowpass(a, w, var, R), w)


@pytest.mark.parametrize("sw,sh", [(211, 173), (256, 192), (166, 135), (480, 270)])
def test_down_smooth(ah, okz, torch, sw, sh):
    rng = np.random.default_rng(sw)
    a, sp = plane(rng, sw, sh)
    dw, dh = sw >> 1, sh >> 1
    dp = (dw + 63) // 64 * 64
    d_dst = torch.zeros((dh, dp), dtype=torch.float32, device="cuda")
    d_sm = torch.zeros((dh, dp), dtype=torch.float32, device="cuda")
    d_a = dev(torch, a)
    ah.check(ah.lib.hak_op_down_smooth(d_a.data_ptr(), d_dst.data_ptr(), d_sm.data_ptr(), sw, sh, sp, dw, dh, dp))
    o_dst, o_sm = okz.down_smooth(a, sw, dw, dh, dp)
    assert eq(d_dst, o_dst, dw) and eq(d_sm, o_sm, dw)


@pytest.mark.parametrize("w,h", SIZES)
def test_kcontrast(ah, okz, torch, w, h):
    rng = np.random.default_rng(w + 1)
    a, p = plane(rng, w, h)
    sm = okz.lowpass(a, w, 1.0, 2)
    kc, hmax = C.c_float(), C.c_float()
    hist = np.zeros(300, np.int32)
    d_sm = dev(torch, sm)
    ah.check(ah.lib.hak_op_kcontrast(d_sm.data_ptr(), w, h, p, 0.7, C.byref(kc), C.byref(hmax),
                                     hist.ctypes.data_as(C.POINTER(C.c_int))))
    okc, ohmax, ohist = okz.kcontrast(okz.scharr_grad(sm, w), w, 0.7)
    assert np.float32(hmax.value) == ohmax
    extra = ((w + 31) // 32 * 32 - w) * h + ((h + 15) // 16 * 16 - h) * w          # akazed.cu:909: threads outside the image count zeros
    assert np.array_equal(hist, ohist) and hist.sum() == w * h + extra
    assert np.float32(kc.value).view(np.uint32) == okc.view(np.uint32)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("diff", [0, 1, 2, 3])
def test_flow(ah, okz, torch, w, h, diff):
    rng = np.random.default_rng(w + 2)
    a, p = plane(rng, w, h)
    d_dst = torch.zeros((h, p), dtype=torch.float32, device="cuda")
    d_a = dev(torch, a)
    ah.check(ah.lib.hak_op_flow(d_a.data_ptr(), d_dst.data_ptr(), w, h, p, diff, 0.37))
    assert eq(d_dst, okz.flow(a, w, diff, 0.37), w)


@pytest.mark.parametrize("w,h", SIZES + [(1024, 40), (257, 90), (255, 90), (4, 3 * 30)])
@pytest.mark.parametrize("taus", [[0.07], [0.1, 0.68, 0.08, 0.19], [5.0, 41.0, 0.3]])
def test_fed_steps(ah, okz, torch, w, h, taus):
    rng = np.random.default_rng(w + 3)
    a, p = plane(rng, w, h)
    g, _ = plane(rng, w, h, 0.01, 1.0)
    d_dst = torch.zeros((h, p), dtype=torch.float32, device="cuda")
    d_tmp = torch.zeros((h, p), dtype=torch.float32, device="cuda")
    t = np.array(taus, np.float32)
    d_a, d_g = dev(torch, a), dev(torch, g)          # keep both alive: a freed temporary's memory is reused
    ah.check(ah.lib.hak_op_nld_steps(d_a.data_ptr(), d_g.data_ptr(), d_dst.data_ptr(),
                                     d_tmp.data_ptr(), w, h, p, t.ctypes.data_as(C.POINTER(C.c_float)), len(taus)))
    assert eq(d_dst, okz.nld_steps(a, g, w, taus), w)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("step", [2, 3, 4, 6])
def test_hessian(ah, okz, torch, w, h, step):
    rng = np.random.default_rng(w + 4)
    a, p = plane(rng, w, h)
    outs = [torch.zeros((h, p), dtype=torch.float32, device="cuda") for _ in range(3)]
    d_a = dev(torch, a)
    ah.check(ah.lib.hak_op_hessian(d_a.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                   w, h, p, step))
    lx, ly, det = okz.hessian(a, w, step)
    assert eq(outs[0], lx, w) and eq(outs[1], ly, w) and eq(outs[2], det, w)


@pytest.mark.parametrize("w,h", SIZES + [(64, 32), (65, 33), (1000, 70)])
@pytest.mark.parametrize("diff", [0, 1, 2, 3])
def test_smooth_flow_fused(ah, okz, torch, w, h, diff):
    """the fused sigma=1 low-pass + conductivity product kernel == oracle lowpass followed by oracle flow"""
    rng = np.random.default_rng(w + 5)
    a, p = plane(rng, w, h)
    d_a = dev(torch, a)
    d_sm = torch.zeros((h, p), dtype=torch.float32, device="cuda")
    d_g = torch.zeros((h, p), dtype=torch.float32, device="cuda")
    ah.check(ah.lib.hak_op_smooth_flow(d_a.data_ptr(), d_sm.data_ptr(), d_g.data_ptr(), w, h, p, diff, 0.41))
    sm = okz.lowpass(a, w, 1.0, 2)
    assert eq(d_sm, sm, w) and eq(d_g, okz.flow(sm, w, diff, 0.41), w)


def test_fast_reciprocal_is_the_ieee_quotient(ah):
    """k_fed_sf computes g = 1 / (1 + dif2) with v_rcp_f32 + one Newton step instead of the 11-instruction IEEE division;
    that is only legitimate because the two agree on every float of [1, 2^64): checked exhaustively (2^29 values)"""
    n = C.c_ulonglong(123)
    ah.check(ah.lib.hak_op_rcp_check(0x3F800000, 0x5F800000, C.byref(n)))
    assert n.value == 0
    ah.check(ah.lib.hak_op_rcp_check(0x5F800000, 0x7F800000, C.byref(n)))       # beyond 2^64 the sequence is NOT exact ...
    assert n.value > 0                                                           # ... which is why the kernel range-checks
