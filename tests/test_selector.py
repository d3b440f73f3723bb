"""PixelSelector::makeMaps (PixelSelector2.cpp:84-540): oracle sanity on CPU, identical maps on the GPU through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi
import synth


def _frame(w, h, seed):
    prob = synth.tracker_problem(w=w, h=h, npts=50, seed=seed)
    return prob["pyr_ref"]


def _oracle_select(oracle, pyr, density, rec, thf, pot):
    keep = [np.ascontiguousarray(pyr[l]) for l in range(3)]
    ptrs = (abi.c_float_p * 3)(*[abi.fp(a) for a in keep])
    h, w, _ = pyr[0].shape
    m = np.zeros((h, w), np.float32)
    p = C.c_int(pot)
    n = oracle.orc_pixel_select(ptrs, w, h, density, rec, thf, C.byref(p), abi.fp(m))
    return n, p.value, m


def test_oracle_selector_density_and_spread(oracle):
    pyr = _frame(640, 480, 7)
    n, pot, m = _oracle_select(oracle, pyr, 1500.0, 1, 1.0, 3)
    assert n == int((m != 0).sum()) and 0.7 * 1500 < n < 1.3 * 1500            # makeMaps steers the potential towards the wanted density
    assert set(np.unique(m)) <= {0.0, 1.0, 2.0, 4.0} and (m == 1).sum() > (m == 2).sum() > 0
    ys, xs = np.nonzero(m)
    assert xs.min() >= 4 and xs.max() < 640 - 5 and ys.min() >= 4 and ys.max() <= 480 - 4
    quad = [((xs < 320) & (ys < 240)).sum(), ((xs >= 320) & (ys < 240)).sum(), ((xs < 320) & (ys >= 240)).sum(), ((xs >= 320) & (ys >= 240)).sum()]
    assert min(quad) > 0.1 * n                                                   # selected all over the image, not in one corner
    g = np.sqrt(pyr[0][..., 1] ** 2 + pyr[0][..., 2] ** 2)
    assert np.median(g[m != 0]) > 2 * np.median(g)                               # and on high-gradient pixels


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(640, 480, 7, 1500.0, 1, 1.0, 3), (640, 480, 8, 600.0, 2, 2.0, 3), (1232, 368, 9, 2000.0, 1, 1.0, 3),
                                  (640, 480, 7, 30000.0, 2, 1.0, 5), (640, 480, 7, 100.0, 0, 1.0, 2)])
def test_gpu_selector_identical_map(gpu_ctx, oracle, case):
    w, h, seed, density, rec, thf, pot0 = case
    pyr = _frame(w, h, seed)
    no, po, mo = _oracle_select(oracle, pyr, density, rec, thf, pot0)
    gpu_ctx.upload_pyramid(91, pyr)
    mg = np.zeros((h, w), np.float32)
    pg = C.c_int(pot0); ng = C.c_int(0)
    gpu_ctx.check(gpu_ctx.L.sdso_pixel_select(gpu_ctx.h, 91, density, rec, thf, C.byref(pg), abi.fp(mg), C.byref(ng)))
    assert ng.value == no and pg.value == po
    assert np.array_equal(mg, mo)                                                # every pixel, every level label
    # the device-built pyramid (sdso_make_pyramid) carries the same absSquaredGrad channel
    gpu_ctx.check(gpu_ctx.L.sdso_make_pyramid(gpu_ctx.h, 92, w, h, abi.fp(np.ascontiguousarray(pyr[0][..., 0]))))
    mg2 = np.zeros((h, w), np.float32); pg2 = C.c_int(pot0)
    gpu_ctx.check(gpu_ctx.L.sdso_pixel_select(gpu_ctx.h, 92, density, rec, thf, C.byref(pg2), abi.fp(mg2), C.byref(ng)))
    assert np.array_equal(mg2, mo) and pg2.value == po


# ---- photometric calibration: gamma-weighted absSquaredGrad (setting_gammaWeightsPixelSelect == 1, HessianBlocks.cpp:194-198)
def _binv():
    """a smooth monotone inverse response G^-1 on [0, 255] (a gamma-2.2-like curve)"""
    x = np.arange(256, dtype=np.float64) / 255.0
    return (255.0 * x ** 2.2).astype(np.float32)


def test_gamma_table_from_inverse_response(oracle):
    """FullSystem::setGammaFunction (FullSystem.cpp:210-234): the library's host function, the oracle's and the numpy restatement
    agree bit for bit, and B inverts Binv."""
    L = abi.load()
    BInv = _binv()
    Bl, Bo = np.zeros(256, np.float32), np.zeros(256, np.float32)
    assert L.sdso_gamma_from_binv(abi.fp(BInv), abi.fp(Bl)) == 0
    oracle.orc_gamma_from_binv(abi.fp(BInv), abi.fp(Bo))
    Bn = synth.gamma_from_binv(BInv)
    assert np.array_equal(Bl, Bo) and np.array_equal(Bl, Bn)
    assert Bl[0] == 0 and Bl[255] == 255 and np.all(np.diff(Bl[1:255]) > 0)
    back = np.interp(Bl[5:250], np.arange(256), BInv)                     # Binv(B(i)) == i
    assert np.abs(back - np.arange(5, 250)).max() < 1e-3


def test_oracle_selector_gamma_weights(oracle):
    pyr = _frame(640, 480, 7)
    n0, p0, m0 = _oracle_select(oracle, pyr, 1500.0, 1, 1.0, 3)
    try:
        ident = np.arange(256, dtype=np.float32)
        oracle.orc_set_gamma(abi.fp(ident))                                # identity response: every weight is exactly 1
        n1, p1, m1 = _oracle_select(oracle, pyr, 1500.0, 1, 1.0, 3)
        assert (n1, p1) == (n0, p0) and np.array_equal(m1, m0)
        B = synth.gamma_from_binv(_binv())
        oracle.orc_set_gamma(abi.fp(B))
        n2, p2, m2 = _oracle_select(oracle, pyr, 1500.0, 1, 1.0, 3)
        assert not np.array_equal(m2, m0)                                  # dark regions (steep response) gain weight, bright ones lose it
        g = pyr[0][..., 0]
        assert g[m2 != 0].mean() < g[m0 != 0].mean()
    finally:
        oracle.orc_set_gamma(None)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(640, 480, 7, 1500.0, 1, 1.0, 3), (1232, 368, 9, 2000.0, 1, 1.0, 3)])
def test_gpu_gamma_weighted_abs_grad_and_selection(gpu_ctx, oracle, case):
    w, h, seed, density, rec, thf, pot0 = case
    pyr = _frame(w, h, seed)
    B = synth.gamma_from_binv(_binv())
    try:
        oracle.orc_set_gamma(abi.fp(B))
        gpu_ctx.check(gpu_ctx.L.sdso_set_gamma(gpu_ctx.h, abi.fp(B)))
        no, po, mo = _oracle_select(oracle, pyr, density, rec, thf, pot0)
        # makeImages on the device: the {I, dx, dy} channels are untouched, absSquaredGrad carries the squared response gradient
        gpu_ctx.check(gpu_ctx.L.sdso_make_pyramid(gpu_ctx.h, 93, w, h, abi.fp(np.ascontiguousarray(pyr[0][..., 0]))))
        for l in range(3):
            hl, wl = h >> l, w >> l
            out3, a = np.zeros((hl, wl, 3), np.float32), np.zeros((hl, wl), np.float32)
            gpu_ctx.check(gpu_ctx.L.sdso_download_pyramid_level(gpu_ctx.h, 93, l, abi.fp(out3)))
            gpu_ctx.check(gpu_ctx.L.sdso_download_abs_grad(gpu_ctx.h, 93, l, abi.fp(a)))
            assert np.array_equal(out3, pyr[l])
            ref = synth.abs_squared_grad(pyr[l], B)
            assert np.array_equal(a[1:-1], ref[1:-1])                      # rows 1 .. h-2: the rows makeImages writes
            assert not np.array_equal(a[1:-1], synth.abs_squared_grad(pyr[l])[1:-1])
        for slot, build in ((93, None), (94, "upload")):                   # device-built and uploaded pyramids select the same pixels
            if build:
                gpu_ctx.upload_pyramid(slot, pyr)
            mg = np.zeros((h, w), np.float32); pg = C.c_int(pot0); ng = C.c_int(0)
            gpu_ctx.check(gpu_ctx.L.sdso_pixel_select(gpu_ctx.h, slot, density, rec, thf, C.byref(pg), abi.fp(mg), C.byref(ng)))
            assert ng.value == no and pg.value == po and np.array_equal(mg, mo)
    finally:
        oracle.orc_set_gamma(None)
        gpu_ctx.check(gpu_ctx.L.sdso_set_gamma(gpu_ctx.h, None))
    # back to the identity response: the round-1 behaviour, bit for bit
    gpu_ctx.check(gpu_ctx.L.sdso_make_pyramid(gpu_ctx.h, 93, w, h, abi.fp(np.ascontiguousarray(pyr[0][..., 0]))))
    a = np.zeros((h, w), np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_download_abs_grad(gpu_ctx.h, 93, 0, abi.fp(a)))
    assert np.array_equal(a[1:-1], synth.abs_squared_grad(pyr[0])[1:-1])
