"""PixelSelector::makeMaps (PixelSelector2.cpp:84-540): oracle sanity on CPU, identical maps on the GPU through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi, synth


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
