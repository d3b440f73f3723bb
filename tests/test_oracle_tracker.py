"""CPU-only checks of the oracle's tracker restatement (parity unpinned by reference fixtures:
the reference has none for this path).  Independent evidence used instead:
  * the 8 Jacobian columns accumulated into b = J^T W r agree with a finite-difference derivative
    of the Huber energy w.r.t. the left-multiplied increment (double re-evaluation);
  * the DSO-native LM recovers the synthetic motion;
  * the pyramid rule of the generator equals the oracle's makeImages restatement bit-for-bit."""
import ctypes as C

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth


@pytest.fixture(scope="module")
def prob():
    return synth.tracker_problem(w=640, h=480, npts=2000, seed=2001)


def test_pyramid_rule_matches_oracle(oracle, prob):
    img = np.ascontiguousarray(prob["pyr_new"][0][..., 0])
    h, w = img.shape
    L = oracle.orc_pyramid_levels(w, h)
    assert L == prob["levels"] == synth.pyramid_levels(w, h)
    outs = [np.zeros(((h >> l), (w >> l), 3), np.float32) for l in range(L)]
    ptrs = (abi.c_float_p * L)(*[abi.fp(o) for o in outs])
    oracle.orc_make_images(abi.fp(img), w, h, L, ptrs)
    for l in range(L):
        assert np.array_equal(outs[l], prob["pyr_new"][l])


def test_pyramid_level_rule(oracle):
    assert oracle.orc_pyramid_levels(1241, 376) == 1          # odd width: no coarser level (globalCalib.cpp:52-58)
    assert oracle.orc_pyramid_levels(1232, 368) == 5
    assert oracle.orc_pyramid_levels(640, 480) == 4


def _energy_double(prob, lvl, T, aff, prm):
    """Huber energy of calcRes in double, fixed inlier set irrelevant: sum over in-bounds points."""
    pc = prob["pc"][lvl]
    img = prob["pyr_new"][lvl].astype(np.float64)
    h, w, _ = img.shape
    fx, fy, cx, cy = [float(prob[k][lvl]) for k in ("fx", "fy", "cx", "cy")]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    R, t = T
    x, y, idp, col = [pc[k].astype(np.float64) for k in ("u", "v", "idepth", "color")]
    P = (R @ np.linalg.inv(K) @ np.stack([x, y, np.ones_like(x)])) + t[:, None] * idp[None, :]
    u, v = P[0] / P[2], P[1] / P[2]
    Ku, Kv = fx * u + cx, fy * v + cy
    ok = (Ku > 2) & (Kv > 2) & (Ku < w - 3) & (Kv < h - 3) & (idp / P[2] > 0)
    Ku, Kv, col = Ku[ok], Kv[ok], col[ok]
    ix, iy = Ku.astype(int), Kv.astype(int)
    dx, dy = Ku - ix, Kv - iy
    I = (dx * dy * img[iy + 1, ix + 1, 0] + (dy - dx * dy) * img[iy + 1, ix, 0] + (dx - dx * dy) * img[iy, ix + 1, 0]
         + (1 - dx - dy + dx * dy) * img[iy, ix, 0])
    a = np.exp(aff[0] - 0.0)
    r = I - (a * col + (aff[1] - a * 0.0))
    ar = np.abs(r)
    e = np.where(ar < 9, r * r, 9 * (2 * ar - 9))       # hw*r^2*(2-hw)
    return e.sum(), len(r)


def _smooth_problem():
    """A smooth image (wavelength >= 40 px, no edges, no noise) and arbitrary template points: the
    Jacobian check needs no photo-consistency, only that the stored central-difference gradient
    equals the derivative of the bilinear interpolant to a few per cent."""
    w, h = 320, 240
    rs = np.random.RandomState(77)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    img = np.full((h, w), 120.0)
    for _ in range(12):
        lam = rs.uniform(40, 160)
        ang = rs.uniform(0, 2 * np.pi)
        img += rs.uniform(4, 10) * np.sin(2 * np.pi / lam * (np.cos(ang) * xs + np.sin(ang) * ys) + rs.uniform(0, 6.28))
    pyr = synth.make_pyramid(img.astype(np.float32), levels=2)
    n = 3000
    u = rs.randint(8, w // 2 - 8, n).astype(np.float32)
    v = rs.randint(8, h // 2 - 8, n).astype(np.float32)
    pc1 = dict(u=u, v=v, idepth=rs.uniform(0.02, 0.2, n).astype(np.float32),
               color=(pyr[1][v.astype(int), u.astype(int), 0] + rs.normal(0, 6, n)).astype(np.float32))
    cal = synth.kitti_calib(w, h)
    fxs, fys, cxs, cys = synth.level_intrinsics(cal["fx"], cal["fy"], cal["cx"], cal["cy"], 2)
    return dict(levels=2, pyr_ref=pyr, pyr_new=pyr, pc=[pc1, pc1], fx=fxs, fy=fys, cx=cxs, cy=cys)


def test_b_is_energy_gradient(oracle):
    prob = _smooth_problem()
    lvl = 1
    prm = helpers.track_params(prob)
    T0 = synth.se3_exp(np.array([0.015, -0.008, 0.3, 0.003, -0.005, 0.0015]))
    aff0 = (0.01, 1.0)
    ev = abi.TrackEval()
    oracle.orc_track_make_eval(C.byref(prm), lvl, C.byref(abi.SE3.from_Rt(*T0)), C.byref(abi.Aff(*aff0)), 10.0, C.byref(ev))
    H, b, res, nw, mask = helpers.oracle_eval(oracle, prob["pc"][lvl], prob["pyr_new"][lvl], ev)
    assert res[5] == 0 and (int(mask.sum()) + 3) // 4 * 4 == nw      # cutoff 200: nothing saturated
    n = nw
    # b is scaled by [ROT,ROT,ROT,TRANS,TRANS,TRANS,A,B] (the reference's swapped order, CoarseTracker.cpp:584-595)
    SC = np.array([1, 1, 1, 0.5, 0.5, 0.5, 10, 1000.0])
    g = np.zeros(8)
    for k in range(8):
        eps = 1e-6
        d = np.zeros(8)
        d[k] = eps
        Tp = synth.se3_mul(synth.se3_exp(d[:6]), T0)
        Tm = synth.se3_mul(synth.se3_exp(-d[:6]), T0)
        ep, _ = _energy_double(prob, lvl, Tp, (aff0[0] + d[6], aff0[1] + d[7]), prm)
        em, _ = _energy_double(prob, lvl, Tm, (aff0[0] - d[6], aff0[1] - d[7]), prm)
        g[k] = (ep - em) / (2 * eps) / 2.0 / n            # dE/dx / 2 = J^T w r ; /n as calcGSSSE :581-582
    # bilinear interpolation makes E piecewise smooth; the analytic J uses the interpolated image
    # gradient instead of the exact derivative of the interpolant, so agreement is approximate.
    assert np.allclose(b / SC, g, rtol=0.03, atol=2e-3 * np.abs(g).max())


def test_lm_recovers_motion(oracle, prob):
    prm = helpers.track_params(prob)
    T, aff, out = helpers.oracle_track(oracle, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0))
    assert out.good == 1
    R, t = T.Rt()
    Rt, tt = prob["refToNew_true"]
    assert np.abs(t - tt).max() < 5e-3
    assert np.abs(R - Rt).max() < 1e-3
    assert out.lastResiduals[0] < 8.0
    assert out.evaluations >= 8 and out.point_evals > 0


def test_abort_on_min_res(oracle, prob):
    prm = helpers.track_params(prob)
    for i in range(5):
        prm.minResForAbort[i] = 0.01      # every level is "worse than achieved" -> abort at the coarsest level
    T, aff, out = helpers.oracle_track(oracle, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0))
    assert out.good == 0
    assert np.isnan(out.lastResiduals[0]) and not np.isnan(out.lastResiduals[prob["levels"] - 1])
    assert np.array_equal(T.Rt()[0], np.eye(3))           # outputs untouched on abort (CoarseTracker.cpp:1032-1047)


def test_make_coarse_depth_oracle_equals_the_generator(oracle):
    """makeCoarseDepthL0 STEP1-splat .. STEP5 (CoarseTracker.cpp:352-534): the oracle's C++ restatement (the checker of the device
    kernels) and the numpy generator behind every synthetic tracking problem (sdso_amd/synth.py::make_pc) are two independent
    restatements of the same loops — identical pc_n, order and floats, including pixels hit by several points and weights != 1."""
    import pyoracle
    import synth
    prob = synth.tracker_problem(w=320, h=240, npts=500, seed=2107)
    u, v, idp = prob["points"]
    rs = np.random.RandomState(8)
    u = u.astype(np.int32).copy(); v = v.astype(np.int32).copy()
    for dst, src in ((10, 400), (11, 400), (12, 400), (20, 300), (450, 3)):
        u[dst], v[dst] = u[src], v[src]
    wgt = rs.uniform(0.2, 3.0, len(u)).astype(np.float32)
    idp = (idp * rs.uniform(0.9, 1.1, len(u))).astype(np.float32)
    a = synth.make_pc(u, v, idp, wgt, prob["pyr_ref"])
    b = pyoracle.make_coarse_depth(oracle, u, v, idp, wgt, prob["pyr_ref"])
    assert len(a) == len(b) == prob["levels"]
    for l in range(prob["levels"]):
        assert len(a[l]["u"]) == len(b[l]["u"]) > 0
        for k in ("u", "v", "idepth", "color"):
            assert np.array_equal(a[l][k], b[l][k]), (l, k)
