"""CPU checks of the oracle's post-state of FullSystem::optimize (orc_ba_get_post_state) and of the order sensitivity of the per-point
sums — independent of the product: every quantity is recomputed here in numpy from the reference's formulas.

  FullSystem::linearizeAll_Reductor, fixLinearization branch      src/FullSystem/FullSystemOptimize.cpp:62-84
  AccumulatedSCHessianSSE::addPoint, per-point part               src/OptimizationBackend/AccumulatedSCHessian.cpp:34-60
  EnergyFunctional::dropResidual (swap-with-last)                 src/OptimizationBackend/EnergyFunctional.cpp:524-533"""
import ctypes as C

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth


@pytest.fixture(scope="module")
def win():
    return synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001)


def _oracle_post(oracle, w, its=6):
    nf, npts, nr = w["nf"], w["np"], w["nr"]
    W, keep = abi.make_ba_window(w, frame_slots=list(range(nf)), dI_list=[p[0] for p in w["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    oo = abi.BAOptResult()
    oracle.orc_ba_optimize(h, its, None, None, None, C.byref(oo))
    P, d = abi.make_post_state(nf, npts, nr)
    oracle.orc_ba_get_post_state(h, C.byref(P))
    tab = np.zeros(nf * nf * 27, np.float32)
    oracle.orc_ba_get_tables(h, abi.fp(tab), None, None, None)
    oracle.orc_ba_destroy(h)
    return P, d, tab.reshape(nf, nf, 27), oo


def test_oracle_post_state_follows_the_reference_formulas(oracle, win):
    w = dict(win)
    nf, npts, nr = w["nf"], w["np"], w["nr"]
    rs = np.random.RandomState(3)
    w["numGoodResiduals"] = rs.randint(0, 5, npts).astype(np.int32)
    w["maxRelBaseline"] = (rs.uniform(0, 0.2, npts) * (rs.rand(npts) < 0.5)).astype(np.float32)
    w["res_isNew"] = (rs.rand(nr) < 0.7).astype(np.uint8)
    P, d, pre, oo = _oracle_post(oracle, w)
    act = d["isActiveAndIsGoodNEW"].astype(bool)
    # toRemove = every residual of activeResiduals that is not active after applyRes (:80-84); state IN <=> active
    assert np.array_equal(d["toRemove"].astype(bool), ~act) and P.n_toRemove == int((~act).sum())
    assert np.array_equal(d["state_state"] == 0, act)
    assert P.result.iterations == oo.iterations and P.resInA == oo.resInA
    # numGoodResiduals: + 1 per active new residual (:67-76)
    inc = np.bincount(w["res_point"], weights=(act & (w["res_isNew"] == 1)).astype(np.float64), minlength=npts).astype(np.int32)
    assert np.array_equal(d["numGoodResiduals"], w["numGoodResiduals"] + inc)
    # HdiF = 1.0 / idepth_hessian in double, rounded to float (AccumulatedSCHessian.cpp:56-58); both 0 without an active residual (:44-48)
    nz = d["idepth_hessian"] != 0
    assert np.array_equal(d["HdiF"][nz], (1.0 / d["idepth_hessian"][nz].astype(np.float64)).astype(np.float32))
    assert not d["HdiF"][~nz].any() and not d["bdSumF"][~nz].any()
    # maxRelBaseline (:64-74) from the final precalc tables and idepths, in float like the reference
    host = w["host"][w["res_point"]]
    f32 = np.float32
    expect = w["maxRelBaseline"].copy()
    # (a point that had no active residual in one of the loop's solves was reset to 0 by addPoint; take those from the oracle's own flag)
    for i in np.nonzero(act & (w["res_isNew"] == 1))[0]:
        p = w["res_point"][i]
        T = pre[host[i], w["res_target"][i]]
        KRKi, Kt = T[:9].reshape(3, 3), T[9:12]
        u, v, idp = f32(w["u"][p]), f32(w["v"][p]), f32(d["idepth"][p])
        pinf = np.array([f32(f32(f32(KRKi[r, 0] * u) + f32(KRKi[r, 1] * v)) + f32(KRKi[r, 2] * f32(1))) for r in range(3)], f32)
        pp = np.array([f32(pinf[r] + f32(Kt[r] * idp)) for r in range(3)], f32)
        dx = f32(f32(pinf[0] / pinf[2]) - f32(pp[0] / pp[2])); dy = f32(f32(pinf[1] / pinf[2]) - f32(pp[1] / pp[2]))
        rel = f32(0.01 * float(np.sqrt(f32(f32(dx * dx) + f32(dy * dy)))))
        expect[p] = max(expect[p], rel)
    same = d["maxRelBaseline"] == expect
    reset = (~same) & (d["maxRelBaseline"] < w["maxRelBaseline"])            # history wiped by a solve that saw no active residual
    assert (same | reset).all() and same.sum() > 0.9 * npts
    # centerProjectedTo of the active residuals = projection of the point at idepth_zero with PRE_RTll_0 / PRE_tTll_0 (Residuals.cpp:117-131)
    i = int(np.nonzero(act)[0][0]); p = w["res_point"][i]
    T = pre[host[i], w["res_target"][i]]
    R0, t0 = T[12:21].reshape(3, 3).astype(np.float64), T[21:24].astype(np.float64)
    fx, fy, cx, cy = [float(x) for x in P.calib_value_scaled[:]]
    X = R0 @ np.array([(w["u"][p] - cx) / fx, (w["v"][p] - cy) / fy, 1.0]) + t0 * float(d["idepth"][p])
    assert abs(X[0] / X[2] * fx + cx - d["centerProjectedTo"][i, 0]) < 1e-2 and abs(d["idepth"][p] / X[2] - d["centerProjectedTo"][i, 2]) < 1e-5
    assert not d["centerProjectedTo"][~act].any()


def test_per_point_sums_depend_on_residualsall_order(oracle, win):
    """Sorting a dropResidual-permuted window back into target order changes some per-point float sums: the GPU test that asks for
    bit-identical Hdd / bd / Hcd on such windows (tests/test_ba_post_state_gpu.py) is sensitive to the order."""
    wd, kept = helpers.drop_residuals(win, seed=11, drop_frac=0.25)
    nf, npts = wd["nf"], wd["np"]
    srt = np.argsort(wd["res_point"].astype(np.int64) * 16 + wd["res_target"], kind="stable")
    assert not np.array_equal(srt, np.arange(wd["nr"]))
    ws = dict(wd)
    for k in ("res_point", "res_target", "res_state"):
        ws[k] = np.ascontiguousarray(wd[k][srt])
    outs = []
    for w in (wd, ws):
        W, keep = abi.make_ba_window(w, frame_slots=list(range(nf)), dI_list=[p[0] for p in w["pyrs"]])
        h = oracle.orc_ba_create(C.byref(W))
        oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
        po = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
        oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in po])
        oracle.orc_ba_destroy(h)
        outs.append(po)
    assert sum(int((a != b).sum()) for a, b in zip(outs[0], outs[1])) > 0
    assert all(np.allclose(a, b, rtol=1e-5, atol=1e-12) for a, b in zip(outs[0], outs[1]))


def test_drop_helper_matches_swap_with_last():
    lists = [[0, 1, 2, 3], [4, 5], [6]]
    assert helpers.apply_drops(lists, [1, 4, 0]) == [[2, 3], [5], [6]]      # [0,1,2,3] -drop 1-> [0,3,2] -drop 0-> [2,3]
