"""The complete post-state of FullSystem::optimize through the C-ABI (sdso_ba_get_post_state) and the per-point sums on windows whose
residual lists went through EnergyFunctional::dropResidual (round-3 verdict, "Missing" #1 and #2).

Reference: FullSystem::linearizeAll(true) at the end of optimize (src/FullSystem/FullSystemOptimize.cpp:52-87, :142-203, :997-1041) sets
PointHessian::maxRelBaseline / numGoodResiduals, fills toRemove, updates lastResiduals[].second and drops the residuals that did not
survive; the last solveSystemF's AccumulatedSCHessianSSE::addPoint left EFPoint::HdiF / bdSumF and PointHessian::idepth_hessian
(src/OptimizationBackend/AccumulatedSCHessian.cpp:34-60).  Consumers: CoarseTracker::makeCoarseDepthL0 (CoarseTracker.cpp:295-350),
FullSystem::flagPointsForRemoval (FullSystem.cpp:997-1040).

Bars: flags / counts / states identical except where a residual's energy sits on its threshold (the same allowance as
tests/test_ba_gpu.py::test_optimize_full_gn_loop); floats within the loop's bars (the states they are computed from agree to 1e-5).
Per-point sums on permuted windows: bit-exact (one accumulate at the uploaded state, where every input is bit-identical)."""
import ctypes as C

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def win_c3():
    return synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)      # configs[2]: 8 KF x 2000 points


@pytest.fixture(scope="module")
def win_dropped(win_c3):
    w2, kept = helpers.drop_residuals(win_c3, seed=11, drop_frac=0.25)
    starts = np.searchsorted(w2["res_point"], np.arange(w2["np"]), side="left")
    ends = np.searchsorted(w2["res_point"], np.arange(w2["np"]), side="right")
    unsorted = sum(1 for p in range(w2["np"]) if np.any(np.diff(w2["res_target"][starts[p]:ends[p]]) < 0))
    assert unsorted > 0.3 * w2["np"]                                          # most points are no longer in target order
    return w2


def _upload_both(ctx, oracle, win, wid=3, slot0=40):
    for f in range(win["nf"]):
        ctx.upload_pyramid(slot0 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[slot0 + f for f in range(win["nf"])], dI_list=[p[0] for p in win["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
    return W, keep, oracle.orc_ba_create(C.byref(W))


def _point_terms(ctx, oracle, h, wid, npts):
    po = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    pg = [np.zeros_like(a) for a in po]
    oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in po])
    ctx.check(ctx.L.sdso_ba_get_point_terms(ctx.h, wid, *[abi.fp(a) for a in pg]))
    return po, pg


def test_per_point_sums_in_residualsall_order(gpu_ctx, oracle, win_dropped):
    """Hdd / bd / Hcd, HdiF, bdSumF and the back-substituted point steps are bit-identical to the CPU path on a window whose
    residualsAll lists are permuted by dropResidual's swap-with-last (EnergyFunctional.cpp:524-533)."""
    ctx, win = gpu_ctx, win_dropped
    W, keep, h = _upload_both(ctx, oracle, win)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h)
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 3, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 3))
    xo = np.zeros(n)
    oracle.orc_ba_solve(h, 0, 0.1, abi.dp(xo), None, None, None, None)       # accumulate + stitch + solve + resubstitute
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 3))
    po, pg = _point_terms(ctx, oracle, h, 3, npts)
    for name, a, b in zip(("HdiF", "bdSumF", "Hdd_accAF", "bd_accAF", "Hcd_accAF"), po, pg):
        assert np.array_equal(a, b), name
    assert (po[0] != 0).sum() > 0.8 * npts
    # the same sums in target order would NOT be identical: the test is sensitive to the order
    so = np.zeros(npts, np.float32)
    oracle.orc_ba_get_point_steps(h, abi.fp(so))
    # back-substitution with the oracle's x: every input of resubstituteFPt is then bit-identical, so the steps must be
    fs, cs = np.zeros(nf * 8), np.zeros(4)
    ctx.check(ctx.L.sdso_ba_resubstitute(ctx.h, 3, abi.dp(xo), abi.dp(fs), abi.dp(cs)))
    sg = np.zeros(npts, np.float32)
    ctx.check(ctx.L.sdso_ba_get_point_steps(ctx.h, 3, abi.fp(sg)))
    assert np.array_equal(so, sg)
    assert np.array_equal(fs, -xo[4:]) and np.array_equal(cs, -xo[:4])
    oracle.orc_ba_destroy(h)
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))


def _post_both(ctx, oracle, win, wid, its, h):
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    oo, og = abi.BAOptResult(), abi.BAOptResult()
    oracle.orc_ba_optimize(h, its, None, None, None, C.byref(oo))
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, wid, its, None, None, None, C.byref(og)))
    Po, do = abi.make_post_state(nf, npts, nr)
    Pg, dg = abi.make_post_state(nf, npts, nr)
    oracle.orc_ba_get_post_state(h, C.byref(Po))
    ctx.check(ctx.L.sdso_ba_get_post_state(ctx.h, wid, C.byref(Pg)))
    return Po, do, Pg, dg


def check_post_state(win, Po, do, Pg, dg):
    """Every field of the two post-states; returns the indices of residuals whose final state differs (threshold flips)."""
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    assert Pg.result.iterations == Po.result.iterations
    flips = np.nonzero(dg["state_state"] != do["state_state"])[0]
    assert len(flips) <= max(2, nr // 2000)                  # IN / OUTLIER flips only where an energy sits on its threshold
    same = np.ones(nr, bool); same[flips] = False
    pt_same = np.ones(npts, bool); pt_same[win["res_point"][flips]] = False
    # ---- flags / counts: identical
    assert np.array_equal(dg["isActiveAndIsGoodNEW"][same], do["isActiveAndIsGoodNEW"][same])
    assert np.array_equal(dg["toRemove"][same], do["toRemove"][same])
    assert np.array_equal(dg["toRemove"], 1 - dg["isActiveAndIsGoodNEW"])           # every residual of a fresh window is in activeResiduals
    assert np.array_equal(dg["numGoodResiduals"][pt_same], do["numGoodResiduals"][pt_same])
    assert abs(Pg.n_toRemove - Po.n_toRemove) <= len(flips) and Pg.n_toRemove == int(dg["toRemove"].sum())
    assert (Pg.resInA == Po.resInA or len(flips) > 0) and Pg.resInL == Po.resInL == 0 and Pg.resInM == Po.resInM
    assert do["toRemove"].sum() > 0 and do["isActiveAndIsGoodNEW"].sum() > 0.4 * nr  # not vacuous
    # ---- frames / calibration (the loop's bars: tests/test_ba_gpu.py::test_optimize_full_gn_loop)
    assert np.abs(dg["state"] - do["state"]).max() <= 1e-4
    assert np.abs(dg["state_zero"] - do["state_zero"]).max() <= 1e-4
    assert np.array_equal(dg["state_zero"][:nf - 1], do["state_zero"][:nf - 1])     # only the newest frame's changes (setEvalPT, :1000-1003)
    assert np.array_equal(dg["evalPT"][:nf - 1], do["evalPT"][:nf - 1])
    assert np.abs(dg["evalPT"] - do["evalPT"]).max() <= 1e-4 and np.array_equal(dg["evalPT"][nf - 1], dg["PRE_worldToCam"][nf - 1])
    assert np.abs(dg["PRE_worldToCam"] - do["PRE_worldToCam"]).max() <= 1e-4
    assert np.all(dg["state_zero"][nf - 1, :6] == 0) and np.array_equal(dg["state_zero"][nf - 1, 6:8], dg["state"][nf - 1, 6:8])
    assert np.abs(dg["frameEnergyTH"] - do["frameEnergyTH"]).max() <= 1e-3 * np.abs(do["frameEnergyTH"]).max()
    assert np.abs(np.array(Pg.calib_value[:]) - np.array(Po.calib_value[:])).max() <= 1e-5
    assert np.abs(np.array(Pg.calib_value_scaled[:]) - np.array(Po.calib_value_scaled[:])).max() <= 1e-3
    # ---- lastX / lastHS / lastbS / steps of the last solve, whitened like the solver tests
    d = np.sqrt(np.abs(np.diag(do["lastHS"]))) + 1e-30
    assert np.abs((dg["lastHS"] - do["lastHS"]) / np.outer(d, d)).max() <= 5e-4
    assert np.abs((dg["lastbS"] - do["lastbS"]) / d).max() <= 2e-3 * max(1.0, np.abs(do["lastbS"] / d).max())   # (b of the LAST solve: its states differ by ~1e-5; measured 5.2e-4)
    # lastX of the LAST solve is a step at the noise floor of the float accumulators (|x| ~ 1e-4 .. 1e-3 in state units): absolute bar, like the states
    assert np.abs(dg["lastX"] - do["lastX"]).max() <= 2e-4
    assert np.array_equal(dg["frame_step"][:, :8].ravel(), -dg["lastX"][4:]) and np.all(dg["frame_step"][:, 8:] == 0)
    assert np.array_equal(np.array(Pg.calib_step[:]), -dg["lastX"][:4])
    # ---- points
    assert np.abs(dg["idepth"] - do["idepth"]).max() <= 5e-5
    for k, rel in (("HdiF", 2e-3), ("idepth_hessian", 2e-3), ("maxRelBaseline", 2e-3)):
        a, b = dg[k][pt_same], do[k][pt_same]
        assert np.array_equal(a == 0, b == 0), k                                  # "no active residual" is a flag, not a float
        assert np.abs(a - b).max() <= rel * np.abs(b).max(), (k, np.abs(a - b).max(), np.abs(b).max())
    nz = dg["idepth_hessian"] != 0
    assert np.array_equal(dg["HdiF"][nz], (1.0 / dg["idepth_hessian"][nz].astype(np.float64)).astype(np.float32))   # HdiF = 1.0 / H (:58)
    assert np.abs(dg["bdSumF"][pt_same] - do["bdSumF"][pt_same]).max() <= 5e-3 * max(1.0, np.abs(do["bdSumF"]).max())
    assert np.abs(dg["step"][pt_same] - do["step"][pt_same]).max() <= 2e-3 * max(np.abs(do["step"]).max(), 1e-6)
    # ---- residuals
    act = (do["isActiveAndIsGoodNEW"] == 1) & same
    assert np.abs(dg["state_energy"][same] - do["state_energy"][same]).max() <= 2e-2 * np.abs(do["state_energy"]).max()
    assert np.abs(dg["centerProjectedTo"][act][:, :2] - do["centerProjectedTo"][act][:, :2]).max() <= 2e-2          # pixels
    assert np.abs(dg["centerProjectedTo"][act][:, 2] - do["centerProjectedTo"][act][:, 2]).max() <= 1e-4             # idepth in the target
    assert np.abs(dg["projectedTo"][act] - do["projectedTo"][act]).max() <= 2e-2
    assert not dg["centerProjectedTo"][dg["isActiveAndIsGoodNEW"] == 0].any() and not do["centerProjectedTo"][do["isActiveAndIsGoodNEW"] == 0].any()
    return flips


@pytest.fixture(scope="module")
def win_small_dropped():
    """A younger window (5 keyframes at 640x480, affine parameters free, points started 10 % off) whose lists were dropped from twice"""
    w = synth.ba_window(w=640, h=480, nf=5, pts_per_kf=150, seed=5113, idepth_noise=0.1)
    w, _ = helpers.drop_residuals(w, seed=3, drop_frac=0.2)
    w, _ = helpers.drop_residuals(w, seed=4, drop_frac=0.1)
    return w


@pytest.mark.parametrize("which", ["fresh", "dropped_with_history", "small_dropped_twice"])
def test_post_state_of_optimize_matches_oracle(gpu_ctx, oracle, win_c3, win_dropped, win_small_dropped, which):
    ctx = gpu_ctx
    win = dict(win_c3 if which == "fresh" else win_small_dropped if which == "small_dropped_twice" else win_dropped)
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    if which != "fresh":                        # points with a history: earlier optimize calls left counts / baselines, some residuals are old
        rs = np.random.RandomState(5)
        win["numGoodResiduals"] = rs.randint(0, 9, npts).astype(np.int32)
        win["maxRelBaseline"] = (rs.uniform(0, 0.4, npts) * (rs.rand(npts) < 0.7)).astype(np.float32)
        win["res_isNew"] = (rs.rand(nr) < 0.8).astype(np.uint8)
    W, keep, h = _upload_both(ctx, oracle, win)
    Po, do, Pg, dg = _post_both(ctx, oracle, win, 3, 6, h)
    check_post_state(win, Po, do, Pg, dg)
    if which != "fresh":
        # numGoodResiduals = history + the active new residuals of this call, exactly (FullSystemOptimize.cpp:67-76)
        inc = np.bincount(win["res_point"], weights=(dg["isActiveAndIsGoodNEW"] & win["res_isNew"]).astype(np.float64), minlength=npts).astype(np.int32)
        assert np.array_equal(dg["numGoodResiduals"], win["numGoodResiduals"] + inc)
    # a second call returns the same thing and does not count twice
    Pg2, dg2 = abi.make_post_state(nf, npts, nr)
    ctx.check(ctx.L.sdso_ba_get_post_state(ctx.h, 3, C.byref(Pg2)))
    for k in dg:
        assert np.array_equal(dg[k], dg2[k], equal_nan=True), k
    oracle.orc_ba_destroy(h)
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))


def test_post_state_needs_an_optimize(gpu_ctx, win_c3):
    ctx, win = gpu_ctx, win_c3
    for f in range(win["nf"]):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(win["nf"])])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    P, d = abi.make_post_state(win["nf"], win["np"], win["nr"])
    assert ctx.L.sdso_ba_get_post_state(ctx.h, 3, C.byref(P)) != 0
    assert b"finished" in ctx.L.sdso_last_error(ctx.h)
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))


def test_post_state_after_batch_optimize(gpu_ctx, oracle, win_c3, win_dropped):
    """The same through sdso_ba_batch_optimize (the fused kernels): post-state per member; lastHS only on request."""
    ctx = gpu_ctx
    wins = [win_c3, win_dropped]
    hs, ids = [], []
    for k, win in enumerate(wins):
        W, keep, h = _upload_both(ctx, oracle, win, wid=20 + k, slot0=300 + 10 * k)
        hs.append((W, keep, h)); ids.append(20 + k)
    ids = np.array(ids, np.int32)
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, len(ids), abi.ip(ids)))
    res = (abi.BAOptResult * len(ids))()
    ctx.check(ctx.L.sdso_ba_batch_optimize(ctx.h, 6, res))
    P, d = abi.make_post_state(wins[0]["nf"], wins[0]["np"], wins[0]["nr"])
    assert ctx.L.sdso_ba_get_post_state(ctx.h, 20, C.byref(P)) != 0             # lastHS was not kept
    for k, win in enumerate(wins):
        # re-upload (the loop above moved the states), this time keeping the system
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 20 + k, C.byref(hs[k][0])))
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, len(ids), abi.ip(ids)))
    ctx.check(ctx.L.sdso_ba_batch_keep_system(ctx.h, 1))
    ctx.check(ctx.L.sdso_ba_batch_optimize(ctx.h, 6, res))
    for k, win in enumerate(wins):
        oo = abi.BAOptResult()
        oracle.orc_ba_optimize(hs[k][2], 6, None, None, None, C.byref(oo))
        Po, do = abi.make_post_state(win["nf"], win["np"], win["nr"])
        Pg, dg = abi.make_post_state(win["nf"], win["np"], win["nr"])
        oracle.orc_ba_get_post_state(hs[k][2], C.byref(Po))
        ctx.check(ctx.L.sdso_ba_get_post_state(ctx.h, 20 + k, C.byref(Pg)))
        check_post_state(win, Po, do, Pg, dg)
        assert Pg.result.iterations == res[k].iterations and Pg.result.resInA == res[k].resInA
        oracle.orc_ba_destroy(hs[k][2])
    for k in ids:
        ctx.check(ctx.L.sdso_ba_release_window(ctx.h, int(k)))


def test_post_state_of_ragged_windows(gpu_ctx, oracle):
    """The shapes the reference produces in between: a host keyframe without points, points whose residuals are all gone (they wait for
    flagPointsForRemoval), a window without any point.  The bookkeeping of linearizeAll(true) / addPointSC on them: a point without an
    active residual has HdiF = idepth_hessian = 0 and maxRelBaseline reset (AccumulatedSCHessian.cpp:44-48), contributes no count, and
    the call works on empty arrays."""
    ctx = gpu_ctx
    base = synth.ba_window(w=320, h=240, nf=4, pts_per_kf=30, seed=3074)

    def strip(win, keep_pts=None, drop_res=None, empty=False):
        win = dict(win)
        if keep_pts is not None:
            idx = np.nonzero(keep_pts)[0]
            remap = -np.ones(win["np"], np.int64); remap[idx] = np.arange(len(idx))
            rk = keep_pts[win["res_point"]]
            for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
                win[k] = win[k][idx]
            win["res_point"] = remap[win["res_point"][rk]].astype(np.int32); win["res_target"] = win["res_target"][rk]; win["res_state"] = win["res_state"][rk]
            win["np"], win["nr"] = len(idx), int(rk.sum())
        if drop_res is not None:
            for k in ("res_point", "res_target", "res_state"):
                win[k] = win[k][~drop_res]
            win["nr"] = int((~drop_res).sum())
        if empty:
            for k in ("u", "v", "idepth", "idepth_zero", "host", "hasDepthPrior", "color", "weights", "res_point", "res_target", "res_state"):
                win[k] = win[k][:0]
            win["np"] = win["nr"] = 0
        return win
    bare = np.arange(0, base["np"], 3)
    cases = [("empty_host", strip(base, keep_pts=base["host"] != 1)),
             ("points_without_residuals", strip(base, drop_res=np.isin(base["res_point"], bare))),
             ("no_points", strip(base, empty=True))]
    for name, win in cases:
        nf, npts, nr = win["nf"], win["np"], win["nr"]
        win["maxRelBaseline"] = np.full(npts, 0.25, np.float32)                     # a history, so that the reset is visible
        win["numGoodResiduals"] = np.full(npts, 2, np.int32)
        W, keep, h = _upload_both(ctx, oracle, win, wid=5, slot0=60)
        Po, do, Pg, dg = _post_both(ctx, oracle, win, 5, 4, h)
        assert Pg.result.iterations == Po.result.iterations, name
        assert (Pg.resInA, Pg.resInL, Pg.resInM, Pg.n_toRemove) == (Po.resInA, Po.resInL, Po.resInM, Po.n_toRemove) or nr > 0, name
        if nr:
            flips = dg["state_state"] != do["state_state"]
            assert flips.sum() <= 2, name
            ok = ~flips
            assert np.array_equal(dg["toRemove"][ok], do["toRemove"][ok]) and np.array_equal(dg["isActiveAndIsGoodNEW"][ok], do["isActiveAndIsGoodNEW"][ok]), name
            pt_ok = np.ones(npts, bool); pt_ok[win["res_point"][flips]] = False
            assert np.array_equal(dg["numGoodResiduals"][pt_ok], do["numGoodResiduals"][pt_ok]), name
            assert np.array_equal(dg["idepth_hessian"][pt_ok] == 0, do["idepth_hessian"][pt_ok] == 0), name
            assert np.array_equal(dg["maxRelBaseline"][pt_ok] == 0, do["maxRelBaseline"][pt_ok] == 0), name
            assert np.abs(dg["state"] - do["state"]).max() <= 2e-4, name
        if name == "points_without_residuals":
            assert not dg["HdiF"][bare].any() and not dg["idepth_hessian"][bare].any() and not dg["maxRelBaseline"][bare].any()
            assert np.array_equal(dg["numGoodResiduals"][bare], win["numGoodResiduals"][bare])                  # untouched history
            assert np.array_equal(dg["idepth"][bare], win["idepth"][bare])                                      # and no step (EnergyFunctional.cpp:305-309)
        if name == "no_points":
            assert (Pg.resInA, Pg.n_toRemove) == (0, 0) and np.isfinite(dg["state"]).all()
        oracle.orc_ba_destroy(h)
        ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 5))
