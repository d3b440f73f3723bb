"""Committed fixtures under tests/golden/ (see tests/golden/cases.py: frozen ORACLE outputs on small seeded cases —
the reference holds no vectors for this path).  CPU: the oracle still reproduces them (a change to the
oracle, to its compiler flags or to the synthetic generator shows up here).  GPU: the HIP path, through the C-ABI,
matches them with the same bars as the live oracle comparisons."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402
import helpers  # noqa: E402
from sdso_amd import abi  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", ["tracker", "ba", "ba_dropped", "stereo", "g2o"])
def test_oracle_reproduces_golden(oracle, name):
    exp = _load(name)
    got = cases.CASES[name](oracle)
    assert set(exp) == set(got)
    assert np.array_equal(exp["input_digest"], got["input_digest"]), "the synthetic generator changed: regenerate with tests/golden/make_golden.py"
    for k in exp:
        a, b = np.asarray(exp[k]), np.asarray(got[k])
        if a.dtype.kind in "iub":
            assert np.array_equal(a, b), k
        else:   # same binary and flags give identical bits; allow libm-level differences of another host
            assert np.allclose(a, b, rtol=1e-6, atol=1e-9, equal_nan=True), k


@pytest.mark.gpu
def test_gpu_tracker_matches_golden(gpu_ctx):
    exp = _load("tracker")
    prob, prm, evs = cases.tracker_case()
    assert np.array_equal(exp["input_digest"], cases.digest(prob["pyr_new"][0], prob["pc"][0]["u"], prob["pc"][0]["idepth"]))
    gpu_ctx.upload_pyramid(2, prob["pyr_new"]); gpu_ctx.set_ref(1, prob["pc"])
    for lvl, ev, T, aff in evs:
        gpu_ctx.L.sdso_track_make_eval(C.byref(prm), lvl, C.byref(abi.SE3.from_Rt(*T)), C.byref(abi.Aff(*aff)), 1.0, C.byref(ev))
        n = len(prob["pc"][lvl]["u"])
        H = np.zeros(64); b = np.zeros(8); res = np.zeros(6); nw = C.c_int(0); mask = np.zeros(n, np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_track_calc_res_gs(gpu_ctx.h, 1, 2, C.byref(ev), abi.dp(H), abi.dp(b), abi.dp(res), C.byref(nw), abi.bp(mask)))
        Ho = exp["H%d" % lvl]
        assert nw.value == int(exp["nw%d" % lvl]) and np.array_equal(np.packbits(mask), exp["mask%d" % lvl])      # bookkeeping: exact
        assert np.abs(H.reshape(8, 8) - Ho).max() <= 2e-5 * np.abs(Ho).max()
        assert abs(res[0] - exp["res%d" % lvl][0]) <= 2e-5 * abs(exp["res%d" % lvl][0]) and res[1] == exp["res%d" % lvl][1]
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); out = abi.TrackResult()
    gpu_ctx.check(gpu_ctx.L.sdso_track_newest_coarse(gpu_ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out)))
    R, t = T.Rt()
    assert np.abs(R - exp["track_R"]).max() <= 1e-5 and np.abs(t - exp["track_t"]).max() <= 1e-5                  # north_star: pose within 1e-5
    assert np.array_equal(np.array(list(out.iterations), np.int32), exp["track_iterations"])


@pytest.mark.gpu
def test_gpu_ba_matches_golden(gpu_ctx):
    exp = _load("ba")
    win = cases.ba_case()
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    e = C.c_double(0)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 3, C.byref(e)))
    J = np.zeros((nr, 74), np.float32); ns = np.zeros(nr, np.uint8); ne = np.zeros(nr, np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_linearization(gpu_ctx.h, 3, abi.fp(J), abi.bp(ns), abi.fp(ne), None, None, None))
    J[ns == 1] = 0
    assert np.array_equal(ns, exp["newState"]) and np.array_equal(ne, exp["newEnergy"]) and np.array_equal(J, exp["J"])   # per residual: bit-exact
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 3))
    x = np.zeros(n); H = np.zeros((n, n))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 3, 0, 0.1, abi.dp(x), abi.dp(H), None, None, None))
    d = np.sqrt(np.abs(exp["Hdiag"])) + 1e-30
    assert np.abs((x - exp["x"]) * d).max() <= 2e-4 * max(1.0, np.abs(exp["x"] * d).max())
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    st, idp, rs, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 4, abi.dp(st), abi.fp(idp), abi.bp(rs), C.byref(oo)))
    assert oo.iterations == int(exp["opt_iterations"])
    assert np.abs(st - exp["opt_state"]).max() <= 1e-4 and np.abs(idp - exp["opt_idepth"]).max() <= 1e-4   # order-of-summation spread, see test_ba_gpu
    assert abs(oo.lastEnergy - float(exp["opt_energy"])) <= 1e-4 * float(exp["opt_energy"])


@pytest.mark.gpu
def test_gpu_ba_dropped_matches_golden(gpu_ctx):
    """A window whose residualsAll lists went through dropResidual: per-point sums and the back-substitution bit-exact, the post-state of
    FullSystem::optimize (flags / counts exact, floats within the loop's bars)."""
    exp = _load("ba_dropped")
    win = cases.ba_dropped_case()
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 3, None))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 3))
    pt = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_terms(gpu_ctx.h, 3, *[abi.fp(a) for a in pt]))
    for k, a in zip(("HdiF", "bdSumF", "Hdd_accAF", "bd_accAF", "Hcd_accAF"), pt):
        assert np.array_equal(a, exp[k]), k                                                          # residualsAll order: bit-exact
    gpu_ctx.check(gpu_ctx.L.sdso_ba_resubstitute(gpu_ctx.h, 3, abi.dp(np.ascontiguousarray(exp["x"])), None, None))
    step = np.zeros(npts, np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_steps(gpu_ctx.h, 3, abi.fp(step)))
    assert np.array_equal(step, exp["point_step"])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    oo = abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 4, None, None, None, C.byref(oo)))
    P, d = abi.make_post_state(nf, npts, nr)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_post_state(gpu_ctx.h, 3, C.byref(P)))
    assert oo.iterations == int(exp["opt_iterations"])
    flips = np.nonzero(d["state_state"] != exp["post_state_state"])[0]
    assert len(flips) <= 2
    same = np.ones(nr, bool); same[flips] = False
    psame = np.ones(npts, bool); psame[win["res_point"][flips]] = False
    for k in ("isActiveAndIsGoodNEW", "toRemove"):
        assert np.array_equal(d[k][same], exp["post_" + k][same]), k
    assert np.array_equal(d["numGoodResiduals"][psame], exp["post_numGoodResiduals"][psame])
    assert abs(P.n_toRemove - int(exp["opt_n_toRemove"])) <= len(flips)
    assert np.abs(d["state"] - exp["post_state"]).max() <= 1e-4 and np.abs(d["idepth"] - exp["post_idepth"]).max() <= 1e-4
    assert np.abs(d["evalPT"] - exp["post_evalPT"]).max() <= 1e-4 and np.array_equal(d["state_zero"][:nf - 1], exp["post_state_zero"][:nf - 1])
    for k in ("HdiF", "idepth_hessian", "maxRelBaseline"):
        a, b = d[k][psame], exp["post_" + k][psame]
        assert np.array_equal(a == 0, b == 0) and np.abs(a - b).max() <= 2e-3 * np.abs(b).max(), k
    act = (exp["post_isActiveAndIsGoodNEW"] == 1) & same
    assert np.abs(d["centerProjectedTo"][act] - exp["post_centerProjectedTo"][act]).max() <= 2e-2


@pytest.mark.gpu
def test_gpu_stereo_matches_golden(gpu_ctx):
    exp = _load("stereo")
    pr = cases.stereo_case()
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    gpu_ctx.upload_pyramid(80, [left]); gpu_ctx.upload_pyramid(81, [right])
    n = len(pr["u"])
    col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_immature_init_batch(gpu_ctx.h, 80, n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth)))
    for k, a in (("color", col), ("weights", wgt), ("gradH", gH), ("energyTH", eth)):
        assert np.array_equal(a, exp[k]), k
    P, d = abi.make_trace_points(n, pr["u"], pr["v"], col, wgt, gH, eth)
    st = np.zeros(n, np.uint8)
    K = np.array(pr["K"], np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 81, abi.fp(K), float(pr["calib"]["baseline"]), 1, C.byref(P), abi.bp(st)))
    assert np.array_equal(st, exp["status"])
    for k in ("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "quality", "lastTraceStatus", "lastTraceUV", "lastTracePixelInterval"):
        assert np.array_equal(d[k], exp[k], equal_nan=True), k          # every traceStereo output: bit-exact


@pytest.mark.gpu
def test_gpu_g2o_factors_match_golden(gpu_ctx):
    """The fork's live factors against the frozen oracle outputs: per-edge doubles bit-exact, sums over edges to 1e-12."""
    import synth
    exp = _load("g2o")
    prob, prm, _ = cases.tracker_case()
    assert np.array_equal(exp["input_digest"], cases.digest(prob["pyr_new"][0], prob["pc"][0]["u"], prob["pc"][0]["idepth"]))
    gpu_ctx.upload_pyramid(12, prob["pyr_new"]); gpu_ctx.set_ref(11, prob["pc"])
    Tc, Tv = synth.se3_exp(cases.G2O_CULL), synth.se3_exp(cases.G2O_VERTEX)
    for lvl in range(prob["levels"]):
        ev = cases.g2o_eval(gpu_ctx.L, "sdso_", prm, lvl, Tc, Tv, (0.01, 1.0))
        n = len(prob["pc"][lvl]["u"])
        res = np.zeros(6); ne = C.c_int(0); mask = np.zeros(n, np.uint8); X = np.zeros((n, 3), np.float32)
        gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_add_edges(gpu_ctx.h, 11, 12, C.byref(ev), abi.dp(res), C.byref(ne), abi.bp(mask), abi.fp(X)))
        assert np.array_equal(np.packbits(mask), exp["mask%d" % lvl]) and np.array_equal(X[::cases.G2O_STRIDE], exp["Xref%d" % lvl])
        assert np.allclose(res, exp["res%d" % lvl], rtol=1e-6, equal_nan=True) and res[1] == exp["res%d" % lvl][1]
        H = np.zeros(64); b = np.zeros(8); chi = np.zeros(2); err = np.zeros(n); J = np.zeros((n, 8))
        gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_linearize(gpu_ctx.h, 11, 12, C.byref(ev), abi.dp(H), abi.dp(b), abi.dp(chi), abi.dp(err), abi.dp(J)))
        sd = cases.G2O_STRIDE
        assert np.allclose(err[::sd], exp["err%d" % lvl], rtol=1e-12, atol=1e-12) and np.allclose(J[::sd], exp["J%d" % lvl], rtol=1e-12, atol=1e-12)
        Ho = exp["H%d" % lvl]
        d = np.sqrt(np.diag(Ho)) + 1e-300
        assert np.abs((H.reshape(8, 8) - Ho) / np.outer(d, d)).max() < 1e-11
        assert np.allclose(chi, exp["chi%d" % lvl], rtol=1e-11)
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); out = abi.TrackResult()
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_newest_coarse(gpu_ctx.h, 11, 12, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out)))
    R, t = T.Rt()
    assert np.abs(R - exp["track_R"]).max() <= 1e-8 and np.abs(t - exp["track_t"]).max() <= 1e-8
    assert np.array_equal(np.array(list(out.iterations), np.int32), exp["track_iterations"]) and out.evaluations == int(exp["track_evaluations"])
    # window edge
    d = cases.g2o_lba_case()
    nf, nr = d["nf"], d["nr"]
    for f in range(nf):
        gpu_ctx.upload_pyramid(20 + f, d["win"]["pyrs"][f][:1])
    S, keep = cases.g2o_lba_struct(d, frame_slots=[20 + f for f in range(nf)])
    e = np.zeros((nr, 8)); Jl = np.zeros((nr, 8, 13)); st = np.zeros(nr, np.uint8); en = np.zeros((nr, 2), np.float32)
    cpt = np.zeros((nr, 3), np.float32); ih = np.zeros(nr, np.float32); lv = np.zeros(nr, np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_lba_eval(gpu_ctx.h, C.byref(S), abi.dp(e), abi.dp(Jl), abi.bp(st), abi.fp(en), abi.fp(cpt), abi.fp(ih), abi.bp(lv)))
    assert np.array_equal(st, exp["lba_state"]) and np.array_equal(lv, exp["lba_level"])
    for got, key in ((e, "lba_error"), (Jl[::cases.G2O_STRIDE], "lba_J"), (en, "lba_energy"), (cpt, "lba_cpt"), (ih, "lba_idepth_hessian")):
        assert np.allclose(got, exp[key], rtol=1e-6, atol=1e-9), key      # the fixture may come from another host's libm; live comparison is bit-exact
    # trace refinement, g2o mode
    pr = cases.stereo_case()
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    gpu_ctx.upload_pyramid(30, [left]); gpu_ctx.upload_pyramid(31, [right])
    n = len(pr["u"])
    col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_immature_init_batch(gpu_ctx.h, 30, n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth)))
    P, dd = abi.make_trace_points(n, pr["u"], pr["v"], col, wgt, gH, eth)
    stt = np.zeros(n, np.uint8)
    K = np.array(pr["K"], np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_set_gn_mode(gpu_ctx.h, 1))
    try:
        gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 31, abi.fp(K), float(pr["calib"]["baseline"]), 1, C.byref(P), abi.bp(stt)))
    finally:
        gpu_ctx.check(gpu_ctx.L.sdso_trace_set_gn_mode(gpu_ctx.h, 0))
    assert np.array_equal(stt, exp["trace_status"])
    for k in ("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "lastTraceUV", "lastTracePixelInterval"):
        assert np.array_equal(dd[k], exp["trace_" + k], equal_nan=True), k
