"""The C++ host layer (stereo-dso-g2o_amd/host/sdso_shim.h) driven the way the reference's classes call it.

host/test_shim.cpp builds CoarseTracker / EnergyFunctional pointer graphs / ImmaturePoint vectors from
stand-ins that carry the reference's member names, goes through the shim, and prints the results; the same
problems pushed through the C-ABI from Python must give the same numbers (identical code path on the
device, so equality is exact up to the %.17g / %.9g round trip of the printout)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stereo-dso-g2o_amd", "host")
EXE = os.path.join(HOST, "test_shim")


def _dump(d, **arrays):
    for k, a in arrays.items():
        np.ascontiguousarray(a).tofile(os.path.join(d, k + ".bin"))


def _run(d, what):
    r = subprocess.run([EXE, str(d), what], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip().splitlines()


def test_shim_header_compiles():
    """CPU: the shim + driver compile against the ABI header with the plain host compiler."""
    r = subprocess.run(["make", "-C", HOST, "-B", "test_shim"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_shim_tracker(gpu_ctx, tmp_path):
    prob = synth.tracker_problem(w=640, h=480, npts=1500, seed=2011)
    L = prob["levels"]
    prm = helpers.track_params(prob)
    gpu_ctx.upload_pyramid(2, prob["pyr_new"]); gpu_ctx.set_ref(1, prob["pc"])
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); out = abi.TrackResult()
    gpu_ctx.check(gpu_ctx.L.sdso_track_newest_coarse(gpu_ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out)))
    arrays = dict(meta=np.array([L, 640, 480, prm.coarsestLvl], np.int32), calib=np.array(prob["K"], np.float64),
                  misc=np.array([1.0, 1.0, 0.0, 0.0] + list(np.eye(3).ravel()) + [0, 0, 0] + [0.0, 0.0] + [np.nan] * 5, np.float64))
    for l in range(L):
        arrays["ref_l%d" % l] = prob["pyr_ref"][l]; arrays["new_l%d" % l] = prob["pyr_new"][l]
        for k in ("u", "v", "idepth", "color"):
            arrays["pc_%s_l%d" % (k, l)] = prob["pc"][l][k]
    _dump(tmp_path, **arrays)
    lines = _run(tmp_path, "tracker")
    assert lines[0] == "good 1" and out.good == 1
    Tc = np.array(lines[1].split()[1:], np.float64)
    R, t = T.Rt()
    assert np.array_equal(Tc[:9].reshape(3, 3), R) and np.array_equal(Tc[9:], t)
    affc = np.array(lines[2].split()[1:], np.float64)
    assert affc[0] == aff.a and affc[1] == aff.b
    resc = np.array(lines[3].split()[1:], np.float64)
    assert np.array_equal(resc, np.array(list(out.lastResiduals)), equal_nan=True)
    flow = np.array(lines[4].split()[1:], np.float64)
    assert np.array_equal(flow, np.array(list(out.lastFlowIndicators)))


@pytest.mark.gpu
def test_shim_fork_live_modes(gpu_ctx, tmp_path):
    """CoarseTracker::forkLive and Device::setForkLiveTraceRefinement select the fork's g2o factors (sdso_g2o_*)."""
    prob = synth.tracker_problem(w=640, h=480, npts=1500, seed=2011)
    L = prob["levels"]
    prm = helpers.track_params(prob)
    gpu_ctx.upload_pyramid(2, prob["pyr_new"]); gpu_ctx.set_ref(1, prob["pc"])
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); out = abi.TrackResult()
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_newest_coarse(gpu_ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out)))
    arrays = dict(meta=np.array([L, 640, 480, prm.coarsestLvl], np.int32), calib=np.array(prob["K"], np.float64),
                  misc=np.array([1.0, 1.0, 0.0, 0.0] + list(np.eye(3).ravel()) + [0, 0, 0] + [0.0, 0.0] + [np.nan] * 5, np.float64))
    for l in range(L):
        arrays["ref_l%d" % l] = prob["pyr_ref"][l]; arrays["new_l%d" % l] = prob["pyr_new"][l]
        for k in ("u", "v", "idepth", "color"):
            arrays["pc_%s_l%d" % (k, l)] = prob["pc"][l][k]
    _dump(tmp_path, **arrays)
    lines = _run(tmp_path, "tracker_g2o")
    assert lines[0] == "good 1" and out.good == 1
    Tc = np.array(lines[1].split()[1:], np.float64)
    R, t = T.Rt()
    assert np.array_equal(Tc[:9].reshape(3, 3), R) and np.array_equal(Tc[9:], t)       # same library, same calls: identical
    resc = np.array(lines[3].split()[1:], np.float64)
    assert np.array_equal(resc, np.array(list(out.lastResiduals)), equal_nan=True)
    native = _run(tmp_path, "tracker")
    assert native[0] == "good 1" and native[1] != lines[1]                               # and it is not the native LM


@pytest.mark.gpu
def test_shim_trace_stereo(gpu_ctx, tmp_path):
    pr = synth.stereo_problem(w=640, h=480, npts=1200, seed=4011)
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    gpu_ctx.upload_pyramid(80, [left]); gpu_ctx.upload_pyramid(81, [right])
    n = len(pr["u"])
    col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_immature_init_batch(gpu_ctx.h, 80, n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth)))
    K = np.array(pr["K"], np.float32); bl = float(pr["calib"]["baseline"])
    P, d = abi.make_trace_points(n, pr["u"], pr["v"], col, wgt, gH, eth)
    st = np.zeros(n, np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 81, abi.fp(K), bl, 1, C.byref(P), abi.bp(st)))
    _dump(tmp_path, meta=np.array([640, 480, n, 1], np.int32), K=np.array(list(K) + [bl], np.float32), right_l0=right,
          u_stereo=pr["u"], v_stereo=pr["v"], idepth_min=np.zeros(n, np.float32), idepth_min_stereo=np.zeros(n, np.float32),
          idepth_max_stereo=np.full(n, np.nan, np.float32), color=col, weights=wgt, gradH=gH, energyTH=eth)
    lines = _run(tmp_path, "stereo")
    assert len(lines) == n
    got = np.array([[float(x) for x in ln.split()] for ln in lines])
    assert np.array_equal(got[:, 0].astype(np.uint8), st) and np.array_equal(got[:, 1].astype(np.uint8), d["lastTraceStatus"])
    for j, k in enumerate(("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "quality")):
        assert np.array_equal(got[:, 2 + j].astype(np.float32), d[k], equal_nan=True), k
    assert np.array_equal(got[:, 6:8].astype(np.float32), d["lastTraceUV"], equal_nan=True)
    assert np.array_equal(got[:, 8].astype(np.float32), d["lastTracePixelInterval"], equal_nan=True)


def _load(d, name, dt):
    return np.fromfile(os.path.join(d, "out_" + name + ".bin"), dtype=dt)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["c3_dropped_history", "small_fresh"])
def test_shim_windowed_ba(gpu_ctx, oracle, tmp_path, which):
    """FullSystem::optimize through sdso_shim::WindowedBA on a reference-shaped pointer graph (8 KF / 2000 points): EVERYTHING the compiled
    C++ program finds in its objects afterwards is compared with the ORACLE's post-state — frame / calibration states, per-point idepth,
    idepth_hessian, maxRelBaseline, numGoodResiduals, HdiF, lastResiduals, per-residual state / energy / centerProjectedTo, the residuals
    that were dropped and the order dropResidual / deleteOut left the survivors in — and so are the two consumers' decisions
    (makeCoarseDepthL0 STEP1 weights, flagPointsForRemoval)."""
    from test_ba_post_state_gpu import check_post_state
    if which == "small_fresh":
        win = synth.ba_window(w=640, h=480, nf=4, pts_per_kf=100, seed=3011)
    else:
        base = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
        win, kept = helpers.drop_residuals(base, seed=11, drop_frac=0.25)
        rs = np.random.RandomState(5)
        win["numGoodResiduals"] = rs.randint(0, 9, win["np"]).astype(np.int32)
        win["maxRelBaseline"] = (rs.uniform(0, 0.4, win["np"]) * (rs.rand(win["np"]) < 0.7)).astype(np.float32)
        win["res_isNew"] = (rs.rand(win["nr"]) < 0.8).astype(np.uint8)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    win.setdefault("numGoodResiduals", np.zeros(npts, np.int32)); win.setdefault("maxRelBaseline", np.zeros(npts, np.float32))
    win.setdefault("res_isNew", np.ones(nr, np.uint8))
    # ---- the oracle's post-state
    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    oo = abi.BAOptResult()
    oracle.orc_ba_optimize(h, 6, None, None, None, C.byref(oo))
    Po, do = abi.make_post_state(nf, npts, nr)
    oracle.orc_ba_get_post_state(h, C.byref(Po))
    # EnergyFunctional::accumulate{AF,LF,SCF}_MT at the state the loop left: the three stitched systems (the shim's members of the same name)
    oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
    st_o = [(np.zeros((n, n)), np.zeros(n)) for _ in range(3)]
    oracle.orc_ba_get_stitched(h, *[abi.dp(a) for pair in st_o for a in pair])
    oracle.orc_ba_destroy(h)
    # ---- the C++ program
    arrays = dict(meta=np.array([nf, npts, nr, win["w"], win["h"], 6, win["solverMode"]], np.int32),
                  calib=np.concatenate([win["calib_value_scaled"], win["calib_value_zero"]]).astype(np.float64))
    for k in ("evalPT", "state", "state_zero", "HM", "bM"):
        arrays[k] = np.asarray(win[k], np.float64)
    for k in ("ab_exposure", "frameEnergyTH", "u", "v", "idepth", "idepth_zero", "color", "weights", "maxRelBaseline"):
        arrays[k] = np.asarray(win[k], np.float32)
    for k in ("frameID", "host", "res_point", "res_target", "numGoodResiduals"):
        arrays[k] = np.asarray(win[k], np.int32)
    for k in ("hasDepthPrior", "res_state", "res_isNew"):
        arrays[k] = np.asarray(win[k], np.uint8)
    for f in range(nf):
        arrays["img%d_l0" % f] = win["pyrs"][f][0]
    _dump(tmp_path, **arrays)
    lines = _run(tmp_path, "ba")
    head = lines[0].split()
    d = str(tmp_path)
    counts = _load(d, "counts", np.int32)          # resInA resInL resInM nResiduals removed iterations
    pt = _load(d, "pt", np.float32).reshape(npts, 8); pi = _load(d, "pi", np.int32).reshape(npts, 8)
    alive = _load(d, "alive", np.int32).astype(bool)
    # ---- pack what the program holds into the post-state layout and run the common field-by-field check
    cal = _load(d, "calib", np.float64)
    dg = dict(idepth=pt[:, 0].copy(), step=pt[:, 2].copy(), idepth_hessian=pt[:, 3].copy(), maxRelBaseline=pt[:, 4].copy(), HdiF=pt[:, 5].copy(), bdSumF=pt[:, 6].copy(),
              numGoodResiduals=pi[:, 0].copy(),
              state=_load(d, "state", np.float64).reshape(nf, 10), state_zero=_load(d, "state_zero", np.float64).reshape(nf, 10),
              evalPT=_load(d, "evalPT", np.float64).reshape(nf, 12), frame_step=_load(d, "frame_step", np.float64).reshape(nf, 10),
              frameEnergyTH=_load(d, "frameEnergyTH", np.float32), lastX=_load(d, "lastX", np.float64), lastbS=_load(d, "lastbS", np.float64),
              lastHS=_load(d, "lastHS", np.float64).reshape(n, n))
    assert np.array_equal(pt[:, 0], pt[:, 1])                                       # setIdepth + setIdepthZero (:268-272)
    # the program dropped what linearizeAll(true) would: the survivors carry their state; dropped residuals are gone from both lists
    rstate, ract = _load(d, "rstate", np.int32), _load(d, "ract", np.int32)
    dg["toRemove"] = (~alive).astype(np.uint8)
    dg["isActiveAndIsGoodNEW"] = np.where(alive, ract, 0).astype(np.uint8)
    dg["state_state"] = np.where(alive, rstate, do["state_state"]).astype(np.uint8)   # (a dropped residual's state left with it)
    dg["state_energy"] = np.where(alive, _load(d, "renergy", np.float32), do["state_energy"]).astype(np.float32)
    dg["centerProjectedTo"] = _load(d, "cpt", np.float32).reshape(nr, 3); dg["projectedTo"] = _load(d, "prj", np.float32).reshape(nr, 16)
    dg["PRE_worldToCam"] = do["PRE_worldToCam"].copy(); dg["PRE_worldToCam"][nf - 1] = dg["evalPT"][nf - 1]   # (host math of the reference's setState; the newest frame's is its evalPT)

    class G:                                                                          # the scalar members of the post-state
        pass
    Pg = G()
    Pg.result = G(); Pg.result.iterations = int(counts[5]); Pg.result.resInA = int(counts[0])
    Pg.n_toRemove = int(counts[4]); Pg.resInA, Pg.resInL, Pg.resInM = int(counts[0]), int(counts[1]), int(counts[2])
    Pg.calib_value, Pg.calib_value_scaled, Pg.calib_step = list(cal[0:4]), list(cal[4:8]), list(cal[8:12])
    flips = check_post_state(win, Po, do, Pg, dg)
    # accumulateAF_MT / accumulateLF_MT / accumulateSCF_MT through the shim: H, b of the top-A, top-L (priors) and Schur systems, each at its
    # own side's final state (they differ by the loop bars)
    st = _load(d, "stitched", np.float64).reshape(3, n * n + n)
    dd = np.sqrt(np.abs(np.diag(do["lastHS"]))) + 1e-30
    for k, name in enumerate(("A", "L", "SC")):
        Hs, bs = st[k, :n * n].reshape(n, n), st[k, n * n:]
        Ho, bo = st_o[k]
        assert np.abs(Hs - Hs.T).max() <= 1e-9 * max(1.0, np.abs(Hs).max()), name
        assert np.abs((Hs - Ho) / np.outer(dd, dd)).max() <= 2e-3, (name, np.abs((Hs - Ho) / np.outer(dd, dd)).max())
        # (b at two final states ~1e-5 apart, a flipped residual included: tests/test_ba_gpu.py::test_stitched_systems_of_the_three_accumulations
        #  holds the same-state bars, 1e-4)
        assert np.abs((bs - bo) / dd).max() <= 2e-2 * max(1.0, np.abs(bo / dd).max()), (name, np.abs((bs - bo) / dd).max())
    assert np.abs(st[0, :n * n]).max() > 0 and np.abs(st[2, :n * n]).max() > 0
    assert int(head[3]) == oo.iterations and int(head[9]) == Pg.n_toRemove and int(head[11]) == nr - Pg.n_toRemove
    # every survivor is IN and active: nothing else stays in ph->residuals after the closing linearizeAll (:80-84, :176-195)
    assert np.all(rstate[alive] == 0) and np.all(ract[alive] == 1)
    # ---- the lists dropResidual / deleteOut left behind: the swap-with-last of the reference on the oracle's toRemove flags
    starts = np.searchsorted(win["res_point"], np.arange(npts), side="left"); ends = np.searchsorted(win["res_point"], np.arange(npts), side="right")
    expect = helpers.apply_drops([list(range(int(starts[p]), int(ends[p]))) for p in range(npts)], [int(i) for i in np.nonzero(dg["toRemove"])[0]])
    lists = _load(d, "lists", np.int32)
    pos = 0
    for p in range(npts):
        e1 = pos + int(np.nonzero(lists[pos:] == -1)[0][0]); e2 = e1 + 1 + int(np.nonzero(lists[e1 + 1:] == -2)[0][0])
        assert list(lists[pos:e1]) == expect[p] and list(lists[e1 + 1:e2]) == expect[p], p      # PointHessian::residuals and EFPoint::residualsAll
        assert pi[p, 1] == len(expect[p])
        pos = e2 + 1
    assert (np.array([len(e) for e in expect]) != (ends - starts)).any()
    # ---- lastResiduals (FullSystemOptimize.cpp:165-185): [0] = the residual into the newest frame, [1] = into the one before
    tgt = win["res_target"]
    for k, t in ((0, nf - 1), (1, nf - 2)):
        rid = np.full(npts, -1, np.int64)
        sel = np.nonzero(tgt == t)[0]
        rid[win["res_point"][sel]] = sel
        had = rid >= 0
        ok = np.ones(npts, bool); ok[win["res_point"][flips]] = False
        assert np.array_equal(pi[had & ok, 2 + 2 * k] == 1, do["toRemove"][rid[had & ok]] == 0)          # .first cleared iff the residual was dropped
        assert np.array_equal(pi[had & ok, 3 + 2 * k], do["state_state"][rid[had & ok]].astype(np.int32))  # .second = state_state
        assert np.all(pi[~had, 2 + 2 * k] == 0)
    # ---- consumers.  makeCoarseDepthL0 STEP1: points whose newest-frame residual is alive and IN; pixel and weight from the oracle's post-state
    rid0 = np.full(npts, -1, np.int64); sel = np.nonzero(tgt == nf - 1)[0]; rid0[win["res_point"][sel]] = sel
    ok = np.ones(npts, bool); ok[win["res_point"][flips]] = False
    enters_o = (rid0 >= 0) & (do["toRemove"][np.maximum(rid0, 0)] == 0) & (do["state_state"][np.maximum(rid0, 0)] == 0)
    assert np.array_equal((pi[:, 6] >= 0)[ok], enters_o[ok]) and enters_o.sum() > 0.2 * npts
    use = enters_o & ok
    uo = (do["centerProjectedTo"][rid0[use], 0] + np.float32(0.5)).astype(np.int32); vo = (do["centerProjectedTo"][rid0[use], 1] + np.float32(0.5)).astype(np.int32)
    ug, vg = pi[use, 6] % 65536, pi[use, 6] // 65536
    assert (np.abs(ug - uo) + np.abs(vg - vo) > 0).sum() <= max(2, use.sum() // 200)             # a pixel changes only when the projection sits on x.5
    wo = np.sqrt(np.float32(1e-3) / (do["HdiF"][use].astype(np.float64) + 1e-12)).astype(np.float32)
    assert np.abs(pt[use, 7] - wo).max() <= 1e-3 * wo.max() and pt[use, 7].min() > 0
    assert np.all(pt[use, 7] < 0.9 * np.float32(np.sqrt(1e-3 / 1e-12)))                          # NOT the weight of HdiF = 0 (what the round-3 binding produced)
    # flagPointsForRemoval for a flagged host: marginalise iff isInlierNew() && idepth_hessian > setting_minIdepthH_marg (FullSystem.cpp:1008-1031)
    nres_left = np.array([len(e) for e in expect])
    dec_o = np.where((do["idepth"] < 0) | (nres_left == 0), 0,
                     np.where((nres_left >= 3) & (do["numGoodResiduals"] >= 4), np.where(do["idepth_hessian"] > 50.0, 1, 2), 3))
    edge = np.abs(do["idepth_hessian"] - 50.0) < 0.5                                            # a Hessian sitting on the threshold may fall either way
    assert np.array_equal(pi[ok & ~edge, 7], dec_o[ok & ~edge])
    if which != "small_fresh":                                                                  # (a fresh 4-frame window has no point with 4 good residuals yet)
        assert (dec_o == 1).sum() > 0.3 * npts and (dec_o != 1).sum() > 0                        # both branches occur


@pytest.mark.gpu
def test_shim_selector_and_marginalize_frame(gpu_ctx, tmp_path):
    prob = synth.tracker_problem(w=640, h=480, npts=50, seed=7)
    pyr = prob["pyr_ref"]
    gpu_ctx.upload_pyramid(91, pyr)
    m = np.zeros((480, 640), np.float32); p = C.c_int(3); n = C.c_int(0)
    gpu_ctx.check(gpu_ctx.L.sdso_pixel_select(gpu_ctx.h, 91, 1500.0, 1, 1.0, C.byref(p), abi.fp(m), C.byref(n)))
    arrays = dict(meta=np.array([640, 480, prob["levels"], 3, 1], np.int32), par=np.array([1500.0, 1.0], np.float32))
    for l in range(prob["levels"]):
        arrays["img_l%d" % l] = pyr[l]
    _dump(tmp_path, **arrays)
    lines = _run(tmp_path, "selector")
    f = lines[0].split()
    assert int(f[1]) == n.value and int(f[3]) == p.value
    assert [int(f[5]), int(f[7]), int(f[9])] == [int((m == 1).sum()), int((m == 2).sum()), int((m == 4).sum())]
    h = 0
    for i, v in enumerate(m.ravel().astype(np.int64)):
        if v or True:
            h = (h * 1000003 + (int(v) * 7 + 1) * (i + 1)) % (1 << 64)
    assert int(f[11]) == h
    HM = np.array([[50.0 + i if i == j else 1.0 / (1 + i + j) for j in range(20)] for i in range(20)])
    bM = 0.1 * (np.arange(20) + 1)
    prior = np.array([1e3, 0, 1e3, 0, 1e2, 0, 1e6, 1e6]); dprior = np.array([1e-3, 0, -2e-3, 0, 1e-3, 0, 1e-4, -1e-4])
    Ho = np.zeros((12, 12)); bo = np.zeros(12)
    assert gpu_ctx.L.sdso_ba_marginalize_frame(2, 0, abi.dp(prior), abi.dp(dprior), abi.dp(HM), abi.dp(bM), abi.dp(Ho), abi.dp(bo)) == 0
    g = lines[1].split()
    assert int(g[1]) == 144 and int(g[2]) == 12 and float(g[3]) == Ho[0, 0] and float(g[4]) == bo[11]


def _ba_arrays(win):
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    arrays = dict(meta=np.array([nf, npts, nr, win["w"], win["h"], 6, win["solverMode"]], np.int32),
                  calib=np.concatenate([win["calib_value_scaled"], win["calib_value_zero"]]).astype(np.float64))
    for k in ("evalPT", "state", "state_zero", "HM", "bM"):
        arrays[k] = np.asarray(win[k], np.float64)
    for k in ("ab_exposure", "frameEnergyTH", "u", "v", "idepth", "idepth_zero", "color", "weights", "maxRelBaseline"):
        arrays[k] = np.asarray(win[k], np.float32)
    for k in ("frameID", "host", "res_point", "res_target", "numGoodResiduals"):
        arrays[k] = np.asarray(win[k], np.int32)
    for k in ("hasDepthPrior", "res_state", "res_isNew"):
        arrays[k] = np.asarray(win[k], np.uint8)
    for f in range(nf):
        arrays["img%d_l0" % f] = win["pyrs"][f][0]
    return arrays


@pytest.mark.gpu
def test_shim_energy_functional_members(oracle, tmp_path):
    """The members the round-4 verdict found missing from the shim, with the reference's signatures, driven by the bodies of the reference's
    own callers (host/test_shim.cpp::run_ba_members) on an 8-keyframe / 2000-point window and compared with the ORACLE:
      PointFrameResidual::linearize(CalibHessian*) / applyRes(bool) per object   (Residuals.h:103, :113; FullSystemOptimize.cpp:52-96)
      EnergyFunctional::setAdjointsF / setDeltaF                                   (EnergyFunctional.cpp:41-119, :173-207)
      AccumulatedTopHessianSSE / AccumulatedSCHessianSSE::{setZero, addPoint<mode>, addPointsInternal<mode>, stitchDouble[MT]}
                                                                                   (accumulateAF/LF/SCF_MT :212-269; marginalizePointsF :663-736)
      EnergyFunctional::solveSystemF(int, double, CalibHessian*) writing into the objects (:838-995, :272-341)
      EnergyFunctional::calcLEnergyF_MT / calcMEnergyF                             (:344-442)"""
    base = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
    win, kept = helpers.drop_residuals(base, seed=11, drop_frac=0.25)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    rs = np.random.RandomState(9)
    A = rs.normal(0, 1, (n, n))
    win["HM"] = (A @ A.T) * 1e3                  # a marginalisation prior, so that calcMEnergyF is not trivially zero
    win["bM"] = rs.normal(0, 1e2, n)
    win.setdefault("numGoodResiduals", np.zeros(npts, np.int32)); win.setdefault("maxRelBaseline", np.zeros(npts, np.float32))
    win.setdefault("res_isNew", np.ones(nr, np.uint8))
    # ---- the oracle, step by step
    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    Eo = C.c_double(0)
    oracle.orc_ba_linearize(h, C.byref(Eo))
    ns, ne, nw = np.zeros(nr, np.uint8), np.zeros(nr, np.float32), np.zeros(nr, np.float32)
    oracle.orc_ba_get_linearization(h, None, abi.bp(ns), abi.fp(ne), abi.fp(nw), None, None)
    oracle.orc_ba_apply_res(h)
    st, act = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8)
    oracle.orc_ba_get_residual_state(h, abi.bp(st), abi.bp(act), None)
    adH, adT, htd = np.zeros((nf * nf, 64)), np.zeros((nf * nf, 64)), np.zeros((nf * nf, 8), np.float32)
    oracle.orc_ba_get_tables(h, None, abi.dp(adH), abi.dp(adT), abi.fp(htd))
    cd, fd, fdp, pd = np.zeros(4, np.float32), np.zeros((nf, 8)), np.zeros((nf, 8)), np.zeros(npts, np.float32)
    oracle.orc_ba_get_deltas(h, abi.fp(cd), abi.dp(fd), abi.dp(fdp), abi.fp(pd))
    oracle.orc_ba_accumulate(h)
    st_o = [(np.zeros((n, n)), np.zeros(n)) for _ in range(3)]
    oracle.orc_ba_get_stitched(h, *[abi.dp(a) for pair in st_o for a in pair])
    xo, HSo, bSo, fso, cso = np.zeros(n), np.zeros((n, n)), np.zeros(n), np.zeros((nf, 8)), np.zeros(4)
    oracle.orc_ba_solve(h, 0, 1e-1, abi.dp(xo), abi.dp(HSo), abi.dp(bSo), abi.dp(fso), abi.dp(cso))
    pso = np.zeros(npts, np.float32)
    oracle.orc_ba_get_point_steps(h, abi.fp(pso))
    pto = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in pto])
    EL0, EM0, EL1, EM1 = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    oracle.orc_ba_calc_energies(h, C.byref(EL0), C.byref(EM0))
    flag = (win["host"] == 0).astype(np.uint8)
    HMo, bMo = np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HMo), abi.dp(bMo))
    oracle.orc_ba_calc_energies(h, C.byref(EL1), C.byref(EM1))
    Po, do = abi.make_post_state(nf, npts, nr)
    oracle.orc_ba_get_post_state(h, C.byref(Po))
    oracle.orc_ba_destroy(h)
    # ---- the C++ program
    _dump(tmp_path, **_ba_arrays(win))
    lines = _run(tmp_path, "ba_members")
    assert lines[-1].startswith("members ok")
    d = str(tmp_path)
    m = lambda name, dt: _load(d, "m_" + name, dt)      # noqa: E731
    # per-object linearize / applyRes: decisions and energies of every residual, bit for bit
    assert np.array_equal(m("newState", np.int32), ns) and np.array_equal(m("state", np.int32), st) and np.array_equal(m("act", np.int32), act)
    lin = ns != 1
    assert np.array_equal(m("newEnergy", np.float64)[lin], ne[lin].astype(np.float64)) and np.array_equal(m("newEnergyWO", np.float64), nw.astype(np.float64))
    en = m("energies", np.float64)     # E, EL, EM, resInA, resInL | EL, EM after the marginalisation, resInM, flagged
    assert abs(en[0] - Eo.value) <= 1e-6 * Eo.value
    # adjoints and deltas: host tables, identical
    ad = m("adjoints", np.float64).reshape(nf * nf, 2, 64)
    assert np.array_equal(ad[:, 0], adH) and np.array_equal(ad[:, 1], adT)
    adf = m("adjointsF", np.float32).reshape(nf * nf, 2, 64)
    assert np.array_equal(adf[:, 0], adH.astype(np.float32)) and np.array_equal(adf[:, 1], adT.astype(np.float32))
    assert np.array_equal(m("adHTdeltaF", np.float32).reshape(nf * nf, 8), htd) and np.array_equal(m("cDeltaF", np.float32), cd)
    fdg = m("frame_delta", np.float64).reshape(nf, 2, 8)
    assert np.array_equal(fdg[:, 0], fd) and np.array_equal(fdg[:, 1], fdp) and np.array_equal(m("point_delta", np.float32), pd)
    # the stitched systems of the three accumulator objects
    blk = n * n + n
    stg = m("stitched", np.float64)
    dsc = np.sqrt(np.abs(np.diag(st_o[0][0] + st_o[1][0]))) + 1e-30
    for k, name in enumerate(("A", "L", "SC")):
        Hg, bg = stg[k * blk:k * blk + n * n].reshape(n, n), stg[k * blk + n * n:(k + 1) * blk]
        Ho, bo = st_o[k]
        assert np.abs((Hg - Ho) / np.outer(dsc, dsc)).max() <= 1e-4, name
        assert np.abs((bg - bo) / dsc).max() <= 1e-4 * max(1.0, np.abs(bo / dsc).max()), name
        assert np.abs(Hg).max() > 0, name
    assert int(en[3]) == int(act.sum()) and int(en[4]) == 0
    # solveSystemF into the objects
    dd = np.sqrt(np.abs(np.diag(HSo))) + 1e-30
    sc = max(1.0, np.abs(xo * dd).max())
    xg = m("lastX", np.float64)
    assert np.abs((xg - xo) * dd).max() / sc <= 2e-4
    assert np.abs((m("lastHS", np.float64).reshape(n, n) - HSo) / np.outer(dd, dd)).max() <= 1e-4
    fsg = m("frame_step", np.float64).reshape(nf, 10)
    assert np.array_equal(fsg[:, :8].ravel(), -xg[4:]) and not fsg[:, 8:].any()              # step.head<8>() = -x.segment<8>(..), tail<2>() = 0
    assert np.array_equal(m("calib_step", np.float64), -xg[:4])
    psg = m("point_step", np.float32).reshape(npts, 3)
    assert np.array_equal(psg[:, 1], pto[0]) and np.array_equal(psg[:, 2], pto[1])           # EFPoint::HdiF, bdSumF: bit-exact per-point terms
    assert helpers.idepths_close(psg[:, 0], pso, 2e-4 * max(1.0, float(np.abs(pso).max())))
    # the energies, before and after the marginalisation linearised the oldest keyframe's points
    for got, ref in ((en[1], EL0.value), (en[2], EM0.value), (en[5], EL1.value), (en[6], EM1.value)):
        assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (got, ref)
    assert abs(EL1.value - EL0.value) > 0 and int(en[8]) == int(flag.sum())
    # marginalizePointsF through the accumulator objects: M, Msc as stitched, and the prior the device keeps = the oracle's HM / bM
    mg = m("marg", np.float64)
    M, Mb, Msc, Mbsc, HMg, bMg = mg[:n * n].reshape(n, n), mg[n * n:blk], mg[blk:blk + n * n].reshape(n, n), mg[blk + n * n:2 * blk], mg[2 * blk:2 * blk + n * n].reshape(n, n), mg[2 * blk + n * n:]
    dm = np.sqrt(np.abs(np.diag(HMo))) + 1e-30
    assert np.abs((HMg - HMo) / np.outer(dm, dm)).max() <= 1e-4 and np.abs((bMg - bMo) / dm).max() <= 1e-4 * max(1.0, np.abs(bMo / dm).max())
    HM0 = np.asarray(win["HM"], np.float64).reshape(n, n)
    assert np.abs((HM0 + 0.25 * (M - Msc) - HMg) / np.outer(dm, dm)).max() <= 1e-9          # HM += setting_margWeightFac * (M - Msc)  (:727)
    assert np.abs((np.asarray(win["bM"]) + 0.25 * (Mb - Mbsc) - bMg) / dm).max() <= 1e-9 * max(1.0, np.abs(bMg / dm).max())
    assert int(en[7]) == Po.resInM and Po.resInM > 0


@pytest.mark.gpu
def test_shim_set_coarse_tracking_ref(oracle, tmp_path):
    """CoarseTracker::setCoarseTrackingRef(std::vector<FrameHessian*>, FrameHessian* fh_right, CalibHessian) — CoarseTracker.h:71-72 — and
    setCTRefForFirstFrame(std::vector<FrameHessian*>) — CoarseTracker.cpp:794-805 — with the reference's signatures on a pointer graph:
    makeCoarseDepthL0 STEP1 (:288-356: the static-stereo re-observation of every point whose last residual is IN, there and back, the
    accept rule, the inverse-covariance weight) and STEP2-5 run on the device; the template levels must equal, bit for bit and in
    order, what the oracle builds from the same points (ImmaturePoint ctor, traceStereo L->R, ctor at lastTraceUV, traceStereo R->L,
    orc_make_coarse_depth)."""
    from test_stereo import _oracle_init, _oracle_trace
    import pyoracle
    pr = synth.stereo_problem(w=640, h=480, npts=1800, seed=4011)
    L = abi.load().sdso_pyramid_levels(640, 480)
    n = len(pr["u"])
    rs = np.random.RandomState(7)
    left = [np.ascontiguousarray(a) for a in pr["pyr_l"][:L]]
    right0 = np.ascontiguousarray(pr["pyr_r"][0])
    # what FullSystem::optimize left on the points: centerProjectedTo (sub-pixel position in the newest keyframe, idepth there),
    # lastResiduals[0] (some absent, some not IN), efPoint->HdiF
    cpt = np.stack([pr["u"] + rs.uniform(-0.45, 0.45, n), pr["v"] + rs.uniform(-0.45, 0.45, n), pr["idepth_true"] * rs.uniform(0.8, 1.25, n)], axis=1).astype(np.float32)
    has_last = (rs.rand(n) < 0.9).astype(np.uint8)
    rstate = np.where(rs.rand(n) < 0.85, 0, rs.randint(1, 3, n)).astype(np.int32)
    hdi = (1.0 / rs.uniform(50, 5000, n)).astype(np.float32)
    frame_of = rs.randint(0, 3, n).astype(np.int32)
    K = np.array(pr["K"], np.float32); bl = float(pr["calib"]["baseline"])
    sel = np.nonzero((has_last == 1) & (rstate == 0))[0]
    # ---- expected, from the oracle
    ui = (cpt[sel, 0] + np.float32(0.5)).astype(np.int32); vi = (cpt[sel, 1] + np.float32(0.5)).astype(np.int32)
    uf, vf = ui.astype(np.float32), vi.astype(np.float32)
    imin, imax = (cpt[sel, 2] * np.float32(0.1)).astype(np.float32), (cpt[sel, 2] * np.float32(1.9)).astype(np.float32)
    co, wo, go, eo = _oracle_init(oracle, pr, left[0], uf, vf)
    Pf, dfw = abi.make_trace_points(len(sel), uf, vf, co, wo, go, eo, imin, imax)
    sf = _oracle_trace(oracle, pr, right0, Pf, 1)
    good = np.nonzero(sf == 0)[0]
    ub, vb = dfw["lastTraceUV"][good, 0].copy(), dfw["lastTraceUV"][good, 1].copy()
    c2, w2, g2, e2 = _oracle_init(oracle, pr, right0, ub, vb)
    Pb, db = abi.make_trace_points(len(good), ub, vb, c2, w2, g2, e2, imin[good], imax[good])
    _oracle_trace(oracle, pr, left[0], Pb, 0)
    new_idepth = cpt[sel, 2].copy()
    ids = dfw["idepth_stereo"][good]
    with np.errstate(divide="ignore"):
        depth = np.float32(1.0) / ids
    ok = (np.abs(uf[good] - db["lastTraceUV"][:, 0]) < 1) & (depth > 0) & (depth < 50)
    new_idepth[good[ok]] = ids[ok]
    assert ok.sum() > 0.3 * len(sel) and (~ok).sum() + (sf != 0).sum() > 0                 # both branches of the accept rule are taken
    weight = np.sqrt((1e-3 / (hdi[sel].astype(np.float64) + 1e-12)).astype(np.float32)).astype(np.float32)
    # (the reference iterates frame by frame, point by point: the splat order — which decides the float sums of colliding pixels)
    order = np.concatenate([np.nonzero(frame_of[sel] == f)[0] for f in range(3)])
    exp0 = pyoracle.make_coarse_depth(oracle, ui[order], vi[order], new_idepth[order], weight[order], left)
    lastf = np.nonzero(frame_of == 2)[0]
    pu, pv, pid = (pr["u"] + rs.uniform(-0.4, 0.4, n)).astype(np.float32), (pr["v"] + rs.uniform(-0.4, 0.4, n)).astype(np.float32), pr["idepth_true"].astype(np.float32)
    w1 = np.sqrt((1e-3 / (hdi[lastf].astype(np.float64) + 1e-12)).astype(np.float32)).astype(np.float32)
    exp1 = pyoracle.make_coarse_depth(oracle, (pu[lastf] + np.float32(0.5)).astype(np.int32), (pv[lastf] + np.float32(0.5)).astype(np.int32), pid[lastf], w1, left)
    # ---- the C++ program
    arrays = dict(meta=np.array([L, 640, 480, n, 3], np.int32), K=np.array(list(K) + [bl], np.float32), cpt=cpt, HdiF=hdi, pu=pu, pv=pv, pidepth=pid,
                  rstate=rstate, frame_of=frame_of, has_last=has_last, right_l0=right0)
    for l in range(L):
        arrays["left_l%d" % l] = left[l]
    _dump(tmp_path, **arrays)
    lines = _run(tmp_path, "tracker_ref")
    assert lines[-1] == "tracker_ref ok"
    pcn = _load(str(tmp_path), "pcn", np.int32).reshape(2, L)
    for variant, exp in enumerate((exp0, exp1)):
        for l in range(L):
            npc = len(exp[l]["u"])
            assert pcn[variant, l] == npc and npc > 0, (variant, l)
            got = _load(str(tmp_path), "pc_%d_l%d" % (variant, l), np.float32).reshape(4, npc)
            for a, k in zip(got, ("u", "v", "idepth", "color")):
                assert np.array_equal(a, exp[l][k]), (variant, l, k)
