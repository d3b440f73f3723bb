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
from sdso_amd import abi, synth

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


@pytest.mark.gpu
def test_shim_windowed_ba(gpu_ctx, tmp_path):
    win = synth.ba_window(w=640, h=480, nf=4, pts_per_kf=100, seed=3011)
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    sg, ig, rg, og = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 4, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
    arrays = dict(meta=np.array([nf, npts, nr, 640, 480, 4, win["solverMode"]], np.int32),
                  calib=np.concatenate([win["calib_value_scaled"], win["calib_value_zero"]]).astype(np.float64))
    for k in ("evalPT", "state", "state_zero", "HM", "bM"):
        arrays[k] = np.asarray(win[k], np.float64)
    for k in ("ab_exposure", "frameEnergyTH", "u", "v", "idepth", "idepth_zero", "color", "weights"):
        arrays[k] = np.asarray(win[k], np.float32)
    for k in ("frameID", "host", "res_point", "res_target"):
        arrays[k] = np.asarray(win[k], np.int32)
    for k in ("hasDepthPrior", "res_state"):
        arrays[k] = np.asarray(win[k], np.uint8)
    for f in range(nf):
        arrays["img%d_l0" % f] = win["pyrs"][f][0]
    _dump(tmp_path, **arrays)
    lines = _run(tmp_path, "ba")
    head = lines[0].split()
    assert int(head[3]) == og.iterations and int(head[5]) == og.resInA and float(head[7]) == og.lastEnergy
    assert np.float32(head[1]) == np.float32(og.rmse)
    st = np.array([ln.split()[1:] for ln in lines[1:1 + nf]], np.float64)
    assert np.array_equal(st, sg)
    assert np.array_equal(np.array(lines[1 + nf].split()[1:], np.float32), ig)
    assert np.array_equal(np.array(lines[2 + nf].split()[1:], np.uint8), rg)


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
