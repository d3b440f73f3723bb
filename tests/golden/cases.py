"""The small seeded cases behind tests/golden/*.npz: shared by make_golden.py (writes the expected outputs,
computed by the oracle) and tests/test_golden.py (checks oracle and HIP path against them).

These fixtures are NOT reference outputs — the reference cannot be built in this image and holds no
vectors for this path; they freeze the oracle's behaviour so that a change to it (or to the generator) is visible."""
import ctypes as C
import hashlib

import numpy as np

import helpers
from sdso_amd import abi, synth


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], np.uint8).copy()


def tracker_case():
    prob = synth.tracker_problem(w=320, h=240, npts=500, seed=2101)
    prm = helpers.track_params(prob)
    T = synth.se3_exp(np.array([0.015, -0.008, 0.3, 0.003, -0.005, 0.0015]))
    evs = []
    for lvl in range(prob["levels"]):
        ev = abi.TrackEval()
        evs.append((lvl, ev, T, (0.01, 1.0)))
    return prob, prm, evs


def tracker_expected(orc):
    prob, prm, evs = tracker_case()
    out = {"input_digest": digest(prob["pyr_new"][0], prob["pc"][0]["u"], prob["pc"][0]["idepth"])}
    for lvl, ev, T, aff in evs:
        orc.orc_track_make_eval(C.byref(prm), lvl, C.byref(abi.SE3.from_Rt(*T)), C.byref(abi.Aff(*aff)), 1.0, C.byref(ev))
        H, b, res, nw, mask = helpers.oracle_eval(orc, prob["pc"][lvl], prob["pyr_new"][lvl], ev)
        out["H%d" % lvl], out["b%d" % lvl], out["res%d" % lvl], out["nw%d" % lvl], out["mask%d" % lvl] = H, b, res, np.int32(nw), np.packbits(mask)
    To, affo, oo = helpers.oracle_track(orc, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0))
    out["track_R"], out["track_t"] = To.Rt()
    out["track_aff"] = np.array([affo.a, affo.b])
    out["track_iterations"] = np.array(list(oo.iterations), np.int32)
    out["track_lastResiduals"] = np.array(list(oo.lastResiduals))
    return out


def ba_case():
    return synth.ba_window(w=320, h=240, nf=4, pts_per_kf=60, seed=3101)


def ba_expected(orc):
    win = ba_case()
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
    h = orc.orc_ba_create(C.byref(W))
    e = C.c_double(0)
    orc.orc_ba_linearize(h, C.byref(e))
    J = np.zeros((nr, 74), np.float32); ns = np.zeros(nr, np.uint8); ne = np.zeros(nr, np.float32)
    orc.orc_ba_get_linearization(h, abi.fp(J), abi.bp(ns), abi.fp(ne), None, None, None)
    orc.orc_ba_apply_res(h)
    orc.orc_ba_accumulate(h)
    acc = np.zeros(abi.accum_floats(nf), np.float32)
    orc.orc_ba_get_accumulators(h, abi.fp(acc))
    x = np.zeros(n); H = np.zeros((n, n)); b = np.zeros(n)
    orc.orc_ba_solve(h, 0, 0.1, abi.dp(x), abi.dp(H), abi.dp(b), None, None)
    step = np.zeros(npts, np.float32)
    orc.orc_ba_get_point_steps(h, abi.fp(step))
    orc.orc_ba_destroy(h)
    h = orc.orc_ba_create(C.byref(W))
    st, idp, rs, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    orc.orc_ba_optimize(h, 4, abi.dp(st), abi.fp(idp), abi.bp(rs), C.byref(oo))
    orc.orc_ba_destroy(h)
    J[ns == 1] = 0          # J of an OOB residual is unspecified
    return dict(input_digest=digest(win["pyrs"][0][0], win["u"], win["idepth"], win["res_target"]), energy=np.float64(e.value), J=J, newState=ns,
                newEnergy=ne, accum=acc, x=x, Hdiag=np.diag(H).copy(), b=b, point_step=step, opt_state=st, opt_idepth=idp, opt_res_state=rs,
                opt_iterations=np.int32(oo.iterations), opt_energy=np.float64(oo.lastEnergy))


def stereo_case():
    return synth.stereo_problem(w=320, h=240, npts=300, seed=4101)


def stereo_expected(orc):
    pr = stereo_case()
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    n = len(pr["u"])
    col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    orc.orc_immature_init_batch(abi.fp(left), pr["w"], pr["h"], n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth))
    P, d = abi.make_trace_points(n, pr["u"], pr["v"], col, wgt, gH, eth)
    st = np.zeros(n, np.uint8)
    K = np.array(pr["K"], np.float32)
    orc.orc_trace_stereo_batch(abi.fp(right), pr["w"], pr["h"], abi.fp(K), float(pr["calib"]["baseline"]), 1, C.byref(P), abi.bp(st))
    out = dict(input_digest=digest(left, right, pr["u"], pr["v"]), color=col, weights=wgt, gradH=gH, energyTH=eth, status=st)
    for k in ("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "quality", "lastTraceStatus", "lastTraceUV", "lastTracePixelInterval"):
        out[k] = d[k]
    return out


CASES = {"tracker": tracker_expected, "ba": ba_expected, "stereo": stereo_expected}
