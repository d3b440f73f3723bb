"""The small seeded cases behind tests/golden/*.npz: shared by make_golden.py (writes the expected outputs,
computed by the oracle) and tests/test_golden.py (checks oracle and HIP path against them).

These fixtures are NOT reference outputs — the reference cannot be built in this image and holds no
vectors for this path; they freeze the oracle's behaviour so that a change to it (or to the generator) is visible."""
import ctypes as C
import hashlib

import numpy as np

import helpers
from sdso_amd import abi
import synth


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


def g2o_lba_case():
    """The active residuals of ba_case() laid out as the fork builds its window graph (FullSystemOptimize.cpp:455-542)."""
    win = ba_case()
    nf = win["nf"]
    pair_R = np.zeros((nf * nf, 9), np.float32); pair_t = np.zeros((nf * nf, 3), np.float32); pair_ab = np.zeros((nf * nf, 2), np.float32)
    for h in range(nf):
        for t in range(nf):
            T = synth.se3_mul(win["poses"][t], synth.se3_inv(win["poses"][h]))
            pair_R[h * nf + t] = T[0].astype(np.float32).ravel(); pair_t[h * nf + t] = T[1].astype(np.float32)
            a = np.exp(win["affs"][t][0] - win["affs"][h][0])
            pair_ab[h * nf + t] = (a, win["affs"][t][1] - a * win["affs"][h][1])
    rp = win["res_point"]
    return dict(nf=nf, nr=len(rp), w=win["w"], h=win["h"], win=win, pair_R=pair_R, pair_t=pair_t, pair_ab=pair_ab,
                host_b0=np.array([win["affs"][h][1] for h in range(nf)], np.float64), frameEnergyTH=win["frameEnergyTH"].copy(),
                cam=[float(x) for x in win["K"]], host=np.ascontiguousarray(win["host"][rp], np.int32),
                target=np.ascontiguousarray(win["res_target"], np.int32), u=np.ascontiguousarray(win["u"][rp]), v=np.ascontiguousarray(win["v"][rp]),
                idepth=np.ascontiguousarray(win["idepth"][rp], np.float64), color=np.ascontiguousarray(win["color"][rp]),
                weights=np.ascontiguousarray(win["weights"][rp]))


def g2o_lba_struct(d, frame_slots=None, dI=None):
    S = abi.G2oLba()
    keep = []
    S.nf, S.nr, S.w, S.h = d["nf"], d["nr"], d["w"], d["h"]
    S.cam[:] = d["cam"]
    for k in ("pair_R", "pair_t", "pair_ab", "frameEnergyTH", "u", "v", "color", "weights"):
        a = np.ascontiguousarray(d[k], np.float32); keep.append(a); setattr(S, k, abi.fp(a))
    for k in ("host_b0", "idepth"):
        a = np.ascontiguousarray(d[k], np.float64); keep.append(a); setattr(S, k, abi.dp(a))
    for k in ("host", "target"):
        a = np.ascontiguousarray(d[k], np.int32); keep.append(a); setattr(S, k, abi.ip(a))
    if frame_slots is not None:
        fs = np.ascontiguousarray(frame_slots, np.int32); keep.append(fs); S.frame_slot = abi.ip(fs)
    if dI is not None:
        ptrs = (abi.c_float_p * len(dI))()
        for i, a in enumerate(dI):
            a = np.ascontiguousarray(a, np.float32); keep.append(a); ptrs[i] = abi.fp(a)
        keep.append(ptrs); S.dI = C.cast(ptrs, C.POINTER(abi.c_float_p))
    return S, keep


def g2o_eval(L, prefix, prm, lvl, T_cull, T_vertex, aff_vertex):
    """sdso_g2o_track_eval_t the way the fork-live calcRes / trackNewestCoarse derive it (host-only helper of either library)."""
    base = abi.TrackEval()
    getattr(L, prefix + "track_make_eval")(C.byref(prm), lvl, C.byref(abi.SE3.from_Rt(*T_cull)), C.byref(abi.Aff(*aff_vertex)), 1.0, C.byref(base))
    ev = abi.G2oTrackEval()
    ev.lvl, ev.w, ev.h = lvl, base.w, base.h
    ev.fx, ev.fy, ev.cx, ev.cy = base.fx, base.fy, base.cx, base.cy
    ev.Ki[:] = base.Ki[:]; ev.RKi[:] = base.RKi[:]; ev.t_cull[:] = base.t[:]
    ev.R[:] = np.asarray(T_vertex[0], np.float64).ravel().tolist(); ev.t[:] = np.asarray(T_vertex[1], np.float64).tolist()
    ev.ab[:] = base.affLL[:]
    ev.b0 = prm.ref_aff_g2l.b
    ev.cutoffTH, ev.huberTH = base.cutoffTH, base.huberTH
    return ev


G2O_STRIDE = 4
G2O_CULL = np.array([0.05, -0.02, 0.4, 0.004, 0.02, -0.003])
G2O_VERTEX = np.array([0.015, -0.008, 0.3, 0.003, -0.005, 0.0015])


def g2o_expected(orc):
    """The fork's live factors (oracle/orc_g2o.cpp): tracker edge system per level, the fork-live tracker, the window edge, the g2o trace refinement."""
    prob, prm, _ = tracker_case()
    out = {"input_digest": digest(prob["pyr_new"][0], prob["pc"][0]["u"], prob["pc"][0]["idepth"])}
    Tc, Tv = synth.se3_exp(G2O_CULL), synth.se3_exp(G2O_VERTEX)
    for lvl in range(prob["levels"]):
        ev = g2o_eval(orc, "orc_", prm, lvl, Tc, Tv, (0.01, 1.0))
        pc = prob["pc"][lvl]
        n = len(pc["u"])
        u, v, idp, col = [np.ascontiguousarray(pc[k], np.float32) for k in ("u", "v", "idepth", "color")]
        img = np.ascontiguousarray(prob["pyr_new"][lvl], np.float32)
        res = np.zeros(6); mask = np.zeros(max(n, 1), np.uint8); X = np.zeros((max(n, 1), 3), np.float32)
        orc.orc_g2o_track_add_edges(n, abi.fp(u), abi.fp(v), abi.fp(idp), abi.fp(col), abi.fp(img), C.byref(ev), abi.dp(res), abi.bp(mask), abi.fp(X))
        H = np.zeros(64); b = np.zeros(8); chi = np.zeros(2); err = np.zeros(max(n, 1)); J = np.zeros((max(n, 1), 8))
        orc.orc_g2o_track_linearize(n, abi.bp(mask), abi.fp(X), abi.fp(col), abi.fp(img), C.byref(ev), abi.dp(H), abi.dp(b), abi.dp(chi), abi.dp(err), abi.dp(J))
        out["res%d" % lvl], out["mask%d" % lvl], out["Xref%d" % lvl] = res, np.packbits(mask[:n]), X[:n][::G2O_STRIDE]
        out["H%d" % lvl], out["b%d" % lvl], out["chi%d" % lvl] = H.reshape(8, 8), b, chi
        out["err%d" % lvl], out["J%d" % lvl] = err[:n][::G2O_STRIDE], J[:n][::G2O_STRIDE]       # every 4th edge keeps the fixture small
    To, affo, oo = helpers.oracle_track(orc, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0), fn="orc_g2o_track_newest_coarse")
    out["track_R"], out["track_t"] = To.Rt()
    out["track_aff"] = np.array([affo.a, affo.b])
    out["track_iterations"] = np.array(list(oo.iterations), np.int32)
    out["track_evaluations"] = np.int32(oo.evaluations)
    out["track_lastResiduals"] = np.array(list(oo.lastResiduals))
    # window edge
    d = g2o_lba_case()
    S, keep = g2o_lba_struct(d, dI=[p[0] for p in d["win"]["pyrs"]])
    nr = d["nr"]
    e = np.zeros((nr, 8)); Jl = np.zeros((nr, 8, 13)); st = np.zeros(nr, np.uint8); en = np.zeros((nr, 2), np.float32)
    cpt = np.zeros((nr, 3), np.float32); ih = np.zeros(nr, np.float32); lv = np.zeros(nr, np.uint8)
    orc.orc_g2o_lba_eval(C.byref(S), abi.dp(e), abi.dp(Jl), abi.bp(st), abi.fp(en), abi.fp(cpt), abi.fp(ih), abi.bp(lv))
    out.update(lba_error=e, lba_J=Jl[::G2O_STRIDE], lba_state=st, lba_energy=en, lba_cpt=cpt, lba_idepth_hessian=ih, lba_level=lv)
    # trace refinement, g2o mode
    pr = stereo_case()
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    n = len(pr["u"])
    col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    orc.orc_immature_init_batch(abi.fp(left), pr["w"], pr["h"], n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth))
    P, dd = abi.make_trace_points(n, pr["u"], pr["v"], col, wgt, gH, eth)
    stt = np.zeros(n, np.uint8)
    K = np.array(pr["K"], np.float32)
    orc.orc_trace_stereo_batch_gn(abi.fp(right), pr["w"], pr["h"], abi.fp(K), float(pr["calib"]["baseline"]), 1, C.byref(P), abi.bp(stt), 1)
    out["trace_status"] = stt
    for k in ("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "lastTraceUV", "lastTracePixelInterval"):
        out["trace_" + k] = dd[k]
    return out


def ba_dropped_case():
    """ba_case() after EnergyFunctional::dropResidual went over it (EnergyFunctional.cpp:524-533): residualsAll lists no longer in target
    order, points with a history (numGoodResiduals, maxRelBaseline) and some old residuals (isNew = false)."""
    win, kept = helpers.drop_residuals(ba_case(), seed=17, drop_frac=0.3)
    rs = np.random.RandomState(9)
    win["numGoodResiduals"] = rs.randint(0, 6, win["np"]).astype(np.int32)
    win["maxRelBaseline"] = (rs.uniform(0, 0.3, win["np"]) * (rs.rand(win["np"]) < 0.6)).astype(np.float32)
    win["res_isNew"] = (rs.rand(win["nr"]) < 0.75).astype(np.uint8)
    return win


def ba_dropped_expected(orc):
    """Per-point sums / back-substitution in residualsAll order and the post-state of FullSystem::optimize (FullSystemOptimize.cpp:52-87,
    :142-203, :997-1041)."""
    win = ba_dropped_case()
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
    h = orc.orc_ba_create(C.byref(W))
    orc.orc_ba_linearize(h, None); orc.orc_ba_apply_res(h)
    x = np.zeros(n)
    orc.orc_ba_solve(h, 0, 0.1, abi.dp(x), None, None, None, None)
    pt = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    orc.orc_ba_get_point_terms(h, *[abi.fp(a) for a in pt])
    step = np.zeros(npts, np.float32)
    orc.orc_ba_get_point_steps(h, abi.fp(step))
    orc.orc_ba_destroy(h)
    h = orc.orc_ba_create(C.byref(W))
    oo = abi.BAOptResult()
    orc.orc_ba_optimize(h, 4, None, None, None, C.byref(oo))
    P, d = abi.make_post_state(nf, npts, nr)
    orc.orc_ba_get_post_state(h, C.byref(P))
    orc.orc_ba_destroy(h)
    out = dict(input_digest=digest(win["pyrs"][0][0], win["u"], win["idepth"], win["res_target"], win["res_isNew"]), x=x, point_step=step,
               HdiF=pt[0], bdSumF=pt[1], Hdd_accAF=pt[2], bd_accAF=pt[3], Hcd_accAF=pt[4], opt_iterations=np.int32(oo.iterations),
               opt_n_toRemove=np.int32(P.n_toRemove), opt_resInA=np.int32(P.resInA))
    for k in ("idepth", "HdiF", "idepth_hessian", "maxRelBaseline", "numGoodResiduals", "state_state", "isActiveAndIsGoodNEW", "toRemove", "centerProjectedTo",
              "state", "state_zero", "evalPT", "frameEnergyTH", "lastX"):
        out["post_" + k] = d[k]
    return out


CASES = {"tracker": tracker_expected, "ba": ba_expected, "ba_dropped": ba_dropped_expected, "stereo": stereo_expected, "g2o": g2o_expected}
