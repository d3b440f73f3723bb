"""The fork's LIVE g2o factors: EdgeSE3PosePhotoDSO, EdgeLBASE3PosePhotoIdepthCamDSO,
EdgeTracePointUVDSO (src/FullSystem/dso_g2o_edge.cpp) and the call sites around them.

CPU part: the restatement in oracle/orc_g2o.cpp is checked against finite differences, against the known synthetic motion and
against the DSO-native path (both must land on the same pose / depth).  GPU part: libsdso_hip.so against the oracle —
per-edge values bit-exact (double and float), sums over edges to 1e-12 (order of summation), the LM driver step for step.
g2o itself is not in the reference tree and not version-pinned: everything g2o decides is "parity unpinned".
"""
import ctypes as C

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth


# ---------------------------------------------------------------------------------------------------------------- T5
@pytest.fixture(scope="module")
def prob():
    return synth.tracker_problem(w=640, h=480, npts=1500, seed=2002)


def _g2o_eval(L, prefix, prm, lvl, T_cull, T_vertex, aff_vertex):
    """sdso_g2o_track_eval_t the way trackNewestCoarse / calcRes derive it (host-only helper of either library)."""
    base = abi.TrackEval()
    Tc = abi.SE3.from_Rt(*T_cull)
    a = abi.Aff(*aff_vertex)
    getattr(L, prefix + "track_make_eval")(C.byref(prm), lvl, C.byref(Tc), C.byref(a), 1.0, C.byref(base))
    ev = abi.G2oTrackEval()
    ev.lvl, ev.w, ev.h = lvl, base.w, base.h
    ev.fx, ev.fy, ev.cx, ev.cy = base.fx, base.fy, base.cx, base.cy
    ev.Ki[:] = base.Ki[:]; ev.RKi[:] = base.RKi[:]; ev.t_cull[:] = base.t[:]
    ev.R[:] = np.asarray(T_vertex[0], np.float64).ravel().tolist(); ev.t[:] = np.asarray(T_vertex[1], np.float64).tolist()
    ev.ab[:] = base.affLL[:]
    ev.b0 = prm.ref_aff_g2l.b
    ev.cutoffTH, ev.huberTH = base.cutoffTH, base.huberTH
    return ev


def _oracle_edges(L, prob, lvl, ev):
    pc = prob["pc"][lvl]
    n = len(pc["u"])
    u, v, idp, col = [np.ascontiguousarray(pc[k], np.float32) for k in ("u", "v", "idepth", "color")]
    img = np.ascontiguousarray(prob["pyr_new"][lvl], np.float32)
    res = np.zeros(6); mask = np.zeros(max(n, 1), np.uint8); X = np.zeros((max(n, 1), 3), np.float32)
    ne = L.orc_g2o_track_add_edges(n, abi.fp(u), abi.fp(v), abi.fp(idp), abi.fp(col), abi.fp(img), C.byref(ev), abi.dp(res), abi.bp(mask), abi.fp(X))
    return ne, res, mask[:n], X[:n], (col, img)


def _oracle_lin(L, ev, mask, X, col, img, want=True):
    n = len(mask)
    H = np.zeros(64); b = np.zeros(8); chi = np.zeros(2)
    err = np.zeros(max(n, 1)); J = np.zeros((max(n, 1), 8))
    m = np.ascontiguousarray(mask); Xc = np.ascontiguousarray(X)
    L.orc_g2o_track_linearize(n, abi.bp(m), abi.fp(Xc), abi.fp(col), abi.fp(img), C.byref(ev), abi.dp(H), abi.dp(b), abi.dp(chi),
                              abi.dp(err) if want else None, abi.dp(J) if want else None)
    return H.reshape(8, 8), b, chi, err[:n], J[:n]


def test_oracle_track_edge_jacobian_is_derivative(oracle):
    """linearizeOplus against central differences of computeError under the vertices' own oplus (exp(d) * T, a += d, b += d).
    On a smooth image the analytic row uses the interpolated gradient, so agreement is to the interpolation error."""
    from test_oracle_tracker import _smooth_problem
    prob = _smooth_problem()
    prm = helpers.track_params(prob)
    prm.coarseCutoffTH = 200.0                                           # keep every edge: no photo-consistency in this problem
    lvl = 1
    T = synth.se3_exp(np.array([0.015, -0.008, 0.3, 0.003, -0.005, 0.0015]))
    aff = (0.01, 0.5)
    ev = _g2o_eval(oracle, "orc_", prm, lvl, T, T, aff)
    ne, res, mask, X, (col, img) = _oracle_edges(oracle, prob, lvl, ev)
    assert ne > 200 and res[1] == ne
    _, _, _, err0, J = _oracle_lin(oracle, ev, mask, X, col, img)
    eps = 1e-4
    num = np.zeros_like(J)
    for k in range(8):
        es = []
        for s in (+1, -1):
            d = np.zeros(6)
            a = list(aff)
            if k < 6:
                d[k] = s * eps
            else:
                a[k - 6] += s * eps
            Tp = synth.se3_mul(synth.se3_exp(d), T)
            evp = _g2o_eval(oracle, "orc_", prm, lvl, T, Tp, a)
            es.append(_oracle_lin(oracle, evp, mask, X, col, img)[3])
        num[:, k] = (es[0] - es[1]) / (2 * eps)
    on = mask.astype(bool)
    # rows: translation, rotation, (a, b).  The analytic photometric column is ab0 * (b0 - meas) = d/da of -(exp(a1-a0) meas + b1 - exp(.) b0)
    scale = np.abs(J[on]).max(axis=0)
    bad = np.abs(J[on] - num[on]) > 0.05 * scale + 0.15 * np.abs(J[on])
    assert bad.mean() < 0.03          # bilinear kinks hit a few percent of the edges
    assert np.allclose(J[on][:, 7], -1.0) and np.allclose(num[on][:, 7], -1.0, atol=1e-3)   # ab is cast to float


def test_oracle_g2o_tracker_recovers_motion_like_native(oracle, prob):
    prm = helpers.track_params(prob)
    Tn, affn, outn = helpers.oracle_track(oracle, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0))
    Tg, affg, outg = helpers.oracle_track(oracle, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0), fn="orc_g2o_track_newest_coarse")
    assert outg.good == 1 and outn.good == 1
    Rt, tt = prob["refToNew_true"]
    R, t = Tg.Rt()
    assert np.abs(t - tt).max() < 5e-3 and np.abs(R - Rt).max() < 1e-3
    Rn, tn = Tn.Rt()
    assert np.abs(t - tn).max() < 2e-3                                   # both optimisers land on the same pose
    L = prob["levels"]
    assert list(outg.iterations)[:L] == [2] * min(L, 5)                  # the fork hard-codes 2 iterations per level (:861)
    assert np.isfinite(list(outg.lastResiduals)[:min(L, 5)]).all()


def test_oracle_g2o_tracker_abort_leaves_outputs(oracle, prob):
    prm = helpers.track_params(prob)
    for i in range(5):
        prm.minResForAbort[i] = 0.01
    T, aff, out = helpers.oracle_track(oracle, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0), fn="orc_g2o_track_newest_coarse")
    assert out.good == 0 and np.array_equal(T.Rt()[0], np.eye(3)) and aff.a == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("lvl", [0, 2])
def test_gpu_track_edges_bit_exact(gpu_ctx, oracle, prob, lvl):
    prm = helpers.track_params(prob)
    prm.coarseCutoffTH = 1.5                                              # errors above +15 are dropped at creation
    gpu_ctx.upload_pyramid(410, prob["pyr_new"])
    gpu_ctx.set_ref(41, prob["pc"])
    T0 = synth.se3_exp(np.array([0.4, -0.1, 1.5, 0.01, 0.25, -0.01]))     # cull pose: part of the template leaves the image
    Tv = synth.se3_exp(np.array([0.01, -0.004, 0.2, 0.002, -0.003, 0.001]))
    aff = (0.015, 1.0)
    evo = _g2o_eval(oracle, "orc_", prm, lvl, T0, Tv, aff)
    evg = _g2o_eval(gpu_ctx.L, "sdso_", prm, lvl, T0, Tv, aff)
    assert bytes(evo) == bytes(evg)                                       # host tables identical
    ne, res, mask, X, (col, img) = _oracle_edges(oracle, prob, lvl, evo)
    n = len(mask)
    resg = np.zeros(6); neg = C.c_int(0); mg = np.zeros(n, np.uint8); Xg = np.zeros((n, 3), np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_add_edges(gpu_ctx.h, 41, 410, C.byref(evg), abi.dp(resg), C.byref(neg), abi.bp(mg), abi.fp(Xg)))
    assert neg.value == ne and np.array_equal(mg, mask) and np.array_equal(Xg, X)
    assert np.array_equal(resg, res, equal_nan=True)                      # counts and the flow indicators (summed in point order)
    assert 0 < ne < n and resg[5] > 0 and (mask == 0).sum() > resg[5] * ne + 0.5   # the cull and the saturation test both bite
    H, b, chi, err, J = _oracle_lin(oracle, evo, mask, X, col, img)
    Hg = np.zeros(64); bg = np.zeros(8); chig = np.zeros(2); errg = np.zeros(n); Jg = np.zeros((n, 8))
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_linearize(gpu_ctx.h, 41, 410, C.byref(evg), abi.dp(Hg), abi.dp(bg), abi.dp(chig), abi.dp(errg), abi.dp(Jg)))
    assert np.array_equal(errg, err) and np.array_equal(Jg, J)            # every edge: same doubles
    Hg = Hg.reshape(8, 8)
    d = np.sqrt(np.diag(H))
    assert np.abs((Hg - H) / np.outer(d, d)).max() < 1e-12                # sums over edges in another order
    assert np.abs((bg - b) / d).max() < 1e-10 * max(1.0, np.abs(b / d).max())
    assert np.allclose(chig, chi, rtol=1e-12)
    assert chi[1] < chi[0]                                                # some edges sit on the Huber branch


@pytest.mark.gpu
def test_gpu_g2o_tracker_matches_oracle(gpu_ctx, oracle, prob):
    prm = helpers.track_params(prob)
    gpu_ctx.upload_pyramid(411, prob["pyr_new"])
    gpu_ctx.set_ref(42, prob["pc"])
    To, affo, outo = helpers.oracle_track(oracle, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0), fn="orc_g2o_track_newest_coarse")
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0.0, 0.0); out = abi.TrackResult()
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_newest_coarse(gpu_ctx.h, 42, 411, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out)))
    assert out.good == outo.good == 1
    assert list(out.iterations) == list(outo.iterations) and out.evaluations == outo.evaluations and out.point_evals == outo.point_evals
    Ro, to = To.Rt(); R, t = T.Rt()
    assert np.abs(R - Ro).max() < 1e-9 and np.abs(t - to).max() < 1e-9    # double LM on sums that differ in the last bits
    assert abs(aff.a - affo.a) < 1e-9 and abs(aff.b - affo.b) < 1e-7
    assert np.allclose(list(out.lastResiduals), list(outo.lastResiduals), rtol=1e-6, equal_nan=True)
    assert np.array_equal(list(out.lastFlowIndicators), list(outo.lastFlowIndicators))
    # abort path: outputs untouched
    for i in range(5):
        prm.minResForAbort[i] = 0.01
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0.0, 0.0)
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_track_newest_coarse(gpu_ctx.h, 42, 411, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out)))
    assert out.good == 0 and np.array_equal(T.Rt()[0], np.eye(3))


# ---------------------------------------------------------------------------------------------------------------- B13
def _lba_case(nf=5, pts_per_kf=60, seed=3061):
    win = synth.ba_window(w=640, h=480, nf=nf, pts_per_kf=pts_per_kf, seed=seed)
    pair_R = np.zeros((nf * nf, 9), np.float32); pair_t = np.zeros((nf * nf, 3), np.float32); pair_ab = np.zeros((nf * nf, 2), np.float32)
    for h in range(nf):
        for t in range(nf):
            T = synth.se3_mul(win["poses"][t], synth.se3_inv(win["poses"][h]))           # Tth = Ttw * Twh (dso_g2o_edge.cpp:25-27)
            pair_R[h * nf + t] = T[0].astype(np.float32).ravel(); pair_t[h * nf + t] = T[1].astype(np.float32)
            a = np.exp(win["affs"][t][0] - win["affs"][h][0])
            pair_ab[h * nf + t] = (a, win["affs"][t][1] - a * win["affs"][h][1])
    rp = win["res_point"]
    d = dict(nf=nf, nr=len(rp), win=win, pair_R=pair_R, pair_t=pair_t, pair_ab=pair_ab,
             host_b0=np.array([win["affs"][h][1] for h in range(nf)], np.float64),
             frameEnergyTH=win["frameEnergyTH"].copy(), cam=[float(x) for x in win["K"]],
             host=np.ascontiguousarray(win["host"][rp], np.int32), target=np.ascontiguousarray(win["res_target"], np.int32),
             u=np.ascontiguousarray(win["u"][rp]), v=np.ascontiguousarray(win["v"][rp]),
             idepth=np.ascontiguousarray(win["idepth"][rp], np.float64),
             color=np.ascontiguousarray(win["color"][rp]), weights=np.ascontiguousarray(win["weights"][rp]))
    return d


def _lba_struct(d, frame_slots=None, dI=None):
    S = abi.G2oLba()
    keep = []
    S.nf, S.nr, S.w, S.h = d["nf"], d["nr"], 640, 480
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


def _lba_out(nr):
    return dict(error=np.zeros((nr, 8)), J=np.zeros((nr, 8, 13)), state=np.zeros(nr, np.uint8), energy=np.zeros((nr, 2), np.float32),
                cpt=np.zeros((nr, 3), np.float32), ih=np.zeros(nr, np.float32), lvl=np.zeros(nr, np.uint8))


def _lba_args(o):
    return (abi.dp(o["error"]), abi.dp(o["J"]), abi.bp(o["state"]), abi.fp(o["energy"]), abi.fp(o["cpt"]), abi.fp(o["ih"]), abi.bp(o["lvl"]))


def test_oracle_lba_edge_idepth_jacobian_and_energy(oracle):
    d = _lba_case()
    dI = [p[0] for p in d["win"]["pyrs"]]
    S, keep = _lba_struct(d, dI=dI)
    o = _lba_out(d["nr"])
    assert oracle.orc_g2o_lba_eval(C.byref(S), *_lba_args(o)) == 0
    inl = o["state"] == 0
    assert inl.mean() > 0.6
    ana = o["J"][:, :, 8]
    assert np.allclose(o["ih"][inl], np.maximum((ana[inl] ** 2).sum(1), 1e-10), rtol=1e-5)
    # d error / d idepth by central differences (one idepth vertex per residual), on smooth images: the stored central-difference
    # gradient then equals the derivative of the bilinear interpolant to a few per cent (photo-consistency is not needed here)
    ys, xs = np.meshgrid(np.arange(480, dtype=np.float64), np.arange(640, dtype=np.float64), indexing="ij")
    rs = np.random.RandomState(5)
    smooth = []
    for f in range(d["nf"]):
        im = np.full((480, 640), 120.0)
        for _ in range(10):
            lam, ang = rs.uniform(50, 200), rs.uniform(0, 2 * np.pi)
            im += rs.uniform(4, 10) * np.sin(2 * np.pi / lam * (np.cos(ang) * xs + np.sin(ang) * ys) + rs.uniform(0, 6.28))
        smooth.append(synth.make_pyramid(im.astype(np.float32), levels=1)[0])
    eps = 1e-4
    outs = []
    for s in (0, +1, -1):
        d2 = dict(d); d2["idepth"] = d["idepth"] + s * eps
        S2, k2 = _lba_struct(d2, dI=smooth)
        o2 = _lba_out(d["nr"])
        oracle.orc_g2o_lba_eval(C.byref(S2), *_lba_args(o2))
        outs.append(o2)
    ok = (outs[0]["state"] != 1) & (outs[1]["state"] != 1) & (outs[2]["state"] != 1)
    assert ok.mean() > 0.9
    num = (outs[1]["error"] - outs[2]["error"]) / (2 * eps)
    ana = outs[0]["J"][:, :, 8]
    sc = np.abs(ana[ok]).max()
    bad = np.abs(ana[ok] - num[ok]) > 0.03 * sc + 0.1 * np.abs(ana[ok])
    assert bad.mean() < 0.03
    # the 8x6 pose block is the derivative under a LEFT perturbation of Tth = Ttw * Twh (the target-frame convention of
    # Residuals.cpp:173-185).  (The fork attaches it to a vertex that perturbs Twh, and its photometric block keeps the
    # tracker's sign although the vertex here is the HOST's (a, b): the window optimiser built on these edges is steered by
    # g2o's accept / reject logic, which is one more reason it is not a parity target — DESIGN.md section 1.)
    win, nf = d["win"], d["nf"]

    def tables(delta):
        pR = np.zeros((nf * nf, 9), np.float32); pt = np.zeros((nf * nf, 3), np.float32)
        for h in range(nf):
            for t in range(nf):
                T = synth.se3_mul(synth.se3_exp(delta), synth.se3_mul(win["poses"][t], synth.se3_inv(win["poses"][h])))
                pR[h * nf + t] = T[0].astype(np.float32).ravel(); pt[h * nf + t] = T[1].astype(np.float32)
        return pR, pt
    num6 = np.zeros((d["nr"], 8, 6))
    ok6 = ok.copy()
    for k in range(6):
        es = []
        for s in (+1, -1):
            d2 = dict(d); v = np.zeros(6); v[k] = s * eps
            d2["pair_R"], d2["pair_t"] = tables(v)
            S2, k2 = _lba_struct(d2, dI=smooth)
            o2 = _lba_out(d["nr"])
            oracle.orc_g2o_lba_eval(C.byref(S2), *_lba_args(o2))
            es.append(o2)
            ok6 &= o2["state"] != 1
        num6[:, :, k] = (es[0]["error"] - es[1]["error"]) / (2 * eps)
    a6 = outs[0]["J"][:, :, :6][ok6].reshape(-1, 6); n6 = num6[ok6].reshape(-1, 6)
    for k in range(6):
        assert np.corrcoef(a6[:, k], n6[:, k])[0, 1] > 0.999 and abs(np.polyfit(a6[:, k], n6[:, k], 1)[0] - 1) < 0.01, k
    assert np.all(o["J"][inl][:, :, 7] == -1.0)
    # centre pixel projection is inside the target image
    assert np.all(o["cpt"][inl][:, 0] > 2) and np.all(o["cpt"][inl][:, 0] < 640 - 3)
    # residual energy = Huber energy of the 8 errors with the blended gradient weight, below the frame threshold for inliers
    assert np.all(o["energy"][inl][:, 0] <= 8 * 8 * 8) and np.all(o["energy"][inl][:, 0] == o["energy"][inl][:, 1])


@pytest.mark.gpu
def test_gpu_lba_edge_bit_exact(gpu_ctx, oracle):
    d = _lba_case()
    nf, nr = d["nf"], d["nr"]
    # provoke the early exits: pattern leaving the image at different pattern indices, points behind the camera, a tight energy threshold
    rs = np.random.RandomState(7)
    d["u"] = d["u"].copy(); d["v"] = d["v"].copy(); d["idepth"] = d["idepth"].copy()
    k = rs.choice(nr, 60, replace=False)
    d["u"][k[:15]] = rs.uniform(-30, 3, 15).astype(np.float32); d["v"][k[15:30]] = rs.uniform(470, 520, 15).astype(np.float32)
    d["idepth"][k[30:45]] *= rs.uniform(20, 200, 15)                    # far along the ray: leaves the image or flips behind the camera
    d["idepth"][k[45:60]] *= -rs.uniform(20, 200, 15)
    d["frameEnergyTH"][1] = 40.0
    for f in range(nf):
        gpu_ctx.upload_pyramid(420 + f, d["win"]["pyrs"][f][:1])
    S, keep = _lba_struct(d, frame_slots=[420 + f for f in range(nf)], dI=[p[0] for p in d["win"]["pyrs"]])
    oo, og = _lba_out(nr), _lba_out(nr)
    assert oracle.orc_g2o_lba_eval(C.byref(S), *_lba_args(oo)) == 0
    gpu_ctx.check(gpu_ctx.L.sdso_g2o_lba_eval(gpu_ctx.h, C.byref(S), *_lba_args(og)))
    for key in ("state", "lvl", "error", "J", "energy", "cpt", "ih"):
        assert np.array_equal(oo[key], og[key]), key
    assert set(np.unique(oo["state"])) == {0, 1, 2} and set(np.unique(oo["lvl"])) == {0, 1}


# ---------------------------------------------------------------------------------------------------------------- S3
@pytest.fixture(scope="module")
def pair():
    return synth.stereo_problem(w=640, h=480, npts=3000, seed=4001)


def _trace(oracle, pr, img, P, mode_right, gn_mode):
    K = np.array(pr["K"], np.float32)
    st = np.zeros(P.n, np.uint8)
    oracle.orc_trace_stereo_batch_gn(abi.fp(img), pr["w"], pr["h"], abi.fp(K), float(pr["calib"]["baseline"]), mode_right, C.byref(P), abi.bp(st), gn_mode)
    return st


def _points(oracle, pr, img, u, v):
    n = len(u)
    c = np.zeros((n, 8), np.float32); w_ = np.zeros((n, 8), np.float32); g = np.zeros((n, 4), np.float32); e = np.zeros(n, np.float32)
    oracle.orc_immature_init_batch(abi.fp(img), pr["w"], pr["h"], n, abi.fp(u), abi.fp(v), abi.fp(c), abi.fp(w_), abi.fp(g), abi.fp(e))
    return c, w_, g, e


def test_oracle_g2o_trace_refinement_close_to_native(oracle, pair):
    pr = pair
    img_l = np.ascontiguousarray(pr["pyr_l"][0], np.float32); img_r = np.ascontiguousarray(pr["pyr_r"][0], np.float32)
    u, v = pr["u"].astype(np.float32), pr["v"].astype(np.float32)
    c, w_, g, e = _points(oracle, pr, img_l, u, v)
    res = {}
    for mode in (0, 1):
        P, keep = abi.make_trace_points(len(u), u, v, c, w_, g, e)
        st = _trace(oracle, pr, img_r, P, 1, mode)
        res[mode] = (st, keep)
    good = (res[0][0] == 0) & (res[1][0] == 0)
    assert good.mean() > 0.6
    duv = np.abs(res[0][1]["lastTraceUV"].reshape(-1, 2)[good] - res[1][1]["lastTraceUV"].reshape(-1, 2)[good])
    assert np.median(duv[:, 0]) < 0.05 and np.percentile(duv[:, 0], 95) < 0.6      # same sub-pixel optimum, different damping
    true_id = pr["idepth_true"][good]
    rel = np.abs(res[1][1]["idepth_stereo"][good] - true_id) / true_id
    assert np.median(rel) < 0.05


@pytest.mark.gpu
def test_gpu_g2o_trace_refinement_bit_exact(gpu_ctx, oracle, pair):
    pr = pair
    img_l = np.ascontiguousarray(pr["pyr_l"][0], np.float32); img_r = np.ascontiguousarray(pr["pyr_r"][0], np.float32)
    gpu_ctx.upload_pyramid(430, pr["pyr_r"][:1])
    u, v = pr["u"].astype(np.float32), pr["v"].astype(np.float32)
    c, w_, g, e = _points(oracle, pr, img_l, u, v)
    K = np.array(pr["K"], np.float32)
    Po, ko = abi.make_trace_points(len(u), u, v, c, w_, g, e)
    Pg, kg = abi.make_trace_points(len(u), u, v, c, w_, g, e)
    so = _trace(oracle, pr, img_r, Po, 1, 1)
    sg = np.zeros(len(u), np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_set_gn_mode(gpu_ctx.h, 1))
    try:
        gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 430, abi.fp(K), float(pr["calib"]["baseline"]), 1, C.byref(Pg), abi.bp(sg)))
    finally:
        gpu_ctx.check(gpu_ctx.L.sdso_trace_set_gn_mode(gpu_ctx.h, 0))
    assert np.array_equal(so, sg)
    for key in ("lastTraceUV", "lastTracePixelInterval", "idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "quality", "lastTraceStatus"):
        assert np.array_equal(ko[key], kg[key], equal_nan=True), key
    # and the mode really is a different refinement than the native one
    Pn, kn = abi.make_trace_points(len(u), u, v, c, w_, g, e)
    sn = np.zeros(len(u), np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 430, abi.fp(K), float(pr["calib"]["baseline"]), 1, C.byref(Pn), abi.bp(sn)))
    assert not np.array_equal(kn["lastTraceUV"], kg["lastTraceUV"])
    assert gpu_ctx.L.sdso_trace_set_gn_mode(gpu_ctx.h, 7) != 0
