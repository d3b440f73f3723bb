"""Independent checks of the windowed-BA ORACLE (oracle/orc_ba.cpp) — CPU only.

The reference holds no golden vectors for this path and cannot be built here, so the oracle is a restatement by the same author as
the HIP kernels: GPU == oracle alone would only show that two restatements agree.  This file pins the oracle against things that
do not come from it: a double-precision numpy model of the photometric residual written from the reference's definitions
(src/FullSystem/Residuals.cpp:83-336, ResidualProjections.h:45-96, HessianBlocks.h:161-181, util/NumType.h:159-170), finite
differences of that model, a dense (frames + calibration + every inverse depth) normal-equation solve, the gauge freedoms of the
problem, and the scene the synthetic window was rendered from.

  1. every block of RawResidualJacobian (Jpdxi, Jpdc incl. SCALE_F/SCALE_C, Jpdd, JIdx, JabF, resF, the Huber/gradient weights and the
     energy) against central differences / direct evaluation of the model; the frame-state chain through adHost / adTarget
     (EnergyFunctional.cpp:41-119) against central differences with respect to FrameHessian::state itself
  2. accumulate (top + Schur) -> stitch -> solve -> resubstitute equals ONE dense double solve over all unknowns
  3. H * nullspace ~ 0 for the 6 pose gauge vectors and the scale vector (EnergyFunctional.cpp:775-835, HessianBlocks.cpp:78-123)
  4. FullSystem::optimize reduces the energy under the energy gate and moves poses / inverse depths towards the rendered truth
  5. marginalizePointsF followed by a solve on the remaining points equals the dense solve with the marginalised rows kept
     (weighted by setting_margWeightFac) (EnergyFunctional.cpp:663-736)
  6. truth mode: float accumulators of the oracle against double accumulation of the same terms
"""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi
import synth

PAT = synth.PATTERN.astype(np.float64)
SC = np.array([0.5, 0.5, 0.5, 1.0, 1.0, 1.0, 10.0, 1000.0])          # SCALE_XI_TRANS x3, SCALE_XI_ROT x3, SCALE_A, SCALE_B  (HessianBlocks.h:54-61)
SCALE_F = SCALE_C = 50.0
HUBER, C2 = 9.0, 2500.0
USE_GN, FIX_LAMBDA, ORTH_X_LATER = 64, 128, 2048
# RawResidualJacobian field offsets in the 74-float ABI order
O_RES, O_XI0, O_XI1, O_C0, O_C1, O_DD, O_I0, O_I1, O_A0, O_A1, O_I2, O_AI, O_A2 = 0, 8, 14, 20, 24, 28, 30, 38, 46, 54, 62, 66, 70


# ------------------------------------------------------------------ the model (numpy, float64; nothing from oracle/)
def frame_pose(win, f, state=None):
    """PRE_worldToCam = exp(state_scaled[0:6]) * worldToCam_evalPT   (HessianBlocks.h:161-181)."""
    st = win["state"][f] if state is None else state
    T0 = (win["evalPT"][f][:9].reshape(3, 3), win["evalPT"][f][9:])
    return synth.se3_mul(synth.se3_exp(st[:6] * SC[:6]), T0)


def rel_pose(Th, Tt):
    return synth.se3_mul(Tt, synth.se3_inv(Th))                          # host -> target


def project(K, T, u, v, idepth):
    """ResidualProjections.h: K * (R * K^-1 (u,v,1) + t * idepth), dehomogenised."""
    fx, fy, cx, cy = K
    X = T[0] @ np.array([(u - cx) / fx, (v - cy) / fy, 1.0]) + T[1] * idepth
    return np.array([fx * X[0] / X[2] + cx, fy * X[1] / X[2] + cy])


def bilinear(img, x, y):
    """getInterpolatedElement33 (util/globalFuncs.h:73-86) in double; img [h, w, 3]."""
    ix, iy = int(x), int(y)
    dx, dy = x - ix, y - iy
    p = img.astype(np.float64)
    return (dx * dy * p[iy + 1, ix + 1] + (dy - dx * dy) * p[iy + 1, ix] + (dx - dx * dy) * p[iy, ix + 1] + (1 - dx - dy + dx * dy) * p[iy, ix])


def aff_pair(win, h, t, state_h=None, state_t=None):
    """AffLight::fromToVecExposure (util/NumType.h:159-170), exposures 1."""
    sh = win["state"][h] if state_h is None else state_h
    st = win["state"][t] if state_t is None else state_t
    a = np.exp(st[6] * SC[6] - sh[6] * SC[6])
    return a, st[7] * SC[7] - a * sh[7] * SC[7]


def fej_window(win):
    """The same window with the linearisation point moved onto the current state (evalPT := PRE_worldToCam, pose part of the state
    0, state_zero := state, idepth_zero := idepth): FEJ Jacobians then ARE the Jacobians at the current state."""
    w = dict(win)
    ev, st = win["evalPT"].copy(), win["state"].copy()
    for f in range(win["nf"]):
        ev[f] = synth.se3_pack(frame_pose(win, f))
        st[f, :6] = 0
    w["evalPT"], w["state"], w["state_zero"] = ev, st, st.copy()
    w["idepth_zero"] = win["idepth"].copy()
    return w


def oracle_linearize(oracle, win, apply=True):
    W, keep = abi.make_ba_window(win, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    nr = win["nr"]
    J, ns, ne, nw = np.zeros((nr, 74), np.float32), np.zeros(nr, np.uint8), np.zeros(nr, np.float32), np.zeros(nr, np.float32)
    e = C.c_double(0)
    oracle.orc_ba_linearize(h, C.byref(e))
    oracle.orc_ba_get_linearization(h, abi.fp(J), abi.bp(ns), abi.fp(ne), abi.fp(nw), None, None)
    act = np.zeros(nr, np.uint8)
    if apply:
        oracle.orc_ba_apply_res(h)
        oracle.orc_ba_get_ef_jacobians(h, abi.fp(J))
        oracle.orc_ba_get_residual_state(h, None, abi.bp(act), None)
    return h, (W, keep), J.astype(np.float64), ns, ne, nw, act, e.value


@pytest.fixture(scope="module")
def win_small():
    return synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001)


# ------------------------------------------------------------------ 1. Jacobian blocks
def test_raw_residual_jacobian_blocks_against_central_differences(oracle, win_small):
    win = fej_window(win_small)
    h, keep, J, ns, ne, nw, act, e = oracle_linearize(oracle, win, apply=False)
    nf = win["nf"]
    adH, adT = np.zeros(nf * nf * 64), np.zeros(nf * nf * 64)
    oracle.orc_ba_get_tables(h, None, abi.dp(adH), abi.dp(adT), None)
    oracle.orc_ba_destroy(h)
    adH, adT = adH.reshape(nf * nf, 8, 8), adT.reshape(nf * nf, 8, 8)
    K = np.array(win["calib_value_scaled"], np.float64)
    rs = np.random.RandomState(0)
    cand = np.nonzero(ns == 0)[0]
    assert len(cand) > 500
    worst = dict(xi=0.0, c=0.0, d=0.0, idx=0.0, hw=0.0, res=0.0, ab=0.0, chain=0.0, energy=0.0)
    for i in rs.choice(cand, 60, replace=False):
        p, t = win["res_point"][i], win["res_target"][i]
        hst = win["host"][p]
        u, v, idp = float(win["u"][p]), float(win["v"][p]), float(win["idepth"][p])
        Th, Tt = frame_pose(win, hst), frame_pose(win, t)
        Tth = rel_pose(Th, Tt)
        img = win["pyrs"][t][0]
        # --- geometry at the centre pixel: Jpdxi (left perturbation of host->target), Jpdc (unscaled calibration value), Jpdd
        eps = 1e-6
        Jxi = np.zeros((2, 6))
        for k in range(6):
            d = np.zeros(6); d[k] = eps
            Jxi[:, k] = (project(K, synth.se3_mul(synth.se3_exp(d), Tth), u, v, idp) - project(K, synth.se3_mul(synth.se3_exp(-d), Tth), u, v, idp)) / (2 * eps)
        Jc = np.zeros((2, 4))
        for k in range(4):
            d = np.zeros(4); d[k] = 1e-4
            Jc[:, k] = (project(K + d, Tth, u, v, idp) - project(K - d, Tth, u, v, idp)) / 2e-4 * (SCALE_F if k < 2 else SCALE_C)
        Jd = (project(K, Tth, u, v, idp + 1e-7) - project(K, Tth, u, v, idp - 1e-7)) / 2e-7
        oxi = np.stack([J[i, O_XI0:O_XI0 + 6], J[i, O_XI1:O_XI1 + 6]])
        oc = np.stack([J[i, O_C0:O_C0 + 4], J[i, O_C1:O_C1 + 4]])
        od = J[i, O_DD:O_DD + 2]
        worst["xi"] = max(worst["xi"], np.abs(oxi - Jxi).max() / np.abs(Jxi).max())
        worst["c"] = max(worst["c"], np.abs(oc - Jc).max() / np.abs(Jc).max())
        worst["d"] = max(worst["d"], np.abs(od - Jd).max() / max(np.abs(Jd).max(), 1.0))
        # --- photometric part, pixel by pixel: weights, residual, JIdx, JabF, energy
        a, b = aff_pair(win, hst, t)
        b0 = win["state_zero"][hst][7] * SC[7]
        E = 0.0
        hw_all, g_all, raw_all = [], [], []
        for k in range(8):
            uv = project(K, Tth, u + PAT[k, 0], v + PAT[k, 1], idp)
            hit = bilinear(img, uv[0], uv[1])
            raw = hit[0] - (a * win["color"][p, k] + b)
            wgt = 0.5 * (np.sqrt(C2 / (C2 + hit[1] ** 2 + hit[2] ** 2)) + win["weights"][p, k])
            hub = 1.0 if abs(raw) < HUBER else HUBER / abs(raw)
            E += wgt * wgt * hub * raw * raw * (2 - hub)
            hw = (np.sqrt(hub) if hub < 1 else hub) * wgt
            hw_all.append(hw); g_all.append(hit[1:3]); raw_all.append(raw)
            gs = max(1.0, np.hypot(hit[1], hit[2]))
            worst["hw"] = max(worst["hw"], abs(J[i, O_A1 + k] - hw) / hw)
            worst["res"] = max(worst["res"], abs(J[i, O_RES + k] - raw * hw) / (gs * hw))           # in units of (gradient x pixel)
            worst["idx"] = max(worst["idx"], np.abs(np.array([J[i, O_I0 + k], J[i, O_I1 + k]]) - hit[1:3] * hw).max() / (gs * hw))
            worst["ab"] = max(worst["ab"], abs(J[i, O_A0 + k] - (win["color"][p, k] - b0) * hw) / (255 * hw))
        worst["energy"] = max(worst["energy"], abs(nw[i] - E) / max(E, 1.0))
        # --- chain through the adjoints: d r_centre / d FrameHessian::state of host and target (image linearised around the sample)
        k = 4
        uv0 = project(K, Tth, u, v, idp)
        jrel = np.concatenate([J[i, O_I0 + k] * oxi[0] + J[i, O_I1 + k] * oxi[1], [J[i, O_A0 + k], J[i, O_A1 + k]]])

        def r_centre(sh, st_):
            T = rel_pose(frame_pose(win, hst, sh), frame_pose(win, t, st_))
            aa, bb = aff_pair(win, hst, t, sh, st_)
            uv = project(K, T, u, v, idp)
            Ilin = bilinear(img, uv0[0], uv0[1])[0] + g_all[k] @ (uv - uv0)
            return hw_all[k] * (Ilin - (aa * win["color"][p, k] + bb))

        for which, ad in (("h", adH[hst + nf * t]), ("t", adT[hst + nf * t])):
            pred = ad @ jrel                                                               # d r / d state[i] = sum_j A[i][j] Jrel[j]
            for si in range(8):
                d = np.zeros(10); d[si] = 1e-6 / SC[si]
                sh_p, sh_m = win["state"][hst] + (d if which == "h" else 0), win["state"][hst] - (d if which == "h" else 0)
                st_p, st_m = win["state"][t] + (d if which == "t" else 0), win["state"][t] - (d if which == "t" else 0)
                fd = (r_centre(sh_p, st_p) - r_centre(sh_m, st_m)) / (2 * d[si])
                scale = max(np.abs(pred).max(), 1e-9)
                worst["chain"] = max(worst["chain"], abs(fd - pred[si]) / max(abs(fd), scale * 0.1, 1e-6))   # (rows of one block span two decades: float J)
    # J is float32 evaluated from float32 tables; the model is double: agreement to float rounding of the inputs
    assert worst["xi"] < 2e-4 and worst["c"] < 2e-4 and worst["d"] < 2e-4, worst
    assert worst["hw"] < 2e-3 and worst["res"] < 2e-3 and worst["idx"] < 2e-3 and worst["ab"] < 1e-3 and worst["energy"] < 5e-3, worst
    assert worst["chain"] < 2e-3, worst


# ------------------------------------------------------------------ dense system from the oracle's per-residual Jacobians
def dense_system(win, J, act, adH, adT, row_weight=None):
    """Normal equations over [calib 4 | frames 8 nf | idepth np] in double from the per-residual RawResidualJacobian rows.
    Returns H, b (b = J^T r: the reference solves H x = b and steps by -x)."""
    nf, npts = win["nf"], win["np"]
    n = 4 + 8 * nf
    N = n + npts
    H, b = np.zeros((N, N)), np.zeros(N)
    for i in np.nonzero(act)[0]:
        p, t = win["res_point"][i], win["res_target"][i]
        hst = win["host"][p]
        ji = J[i]
        I0, I1 = ji[O_I0:O_I0 + 8], ji[O_I1:O_I1 + 8]
        rel = np.zeros((8, 8))
        rel[:, :6] = np.outer(I0, ji[O_XI0:O_XI0 + 6]) + np.outer(I1, ji[O_XI1:O_XI1 + 6])
        rel[:, 6], rel[:, 7] = ji[O_A0:O_A0 + 8], ji[O_A1:O_A1 + 8]
        rows = np.zeros((8, N))
        rows[:, :4] = np.outer(I0, ji[O_C0:O_C0 + 4]) + np.outer(I1, ji[O_C1:O_C1 + 4])
        rows[:, 4 + 8 * hst:12 + 8 * hst] += rel @ adH[hst + nf * t].T
        rows[:, 4 + 8 * t:12 + 8 * t] += rel @ adT[hst + nf * t].T
        rows[:, n + p] = I0 * ji[O_DD] + I1 * ji[O_DD + 1]
        r = ji[O_RES:O_RES + 8].copy()
        if row_weight is not None:
            rows *= row_weight[i]; r *= row_weight[i]
        H += rows.T @ rows
        b += rows.T @ r
    return H, b


def frame_priors(win):
    """FrameHessian::getPrior (HessianBlocks.h:239-265) and the calibration prior (EnergyFunctional.cpp:117)."""
    nf = win["nf"]
    pr = np.zeros(4 + 8 * nf)
    pr[:4] = 5e9
    for f in range(nf):
        o = 4 + 8 * f
        if win["frameID"][f] == 0:
            pr[o:o + 3], pr[o + 3:o + 6], pr[o + 6], pr[o + 7] = 1e10, 1e11, 1e14, 1e14
        else:
            pr[o + 6], pr[o + 7] = win["affineOptModeA"], win["affineOptModeB"]
    return pr


def schur_solve(H, b, n, x_gauge=None):
    """Double-precision Schur solve of the dense system.  The reduced system is singular along the gauge directions the window does
    not fix (monocular scale; the rigid motion of the world once frame 0 and its pose prior are gone): it is solved on the complement
    (eigenvalues of the whitened matrix > 1e-9 of the largest), and the component of `x_gauge` along the dropped directions is added
    so that the back-substituted inverse-depth steps are comparable.  Returns x_c, x_d, S, g, number of dropped directions and the
    projector data (V_kept, sv)."""
    Hcc, Hcd, Hdd = H[:n, :n], H[:n, n:], np.diag(H[n:, n:]).copy()
    ok = Hdd > 0
    inv = np.where(ok, 1.0 / np.where(ok, Hdd, 1.0), 0.0)
    S = Hcc - (Hcd * inv) @ Hcd.T
    g = b[:n] - (Hcd * inv) @ b[n:]
    sv = 1.0 / np.sqrt(np.diag(S) + 10)
    w, V = np.linalg.eigh(S * np.outer(sv, sv))
    kept = w > 1e-9 * w.max()
    Vk, Vn = V[:, kept], V[:, ~kept]
    y = Vk @ ((Vk.T @ (sv * g)) / w[kept])
    if x_gauge is not None:
        y = y + Vn @ (Vn.T @ (x_gauge / sv))
    xc = sv * y
    xd = inv * (b[n:] - Hcd.T @ xc)
    return xc, xd, S, g, int((~kept).sum()), (Vk, Vn, sv)


def gauge_vectors(win):
    """The seven unobservable directions in the window's frame-state coordinates, from the poses alone: a global rigid motion of the
    world is worldToCam <- worldToCam * exp(eps), i.e. Ad(worldToCam) eps in the left tangent of each frame; a global scale moves the
    translations only (what FrameHessian::setStateZero differentiates numerically, HessianBlocks.cpp:78-123)."""
    nf, n = win["nf"], 4 + 8 * win["nf"]
    vecs = []
    for k in range(7):
        v = np.zeros(n)
        for f in range(nf):
            R, t = frame_pose(win, f, np.zeros(10))      # at the linearisation point (worldToCam_evalPT), where the Jacobians live
            if k < 6:
                Ad = np.zeros((6, 6)); Ad[:3, :3] = R; Ad[3:, 3:] = R; Ad[:3, 3:] = synth.hat(t) @ R   # Sophus SE3::Adj (se3.hpp:131-176)
                tw = Ad[:, k]
            else:
                tw = np.concatenate([t, np.zeros(3)])
            v[4 + 8 * f:4 + 8 * f + 6] = tw / SC[:6]
        vecs.append(v / np.linalg.norm(v))
    return np.array(vecs).T


def _tables(oracle, h, nf):
    adH, adT = np.zeros(nf * nf * 64), np.zeros(nf * nf * 64)
    oracle.orc_ba_get_tables(h, None, abi.dp(adH), abi.dp(adT), None)
    return adH.reshape(nf * nf, 8, 8), adT.reshape(nf * nf, 8, 8)


# ------------------------------------------------------------------ 2. accumulate + stitch + Schur + solve == one dense solve
@pytest.mark.parametrize("first_id", [0, 3])
def test_schur_reduced_solve_equals_dense_solve(oracle, win_small, first_id):
    win = dict(win_small)
    win["solverMode"] = USE_GN                                          # lambda = 0: the reduced and the full system are the same problem
    win["frameID"] = (np.arange(win["nf"]) + first_id).astype(np.int32)
    h, keep, J, ns, ne, nw, act, e = oracle_linearize(oracle, win)
    nf, npts, n = win["nf"], win["np"], 4 + 8 * win["nf"]
    adH, adT = _tables(oracle, h, nf)
    xo, Ho, bo, step = np.zeros(n), np.zeros((n, n)), np.zeros(n), np.zeros(npts, np.float32)
    oracle.orc_ba_solve(h, 0, 0.0, abi.dp(xo), abi.dp(Ho), abi.dp(bo), None, None)
    oracle.orc_ba_get_point_steps(h, abi.fp(step))
    oracle.orc_ba_destroy(h)
    H, b = dense_system(win, J, act, adH, adT)
    pr = frame_priors(win)
    dp = np.zeros(n)                                                    # delta_prior = state (EnergyFunctional.cpp:198), cDelta = 0 here
    for f in range(nf):
        dp[4 + 8 * f:12 + 8 * f] = win["state"][f][:8]
    H[:n, :n] += np.diag(pr)
    b[:n] += pr * dp
    xc, xd, S, g, ndrop, (Vk, Vn, sv) = schur_solve(H, b, n, x_gauge=xo)
    d = np.sqrt(np.abs(np.diag(S)))
    assert np.abs((Ho - S) / np.outer(d, d)).max() < 2e-5               # lastHS = stitched top - Schur, float accumulators vs double
    assert np.abs((bo - g) / d).max() < 2e-5 * max(1.0, np.abs(g / d).max())
    # exactly the gauge freedoms are undetermined: monocular scale, plus the rigid motion of the world without frame 0's pose prior
    assert ndrop == (1 if first_id == 0 else 7)
    # ... and they are those: the dropped directions lie in the span of the seven gauge vectors built from the poses (with frame 0
    # pinned, the one left is the scaling about camera 0, a combination of the scale and translation vectors)
    Ng = gauge_vectors(win) / sv[:, None]
    assert np.linalg.norm(Vn - Ng @ np.linalg.lstsq(Ng, Vn, rcond=None)[0], axis=0).max() < 1e-4
    tol = 2e-4
    assert np.abs(Vk.T @ ((xo - xc) / sv)).max() < tol * max(1.0, np.abs(Vk.T @ (xc / sv)).max())           # x on the determined subspace
    has = xd != 0
    assert has.sum() > 0.8 * npts
    assert np.abs(step[has] + xd[has]).max() < tol * np.abs(xd).max()   # resubstituteF: step = -x_d


# ------------------------------------------------------------------ 3. gauge freedoms
def test_reduced_system_annihilates_the_seven_gauge_vectors(oracle, win_small):
    win = fej_window(win_small)
    win["frameID"] = (np.arange(win["nf"]) + 3).astype(np.int32)       # frame 0 has left the window: no pose prior, HM = 0
    win["solverMode"] = USE_GN
    h, keep, J, ns, ne, nw, act, e = oracle_linearize(oracle, win)
    nf, n = win["nf"], 4 + 8 * win["nf"]
    Ho, xo = np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_solve(h, 0, 0.0, abi.dp(xo), abi.dp(Ho), None, None, None)
    oracle.orc_ba_destroy(h)
    vecs = list(gauge_vectors(win).T)
    rs = np.random.RandomState(1)
    ref = []
    for _ in range(20):                                                 # typical curvature along random pose directions
        v = np.zeros(n)
        for f in range(nf):
            v[4 + 8 * f:4 + 8 * f + 6] = rs.normal(0, 1, 6) / SC[:6]
        v /= np.linalg.norm(v)
        ref.append(v @ Ho @ v)
    ref = np.median(ref)
    for k, v in enumerate(vecs):
        assert abs(v @ Ho @ v) < 1e-5 * ref, (k, v @ Ho @ v, ref)
        assert np.linalg.norm(Ho @ v) < 3e-4 * np.sqrt(ref * np.abs(np.diag(Ho)).max()), k


# ------------------------------------------------------------------ 4. the GN driver against the rendered scene
def _pose_errors(win, state):
    """rotation angle and translation-direction error of the window's relative poses (frame 0 -> f) against the rendered ones."""
    rot, tr = [], []
    T0 = frame_pose(win, 0, state[0]); G0 = win["poses"][0]
    for f in range(1, win["nf"]):
        T = rel_pose(T0, frame_pose(win, f, state[f])); G = rel_pose(G0, win["poses"][f])
        dR = T[0] @ G[0].T
        rot.append(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
        tr.append(np.arccos(np.clip(T[1] @ G[1] / (np.linalg.norm(T[1]) * np.linalg.norm(G[1])), -1, 1)))
    return float(np.max(rot)), float(np.max(tr))


def test_optimize_descends_and_recovers_the_rendered_scene(oracle):
    win = dict(synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001, idepth_noise=0.10))
    rs = np.random.RandomState(3)
    st = win["state"].copy()
    st[1:, 3:6] += rs.normal(0, 2.5e-3, (win["nf"] - 1, 3))            # rotate every keyframe but the first by ~0.15 deg
    st[1:, 0:3] += rs.normal(0, 1.0e-2, (win["nf"] - 1, 3)) / 0.5      # and shift it by ~1 cm
    win["state"] = st
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    imgs = [p[0] for p in win["pyrs"]]

    def robust_energy(state, idepth):
        """Huber energy of all residuals at a given state, with the outlier clamp out of the way (frameEnergyTH = 1e9): a yardstick
        that does not move with setNewFrameEnergyTH, unlike the energies FullSystem::optimize itself compares."""
        w = dict(win)
        w["state"], w["idepth"], w["idepth_zero"] = state, idepth.astype(np.float32), idepth.astype(np.float32)
        w["frameEnergyTH"] = np.full(nf, 1e9, np.float32)
        W, keep = abi.make_ba_window(w, frame_slots=list(range(nf)), dI_list=imgs)
        h = oracle.orc_ba_create(C.byref(W))
        e, ns = C.c_double(0), np.zeros(nr, np.uint8)
        oracle.orc_ba_linearize(h, C.byref(e))
        oracle.orc_ba_get_linearization(h, None, abi.bp(ns), None, None, None, None)
        oracle.orc_ba_destroy(h)
        assert (ns == 1).sum() == 0                                     # same residual set at every state
        return e.value

    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=imgs)      # setting_forceAceptStep = true, the reference default
    energies, states, idepths = [robust_energy(win["state"], win["idepth"])], [win["state"].copy()], [win["idepth"].copy()]
    for its in (1, 2, 3, 4, 6):
        h = oracle.orc_ba_create(C.byref(W))
        so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h, its, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
        oracle.orc_ba_destroy(h)
        assert oo.iterations == its
        energies.append(robust_energy(so, io)); states.append(so.copy()); idepths.append(io.copy())
    for a, b in zip(energies[:-1], energies[1:]):
        assert b < a, energies                                          # every Gauss-Newton step lowers the robust photometric energy
    assert energies[-1] < 0.5 * energies[0], energies
    r0, t0 = _pose_errors(win, states[0])
    r1, t1 = _pose_errors(win, states[-1])
    assert r1 < 0.25 * r0 and t1 < 0.2 * t0, (r0, r1, t0, t1)            # measured: 4.8e-3 -> 7e-4 rad, 2.8e-2 -> 2.7e-3 rad

    def idepth_err(idp):                                                # monocular scale is free: compare after the best common factor
        s = np.median(idp / win["idepth_true"])
        return np.median(np.abs(idp / s - win["idepth_true"]) / win["idepth_true"])
    e0, e1 = idepth_err(idepths[0]), idepth_err(idepths[-1])
    assert e1 < 0.3 * e0, (e0, e1)


# ------------------------------------------------------------------ 5. marginalise-then-solve == keep-and-solve
@pytest.mark.parametrize("pattern", ["host0", "random"])
def test_marginalize_points_then_solve_equals_dense_solve_with_points_kept(oracle, win_small, pattern):
    win = dict(fej_window(win_small))                                   # delta = 0: res_toZeroF = resF and bM_top = bM
    win["solverMode"] = USE_GN
    nf, npts, n = win["nf"], win["np"], 4 + 8 * win["nf"]
    flag = (win["host"] == 0) if pattern == "host0" else (np.random.RandomState(7).uniform(size=npts) < 0.15)
    flag = flag.astype(np.uint8)
    h, keep, J, ns, ne, nw, act, e = oracle_linearize(oracle, win)
    adH, adT = _tables(oracle, h, nf)
    oracle.orc_ba_accumulate(h)                                         # a regular iteration first, as FullSystem does
    HM, bM = np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HM), abi.dp(bM))
    # flagPointsForRemoval re-linearises the residuals of the flagged points (resetOOB + linearize + applyRes, FullSystem.cpp:1012-1021)
    # against the energy threshold setNewFrameEnergyTH left behind: their active set / Jacobians are the ones after that pass
    Jm, actm = np.zeros((win["nr"], 74), np.float32), np.zeros(win["nr"], np.uint8)
    oracle.orc_ba_get_ef_jacobians(h, abi.fp(Jm))
    oracle.orc_ba_get_residual_state(h, None, abi.bp(actm), None)
    oracle.orc_ba_destroy(h)
    fl_r = flag[win["res_point"]] == 1
    J_all, act_all = np.where(fl_r[:, None], Jm.astype(np.float64), J), np.where(fl_r, actm, act)
    # the window that is left: flagged points and their residuals removed, HM / bM carry what they knew
    keep_p = flag == 0
    idx = np.nonzero(keep_p)[0]
    remap = -np.ones(npts, np.int64); remap[idx] = np.arange(len(idx))
    rk = keep_p[win["res_point"]]
    rest = dict(win)
    for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
        rest[k] = win[k][idx]
    rest["res_point"] = remap[win["res_point"][rk]].astype(np.int32); rest["res_target"] = win["res_target"][rk]; rest["res_state"] = win["res_state"][rk]
    rest["np"], rest["nr"] = len(idx), int(rk.sum())
    rest["HM"], rest["bM"] = HM, bM
    h2, keep2, J2, ns2, ne2, nw2, act2, e2 = oracle_linearize(oracle, rest)
    xo, Ho = np.zeros(n), np.zeros((n, n))
    oracle.orc_ba_solve(h2, 0, 0.0, abi.dp(xo), abi.dp(Ho), None, None, None)
    oracle.orc_ba_destroy(h2)
    assert np.array_equal(J2[act2 == 1], J[rk][act[rk] == 1])           # same linearisation of the kept residuals
    # dense: every residual kept, rows of marginalised points weighted by sqrt(setting_margWeightFac) = 0.5
    wrow = np.where(flag[win["res_point"]] == 1, 0.5, 1.0)
    H, b = dense_system(win, J_all, act_all, adH, adT, row_weight=wrow)
    pr = frame_priors(win)
    dpv = np.zeros(n)
    for f in range(nf):
        dpv[4 + 8 * f:12 + 8 * f] = win["state"][f][:8]
    H[:n, :n] += np.diag(pr); b[:n] += pr * dpv
    xc, xd, S, g, ndrop, (Vk, Vn, sv) = schur_solve(H, b, n, x_gauge=xo)
    d = np.sqrt(np.abs(np.diag(S)))
    assert np.abs((Ho - S) / np.outer(d, d)).max() < 5e-5
    assert ndrop == 1                                                   # scale is free (see the dense-solve test)
    assert np.abs(Vk.T @ ((xo - xc) / sv)).max() < 5e-4 * max(1.0, np.abs(Vk.T @ (xc / sv)).max())


# ------------------------------------------------------------------ 6. float accumulators against double accumulation
def test_float_accumulators_sit_within_float_rounding_of_double_accumulation(oracle, win_small):
    win = win_small
    nf, n = win["nf"], 4 + 8 * win["nf"]
    h, keep, J, ns, ne, nw, act, e = oracle_linearize(oracle, win)
    na = abi.accum_floats(nf)
    xf, Hf = np.zeros(n), np.zeros((n, n))
    oracle.orc_ba_solve(h, 0, 1e-5, abi.dp(xf), abi.dp(Hf), None, None, None)
    af = np.zeros(na, np.float32)
    oracle.orc_ba_get_accumulators(h, abi.fp(af))
    oracle.orc_set_acc64(1)
    try:
        xd, Hd, ad = np.zeros(n), np.zeros((n, n)), np.zeros(na)
        oracle.orc_ba_solve(h, 0, 1e-5, abi.dp(xd), abi.dp(Hd), None, None, None)
        oracle.orc_ba_get_accumulators_f64(h, abi.dp(ad))
    finally:
        oracle.orc_set_acc64(0)
    oracle.orc_ba_destroy(h)
    o0 = 0
    for name, cnt, w in (("topA", nf * nf, 91), ("topL", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8)):
        A, D = af[o0:o0 + cnt * w].reshape(-1, w).astype(np.float64), ad[o0:o0 + cnt * w].reshape(-1, w)
        m = np.abs(D).max(axis=1, keepdims=True)
        assert np.array_equal(m == 0, np.abs(A).max(axis=1, keepdims=True) == 0), name
        assert (np.abs(A - D) / np.maximum(m, 1e-30)).max() < 2e-5, name     # a few hundred float additions per bin
        o0 += cnt * w
    d = np.sqrt(np.abs(np.diag(Hd)))
    assert np.abs((xf - xd) * d).max() < 2e-4 * max(1.0, np.abs(xd * d).max())
    assert np.abs(xf - xd).max() > 0                                          # the mode really changes the arithmetic


# ------------------------------------------------------------------ 7. the solver-mode bits outside solveSystemF (round 6)
MOMENTUM, STEPMOMENTUM, ORTH_POINTMARG, ORTH_FULL = 512, 1024, 4, 8


def _oracle_optimize(oracle, win, its):
    W, keep = abi.make_ba_window(win, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    s, i, o = np.zeros((win["nf"], 10)), np.zeros(win["np"], np.float32), abi.BAOptResult()
    oracle.orc_ba_optimize(h, its, abi.dp(s), abi.fp(i), None, C.byref(o))
    st = np.zeros(32, np.float32)
    ns = oracle.orc_ba_get_step_trace(h, st.ctypes.data_as(C.POINTER(C.c_float)), 32)
    oracle.orc_ba_destroy(h)
    return s, i, o, st[:ns]


@pytest.mark.parametrize("bit", [MOMENTUM, STEPMOMENTUM])
def test_momentum_bits_leave_the_first_iteration_alone_and_act_from_the_second(oracle, win_small, bit):
    """SOLVER_MOMENTUM adds half of the PREVIOUS step (zero before the first iteration, FullSystemOptimize.cpp:327-343); SOLVER_STEPMOMENTUM
    scales by a stepsize that stays 1 while previousX is NaN (:929-938).  One iteration therefore ends exactly where the default mode's
    does; from the second on the trajectories differ, the stepsize stays inside [0.25, 2] and the loop still descends."""
    base = dict(win_small); base["solverMode"] = FIX_LAMBDA | ORTH_X_LATER
    mod = dict(base); mod["solverMode"] = base["solverMode"] | bit
    # (nf = 5: FullSystem::optimize keeps mnumOptIts as given only from four keyframes on, :875-876)
    s1, i1, o1, st1 = _oracle_optimize(oracle, base, 1)
    m1, j1, p1, mt1 = _oracle_optimize(oracle, mod, 1)
    assert o1.iterations == p1.iterations == 1
    assert np.array_equal(s1, m1) and np.array_equal(i1, j1) and mt1[0] == 1.0
    s4, i4, o4, st4 = _oracle_optimize(oracle, base, 4)
    m4, j4, p4, mt4 = _oracle_optimize(oracle, mod, 4)
    assert np.abs(s4 - m4).max() > 1e-7
    assert np.all(st4 == 1.0)
    assert mt4.min() >= 0.25 and mt4.max() <= 2.0
    if bit == STEPMOMENTUM:
        assert np.abs(mt4[1:] - 1.0).max() > 0.01, mt4
    assert p4.lastEnergy <= 1.02 * o1.lastEnergy                          # still a descent


@pytest.mark.parametrize("first_id", [0, 3])
def test_marginalize_points_with_the_nullspace_bits(oracle, win_small, first_id):
    """EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:707-731): SOLVER_ORTHOGONALIZE_POINTMARG projects the marginalised points'
    H, b off the seven gauge directions when frame 0 is not in the window (and is the plain statement when it is); SOLVER_ORTHOGONALIZE_FULL
    projects the whole prior.  Checked against gauge vectors that do not come from the oracle (gauge_vectors: poses alone)."""
    base = dict(win_small)
    nf, npts, n = base["nf"], base["np"], 4 + 8 * base["nf"]
    base["frameID"] = (np.arange(nf) + first_id).astype(np.int32)
    A = np.random.RandomState(5).normal(size=(n, 6))
    base["HM"] = (A @ A.T) * 1e9                                          # as large as the marginalised points' H, and not gauge-free
    base["bM"] = np.random.RandomState(6).normal(size=n) * 1e6
    flag = (base["host"] <= 1).astype(np.uint8)
    N = gauge_vectors(base)
    out = {}
    for name, bits in (("plain", 0), ("marg", ORTH_POINTMARG), ("full", ORTH_FULL)):
        win = dict(base); win["solverMode"] = FIX_LAMBDA | ORTH_X_LATER | bits
        h, keep, J, ns, ne, nw, act, e = oracle_linearize(oracle, win)
        oracle.orc_ba_accumulate(h)
        HM, bM = np.zeros((n, n)), np.zeros(n)
        oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HM), abi.dp(bM))
        oracle.orc_ba_destroy(h)
        out[name] = (HM, bM)
    HM0 = np.array(base["HM"])
    dH = out["plain"][0] - HM0
    lead = np.abs(dH).max()
    assert lead > 0
    P = N @ np.linalg.pinv(N)                                              # projector onto the span of the analytic gauge vectors
    Q = np.eye(n) - P
    if first_id == 0:
        assert np.array_equal(out["marg"][0], out["plain"][0]) and np.array_equal(out["marg"][1], out["plain"][1])
    else:
        # the marginalised points' H is gauge-free by construction up to float (no pose prior in it), so the projection moves it by
        # rounding-sized amounts only: the bit runs (not the same bits), and its result is the projected plain statement
        dm = out["marg"][0] - HM0
        assert not np.array_equal(out["marg"][0], out["plain"][0])
        assert np.abs(dm - (dH - P @ dH @ P)).max() <= 1e-4 * lead            # *H -= NNpiTS * *H * NNpiTS (:826)
        assert np.abs(N.T @ dm @ N).max() <= 1e-4 * lead
    # the whole prior projected: the (deliberately not gauge-free) HM0 loses its gauge part.  The oracle's nullspaces are central differences
    # (FrameHessian::setStateZero), 1e-6 off the analytic vectors, hence 1e-4
    HF, bF = out["full"]
    big = np.abs(out["plain"][0]).max()
    assert np.abs(HF - (out["plain"][0] - P @ out["plain"][0] @ P)).max() <= 1e-4 * big          # H -= P H P (:826), b -= P b (:823)
    assert np.abs(bF - Q @ out["plain"][1]).max() <= 1e-4 * np.abs(out["plain"][1]).max()
    assert np.abs(N.T @ out["plain"][0] @ N).max() > 1e-2 * big and np.abs(N.T @ HF @ N).max() <= 1e-4 * big
