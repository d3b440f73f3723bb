"""Static stereo (BASELINE configs[3]): oracle sanity on CPU, GPU parity through the C-ABI.

traceStereo works per point with a fixed float operation order, so every output — status enum,
lastTraceUV, the idepth interval, idepth_stereo, quality — must be bit-identical to the CPU path."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi
import synth

FIELDS = ("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "quality", "lastTraceStatus", "lastTraceUV", "lastTracePixelInterval")


@pytest.fixture(scope="module")
def pair():
    return synth.stereo_problem(w=640, h=480, npts=3000, seed=4001)


def _oracle_init(oracle, pr, img, u, v):
    n = len(u)
    col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    oracle.orc_immature_init_batch(abi.fp(img), pr["w"], pr["h"], n, abi.fp(u), abi.fp(v), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth))
    return col, wgt, gH, eth


def _oracle_trace(oracle, pr, img, P, mode_right):
    K = np.array(pr["K"], np.float32)
    st = np.zeros(P.n, np.uint8)
    oracle.orc_trace_stereo_batch(abi.fp(img), pr["w"], pr["h"], abi.fp(K), float(pr["calib"]["baseline"]), mode_right, C.byref(P), abi.bp(st))
    return st


def test_oracle_trace_recovers_depth(oracle, pair):
    pr = pair
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    col, wgt, gH, eth = _oracle_init(oracle, pr, left, pr["u"], pr["v"])
    assert np.isfinite(eth).all() and eth[0] == 8 * 144
    P, d = abi.make_trace_points(len(pr["u"]), pr["u"], pr["v"], col, wgt, gH, eth)
    st = _oracle_trace(oracle, pr, right, P, 1)
    good = st == 0
    assert good.mean() > 0.5
    rel = np.abs(d["idepth_stereo"][good] - pr["idepth_true"][good]) / pr["idepth_true"][good]
    assert np.median(rel) < 0.05                       # sub-pixel disparity -> a few per cent in inverse depth
    assert set(np.unique(st)) <= {0, 1, 2, 3, 4}
    # status bookkeeping: a second OUTLIER in a row becomes OOB (ImmaturePoint.cpp:418-421)
    out = np.nonzero(st == 2)[0]
    if len(out):
        P2, d2 = abi.make_trace_points(len(pr["u"]), pr["u"], pr["v"], col, wgt, gH, eth)
        d2["lastTraceStatus"][:] = 2
        st2 = _oracle_trace(oracle, pr, right, P2, 1)
        assert (st2[out] == 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("size", ["small", "kitti"])
def test_gpu_trace_bit_exact(gpu_ctx, oracle, pair, size):
    pr = pair if size == "small" else synth.stereo_problem(w=1232, h=368, npts=20000, seed=4001)     # configs[3]
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    gpu_ctx.upload_pyramid(80, [left]); gpu_ctx.upload_pyramid(81, [right])
    n = len(pr["u"])
    co, wo, go, eo = _oracle_init(oracle, pr, left, pr["u"], pr["v"])
    cg, wg, gg, eg = np.zeros_like(co), np.zeros_like(wo), np.zeros_like(go), np.zeros_like(eo)
    gpu_ctx.check(gpu_ctx.L.sdso_immature_init_batch(gpu_ctx.h, 80, n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(cg), abi.fp(wg), abi.fp(gg), abi.fp(eg)))
    assert np.array_equal(co, cg) and np.array_equal(wo, wg) and np.array_equal(go, gg) and np.array_equal(eo, eg)
    K = np.array(pr["K"], np.float32)
    bl = float(pr["calib"]["baseline"])
    # L -> R, fresh points (idepth_max = NaN: full 0.027*(w+h) pixel search)
    Po, do = abi.make_trace_points(n, pr["u"], pr["v"], co, wo, go, eo)
    Pg, dg = abi.make_trace_points(n, pr["u"], pr["v"], co, wo, go, eo)
    so = _oracle_trace(oracle, pr, right, Po, 1)
    sg = np.zeros(n, np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 81, abi.fp(K), bl, 1, C.byref(Pg), abi.bp(sg)))
    assert np.array_equal(so, sg)
    for k in FIELDS:
        assert np.array_equal(do[k], dg[k], equal_nan=True), k
    assert (so == 0).mean() > 0.5
    # R -> L back-trace from the traced position with a finite interval (FullSystem.cpp:590-600 stereoMatch)
    good = np.nonzero(so == 0)[0]
    ub, vb = do["lastTraceUV"][good, 0].copy(), do["lastTraceUV"][good, 1].copy()
    inb = (ub > 6) & (vb > 6) & (ub < pr["w"] - 7) & (vb < pr["h"] - 7)
    ub, vb, good = ub[inb], vb[inb], good[inb]
    c2, w2, g2, e2 = _oracle_init(oracle, pr, right, ub, vb)
    imin = (do["idepth_stereo"][good] * 0.1).astype(np.float32); imax = (do["idepth_stereo"][good] * 1.9).astype(np.float32)
    Po2, do2 = abi.make_trace_points(len(ub), ub, vb, c2, w2, g2, e2, imin, imax)
    Pg2, dg2 = abi.make_trace_points(len(ub), ub, vb, c2, w2, g2, e2, imin, imax)
    so2 = _oracle_trace(oracle, pr, left, Po2, 0)
    sg2 = np.zeros(len(ub), np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 80, abi.fp(K), bl, 0, C.byref(Pg2), abi.bp(sg2)))
    assert np.array_equal(so2, sg2)
    for k in FIELDS:
        assert np.array_equal(do2[k], dg2[k], equal_nan=True), k
    ok = so2 == 0
    assert ok.mean() > 0.5
    assert np.median(np.abs(do2["lastTraceUV"][ok, 0] - pr["u"][good][ok])) < 1.0      # left-right consistency


@pytest.mark.gpu
def test_gpu_trace_edge_cases(gpu_ctx, oracle, pair):
    pr = pair
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    gpu_ctx.upload_pyramid(81, [right])
    K = np.array(pr["K"], np.float32)
    bl = float(pr["calib"]["baseline"])
    # points at the border (OOB at once), at the far right (search leaves the image), with tiny and huge intervals (SKIPPED / BADCONDITION)
    u = np.array([5.0, 630.0, 320.0, 320.0, 320.0, 100.0, 5.5, 300.0], np.float32)
    v = np.array([240.0, 240.0, 6.0, 240.0, 240.0, 470.0, 5.5, 200.0], np.float32)
    col, wgt, gH, eth = _oracle_init(oracle, pr, left, u, v)
    imin = np.array([0, 0, 0, 0.05, 0.01, 0, 0, 0.02], np.float32)
    imax = np.array([np.nan, np.nan, np.nan, 0.0501, 0.2, np.nan, np.nan, 0.0232], np.float32)
    for prev in (5, 2):
        Po, do = abi.make_trace_points(len(u), u, v, col, wgt, gH, eth, imin, imax)
        Pg, dg = abi.make_trace_points(len(u), u, v, col, wgt, gH, eth, imin, imax)
        do["lastTraceStatus"][:] = prev; dg["lastTraceStatus"][:] = prev
        so = _oracle_trace(oracle, pr, right, Po, 1)
        sg = np.zeros(len(u), np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 81, abi.fp(K), bl, 1, C.byref(Pg), abi.bp(sg)))
        assert np.array_equal(so, sg)
        for k in FIELDS:
            assert np.array_equal(do[k], dg[k], equal_nan=True), k
        assert 1 in so and 3 in so
    # n = 0 and an unknown slot
    P0, d0 = abi.make_trace_points(0, u[:0], v[:0], col[:0], wgt[:0], gH[:0], eth[:0])
    gpu_ctx.check(gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 81, abi.fp(K), bl, 1, C.byref(P0), None))
    assert gpu_ctx.L.sdso_trace_stereo_batch(gpu_ctx.h, 999, abi.fp(K), bl, 1, C.byref(P0), None) == -1


@pytest.mark.gpu
def test_gpu_stereo_match_left_right_left(gpu_ctx, oracle, pair):
    """sdso_stereo_match_batch = the L->R->L pattern of FullSystem::stereoMatch / makeCoarseDepthL0 in one call; every
    intermediate stays on the device.  Must equal the four oracle steps (ctor, trace, ctor at lastTraceUV, trace back)."""
    pr = pair
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    gpu_ctx.upload_pyramid(80, [left]); gpu_ctx.upload_pyramid(81, [right])
    n = len(pr["u"])
    K = np.array(pr["K"], np.float32); bl = float(pr["calib"]["baseline"])
    for interval in (False, True):     # fresh points (stereoMatch) / prior interval 0.1..1.9 x idepth (makeCoarseDepthL0)
        imin = (pr["idepth_true"] * 0.1).astype(np.float32) if interval else None
        imax = (pr["idepth_true"] * 1.9).astype(np.float32) if interval else None
        # oracle, step by step
        co, wo, go, eo = _oracle_init(oracle, pr, left, pr["u"], pr["v"])
        Po, do = abi.make_trace_points(n, pr["u"], pr["v"], co, wo, go, eo, imin, imax)
        sf = _oracle_trace(oracle, pr, right, Po, 1)
        good = np.nonzero(sf == 0)[0]
        ub, vb = do["lastTraceUV"][good, 0].copy(), do["lastTraceUV"][good, 1].copy()
        c2, w2, g2, e2 = _oracle_init(oracle, pr, right, ub, vb)
        Pb, db = abi.make_trace_points(len(good), ub, vb, c2, w2, g2, e2, None if imin is None else imin[good], None if imax is None else imax[good])
        sb = _oracle_trace(oracle, pr, left, Pb, 0)
        # device, one call
        M = abi.StereoMatch()
        out = dict(status_fwd=np.zeros(n, np.uint8), status_back=np.zeros(n, np.uint8), idepth_stereo=np.zeros(n, np.float32),
                   idepth_min_out=np.zeros(n, np.float32), idepth_max_out=np.zeros(n, np.float32), fwd_uv=np.zeros((n, 2), np.float32),
                   back_uv=np.zeros((n, 2), np.float32))
        M.n = n; M.u = abi.fp(pr["u"]); M.v = abi.fp(pr["v"])
        if interval:
            M.idepth_min_stereo = abi.fp(imin); M.idepth_max_stereo = abi.fp(imax)
            M.back_idepth_min_stereo = abi.fp(imin); M.back_idepth_max_stereo = abi.fp(imax)
        for k, a in out.items():
            setattr(M, k, abi.bp(a) if a.dtype == np.uint8 else abi.fp(a))
        gpu_ctx.check(gpu_ctx.L.sdso_stereo_match_batch(gpu_ctx.h, 80, 81, abi.fp(K), bl, 1, C.byref(M)))
        assert np.array_equal(out["status_fwd"], sf)
        assert np.array_equal(out["idepth_stereo"][good], do["idepth_stereo"][good])
        assert np.array_equal(out["idepth_min_out"][good], do["idepth_min_stereo"][good]) and np.array_equal(out["idepth_max_out"][good], do["idepth_max_stereo"][good])
        assert np.array_equal(out["fwd_uv"], do["lastTraceUV"])
        assert (out["status_back"][sf != 0] == 255).all() and np.array_equal(out["status_back"][good], sb)
        assert np.array_equal(out["back_uv"][good], db["lastTraceUV"])
        # the caller's accept rule (FullSystem.cpp:598-601) keeps most points and the kept ones are consistent by construction
        ok = (sb == 0) & (np.abs(pr["u"][good] - db["lastTraceUV"][:, 0]) < 1) & (1.0 / do["idepth_stereo"][good] > 0) & (1.0 / do["idepth_stereo"][good] < 70)
        assert ok.mean() > 0.5


def _trace_on_case(n=3000, seed=2041):
    """A host keyframe, the newest frame displaced by a general motion (rotation + translation, so the epipolar lines are
    neither horizontal nor parallel), immature points on the host."""
    prob = synth.tracker_problem(w=640, h=480, npts=n, seed=seed, motion=(0.25, -0.06, 0.10, 0.006, -0.01, 0.012), aff=(0.02, 1.5))
    u, v, idp = prob["points"]
    R, t = prob["refToNew_true"]
    fx, fy, cx, cy = [np.float32(x) for x in prob["K"]]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    Ki = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    G = abi.TraceGeom()
    KRKi = (K @ R.astype(np.float32) @ Ki).astype(np.float32)
    Kt = (K @ t.astype(np.float32)).astype(np.float32)
    G.KRKi[:] = list(KRKi.ravel()); G.Kt[:] = list(Kt); G.aff[:] = [float(np.exp(np.float32(0.02))), 1.5]
    return prob, u.astype(np.float32), v.astype(np.float32), idp.astype(np.float32), G


def test_oracle_trace_on_brackets_true_depth(oracle):
    prob, u, v, idp, G = _trace_on_case()
    host = np.ascontiguousarray(prob["pyr_ref"][0]); new = np.ascontiguousarray(prob["pyr_new"][0])
    pr = dict(w=640, h=480)
    col, wgt, gH, eth = _oracle_init(oracle, pr, host, u, v)
    P, d = abi.make_trace_points(len(u), u, v, col, wgt, gH, eth)
    st = np.zeros(len(u), np.uint8)
    pg = np.zeros(len(u), np.int32)
    assert oracle.orc_trace_on_batch(abi.fp(new), 640, 480, 1, C.byref(G), abi.ip(pg), C.byref(P), abi.bp(st)) == 0
    good = st == 0
    assert good.mean() > 0.4
    mid = 0.5 * (d["idepth_min_stereo"][good] + d["idepth_max_stereo"][good])
    assert np.median(np.abs(mid - idp[good]) / idp[good]) < 0.08     # sub-pixel match along a general epipolar line
    inside = (d["idepth_min_stereo"][good] <= idp[good] * 1.1) & (d["idepth_max_stereo"][good] >= idp[good] * 0.9)
    assert inside.mean() > 0.7                                      # the new interval brackets the true inverse depth
    assert set(np.unique(st)) <= {0, 1, 2, 3, 4}


@pytest.mark.gpu
def test_gpu_trace_on_bit_exact(gpu_ctx, oracle):
    prob, u, v, idp, G = _trace_on_case()
    host = np.ascontiguousarray(prob["pyr_ref"][0]); new = np.ascontiguousarray(prob["pyr_new"][0])
    gpu_ctx.upload_pyramid(85, [new])
    pr = dict(w=640, h=480)
    n = len(u)
    col, wgt, gH, eth = _oracle_init(oracle, pr, host, u, v)
    # two geometries (the second with a different affine pair) and three kinds of prior state: fresh, a finite interval around
    # the truth (SKIPPED / BADCONDITION paths write the interval midpoint), and points that were OOB / OUTLIER before
    G2 = abi.TraceGeom(); G2.KRKi[:] = list(G.KRKi); G2.Kt[:] = list(G.Kt); G2.aff[:] = [1.0, 0.0]
    geoms = (abi.TraceGeom * 2)(G, G2)
    pg = (np.arange(n) % 7 == 0).astype(np.int32)
    imin = np.zeros(n, np.float32); imax = np.full(n, np.nan, np.float32)
    sel = np.arange(n) % 3 == 1
    imin[sel] = idp[sel] * 0.6; imax[sel] = idp[sel] * 1.5
    sel2 = np.arange(n) % 11 == 2
    imin[sel2] = idp[sel2] * 0.99; imax[sel2] = idp[sel2] * 1.01
    prev = np.full(n, 5, np.uint8); prev[np.arange(n) % 13 == 3] = 1; prev[np.arange(n) % 13 == 4] = 2
    outs = []
    for which in ("oracle", "gpu"):
        P, d = abi.make_trace_points(n, u, v, col, wgt, gH, eth, imin, imax)
        d["lastTraceStatus"][:] = prev
        st = np.zeros(n, np.uint8)
        if which == "oracle":
            assert oracle.orc_trace_on_batch(abi.fp(new), 640, 480, 2, geoms, abi.ip(pg), C.byref(P), abi.bp(st)) == 0
        else:
            gpu_ctx.check(gpu_ctx.L.sdso_trace_on_batch(gpu_ctx.h, 85, 2, geoms, abi.ip(pg), C.byref(P), abi.bp(st)))
        outs.append((st, d))
    (so, do), (sg, dg) = outs
    assert np.array_equal(so, sg)
    for k in ("idepth_min_stereo", "idepth_max_stereo", "quality", "lastTraceStatus", "lastTraceUV", "lastTracePixelInterval"):
        assert np.array_equal(do[k], dg[k], equal_nan=True), k
    assert {0, 1, 2, 3}.issubset(set(np.unique(so)))               # GOOD, OOB (incl. already-OOB), OUTLIER, SKIPPED all occur
    assert gpu_ctx.L.sdso_trace_on_batch(gpu_ctx.h, 85, 1, geoms, abi.ip(pg), C.byref(P), abi.bp(st)) == -1   # point_geom out of range


def _activation_case(oracle, nf=5, per_host=150, seed=3051):
    """nf keyframes of a synthetic window; `per_host` immature points on every host with an idepth interval around the truth."""
    win = synth.ba_window(w=640, h=480, nf=nf, pts_per_kf=4, seed=seed)
    K = win["K"]
    rs = np.random.RandomState(seed)
    pair_R = np.zeros((nf * nf, 9), np.float32); pair_t = np.zeros((nf * nf, 3), np.float32); pair_aff = np.zeros((nf * nf, 2), np.float32)
    for h in range(nf):
        for t in range(nf):
            T = synth.se3_mul(win["poses"][t], synth.se3_inv(win["poses"][h]))           # leftToLeft = target.worldToCam * host.camToWorld
            pair_R[h * nf + t] = T[0].astype(np.float32).ravel(); pair_t[h * nf + t] = T[1].astype(np.float32)
            a = np.exp(win["affs"][t][0] - win["affs"][h][0])                             # AffLight::fromToVecExposure, exposures 1
            pair_aff[h * nf + t] = (a, win["affs"][t][1] - a * win["affs"][h][1])
    hosts, us, vs, imin, imax, cols, wgts, eths, truth = [], [], [], [], [], [], [], [], []
    cal = dict(w=640, h=480)
    scene = synth.Scene(1001)
    for h in range(nf):
        img = np.ascontiguousarray(win["pyrs"][h][0])
        _, idmap = scene.render(640, 480, K, win["poses"][h])
        u, v = synth.select_points(win["pyrs"][h][0], per_host, seed + 17 * h, margin=8, idepth=idmap, min_idepth=0.0075)
        u = u.astype(np.float32); v = v.astype(np.float32)
        c, w_, g, e = _oracle_init(oracle, cal, img, u, v)
        idt = idmap[v.astype(int), u.astype(int)].astype(np.float32)
        lo = idt * rs.uniform(0.6, 1.0, len(u)).astype(np.float32); hi = idt * rs.uniform(1.0, 1.6, len(u)).astype(np.float32)
        hosts.append(np.full(len(u), h, np.int32)); us.append(u); vs.append(v); imin.append(lo); imax.append(hi)
        cols.append(c); wgts.append(w_); eths.append(e); truth.append(idt)
    cat = lambda xs: np.ascontiguousarray(np.concatenate(xs))
    d = dict(nf=nf, win=win, pair_R=pair_R, pair_t=pair_t, pair_aff=pair_aff, host=cat(hosts), u=cat(us), v=cat(vs), idepth_min=cat(imin),
             idepth_max=cat(imax), color=cat(cols), weights=cat(wgts), energyTH=cat(eths), truth=cat(truth))
    return d


def _activate_struct(d, frame_slots=None, dI=None, minObs=2):
    A = abi.Activate()
    keep = []
    A.nf, A.w, A.h, A.n, A.minObs = d["nf"], 640, 480, len(d["u"]), minObs
    A.K[:] = [float(x) for x in d["win"]["K"]]
    for k in ("pair_R", "pair_t", "pair_aff", "u", "v", "idepth_min", "idepth_max", "color", "weights", "energyTH"):
        a = np.ascontiguousarray(d[k], np.float32); keep.append(a); setattr(A, k, abi.fp(a))
    A.host = abi.ip(d["host"])
    if frame_slots is not None:
        fs = np.ascontiguousarray(frame_slots, np.int32); keep.append(fs); A.frame_slot = abi.ip(fs)
    if dI is not None:
        ptrs = (abi.c_float_p * len(dI))()
        for i, a in enumerate(dI):
            a = np.ascontiguousarray(a, np.float32); keep.append(a); ptrs[i] = abi.fp(a)
        keep.append(ptrs); A.dI = C.cast(ptrs, C.POINTER(abi.c_float_p))
    return A, keep


def test_oracle_activation_converges_to_true_idepth(oracle):
    d = _activation_case(oracle)
    A, keep = _activate_struct(d, dI=[p[0] for p in d["win"]["pyrs"]])
    n = A.n
    st = np.zeros(n, np.int8); idp = np.zeros(n, np.float32); rs_ = np.zeros((n, d["nf"]), np.uint8)
    assert oracle.orc_activate_points(C.byref(A), st.ctypes.data_as(C.POINTER(C.c_int8)), abi.fp(idp), abi.bp(rs_)) == 0
    act = st == 1
    assert act.mean() > 0.5 and (st == -1).sum() > 0                      # most points activate, some are rejected
    rel = np.abs(idp[act] - d["truth"][act]) / d["truth"][act]
    start = np.abs(0.5 * (d["idepth_min"][act] + d["idepth_max"][act]) - d["truth"][act]) / d["truth"][act]
    assert np.median(rel) < 0.02 and np.median(rel) < 0.3 * np.median(start)   # the 3 GN steps pull the interval midpoint onto the surface


@pytest.mark.gpu
def test_gpu_activation_bit_exact(gpu_ctx, oracle):
    d = _activation_case(oracle)
    nf = d["nf"]
    for f in range(nf):
        gpu_ctx.upload_pyramid(60 + f, d["win"]["pyrs"][f][:1])
    # degrade some inputs: huge interval (diverging first step), interval far from the truth (outliers), NaN energyTH
    d["idepth_max"][::17] *= 6; d["idepth_min"][5::23] *= 0.05; d["idepth_max"][5::23] *= 0.1; d["energyTH"][9::41] = np.nan
    A, keep = _activate_struct(d, frame_slots=[60 + f for f in range(nf)], dI=[p[0] for p in d["win"]["pyrs"]])
    n = A.n
    so = np.zeros(n, np.int8); io = np.zeros(n, np.float32); ro = np.zeros((n, nf), np.uint8)
    sg = np.zeros(n, np.int8); ig = np.zeros(n, np.float32); rg = np.zeros((n, nf), np.uint8)
    assert oracle.orc_activate_points(C.byref(A), so.ctypes.data_as(C.POINTER(C.c_int8)), abi.fp(io), abi.bp(ro)) == 0
    gpu_ctx.check(gpu_ctx.L.sdso_activate_points_batch(gpu_ctx.h, C.byref(A), sg.ctypes.data_as(C.POINTER(C.c_int8)), abi.fp(ig), abi.bp(rg)))
    assert np.array_equal(so, sg) and np.array_equal(ro, rg)
    assert np.array_equal(io, ig, equal_nan=True)                          # same float operations in the same order: bit-exact
    assert set(np.unique(so)) == {-1, 0, 1}
