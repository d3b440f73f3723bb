"""GPU parity of the windowed BA (BASELINE configs[2]) against the oracle, through the C-ABI.

Bit-exact: host tables (precalc, adjoints, adHTdeltaF), every per-residual quantity (J, energies,
states, JpJdF) and every per-point quantity whose sum runs in the reference's residual order
(Hdd/bd/Hcd, HdiF, bdSumF).  Float tolerance (order of summation only): the packed accumulators.
Double tolerance: stitched H/b, the solution x, and the states after the full GN loop."""
import ctypes as C

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def win_small():
    return synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001)


@pytest.fixture(scope="module")
def win_c3():
    return synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)      # configs[2]: 8 KF x 2000 points


def _both(ctx, oracle, win, slot0=40, wid=3):
    for f in range(win["nf"]):
        ctx.upload_pyramid(slot0 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[slot0 + f for f in range(win["nf"])], dI_list=[p[0] for p in win["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
    h = oracle.orc_ba_create(C.byref(W))
    return W, keep, h


def _lin_both(ctx, oracle, win, h, wid):
    nr = win["nr"]
    eo, eg = C.c_double(0), C.c_double(0)
    oracle.orc_ba_linearize(h, C.byref(eo))
    ctx.check(ctx.L.sdso_ba_keep_projections(ctx.h, wid, 1))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, wid, C.byref(eg)))
    o = dict(J=np.zeros((nr, 74), np.float32), ns=np.zeros(nr, np.uint8), ne=np.zeros(nr, np.float32), nw=np.zeros(nr, np.float32),
             pj=np.zeros((nr, 16), np.float32), cp=np.zeros((nr, 3), np.float32))
    g = {k: np.zeros_like(v) for k, v in o.items()}
    oracle.orc_ba_get_linearization(h, abi.fp(o["J"]), abi.bp(o["ns"]), abi.fp(o["ne"]), abi.fp(o["nw"]), abi.fp(o["pj"]), abi.fp(o["cp"]))
    ctx.check(ctx.L.sdso_ba_get_linearization(ctx.h, wid, abi.fp(g["J"]), abi.bp(g["ns"]), abi.fp(g["ne"]), abi.fp(g["nw"]), abi.fp(g["pj"]), abi.fp(g["cp"])))
    return eo.value, eg.value, o, g


def _check_lin(eo, eg, o, g):
    assert np.array_equal(o["ns"], g["ns"])                       # IN / OOB / OUTLIER decisions
    assert np.array_equal(o["ne"], g["ne"]) and np.array_equal(o["nw"], g["nw"])
    live = o["ns"] != 1                                           # J of an OOB residual is unspecified in the reference too
    assert np.array_equal(o["J"][live], g["J"][live])             # all 74 floats of RawResidualJacobian, bit-exact
    assert np.array_equal(o["pj"][live], g["pj"][live]) and np.array_equal(o["cp"][live], g["cp"][live])
    assert abs(eo - eg) <= 1e-9 * abs(eo)                         # double sum of identical floats, different order


@pytest.mark.parametrize("which", ["small", "c3"])
def test_tables_linearize_apply(gpu_ctx, oracle, win_small, win_c3, which):
    win = win_small if which == "small" else win_c3
    W, keep, h = _both(gpu_ctx, oracle, win)
    nf, nr = win["nf"], win["nr"]
    to = [np.zeros(nf * nf * 27, np.float32), np.zeros(nf * nf * 64), np.zeros(nf * nf * 64), np.zeros(nf * nf * 8, np.float32)]
    tg = [np.zeros_like(a) for a in to]
    oracle.orc_ba_get_tables(h, abi.fp(to[0]), abi.dp(to[1]), abi.dp(to[2]), abi.fp(to[3]))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_tables(gpu_ctx.h, 3, abi.fp(tg[0]), abi.dp(tg[1]), abi.dp(tg[2]), abi.fp(tg[3])))
    for a, b in zip(to, tg):
        assert np.array_equal(a, b)
    eo, eg, o, g = _lin_both(gpu_ctx, oracle, win, h, 3)
    _check_lin(eo, eg, o, g)
    assert (o["ns"] == 0).sum() > 0.4 * nr
    # applyRes + takeDataF
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    so, ao, jo = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8), np.zeros((nr, 8), np.float32)
    sg, ag, jg = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8), np.zeros((nr, 8), np.float32)
    oracle.orc_ba_get_residual_state(h, abi.bp(so), abi.bp(ao), abi.fp(jo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_residual_state(gpu_ctx.h, 3, abi.bp(sg), abi.bp(ag), abi.fp(jg)))
    assert np.array_equal(so, sg) and np.array_equal(ao, ag)
    assert np.array_equal(jo[ao == 1], jg[ao == 1])
    # second linearize: frameEnergyTH of the newest frame was updated by setNewFrameEnergyTH -> same decisions again
    eo, eg, o, g = _lin_both(gpu_ctx, oracle, win, h, 3)
    _check_lin(eo, eg, o, g)
    oracle.orc_ba_destroy(h)


def _accumulate_both(ctx, oracle, win, h, wid):
    nf = win["nf"]
    oracle.orc_ba_accumulate(h)
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, wid))
    na = abi.accum_floats(nf)
    assert na == ctx.L.sdso_ba_accum_floats(nf) == oracle.orc_ba_accum_floats(nf)
    ao, ag = np.zeros(na, np.float32), np.zeros(na, np.float32)
    oracle.orc_ba_get_accumulators(h, abi.fp(ao))
    ctx.check(ctx.L.sdso_ba_get_accumulators(ctx.h, wid, abi.fp(ag)))
    return ao, ag


def _check_accum(ao, ag, nf):
    # sections: topA, topL (per pair 91), accD (per triple 64), accE, accEB, Hcc, bc, nres
    def sec(a, lo, hi, w):
        return a[lo:hi].reshape(-1, w)
    o0 = 0
    for name, cnt, w in (("topA", nf * nf, 91), ("topL", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8), ("Hcc", 1, 16), ("bc", 1, 4)):
        so, sg = sec(ao, o0, o0 + cnt * w, w), sec(ag, o0, o0 + cnt * w, w)
        scale = np.abs(so).max(axis=1, keepdims=True)
        assert np.array_equal(scale == 0, np.abs(sg).max(axis=1, keepdims=True) == 0), name       # same empty bins
        err = np.abs(so - sg) / np.maximum(scale, 1e-30)
        assert err.max() <= 3e-5, (name, err.max())
        o0 += cnt * w
    assert np.array_equal(ao[o0:o0 + 2], ag[o0:o0 + 2])           # nresA, nresL are exact counts


@pytest.mark.parametrize("which", ["small", "c3"])
def test_accumulate_solve(gpu_ctx, oracle, win_small, win_c3, which):
    win = win_small if which == "small" else win_c3
    W, keep, h = _both(gpu_ctx, oracle, win)
    nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
    _lin_both(gpu_ctx, oracle, win, h, 3)
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    ao, ag = _accumulate_both(gpu_ctx, oracle, win, h, 3)
    _check_accum(ao, ag, nf)
    po = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    pg = [np.zeros_like(a) for a in po]
    oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in po])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_terms(gpu_ctx.h, 3, *[abi.fp(a) for a in pg]))
    for a, b in zip(po, pg):
        assert np.array_equal(a, b)                                # HdiF, bdSumF, Hdd_accAF, bd_accAF, Hcd_accAF: bit-exact
    for it, lam in ((0, 0.1), (2, 0.025)):                         # iteration >= 2 orthogonalises x against the 7 gauge directions
        xo, Ho, bo, fso, cso = np.zeros(n), np.zeros((n, n)), np.zeros(n), np.zeros(nf * 8), np.zeros(4)
        xg, Hg, bg, fsg, csg = np.zeros(n), np.zeros((n, n)), np.zeros(n), np.zeros(nf * 8), np.zeros(4)
        oracle.orc_ba_solve(h, it, lam, abi.dp(xo), abi.dp(Ho), abi.dp(bo), abi.dp(fso), abi.dp(cso))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 3))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 3, it, lam, abi.dp(xg), abi.dp(Hg), abi.dp(bg), abi.dp(fsg), abi.dp(csg)))
        d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
        assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-4    # lastHS, entries span 1e0..1e14
        assert np.abs((bg - bo) / d).max() <= 1e-4 * max(1.0, np.abs(bo / d).max())
        # x: compare in the metric of the system (x_i * sqrt(H_ii) is the whitened step)
        assert np.abs((xg - xo) * d).max() <= 2e-4 * max(1.0, np.abs(xo * d).max())
        assert np.allclose(fsg, -xg[4:]) and np.allclose(csg, -xg[:4])
        so, sg = np.zeros(npts, np.float32), np.zeros(npts, np.float32)
        oracle.orc_ba_get_point_steps(h, abi.fp(so))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_steps(gpu_ctx.h, 3, abi.fp(sg)))
        assert np.abs(sg - so).max() <= 2e-4 * max(np.abs(so).max(), 1e-6)
    oracle.orc_ba_destroy(h)


def test_solve_requires_accumulate_and_arg_errors(gpu_ctx, oracle, win_small):
    W, keep, h = _both(gpu_ctx, oracle, win_small, wid=5)
    oracle.orc_ba_destroy(h)
    n = 8 * win_small["nf"] + 4
    x = np.zeros(n)
    assert gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 5, 0, 0.1, abi.dp(x), None, None, None, None) == -1
    assert gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 999, None) == -1
    ok = dict(win_small)
    ok["solverMode"] = 128 | 512 | 1024 | 4 | 8  # the bits outside solveSystemF are accepted since round 6 (tests/test_ba_solver_bits_gpu.py)
    Wb, kb = abi.make_ba_window(ok, frame_slots=[40 + f for f in range(ok["nf"])])
    assert gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 6, C.byref(Wb)) == 0
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 6))
    bad = dict(win_small)
    bad["host"] = win_small["host"][::-1].copy()      # not in allPoints order
    Wb, kb = abi.make_ba_window(bad, frame_slots=[40 + f for f in range(bad["nf"])])
    assert gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 6, C.byref(Wb)) == -1
    assert gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 6, None) == -1          # a refused upload leaves no half-built window behind
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 5))


SVD, ORTH_SYS, CUT7, USE_GN, FIX_LAMBDA, ORTH_X, ORTH_X_LATER = 1, 2, 16, 64, 128, 256, 2048


@pytest.mark.parametrize("mode,first_id", [(SVD | FIX_LAMBDA | ORTH_X_LATER, 0), (SVD | CUT7 | FIX_LAMBDA, 0), (ORTH_SYS | FIX_LAMBDA | ORTH_X_LATER, 0),
                                           (ORTH_SYS | FIX_LAMBDA, 3), (ORTH_SYS | SVD, 3), (USE_GN | ORTH_X, 0), (0, 0)])
def test_solver_mode_variants(gpu_ctx, oracle, win_small, mode, first_id):
    """The other branches of solveSystemF (EnergyFunctional.cpp:838-995): SOLVER_SVD [_CUT7], SOLVER_ORTHOGONALIZE_SYSTEM with and
    without the first frame in the window, SOLVER_USE_GN, SOLVER_ORTHOGONALIZE_X, plain lambda.  Same bars as the default branch."""
    win = dict(win_small)
    win["solverMode"] = mode
    win["frameID"] = (np.arange(win["nf"]) + first_id).astype(np.int32)     # first_id > 0: frame 0 has left the window (no gauge prior)
    W, keep, h = _both(gpu_ctx, oracle, win, wid=7)
    nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
    _lin_both(gpu_ctx, oracle, win, h, 7)
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 7))
    for it, lam in ((0, 0.1), (2, 0.025)):
        xo, Ho, bo = np.zeros(n), np.zeros((n, n)), np.zeros(n)
        xg, Hg, bg = np.zeros(n), np.zeros((n, n)), np.zeros(n)
        oracle.orc_ba_solve(h, it, lam, abi.dp(xo), abi.dp(Ho), abi.dp(bo), None, None)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 7))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 7, it, lam, abi.dp(xg), abi.dp(Hg), abi.dp(bg), None, None))
        d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
        assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-4
        assert np.abs((bg - bo) / d).max() <= 1e-4 * max(1.0, np.abs(bo / d).max())
        # without frame 0 nothing fixes the 7 gauge directions: the projected system is singular there and the solution along
        # them is set by rounding (in the oracle as much as here), so the bar is wider for those windows
        tol = 5e-4 if first_id == 0 else 1e-2
        assert np.abs((xg - xo) * d).max() <= tol * max(1.0, np.abs(xo * d).max())
        so, sg = np.zeros(npts, np.float32), np.zeros(npts, np.float32)
        oracle.orc_ba_get_point_steps(h, abi.fp(so))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_steps(gpu_ctx.h, 7, abi.fp(sg)))
        assert np.abs(sg - so).max() <= tol * max(np.abs(so).max(), 1e-6)
    oracle.orc_ba_destroy(h)
    if mode & (SVD | ORTH_SYS):
        # the batch entry points take these windows too (host-driven solve per window behind one batched accumulate): the batch's x is the
        # single call's x at the same state, and sdso_ba_batch_optimize reproduces sdso_ba_optimize window by window
        ids = np.array([7], np.int32)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 1, abi.ip(ids)))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_accumulate(gpu_ctx.h))            # (linearises, applies and accumulates in one kernel)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_solve(gpu_ctx.h, 0.025, 1))
        xb = np.zeros(n)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_get_x(gpu_ctx.h, abi.dp(xb)))
        xs = np.zeros(n)                                                        # the single call on the state the batch left applied
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 7))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 7, 2, 0.025, abi.dp(xs), None, None, None, None))
        d2 = np.sqrt(np.abs(np.diag(Hg))) + 1e-30
        assert np.abs((xb - xs) * d2).max() <= 1e-6 * max(1.0, np.abs(xs * d2).max())
        res = (abi.BAOptResult * 1)()
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 7, C.byref(W)))       # a fresh copy (the calls above moved the newest frame's threshold)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 1, abi.ip(ids)))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize(gpu_ctx.h, 4, res))
        sb, ib, rb = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(win["nr"], np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, 7, abi.dp(sb), abi.fp(ib), abi.bp(rb)))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 7, C.byref(W)))       # the same window again (the upload releases the batch's binding)
        ss, is_, rs, os_ = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(win["nr"], np.uint8), abi.BAOptResult()
        gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 7, 4, abi.dp(ss), abi.fp(is_), abi.bp(rs), C.byref(os_)))
        assert res[0].iterations == os_.iterations
        assert np.abs(sb - ss).max() <= 1e-9 and np.abs(ib - is_).max() <= 1e-7 and np.array_equal(rb, rs)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 7))


@pytest.mark.parametrize("mode,first_id", [(SVD | FIX_LAMBDA | ORTH_X_LATER, 0), (SVD | CUT7 | FIX_LAMBDA, 0), (ORTH_SYS | FIX_LAMBDA | ORTH_X_LATER, 3), (ORTH_SYS | SVD | FIX_LAMBDA, 0)])
def test_device_solver_of_the_svd_and_orthogonalised_branches(gpu_ctx, oracle, win_c3, mode, first_id, monkeypatch):
    """k_ba_solve_alt (csrc/ba_solve_alt.hip: parallel-order Jacobi eigen-decomposition / projected system + register LDL^T on the device)
    against the host statement of the same arithmetic (solve_system_host behind SDSO_BA_SOLVE_HOST=1, sequential-order Jacobi) on the
    8-keyframe window, and the whole resident loop in these modes — now without a host round trip per iteration, batch included —
    against the oracle's loop."""
    win = dict(win_c3)
    win["solverMode"] = mode
    win["frameID"] = (np.arange(win["nf"]) + first_id).astype(np.int32)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    W, keep, h = _both(gpu_ctx, oracle, win, wid=8)
    _lin_both(gpu_ctx, oracle, win, h, 8)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 8))
    xs = {}
    for where in ("device", "host"):
        if where == "host":
            monkeypatch.setenv("SDSO_BA_SOLVE_HOST", "1")
        else:
            monkeypatch.delenv("SDSO_BA_SOLVE_HOST", raising=False)
        x, H, b = np.zeros(n), np.zeros((n, n)), np.zeros(n)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 8))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 8, 2, 0.025, abi.dp(x), abi.dp(H), abi.dp(b), None, None))
        st = np.zeros(npts, np.float32)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_steps(gpu_ctx.h, 8, abi.fp(st)))
        xs[where] = (x, H, b, st)
    monkeypatch.delenv("SDSO_BA_SOLVE_HOST", raising=False)
    (xd, Hd, bd, sd), (xh, Hh, bh, sh) = xs["device"], xs["host"]
    d = np.sqrt(np.abs(np.diag(Hh))) + 1e-30
    assert np.abs((Hd - Hh) / np.outer(d, d)).max() <= 1e-12            # the same f64 expressions on the same stitched blocks
    assert np.abs((bd - bh) / d).max() <= 1e-12 * max(1.0, np.abs(bh / d).max())
    # x: the same system through a different rotation order (SVD) / the same pivoted LDL^T in a different elimination order; without
    # frame 0 the projected system is singular along the gauge and x there is set by rounding on both sides
    tol = 1e-7 if first_id == 0 else 1e-3
    assert np.abs((xd - xh) * d).max() <= tol * max(1.0, np.abs(xh * d).max()), np.abs((xd - xh) * d).max()
    assert np.abs(xd).max() > 0 and np.isfinite(xd).all()
    assert np.abs(sd - sh).max() <= max(tol, 1e-6) * max(np.abs(sh).max(), 1e-6)
    # the whole loop, device-resident in these modes too: against the oracle's loop with the usual bars
    so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_destroy(h)
    h = oracle.orc_ba_create(C.byref(W))                                 # (a fresh window on both sides)
    oracle.orc_ba_optimize(h, 6, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
    oracle.orc_ba_destroy(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 8, C.byref(W)))
    sg, ig, rg, og = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 8, 6, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
    assert og.iterations == oo.iterations
    bar = 3e-4 if first_id == 0 else 5e-2                                # (no gauge prior: the states drift along the gauge on both sides)
    assert np.abs(sg - so).max() <= bar, np.abs(sg - so).max()
    assert (rg != ro).sum() <= max(2, nr // 1000)
    # and two such windows as a batch: the resident batch loop equals the single calls
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 8, C.byref(W)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 9, C.byref(W)))
    ids = np.array([8, 9], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 2, abi.ip(ids)))
    res = (abi.BAOptResult * 2)()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize(gpu_ctx.h, 6, res))
    for wid in (8, 9):
        sb, ib, rb = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, wid, abi.dp(sb), abi.fp(ib), abi.bp(rb)))
        assert res[wid - 8].iterations == og.iterations
        assert np.abs(sb - sg).max() <= 1e-9 and np.abs(ib - ig).max() <= 1e-7 and np.array_equal(rb, rg)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 8))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 9))


def test_stitched_systems_of_the_three_accumulations(gpu_ctx, oracle, win_c3):
    """EnergyFunctional::accumulateAF_MT / accumulateLF_MT / accumulateSCF_MT (EnergyFunctional.cpp:212-269) return the STITCHED systems
    (AccumulatedTopHessian.h:95-148 without / with priors, AccumulatedSCHessian.h:96-135); sdso_ba_get_stitched hands them out.  Same
    state on both sides: the stitch bars of the default branch per block, exact symmetry, and the sum solveSystemF forms from them
    (HL + HM + HA - Hsc = lastHS, :906-909) to rounding."""
    win = dict(win_c3)
    W, keep, h = _both(gpu_ctx, oracle, win, wid=11)
    n = 8 * win["nf"] + 4
    _lin_both(gpu_ctx, oracle, win, h, 11)
    oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 11))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 11))
    so = [(np.zeros((n, n)), np.zeros(n)) for _ in range(3)]
    sg = [(np.zeros((n, n)), np.zeros(n)) for _ in range(3)]
    oracle.orc_ba_get_stitched(h, *[abi.dp(a) for pair in so for a in pair])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_stitched(gpu_ctx.h, 11, *[abi.dp(a) for pair in sg for a in pair]))
    x, HS, bS = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 11, 0, 0.1, abi.dp(x), abi.dp(HS), abi.dp(bS), None, None))
    d = np.sqrt(np.abs(np.diag(HS))) + 1e-30
    for k, name in enumerate(("A", "L", "SC")):
        (Hg, bg), (Ho, bo) = sg[k], so[k]
        assert np.array_equal(Hg, Hg.T) or np.abs(Hg - Hg.T).max() <= 1e-12 * np.abs(Hg).max(), name
        assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-4, (name, np.abs((Hg - Ho) / np.outer(d, d)).max())
        assert np.abs((bg - bo) / d).max() <= 1e-4 * max(1.0, np.abs(bo / d).max()), name
        assert np.abs(Hg).max() > 0, name
    HM = np.asarray(win["HM"], np.float64).reshape(n, n)
    total = sg[1][0] + HM + sg[0][0] - sg[2][0]
    assert np.abs((total - HS) / np.outer(d, d)).max() <= 1e-9
    # partial requests and the precondition
    only = np.zeros((n, n))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_stitched(gpu_ctx.h, 11, None, None, None, None, abi.dp(only), None))
    assert np.array_equal(only, sg[2][0])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 11, C.byref(W)))
    assert gpu_ctx.L.sdso_ba_get_stitched(gpu_ctx.h, 11, abi.dp(only), None, None, None, None, None) == -1      # nothing accumulated yet
    oracle.orc_ba_destroy(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 11))


@pytest.mark.parametrize("which", ["small", "c3"])
def test_optimize_full_gn_loop(gpu_ctx, oracle, win_small, win_c3, which):
    win = win_small if which == "small" else win_c3
    W, keep, h = _both(gpu_ctx, oracle, win)
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    sg, ig, rg, og = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_optimize(h, 6, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 6, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
    assert og.iterations == oo.iterations
    # The float accumulators make the CPU path itself order-dependent (the reference sums per-thread
    # copies in scheduling order): measure that spread by re-running the oracle on the same window with
    # the points shuffled inside each host group (3 shuffles; one sample is too noisy — over 8 shuffles of the
    # small window the oracle moves by 5e-6 ... 7.7e-5), and require the GPU to sit within 1e-5 (north_star)
    # plus twice that spread.  Measured on MI355X: |gpu-oracle| 7.3e-5 / 8.2e-6 (small / C3).
    import helpers
    spread_s, spread_i = 0.0, 0.0
    for seed in (1, 3, 5):
        w2, order = helpers.permuted_window(win, seed)
        W2, keep2 = abi.make_ba_window(w2, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
        h2 = oracle.orc_ba_create(C.byref(W2))
        sp, ip, rp, op = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h2, 6, abi.dp(sp), abi.fp(ip), abi.bp(rp), C.byref(op))
        oracle.orc_ba_destroy(h2)
        spread_s = max(spread_s, np.abs(sp - so).max())
        spread_i = max(spread_i, np.abs(ip - io[order]).max())
    assert np.abs(sg - so).max() <= 1e-5 + 2.0 * spread_s
    assert helpers.idepths_close(ig, io, 1e-5 + 2.0 * spread_i)
    mism = (rg != ro).sum()
    assert mism <= max(2, nr // 2000)        # IN/OUTLIER flips only where an energy sits on the threshold
    assert helpers.counts_close(og.resInA, oo.resInA, nr)
    assert abs(og.lastEnergy - oo.lastEnergy) <= 1e-4 * oo.lastEnergy
    oracle.orc_ba_destroy(h)


# scale of a frame's 8 state entries: state_scaled = SCALE * state (SCALE_XI_TRANS 0.5, SCALE_XI_ROT 1, SCALE_A 10, SCALE_B 1000; HessianBlocks.h:54-61, :161-169)
_STATE_SCALE = np.array([0.5, 0.5, 0.5, 1.0, 1.0, 1.0, 10.0, 1000.0])


def test_pose_updates_of_every_gn_iteration_c3(gpu_ctx, oracle, win_c3):
    """north_star's bar as it is stated — "pose deltas within 1e-5 of reference" — at BASELINE configs[2] (8 keyframes x 2000 points), per
    Gauss-Newton iteration: the loop of FullSystem::optimize (FullSystemOptimize.cpp:871-1041) runs on the device one iteration at a time
    (sdso_ba_batch_optimize_begin / accumulate / solve_step on a one-window batch: the kernels of the resident loop) and on the oracle
    (orc_ba_get_x_trace: lastX of every solveSystemF), each along its OWN trajectory from the same uploaded state.  For every iteration the
    update of every frame is compared in the units the pose moves in (x * SCALE).  FIXED bars:
      * against the reference arithmetic with its sums carried in f64 (orc_set_acc64: the float summation ORDER of the CPU path out of the
        reference value): translation and rotation entries |dx| <= 1e-5 on EVERY iteration, the first included (measured on MI355X,
        round 6: 4.1e-6, 1.4e-6, 2.3e-7, 4.3e-7).  The device carries its cross-residual / cross-point sums in f64 on the matrix cores (ba_kernels.hip,
        ACC_MODE 1), so this is the comparison in which only the reference's per-residual arithmetic is left;
      * against the reference's float path as it is (4-byte accumulators, sequential): <= 1e-5 from the second iteration on and <= 2e-5 on
        the first (measured 9.4e-6, 1.1e-6, 2.4e-7, 5.9e-7; round 5 with fp32 chains: 1.39e-5 on the first) — what is left there is the CPU
        float path's OWN distance from the order-independent sums (5.3e-6 on this window, median 1.3e-5 over 24 windows against the
        device's 1.4e-6: profiles/r06_truth_updates.txt), asserted below to be no more than that plus the device's 1e-5;
      * the affine a (scaled, dimensionless) <= 1e-5, b (scaled: intensity levels of 0..255) <= 1e-3.  After the loop every frame's pose
        state is within a FIXED 2e-5 of the float oracle's, next to the spread-relative bar of test_optimize_full_gn_loop.
    A failure names the iteration and the residuals whose final state differs."""
    ctx, win = gpu_ctx, win_c3
    W, keep, h = _both(ctx, oracle, win, wid=13)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_optimize(h, 6, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
    xo = np.zeros((8, n))
    its_o = oracle.orc_ba_get_x_trace(h, abi.dp(xo), 8)
    oracle.orc_ba_destroy(h)
    assert its_o == oo.iterations and its_o >= 2
    # the same loop with every accumulator sum of the restatement in double
    oracle.orc_set_acc64(1)
    try:
        h64 = oracle.orc_ba_create(C.byref(W))
        s64, i64, o64 = np.zeros((nf, 10)), np.zeros(npts, np.float32), abi.BAOptResult()
        oracle.orc_ba_optimize(h64, 6, abi.dp(s64), abi.fp(i64), None, C.byref(o64))
        x64 = np.zeros((8, n))
        its_64 = oracle.orc_ba_get_x_trace(h64, abi.dp(x64), 8)
        oracle.orc_ba_destroy(h64)
    finally:
        oracle.orc_set_acc64(0)
    ids = np.array([13], np.int32)
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, 1, abi.ip(ids)))
    ctx.check(ctx.L.sdso_ba_batch_optimize_begin(ctx.h, 1))
    xg = np.zeros((6, n))
    for it in range(6):
        ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
        ctx.check(ctx.L.sdso_ba_batch_solve_step(ctx.h, 0.1 * 0.25 ** it, 1 if it >= 2 else 0))     # (FIX_LAMBDA | ORTHOGONALIZE_X_LATER: settings.cpp:51)
        ctx.check(ctx.L.sdso_ba_batch_get_x(ctx.h, abi.dp(xg[it:it + 1])))
    og = (abi.BAOptResult * 1)()
    ctx.check(ctx.L.sdso_ba_batch_optimize_end(ctx.h, og))
    assert og[0].iterations == oo.iterations
    sg, ig, rg = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8)
    ctx.check(ctx.L.sdso_ba_get_state(ctx.h, 13, abi.dp(sg), abi.fp(ig), abi.bp(rg)))
    flipped = np.nonzero(rg != ro)[0]
    report = []
    for it in range(its_o):      # (a window whose break test fired keeps its last x on the device: only the iterations both ran)
        d = np.abs((xg[it, 4:] - xo[it, 4:]).reshape(nf, 8) * _STATE_SCALE)
        report.append((it, float(d[:, :6].max()), float(d[:, 6].max()), float(d[:, 7].max())))
    msg = "per iteration (it, pose, a, b): %s; residuals whose final state differs: %s" % (report, flipped.tolist())
    truth, cpu_own = [], []
    for it in range(min(its_o, its_64)):
        truth.append(float(np.abs((xg[it, 4:] - x64[it, 4:]).reshape(nf, 8) * _STATE_SCALE)[:, :6].max()))
        cpu_own.append(float(np.abs((xo[it, 4:] - x64[it, 4:]).reshape(nf, 8) * _STATE_SCALE)[:, :6].max()))
    msg += "; pose update vs the f64-accumulator truth per iteration: device %s, CPU float path %s" % (truth, cpu_own)
    assert og[0].iterations == o64.iterations, msg
    for it, dpose in enumerate(truth):
        assert dpose <= 1e-5, msg                      # north_star's bar, every iteration, against the order-independent reference value
    for it, dpose, da, db in report:
        # against the float oracle the first update's distance is the CPU path's own summation noise: the device is within 1e-5 of the truth
        # (above), so |device - cpu| <= |cpu - truth| + 1e-5, and it stays under the 2e-5 of rounds 4-5
        assert dpose <= (min(2e-5, cpu_own[0] + 1e-5) if it == 0 else 1e-5) and da <= 1e-5 and db <= 1e-3, msg
    dstate = np.abs(sg - so)[:, :8] * _STATE_SCALE
    assert dstate[:, :6].max() <= 2e-5, (float(dstate[:, :6].max()), msg)
    assert dstate[:, 6].max() <= 2e-5 and dstate[:, 7].max() <= 2e-3, (dstate[:, 6:].max(axis=0).tolist(), msg)
    assert len(flipped) <= max(2, nr // 2000)
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 13))
    print("pose updates at configs[2]:", msg)


@pytest.mark.parametrize("noise", [dict(), dict(idepth_noise=0.3, state_noise=1e-2)])
def test_optimize_energy_gated_steps(gpu_ctx, oracle, noise):
    """setting_forceAceptStep = false: calcLEnergyF_MT / calcMEnergyF gate every step (FullSystemOptimize.cpp:978),
    rejected steps restore the backup and multiply lambda by 100.  Both windows below reject steps on the CPU path
    (their final energy differs from the forced-accept run); the device path must take the same decisions."""
    win = dict(synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001, **noise))
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    n = 8 * nf + 4
    rs = np.random.RandomState(5)
    A = rs.normal(0, 1, (n, n))
    win["HM"] = (A @ A.T) * 1e3                  # a marginalisation prior, so that calcMEnergyF is not trivially zero
    win["bM"] = rs.normal(0, 1e2, n)
    forced = dict(win)
    win["forceAcceptStep"] = 0
    res = {}
    for name, w in (("gated", win), ("forced", forced)):
        W, keep, h = _both(gpu_ctx, oracle, w)
        so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        sg, ig, rg, og = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h, 6, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 6, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
        oracle.orc_ba_destroy(h)
        res[name] = (so, oo.lastEnergy)
        assert og.iterations == oo.iterations
        assert abs(og.lastEnergy - oo.lastEnergy) <= 1e-3 * oo.lastEnergy
        assert np.abs(sg - so).max() <= 2e-4 and helpers.idepths_close(ig, io, 2e-4)     # same accept / reject sequence, float-order spread only
    assert abs(res["gated"][1] - res["forced"][1]) > 1e-3 * res["forced"][1]         # the gate really rejected something


def test_marginalize_points(gpu_ctx, oracle, win_small):
    win = win_small
    W, keep, h = _both(gpu_ctx, oracle, win)
    nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
    # bring both to the same applied state, one accumulate so that HdiF etc. exist
    _lin_both(gpu_ctx, oracle, win, h, 3)
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    _accumulate_both(gpu_ctx, oracle, win, h, 3)
    flag = (win["host"] == 0).astype(np.uint8)          # marginalise every point hosted in the oldest keyframe
    HMo, bMo, HMg, bMg = np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HMo), abi.dp(bMo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_marginalize_points(gpu_ctx.h, 3, abi.bp(flag), abi.dp(HMg), abi.dp(bMg)))
    assert np.abs(HMo).max() > 0
    d = np.sqrt(np.abs(np.diag(HMo))) + 1e-30
    live = np.abs(np.diag(HMo)) > 0
    assert np.abs(((HMg - HMo) / np.outer(d, d))[np.ix_(live, live)]).max() <= 1e-4
    assert np.abs(((bMg - bMo) / d)[live]).max() <= 1e-4 * max(1.0, np.abs((bMo / d)[live]).max())
    oracle.orc_ba_destroy(h)


def test_batch_equals_single(gpu_ctx, oracle, win_small):
    """Two windows advanced by the batch entry points give the same x as the single-window calls."""
    wins = [win_small, synth.ba_window(w=640, h=480, nf=5, pts_per_kf=90, seed=3011)]
    xs = []
    for i, win in enumerate(wins):
        for f in range(win["nf"]):
            gpu_ctx.upload_pyramid(60 + 10 * i + f, win["pyrs"][f][:1])
        W, keep = abi.make_ba_window(win, frame_slots=[60 + 10 * i + f for f in range(win["nf"])])
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 20 + i, C.byref(W)))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 20 + i, None))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 20 + i))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 20 + i))
        x = np.zeros(8 * win["nf"] + 4)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 20 + i, 0, 0.1, abi.dp(x), None, None, None, None))
        xs.append(x)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 20 + i, C.byref(W)))      # fresh state for the batch run
    ids = np.array([20, 21], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 2, abi.ip(ids)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_accumulate(gpu_ctx.h))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_solve(gpu_ctx.h, 1e-5, 0))
    xb = np.zeros((2, 44))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_get_x(gpu_ctx.h, abi.dp(xb)))
    for i in range(2):
        assert np.array_equal(xb[i], xs[i])             # deterministic kernels: identical bits


def test_ragged_and_degenerate_windows(gpu_ctx, oracle):
    """Edge cases of the window shape: two keyframes only, hosts without points, points without residuals, residual lists of very
    different length in one batch launch.  Everything must agree with the oracle (x in the whitened metric) and nothing may fault."""
    cases = []
    w2 = synth.ba_window(w=320, h=240, nf=2, pts_per_kf=40, seed=3071)                       # minimum window
    cases.append(("nf2", w2))
    w3 = dict(synth.ba_window(w=320, h=240, nf=4, pts_per_kf=30, seed=3072))
    keep_pts = w3["host"] != 1                                                                # a keyframe that hosts no point
    idx = np.nonzero(keep_pts)[0]
    remap = -np.ones(w3["np"], np.int64); remap[idx] = np.arange(len(idx))
    rk = keep_pts[w3["res_point"]]
    for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
        w3[k] = w3[k][idx]
    w3["res_point"] = remap[w3["res_point"][rk]].astype(np.int32); w3["res_target"] = w3["res_target"][rk]; w3["res_state"] = w3["res_state"][rk]
    w3["np"], w3["nr"] = len(idx), int(rk.sum())
    cases.append(("empty_host", w3))
    w4 = dict(synth.ba_window(w=320, h=240, nf=4, pts_per_kf=30, seed=3073))
    drop = np.isin(w4["res_point"], np.arange(0, w4["np"], 3))                                # every third point loses all its residuals
    w4["res_point"] = w4["res_point"][~drop]; w4["res_target"] = w4["res_target"][~drop]; w4["res_state"] = w4["res_state"][~drop]
    w4["nr"] = int((~drop).sum())
    cases.append(("points_without_residuals", w4))
    xs = {}
    for i, (name, win) in enumerate(cases):
        nf, n = win["nf"], 8 * win["nf"] + 4
        for f in range(nf):
            gpu_ctx.upload_pyramid(200 + 10 * i + f, win["pyrs"][f][:1])
        W, keep = abi.make_ba_window(win, frame_slots=[200 + 10 * i + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 20 + i, C.byref(W)))
        h = oracle.orc_ba_create(C.byref(W))
        oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
        xo, Ho = np.zeros(n), np.zeros((n, n))
        oracle.orc_ba_solve(h, 0, 0.1, abi.dp(xo), abi.dp(Ho), None, None, None)
        oracle.orc_ba_destroy(h)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 20 + i, None))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 20 + i))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 20 + i))
        xg = np.zeros(n)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 20 + i, 0, 0.1, abi.dp(xg), None, None, None, None))
        d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
        assert np.abs((xg - xo) * d).max() <= 2e-4 * max(1.0, np.abs(xo * d).max()), name
        xs[name] = xg
    # the two nf = 4 windows (different point / residual counts) in ONE batch launch give the same x as alone
    ids = np.array([21, 22], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 21, C.byref(abi.make_ba_window(cases[1][1], frame_slots=[210 + f for f in range(4)])[0])))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 22, C.byref(abi.make_ba_window(cases[2][1], frame_slots=[220 + f for f in range(4)])[0])))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 2, abi.ip(ids)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_accumulate(gpu_ctx.h))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_solve(gpu_ctx.h, 1e-5, 0))      # sdso_ba_solve applies SOLVER_FIX_LAMBDA (lambda := 1e-5) itself
    xb = np.zeros((2, 36))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_get_x(gpu_ctx.h, abi.dp(xb)))
    for k, name in enumerate(("empty_host", "points_without_residuals")):
        dd = np.abs(xb[k] - xs[name]).max()
        assert dd <= 1e-9 * max(1.0, np.abs(xs[name]).max()), (name, dd)
    # a window with no points at all is accepted and solves to the prior-only system
    w0 = dict(cases[0][1])
    for k in ("u", "v", "idepth", "idepth_zero", "host", "hasDepthPrior"):
        w0[k] = w0[k][:0]
    w0["color"] = w0["color"][:0]; w0["weights"] = w0["weights"][:0]
    w0["res_point"] = w0["res_point"][:0]; w0["res_target"] = w0["res_target"][:0]; w0["res_state"] = w0["res_state"][:0]
    w0["np"] = w0["nr"] = 0
    W0, k0 = abi.make_ba_window(w0, frame_slots=[200, 201], dI_list=[p[0] for p in w0["pyrs"]])
    h0 = oracle.orc_ba_create(C.byref(W0))
    oracle.orc_ba_linearize(h0, None); oracle.orc_ba_apply_res(h0); oracle.orc_ba_accumulate(h0)
    xo0 = np.zeros(20)
    oracle.orc_ba_solve(h0, 0, 0.1, abi.dp(xo0), None, None, None, None)
    oracle.orc_ba_destroy(h0)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 25, C.byref(W0)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 25, None))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 25))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 25))
    x0 = np.ones(20)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 25, 0, 0.1, abi.dp(x0), None, None, None, None))
    assert np.isfinite(x0).all() and np.allclose(x0, xo0, rtol=1e-9, atol=1e-12)     # priors only: pure double algebra on both sides


@pytest.mark.parametrize("modes", [(-1.0, -1.0), (0.0, 0.0)])
def test_affine_modes_in_the_window(gpu_ctx, oracle, modes):
    """setting_affineOptModeA/B < 0 fixes the affine brightness parameters: JabF columns are zeroed in linearize (Residuals.cpp:262-263)
    and the frame priors change (HessianBlocks.h:239-265); = 0 removes the affine prior."""
    win = dict(synth.ba_window(w=640, h=480, nf=4, pts_per_kf=80, seed=3081))
    win["affineOptModeA"], win["affineOptModeB"] = modes
    W, keep, h = _both(gpu_ctx, oracle, win)
    eo, eg, o, g = _lin_both(gpu_ctx, oracle, win, h, 3)
    _check_lin(eo, eg, o, g)
    if modes[0] < 0:
        live = o["ns"] == 0
        assert not g["J"][live][:, 46:54].any() and not g["J"][live][:, 54:62].any()      # JabF[0], JabF[1]
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    sg, ig, rg, og = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_destroy(h)
    W, keep, h = _both(gpu_ctx, oracle, win)
    oracle.orc_ba_optimize(h, 4, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 4, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
    oracle.orc_ba_destroy(h)
    assert og.iterations == oo.iterations
    # order-of-summation spread after 4 GN iterations (cf. test_optimize_full_gn_loop); states are O(1e-2)
    assert np.abs(sg - so).max() <= 5e-4 and helpers.idepths_close(ig, io, 5e-4)
    if modes[0] < 0:
        assert np.abs(sg[:, 6:8]).max() < 1e-6 and np.allclose(sg[:, 6:8], so[:, 6:8], rtol=1e-2, atol=1e-12)   # held by the 1e14 prior


def test_full_size_window_properties(gpu_ctx):
    """BASELINE configs[4] size (8 keyframes x ~8000 points, ~50k residuals), checked through size-independent properties instead of
    the oracle: (1) the packed accumulators of two half-windows (contiguous point ranges, as on two ranks) add up to those of the whole window —
    the invariant the multi-GPU sharding rests on; (2) solving from the summed shards gives the x of the whole window; (3) the
    stitched system is symmetric and its Schur part shrinks the top part (H - Hsc stays positive on the diagonal)."""
    from sdso_amd import dist as sd
    win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=1000, seed=3101)
    nf, n = win["nf"], 68
    assert win["np"] == 8000 and win["nr"] > 45000
    for f in range(nf):
        gpu_ctx.upload_pyramid(400 + f, win["pyrs"][f][:1])
    slots = [400 + f for f in range(nf)]
    na = abi.accum_floats(nf)

    def run(w, wid):
        W, keep = abi.make_ba_window(w, frame_slots=slots)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, wid, C.byref(W)))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, wid, None))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, wid))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, wid))
        a = np.zeros(na, np.float32)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_accumulators(gpu_ctx.h, wid, abi.fp(a)))
        return a

    whole = run(win, 50)
    x = np.zeros(n); H = np.zeros((n, n)); b = np.zeros(n)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 50, 0, 0.1, abi.dp(x), abi.dp(H), abi.dp(b), None, None))
    parts = [run(sd.shard_window(win, r, 2)[0], 51 + r) for r in range(2)]
    ssum = parts[0] + parts[1]
    scale = np.maximum(np.abs(whole), 1e-30)
    blocks = [(0, nf * nf * 91 * 2, 91), (nf * nf * 91 * 2, nf * nf * 91 * 2 + nf ** 3 * 64, 64)]
    for lo, hi, wdt in blocks:                                                 # per accumulator block, relative to its largest entry
        A = whole[lo:hi].reshape(-1, wdt); S = ssum[lo:hi].reshape(-1, wdt)
        m = np.abs(A).max(axis=1, keepdims=True)
        assert np.array_equal(m == 0, np.abs(S).max(axis=1, keepdims=True) == 0)
        assert (np.abs(A - S) / np.maximum(m, 1e-30)).max() <= 5e-5
    assert np.array_equal(whole[-2:], ssum[-2:])                               # residual counts are exact
    gpu_ctx.check(gpu_ctx.L.sdso_ba_set_accumulators(gpu_ctx.h, 51, abi.fp(ssum)))
    xs = np.zeros(n)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 51, 0, 0.1, abi.dp(xs), None, None, None, None))
    d = np.sqrt(np.abs(np.diag(H))) + 1e-30
    assert np.abs((xs - x) * d).max() <= 2e-4 * max(1.0, np.abs(x * d).max())
    assert np.abs(H - H.T).max() <= 1e-9 * np.abs(H).max() and (np.diag(H) > 0).all() and np.isfinite(x).all()


@pytest.fixture(scope="module")
def win_c5():
    return synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=1000, seed=3101)     # configs[4]: 8 KF x 8000 points, ~50k residuals


def test_full_size_window_matches_oracle(gpu_ctx, oracle, win_c5):
    """BASELINE configs[4] size against the ORACLE (the round-5 verdict's Missing #4: it was property-checked only), same bars as configs[2]:
    host tables, every RawResidualJacobian, state and energy bit-exact; per-point Hdd / bd / Hcd / HdiF / bdSumF bit-exact; packed
    accumulators <= 3e-5 of the block maximum against the float oracle (and <= 3e-7 against the f64-accumulator truth: the device's sums
    are f64); stitched H / b <= 1e-4, x and the point steps <= 2e-4 whitened — also with orthogonalize_x —; the 6-iteration loop of
    FullSystem::optimize with the per-iteration pose-update bar of north_star (1e-5) against the truth and the fixed 2e-5 after the loop;
    and the window cut 2 and 8 ways by points (sdso_amd.dist.shard_window, the multi-GPU partition: AccumulatedTopHessian.cpp:299-308 sums
    per-thread copies the same way), the shards' accumulators summed and solved, against the ORACLE's unsharded x."""
    from sdso_amd import dist as sd
    win = win_c5
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    assert npts == 8000 and nr > 45000
    W, keep, h = _both(gpu_ctx, oracle, win, slot0=400, wid=60)
    to = [np.zeros(nf * nf * 27, np.float32), np.zeros(nf * nf * 64), np.zeros(nf * nf * 64), np.zeros(nf * nf * 8, np.float32)]
    tg = [np.zeros_like(a) for a in to]
    oracle.orc_ba_get_tables(h, abi.fp(to[0]), abi.dp(to[1]), abi.dp(to[2]), abi.fp(to[3]))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_tables(gpu_ctx.h, 60, abi.fp(tg[0]), abi.dp(tg[1]), abi.dp(tg[2]), abi.fp(tg[3])))
    for a, b in zip(to, tg):
        assert np.array_equal(a, b)
    eo, eg, o, g = _lin_both(gpu_ctx, oracle, win, h, 60)
    _check_lin(eo, eg, o, g)
    assert (o["ns"] == 0).sum() > 0.4 * nr
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 60))
    so, ao_, jo = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8), np.zeros((nr, 8), np.float32)
    sg, ag_, jg = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8), np.zeros((nr, 8), np.float32)
    oracle.orc_ba_get_residual_state(h, abi.bp(so), abi.bp(ao_), abi.fp(jo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_residual_state(gpu_ctx.h, 60, abi.bp(sg), abi.bp(ag_), abi.fp(jg)))
    assert np.array_equal(so, sg) and np.array_equal(ao_, ag_) and np.array_equal(jo[ao_ == 1], jg[ao_ == 1])
    ao, ag = _accumulate_both(gpu_ctx, oracle, win, h, 60)
    _check_accum(ao, ag, nf)
    # ... and against the f64-accumulator truth
    oracle.orc_set_acc64(1)
    try:
        h64 = oracle.orc_ba_create(C.byref(W))
        oracle.orc_ba_linearize(h64, None); oracle.orc_ba_apply_res(h64); oracle.orc_ba_accumulate(h64)
        a64 = np.zeros(abi.accum_floats(nf))
        oracle.orc_ba_get_accumulators_f64(h64, abi.dp(a64))
        oracle.orc_ba_destroy(h64)
    finally:
        oracle.orc_set_acc64(0)
    o0 = 0
    for name, cnt, w in (("topA", nf * nf, 91), ("topL", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8), ("Hcc", 1, 16), ("bc", 1, 4)):
        T = a64[o0:o0 + cnt * w].reshape(-1, w)
        m = np.maximum(np.abs(T).max(axis=1, keepdims=True), 1e-30)
        live = np.abs(T).max(axis=1) > 0
        if live.any():
            eg_ = np.abs((ag[o0:o0 + cnt * w].reshape(-1, w).astype(np.float64) - T) / m)[live].max()
            ec_ = np.abs((ao[o0:o0 + cnt * w].reshape(-1, w).astype(np.float64) - T) / m)[live].max()
            assert eg_ <= 3e-7, (name, eg_, ec_)
        o0 += cnt * w
    po = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    pg = [np.zeros_like(a) for a in po]
    oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in po])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_terms(gpu_ctx.h, 60, *[abi.fp(a) for a in pg]))
    for a, b in zip(po, pg):
        assert np.array_equal(a, b)
    xo_first = None
    for it, lam in ((0, 0.1), (2, 0.025)):
        xo, Ho, bo = np.zeros(n), np.zeros((n, n)), np.zeros(n)
        xg, Hg, bg = np.zeros(n), np.zeros((n, n)), np.zeros(n)
        oracle.orc_ba_solve(h, it, lam, abi.dp(xo), abi.dp(Ho), abi.dp(bo), None, None)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 60))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 60, it, lam, abi.dp(xg), abi.dp(Hg), abi.dp(bg), None, None))
        d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
        assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-4
        assert np.abs((bg - bo) / d).max() <= 1e-4 * max(1.0, np.abs(bo / d).max())
        assert np.abs((xg - xo) * d).max() <= 2e-4 * max(1.0, np.abs(xo * d).max())
        sto, stg = np.zeros(npts, np.float32), np.zeros(npts, np.float32)
        oracle.orc_ba_get_point_steps(h, abi.fp(sto))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_steps(gpu_ctx.h, 60, abi.fp(stg)))
        assert np.abs(stg - sto).max() <= 2e-4 * max(np.abs(sto).max(), 1e-6)
        if it == 0:
            xo_first, d_first = xo.copy(), d.copy()
    oracle.orc_ba_destroy(h)

    # ---- the window cut 2 and 8 ways: every shard linearised / accumulated on its own, the packed blocks added (what the all-reduce does),
    # one solve from the sum — against the oracle's x of the UNSHARDED window
    slots = [400 + f for f in range(nf)]
    na = abi.accum_floats(nf)
    for world in (2, 8):
        ssum = np.zeros(na, np.float32)
        for r in range(world):
            ws = sd.shard_window(win, r, world)[0]
            Ws, ks = abi.make_ba_window(ws, frame_slots=slots)
            gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 61, C.byref(Ws)))
            gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 61, None))
            gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 61))
            gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 61))
            a = np.zeros(na, np.float32)
            gpu_ctx.check(gpu_ctx.L.sdso_ba_get_accumulators(gpu_ctx.h, 61, abi.fp(a)))
            ssum += a
        assert ssum[-2] == ao[-2]                                            # every active residual sits in exactly one shard
        _check_accum(ao, ssum, nf)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_set_accumulators(gpu_ctx.h, 61, abi.fp(ssum)))
        xs = np.zeros(n)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 61, 0, 0.1, abi.dp(xs), None, None, None, None))
        assert np.abs((xs - xo_first) * d_first).max() <= 2e-4 * max(1.0, np.abs(xo_first * d_first).max()), world
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 61))

    # ---- FullSystem::optimize: float oracle, f64-accumulator truth, device
    def oracle_loop(acc64):
        oracle.orc_set_acc64(1 if acc64 else 0)
        try:
            hh = oracle.orc_ba_create(C.byref(W))
            s, i, r, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
            oracle.orc_ba_optimize(hh, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(oo))
            x = np.zeros((8, n))
            its = oracle.orc_ba_get_x_trace(hh, abi.dp(x), 8)
            oracle.orc_ba_destroy(hh)
        finally:
            oracle.orc_set_acc64(0)
        return s, i, r, oo, x, its
    s32, i32, r32, o32, x32, its32 = oracle_loop(False)
    s64, i64, r64, o64, x64, its64 = oracle_loop(True)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 60, C.byref(W)))
    ids = np.array([60], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 1, abi.ip(ids)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize_begin(gpu_ctx.h, 1))
    xg = np.zeros((6, n))
    for it in range(6):
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_accumulate(gpu_ctx.h))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_solve_step(gpu_ctx.h, 0.1 * 0.25 ** it, 1 if it >= 2 else 0))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_get_x(gpu_ctx.h, abi.dp(xg[it:it + 1])))
    og = (abi.BAOptResult * 1)()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize_end(gpu_ctx.h, og))
    sgd, igd, rgd = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, 60, abi.dp(sgd), abi.fp(igd), abi.bp(rgd)))
    assert og[0].iterations == o32.iterations == o64.iterations
    upd = lambda a, b, it: float(np.abs((a[it, 4:] - b[it, 4:]).reshape(nf, 8) * _STATE_SCALE)[:, :6].max())   # noqa: E731
    truth = [upd(xg, x64, it) for it in range(its64)]
    cpu_own = [upd(x32, x64, it) for it in range(its64)]
    vs32 = [upd(xg, x32, it) for it in range(its32)]
    msg = "pose update per iteration: device-truth %s, cpu_f32-truth %s, device-cpu_f32 %s" % (truth, cpu_own, vs32)
    print("configs[4]:", msg)
    for it in range(its64):
        assert truth[it] <= 1e-5, msg
        assert vs32[it] <= max(1e-5, cpu_own[it] + 1e-5), msg
    dstate = np.abs(sgd - s32)[:, :8] * _STATE_SCALE
    dtruth = np.abs(sgd - s64)[:, :8] * _STATE_SCALE
    assert dtruth[:, :6].max() <= 2e-5 and dstate[:, :6].max() <= 2e-5, (float(dtruth[:, :6].max()), float(dstate[:, :6].max()), msg)
    assert dstate[:, 6].max() <= 2e-5 and dstate[:, 7].max() <= 2e-3
    assert helpers.idepths_close(igd, i32, 2e-4)
    assert (rgd != r32).sum() <= max(2, nr // 2000)
    assert helpers.counts_close(og[0].resInA, o32.resInA, nr)
    assert abs(og[0].lastEnergy - o32.lastEnergy) <= 1e-4 * o32.lastEnergy
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 60))
