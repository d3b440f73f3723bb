"""marginalizePointsF at BASELINE configs[2] size through the product kernels, and the linearised pass that follows it.

flagPointsForRemoval (FullSystem.cpp:1004-1021) re-linearises the residuals of every flagged point, `fixLinearizationF`s them
(EnergyFunctionalStructs.cpp:96-123: res_toZeroF = resF - J*delta, isLinearized = true), and marginalizePointsF
(EnergyFunctional.cpp:663-736) accumulates them with addPoint<2> and the Schur accumulators into HM / bM.  While such residuals stay
in the window every later solveSystemF runs them through the LINEARISED pass accumulateLF = addPoint<1>
(AccumulatedTopHessian.cpp:89-111: resApprox = res_toZeroF + J*delta with the CURRENT deltas) — the pass no other test or
bench line populates.  Patterns: no point, a random 10 %, every point of one host, every point; with and without an incoming prior.
The synthetic windows have state != state_zero and idepth != idepth_zero, so J*delta is not zero."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi
import synth
from test_ba_gpu import _both, _lin_both, _check_lin, _accumulate_both, _check_accum

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def win_c3():
    return synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)


def _flags(win, pattern):
    npts = win["np"]
    if pattern == "none":
        return np.zeros(npts, np.uint8)
    if pattern == "all":
        return np.ones(npts, np.uint8)
    if pattern == "host2":
        return (win["host"] == 2).astype(np.uint8)
    f = (np.random.RandomState(77).uniform(size=npts) < 0.10).astype(np.uint8)
    assert 0 < f.sum() < npts
    return f


def _oracle_sequence(oracle, win, flag):
    """The oracle alone through the sequence of test_marginalize_points_c3; returns the states / idepths after the GN loop."""
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    x = np.zeros(n)
    oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
    oracle.orc_ba_marginalize_points(h, abi.bp(np.ascontiguousarray(flag)), None, None)
    oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
    oracle.orc_ba_solve(h, 0, 0.1, abi.dp(x), None, None, None, None)
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_optimize(h, 4, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o))
    oracle.orc_ba_destroy(h)
    return s, i


@pytest.mark.parametrize("pattern,prior", [("none", False), ("random10", False), ("random10", True), ("host2", False), ("all", False)])
def test_marginalize_points_c3(gpu_ctx, oracle, win_c3, pattern, prior):
    win = dict(win_c3)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    if prior:
        A = np.random.RandomState(5).normal(size=(n, 6))
        win["HM"] = (A @ A.T) * 1e3
        win["bM"] = np.random.RandomState(6).normal(size=n) * 10
    W, keep, h = _both(gpu_ctx, oracle, win)
    HM0 = np.array(win["HM"], np.float64).reshape(n, n) if win.get("HM") is not None else np.zeros((n, n))
    eo, eg, o, g = _lin_both(gpu_ctx, oracle, win, h, 3)
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    _accumulate_both(gpu_ctx, oracle, win, h, 3)
    flag = _flags(win, pattern)

    # ---- marginalizePointsF
    HMo, bMo, HMg, bMg = np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HMo), abi.dp(bMo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_marginalize_points(gpu_ctx.h, 3, abi.bp(flag), abi.dp(HMg), abi.dp(bMg)))
    if pattern == "none":
        assert np.array_equal(HMo, HM0) and np.array_equal(HMg, HM0)          # nothing flagged: HM += 0.25 * (0 - 0)
        assert not bMg.any()
    else:
        assert np.abs(HMo - HM0).max() > 0
        d = np.sqrt(np.abs(np.diag(HMo))) + 1e-30
        live = np.abs(np.diag(HMo)) > 0
        assert live.sum() >= 8
        assert np.abs(((HMg - HMo) / np.outer(d, d))[np.ix_(live, live)]).max() <= 1e-4
        assert np.abs(((bMg - bMo) / d)[live]).max() <= 1e-4 * max(1.0, np.abs((bMo / d)[live]).max())
        assert np.array_equal(HMg == 0, HMo == 0)                              # same frames touched
    so, ao, jo = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8), np.zeros((nr, 8), np.float32)
    sg, ag, jg = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8), np.zeros((nr, 8), np.float32)
    oracle.orc_ba_get_residual_state(h, abi.bp(so), abi.bp(ao), abi.fp(jo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_residual_state(gpu_ctx.h, 3, abi.bp(sg), abi.bp(ag), abi.fp(jg)))
    assert np.array_equal(so, sg) and np.array_equal(ao, ag) and np.array_equal(jo, jg)   # flagged residuals were re-linearised + applied

    # ---- the following GN iteration: linearizeAll + applyRes, accumulateAF (mode 0) + accumulateLF (mode 1) + SC, solve with the new prior
    eo, eg, o, g = _lin_both(gpu_ctx, oracle, win, h, 3)
    _check_lin(eo, eg, o, g)
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    acc_o, acc_g = _accumulate_both(gpu_ctx, oracle, win, h, 3)
    _check_accum(acc_o, acc_g, nf)
    topA, topL = acc_o[:nf * nf * 91], acc_o[nf * nf * 91:2 * nf * nf * 91]
    if pattern == "none":
        assert not topL.any() and topA.any()
    elif pattern == "all":
        assert topL.any() and not topA.any()                                   # every active residual now sits in the linearised pass
    else:
        assert topL.any() and topA.any()
    po = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    pg = [np.zeros_like(a) for a in po]
    oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in po])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_terms(gpu_ctx.h, 3, *[abi.fp(a) for a in pg]))
    for a, b in zip(po, pg):
        assert np.array_equal(a, b)                                            # HdiF, bdSumF, Hdd/bd/Hcd incl. the accLF parts: bit-exact
    xo, Ho, bo = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    xg, Hg, bg = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_solve(h, 0, 0.1, abi.dp(xo), abi.dp(Ho), abi.dp(bo), None, None)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_solve(gpu_ctx.h, 3, 0, 0.1, abi.dp(xg), abi.dp(Hg), abi.dp(bg), None, None))
    d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
    assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-4
    assert np.abs((bg - bo) / d).max() <= 1e-4 * max(1.0, np.abs(bo / d).max())
    assert np.abs((xg - xo) * d).max() <= 2e-4 * max(1.0, np.abs(xo * d).max())
    sto, stg = np.zeros(npts, np.float32), np.zeros(npts, np.float32)
    oracle.orc_ba_get_point_steps(h, abi.fp(sto))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_point_steps(gpu_ctx.h, 3, abi.fp(stg)))
    assert np.abs(stg - sto).max() <= 2e-4 * max(np.abs(sto).max(), 1e-6)

    # ---- and a whole GN loop over the window with linearised residuals in it (deltas grow: J*delta of the linearised pass moves)
    if pattern in ("random10", "host2"):
        s_o, i_o, r_o, o_o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        s_g, i_g, r_g, o_g = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h, 4, abi.dp(s_o), abi.fp(i_o), abi.bp(r_o), C.byref(o_o))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 4, abi.dp(s_g), abi.fp(i_g), abi.bp(r_g), C.byref(o_g)))
        assert o_g.iterations == o_o.iterations
        # float summation order makes the CPU path itself order-dependent: measure that spread by running the oracle through the same
        # sequence with the points shuffled inside each host group (cf. test_optimize_full_gn_loop) and allow 1e-5 + twice the spread
        import helpers
        spread_s, spread_i = 0.0, 0.0
        for seed in (1, 3):
            w2, order = helpers.permuted_window(win, seed)
            s_p, i_p = _oracle_sequence(oracle, w2, flag[order])
            spread_s = max(spread_s, np.abs(s_p - s_o).max())
            spread_i = max(spread_i, np.abs(i_p - i_o[order]).max())
        assert np.abs(s_g - s_o).max() <= 1e-5 + 2.0 * spread_s, (np.abs(s_g - s_o).max(), spread_s)
        assert np.abs(i_g - i_o).max() <= 1e-5 + 2.0 * spread_i, (np.abs(i_g - i_o).max(), spread_i)
        assert abs(o_g.lastEnergy - o_o.lastEnergy) <= 1e-3 * o_o.lastEnergy
    oracle.orc_ba_destroy(h)


def test_marginalize_in_batch_member(gpu_ctx, oracle, win_c3):
    """The fused batch kernel honours isLinearized too: marginalise 10 % of a window, then advance it through
    sdso_ba_batch_accumulate / sdso_ba_batch_solve and compare with the single-window path and the oracle."""
    win = dict(win_c3)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    W, keep, h = _both(gpu_ctx, oracle, win)
    _lin_both(gpu_ctx, oracle, win, h, 3)
    oracle.orc_ba_apply_res(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
    _accumulate_both(gpu_ctx, oracle, win, h, 3)
    flag = _flags(win, "random10")
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), None, None)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_marginalize_points(gpu_ctx.h, 3, abi.bp(flag), None, None))
    oracle.orc_ba_linearize(h, None)
    oracle.orc_ba_apply_res(h)
    oracle.orc_ba_accumulate(h)
    acc_o = np.zeros(abi.accum_floats(nf), np.float32)
    oracle.orc_ba_get_accumulators(h, abi.fp(acc_o))
    xo, Ho = np.zeros(n), np.zeros((n, n))
    oracle.orc_ba_solve(h, 0, 0.1, abi.dp(xo), abi.dp(Ho), None, None, None)
    ids = np.array([3], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 1, abi.ip(ids)))
    for mat in (1, 0):
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_set_materialize(gpu_ctx.h, mat))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_accumulate(gpu_ctx.h))
        acc_g = np.zeros(abi.accum_floats(nf), np.float32)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_accumulators(gpu_ctx.h, 3, abi.fp(acc_g)))
        _check_accum(acc_o, acc_g, nf)
        assert acc_g[nf * nf * 91:2 * nf * nf * 91].any()
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_solve(gpu_ctx.h, 0.1, 0))      # FIX_LAMBDA (the default solverMode) makes it 1e-5 on both sides
        xg = np.zeros(n)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_get_x(gpu_ctx.h, abi.dp(xg)))
        d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
        assert np.abs((xg - xo) * d).max() <= 2e-4 * max(1.0, np.abs(xo * d).max())
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 3))
    oracle.orc_ba_destroy(h)


def test_marginalize_frame_on_the_device_resident_prior(gpu_ctx, oracle, win_c3):
    """EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660) inside the window's life on the device: HM / bM stay resident from
    sdso_ba_marginalize_points through sdso_ba_marginalize_frame_dev into the next window (sdso_ba_adopt_prior) — no host copy on the way.
      (1) the kernel against orc_marginalize_frame on the SAME input (the device's own prior after the points' marginalisation, with the
          frame's EFFrame::prior / delta_prior): <= 1e-9 whitened (the 8 x 8 corner carries the 1e10 .. 1e14 priors of the first keyframe
          and the two elimination orders of its inverse round differently — the bar tests/test_abi_cpu.py holds the host statement
          to), exactly symmetric, and IDENTICAL, bit for bit, to the library's host statement of the same algebra;
      (2) the chain marginalise points -> marginalise frame -> next window -> solve equals the host chain (prior copied out, host algebra,
          uploaded with the next window): the same x;
      (3) the whole chain against the ORACLE's chain at the loop bars (the points' marginalisation carries float accumulation)."""
    import helpers
    ctx = gpu_ctx
    win = dict(win_c3)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    m = n - 8
    A = np.random.RandomState(5).normal(size=(n, 6))
    win["HM"] = (A @ A.T) * 1e4
    win["bM"] = np.random.RandomState(6).normal(size=n) * 1e2
    flag = (win["host"] == 0).astype(np.uint8)            # marginalizeFrame asserts fh->points.size() == 0: all of its points go first
    W, keep, h = _both(ctx, oracle, win, wid=21)
    _lin_both(ctx, oracle, win, h, 21)
    oracle.orc_ba_apply_res(h); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 21))
    _accumulate_both(ctx, oracle, win, h, 21)
    HMo, bMo = np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HMo), abi.dp(bMo))
    ctx.check(ctx.L.sdso_ba_marginalize_points(ctx.h, 21, abi.bp(flag), None, None))       # (no copy out: the prior stays on the device)
    Hg, bg = np.zeros((m, m)), np.zeros(m)
    ctx.check(ctx.L.sdso_ba_marginalize_frame_dev(ctx.h, 21, 0, abi.dp(Hg), abi.dp(bg)))
    # EFFrame::prior of the frame with frameID 0 (settings.cpp:44-48) and delta_prior = its state (EnergyFunctionalStructs.cpp:59-61)
    assert win["frameID"][0] == 0
    prior = np.array([1e10] * 3 + [1e11] * 3 + [1e14, 1e14], np.float32).astype(np.float64)      # (the settings are floats: 1e11f = 99999997952)
    dprior = np.ascontiguousarray(np.asarray(win["state"], np.float64).reshape(nf, 10)[0, :8])
    # (1) the same input through the oracle and through the library's host statement
    HMd, bMd = np.zeros((n, n)), np.zeros(n)
    ctx.check(ctx.L.sdso_ba_marginalize_points(ctx.h, 21, abi.bp(np.zeros(npts, np.uint8)), abi.dp(HMd), abi.dp(bMd)))   # nothing flagged: HM unchanged, copied out
    Ho, bo, Hh, bh = np.zeros((m, m)), np.zeros(m), np.zeros((m, m)), np.zeros(m)
    assert oracle.orc_marginalize_frame(nf, 0, abi.dp(prior), abi.dp(dprior), abi.dp(HMd), abi.dp(bMd), abi.dp(Ho), abi.dp(bo)) == 0
    assert ctx.L.sdso_ba_marginalize_frame(nf, 0, abi.dp(prior), abi.dp(dprior), abi.dp(HMd), abi.dp(bMd), abi.dp(Hh), abi.dp(bh)) == 0
    d = np.sqrt(np.abs(np.diag(Ho))) + 1e-300
    assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-9 and np.abs((bg - bo) / d).max() <= 1e-9 * max(1.0, np.abs(bo / d).max())
    assert np.array_equal(Hg, Hg.T) and np.array_equal(Hg, Hh) and np.array_equal(bg, bh)
    # (3) against the oracle's own chain
    Hoo, boo = np.zeros((m, m)), np.zeros(m)
    oracle.orc_marginalize_frame(nf, 0, abi.dp(prior), abi.dp(dprior), abi.dp(HMo), abi.dp(bMo), abi.dp(Hoo), abi.dp(boo))
    live = np.abs(np.diag(Hoo)) > 0
    do = np.sqrt(np.abs(np.diag(Hoo))) + 1e-300
    assert np.abs(((Hg - Hoo) / np.outer(do, do))[np.ix_(live, live)]).max() <= 1e-4
    oracle.orc_ba_destroy(h)
    # (2) the next window: the seven remaining keyframes; the device chain adopts the prior, the host chain uploads it
    w2 = helpers.drop_frame(win, 0)
    n2 = 8 * w2["nf"] + 4
    assert n2 == m
    xs = []
    for chain in ("device", "host"):
        wk = dict(w2)
        if chain == "host":
            wk["HM"], wk["bM"] = Hh.copy(), bh.copy()
        for f in range(wk["nf"]):
            ctx.upload_pyramid(60 + f, wk["pyrs"][f][:1])
        W2, keep2 = abi.make_ba_window(wk, frame_slots=[60 + f for f in range(wk["nf"])], dI_list=[p[0] for p in wk["pyrs"]])
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 22, C.byref(W2)))
        if chain == "device":
            ctx.check(ctx.L.sdso_ba_adopt_prior(ctx.h, 22, 21))
        ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 22, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 22)); ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 22))
        x, HS = np.zeros(n2), np.zeros((n2, n2))
        ctx.check(ctx.L.sdso_ba_solve(ctx.h, 22, 0, 0.1, abi.dp(x), abi.dp(HS), None, None, None))
        xs.append((x, HS))
    assert np.abs(xs[0][0]).max() > 0 and np.array_equal(xs[0][0], xs[1][0]) and np.array_equal(xs[0][1], xs[1][1])
    assert ctx.L.sdso_ba_adopt_prior(ctx.h, 22, 22) == -1 and ctx.L.sdso_ba_marginalize_frame_dev(ctx.h, 22, 9, None, None) == -1
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 21)); ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 22))


def test_two_frames_leave_at_one_keyframe_on_the_device(gpu_ctx, win_c3):
    """FullSystem::makeKeyFrame marginalises EVERY flagged frame (FullSystem.cpp:1470-1476), each on the prior the previous call left and with
    EFFrame::idx counted after the earlier erase.  On the device: two sdso_ba_marginalize_frame_dev calls in a row chain on the resident
    result, bit-identical to two calls of the host statement (sdso_ba_marginalize_frame); sdso_ba_marginalize_points in between restarts
    from the window's prior; sdso_ba_adopt_prior checks by frameID that the adopting window's leading frames are the surviving ones
    (round-5 advisor finding: a prior attached to other frames was accepted silently)."""
    import helpers
    ctx = gpu_ctx
    win = dict(win_c3)
    nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
    A = np.random.RandomState(15).normal(size=(n, 9))
    win["HM"] = (A @ A.T) * 1e4
    win["bM"] = np.random.RandomState(16).normal(size=n) * 1e2
    for f in range(nf):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 23, C.byref(W)))
    st = np.asarray(win["state"], np.float64).reshape(nf, 10)
    prior0 = np.array([1e10] * 3 + [1e11] * 3 + [1e14, 1e14], np.float32).astype(np.float64)      # frameID 0 (HessianBlocks.h:239-265)
    assert win["frameID"][0] == 0 and win["frameID"][3] != 0
    priork = np.zeros(8)
    priork[6:] = [win["affineOptModeA"], win["affineOptModeB"]]                                    # setting_affineOptModeA / B as the window carries them (HessianBlocks.h:250-258)
    # frame 0 leaves, then the frame that was at index 3 (index 2 of what is left)
    m1, m2 = n - 8, n - 16
    H1, b1, H2, b2 = np.zeros((m1, m1)), np.zeros(m1), np.zeros((m2, m2)), np.zeros(m2)
    ctx.check(ctx.L.sdso_ba_marginalize_frame_dev(ctx.h, 23, 0, abi.dp(H1), abi.dp(b1)))
    assert ctx.L.sdso_ba_marginalize_frame_dev(ctx.h, 23, 7, None, None) == -1                      # seven frames are left: indices 0..6
    ctx.check(ctx.L.sdso_ba_marginalize_frame_dev(ctx.h, 23, 2, abi.dp(H2), abi.dp(b2)))
    Hh1, bh1, Hh2, bh2 = np.zeros((m1, m1)), np.zeros(m1), np.zeros((m2, m2)), np.zeros(m2)
    HM0, bM0 = np.ascontiguousarray(win["HM"], np.float64), np.ascontiguousarray(win["bM"], np.float64)
    assert ctx.L.sdso_ba_marginalize_frame(nf, 0, abi.dp(prior0), abi.dp(np.ascontiguousarray(st[0, :8])), abi.dp(HM0), abi.dp(bM0), abi.dp(Hh1), abi.dp(bh1)) == 0
    assert ctx.L.sdso_ba_marginalize_frame(nf - 1, 2, abi.dp(priork), abi.dp(np.ascontiguousarray(st[3, :8])), abi.dp(Hh1), abi.dp(bh1), abi.dp(Hh2), abi.dp(bh2)) == 0
    assert np.array_equal(H1, Hh1) and np.array_equal(b1, bh1)
    assert np.array_equal(H2, Hh2) and np.array_equal(b2, bh2) and np.abs(H2).max() > 0
    # the next window: the six surviving keyframes (+ nothing new); leading frames checked by frameID
    w2 = helpers.drop_frame(helpers.drop_frame(win, 0), 2)
    for f in range(w2["nf"]):
        ctx.upload_pyramid(60 + f, w2["pyrs"][f][:1])
    W2, keep2 = abi.make_ba_window(w2, frame_slots=[60 + f for f in range(w2["nf"])], dI_list=[p[0] for p in w2["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 24, C.byref(W2)))
    ctx.check(ctx.L.sdso_ba_adopt_prior(ctx.h, 24, 23))
    assert ctx.L.sdso_ba_adopt_prior(ctx.h, 24, 23) == -1                                           # its prior is no longer the uploaded zero
    HMa, bMa = np.zeros((m2, m2)), np.zeros(m2)
    ctx.check(ctx.L.sdso_ba_marginalize_points(ctx.h, 24, abi.bp(np.zeros(w2["np"], np.uint8)), abi.dp(HMa), abi.dp(bMa)))   # nothing flagged: the prior, copied out
    assert np.array_equal(HMa, H2) and np.array_equal(bMa, b2)
    # a window whose leading frames are NOT the survivors (frame 1 dropped instead of frame 0) is refused
    w3 = helpers.drop_frame(helpers.drop_frame(win, 1), 2)
    W3, keep3 = abi.make_ba_window(w3, frame_slots=[60 + f for f in range(w3["nf"])], dI_list=[p[0] for p in w3["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 25, C.byref(W3)))
    assert ctx.L.sdso_ba_adopt_prior(ctx.h, 25, 23) == -1
    # sdso_ba_marginalize_points restarts the chain from the window's own prior
    ctx.check(ctx.L.sdso_ba_marginalize_points(ctx.h, 23, abi.bp(np.zeros(npts, np.uint8)), None, None))
    H1b, b1b = np.zeros((m1, m1)), np.zeros(m1)
    ctx.check(ctx.L.sdso_ba_marginalize_frame_dev(ctx.h, 23, 0, abi.dp(H1b), abi.dp(b1b)))
    assert np.array_equal(H1b, H1) and np.array_equal(b1b, b1)
    for wid in (23, 24, 25):
        ctx.check(ctx.L.sdso_ba_release_window(ctx.h, wid))
