"""The device-resident Gauss-Newton loop (csrc/ba_opt.hip: sdso_ba_optimize's accepted-step path, sdso_ba_batch_optimize,
sdso_ba_batch_optimize_begin / sdso_ba_batch_step / _end) against the oracle's FullSystem::optimize and against the library's own host
loop (SDSO_BA_HOST_LOOP=1, see test_resident_equals_host_loop).

Bars: iteration counts, resInA and the residual states equal; states / idepths within 1e-5 + twice the oracle's own order-of-summation
spread (cf. tests/test_ba_gpu.py::test_optimize_full_gn_loop); setNewFrameEnergyTH's quantile is an order statistic, so the device radix
select must reproduce nth_element exactly — checked through the residual states (IN / OUTLIER decisions) and lastEnergy."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi
import synth
import helpers

pytestmark = pytest.mark.gpu


def _oracle_opt(oracle, win, its):
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    W, keep = abi.make_ba_window(win, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_optimize(h, its, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o))
    oracle.orc_ba_destroy(h)
    return s, i, r, o


def _spread(oracle, win, its, so, io):
    ss, si = 0.0, 0.0
    for seed in (1, 3):
        w2, order = helpers.permuted_window(win, seed)
        sp, ip, rp, op = _oracle_opt(oracle, w2, its)
        ss = max(ss, np.abs(sp - so).max())
        si = max(si, np.abs(ip - io[order]).max())
    return ss, si


def _check(win, got, ref, spread):
    sg, ig, rg, og = got
    so, io, ro, oo = ref
    assert og.iterations == oo.iterations
    assert np.abs(sg - so).max() <= 1e-5 + 2.0 * spread[0], (np.abs(sg - so).max(), spread)
    # idepths: the bulk within the oracle's own spread; the maximum sits on single weakly observed points whose trajectory over the loop
    # amplifies any rounding difference (either path: tests/diag/truth_spread.py) — bounded, not compared point by point
    di = np.abs(ig - io)
    assert np.percentile(di, 99) <= 1e-5 + 2.0 * spread[1], (np.percentile(di, 99), spread)
    assert di.max() <= 1e-5 + 10.0 * spread[1], (di.max(), spread)
    mism = int((rg != ro).sum())
    assert mism <= max(2, win["nr"] // 2000)                  # IN / OUTLIER flips only where an energy sits on the threshold
    assert helpers.counts_close(og.resInA, oo.resInA, win["nr"])     # (resInA counts the LAST solve's residuals: a flip inside the loop moves it by one)
    # (the energy of the closing linearisation: one residual moving across its clamp changes it by ~1e-4 of the total on the small windows)
    assert abs(og.lastEnergy - oo.lastEnergy) <= 1e-3 * oo.lastEnergy
    assert abs(og.rmse - oo.rmse) <= 1e-3 * oo.rmse


WINS = {
    "small": dict(w=640, h=480, nf=5, pts_per_kf=120, seed=3001),
    "c3": dict(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001),
    "noisy": dict(w=640, h=480, nf=6, pts_per_kf=150, seed=3017, idepth_noise=0.3, state_noise=1e-2),   # larger steps, no early break
    "three_kf": dict(w=640, h=480, nf=3, pts_per_kf=150, seed=3019),                                     # nf < 4: 15 iterations
}


@pytest.mark.parametrize("which", list(WINS))
def test_single_window_resident_loop(gpu_ctx, oracle, which):
    """sdso_ba_optimize with forceAcceptStep (the default): the whole loop runs on the device."""
    win = synth.ba_window(**WINS[which])
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    assert win.get("forceAcceptStep", 1)
    ref = _oracle_opt(oracle, win, 6)
    for f in range(nf):
        gpu_ctx.upload_pyramid(500 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[500 + f for f in range(nf)])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 50, C.byref(W)))
    sg, ig, rg, og = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 50, 6, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
    _check(win, (sg, ig, rg, og), ref, _spread(oracle, win, 6, ref[0], ref[1]))
    # the window stays usable through the per-window entry points afterwards (finished flag cleared, tables at the final state)
    e = C.c_double(0)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 50, C.byref(e)))
    assert abs(e.value - og.lastEnergy) <= 1e-6 * og.lastEnergy
    s2 = np.zeros((nf, 10))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, 50, abi.dp(s2), None, None))
    assert np.array_equal(s2, sg)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 50))


def test_resident_equals_host_loop(gpu_ctx, oracle, monkeypatch):
    """Device loop vs the library's own host loop (SDSO_BA_HOST_LOOP=1, read per call) on the same windows: same decisions, states equal
    up to what libm's sin / cos on host vs device and the fused vs un-fused summation order allow."""
    for spec, mode in ((WINS["small"], 128 | 2048), (WINS["noisy"], 0)):       # default solverMode; and the decaying-lambda schedule
        out = {}
        for host in (0, 1):
            win = dict(synth.ba_window(**spec))
            win["solverMode"] = mode
            nf, npts, nr = win["nf"], win["np"], win["nr"]
            for f in range(nf):
                gpu_ctx.upload_pyramid(510 + f, win["pyrs"][f][:1])
            W, keep = abi.make_ba_window(win, frame_slots=[510 + f for f in range(nf)])
            gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 51, C.byref(W)))
            if host:
                monkeypatch.setenv("SDSO_BA_HOST_LOOP", "1")
            else:
                monkeypatch.delenv("SDSO_BA_HOST_LOOP", raising=False)
            s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
            gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 51, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
            out[host] = (s, i, r, o)
            gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 51))
        monkeypatch.delenv("SDSO_BA_HOST_LOOP", raising=False)
        (s1, i1, r1, o1), (s0, i0, r0, o0) = out[0], out[1]
        assert o1.iterations == o0.iterations and o1.resInA == o0.resInA
        assert np.array_equal(r1, r0)
        assert np.abs(s1 - s0).max() <= 2e-6 and np.abs(i1 - i0).max() <= 2e-5, (np.abs(s1 - s0).max(), np.abs(i1 - i0).max())
        assert abs(o1.lastEnergy - o0.lastEnergy) <= 1e-5 * o0.lastEnergy


def test_batch_resident_loop(gpu_ctx, oracle):
    """sdso_ba_batch_optimize over windows that stop at different iterations: each window must match its own oracle run; then the same
    batch through begin / accumulate / solve / step / end."""
    specs = [dict(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001),
             dict(w=1232, h=368, nf=8, pts_per_kf=250, seed=3008, idepth_noise=0.3, state_noise=1e-2),
             dict(w=1232, h=368, nf=8, pts_per_kf=120, seed=3015)]
    wins = [synth.ba_window(**s) for s in specs]
    nf = 8
    refs = [_oracle_opt(oracle, w, 6) for w in wins]
    assert len({r[3].iterations for r in refs}) >= 2, "the batch should mix early and late finishers"
    spreads = [_spread(oracle, w, 6, r[0], r[1]) for w, r in zip(wins, refs)]
    keepalive = []

    def upload():
        for k, win in enumerate(wins):
            for f in range(nf):
                gpu_ctx.upload_pyramid(520 + k * nf + f, win["pyrs"][f][:1])
            W, keep = abi.make_ba_window(win, frame_slots=[520 + k * nf + f for f in range(nf)])
            keepalive.append((W, keep))
            gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 60 + k, C.byref(W)))
        ids = np.array([60 + k for k in range(len(wins))], np.int32)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, len(wins), abi.ip(ids)))

    def readback(k, res):
        win = wins[k]
        s, i, r = np.zeros((nf, 10)), np.zeros(win["np"], np.float32), np.zeros(win["nr"], np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, 60 + k, abi.dp(s), abi.fp(i), abi.bp(r)))
        return s, i, r, res[k]

    upload()
    res = (abi.BAOptResult * len(wins))()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize(gpu_ctx.h, 6, res))
    for k in range(len(wins)):
        _check(wins[k], readback(k, res), refs[k], spreads[k])
    first = [readback(k, res) for k in range(len(wins))]

    upload()                                                   # fresh states; the same loop in pieces, Jacobians kept in registers
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_set_materialize(gpu_ctx.h, 0))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize_begin(gpu_ctx.h, 1))
    for it in range(6):
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_accumulate(gpu_ctx.h))
        orth = 1 if it >= 2 else 0                             # SOLVER_ORTHOGONALIZE_X_LATER of the default solverMode
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_solve(gpu_ctx.h, 0.1 * 0.25 ** it, orth))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_step(gpu_ctx.h))
    res2 = (abi.BAOptResult * len(wins))()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize_end(gpu_ctx.h, res2))
    for k in range(len(wins)):
        s, i, r, o = readback(k, res2)
        assert o.iterations == first[k][3].iterations and o.resInA == first[k][3].resInA
        assert np.array_equal(s, first[k][0]) and np.array_equal(i, first[k][1]) and np.array_equal(r, first[k][2])
    assert gpu_ctx.L.sdso_ba_batch_step(gpu_ctx.h) != 0        # no loop in flight any more
    for k in range(len(wins)):
        gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 60 + k))


@pytest.mark.parametrize("noise", [dict(), dict(idepth_noise=0.3, state_noise=1e-2)])
@pytest.mark.parametrize("mode", [128 | 2048, 0])
def test_gated_loop_on_the_device(gpu_ctx, oracle, monkeypatch, noise, mode):
    """setting_forceAceptStep = false: the energy gate (FullSystemOptimize.cpp:961-990) is taken on the device (k_ba_opt_gate) — against
    the library's host loop the decisions, iteration counts and states must agree (a marginalisation prior makes calcMEnergyF non-trivial;
    solverMode 0 exercises the loop's own lambda, which depends on the decisions), and against the oracle as
    tests/test_ba_gpu.py::test_optimize_energy_gated_steps demands."""
    win = dict(synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001, **noise))
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    A = np.random.RandomState(4).normal(size=(n, 5))
    win["HM"] = (A @ A.T) * 1e3
    win["bM"] = np.random.RandomState(5).normal(size=n) * 5
    win["forceAcceptStep"] = 0
    win["solverMode"] = mode
    so, io, ro, oo = _oracle_opt(oracle, win, 6)
    for f in range(nf):
        gpu_ctx.upload_pyramid(560 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[560 + f for f in range(nf)])
    out = {}
    for host in (0, 1):
        if host:
            monkeypatch.setenv("SDSO_BA_HOST_LOOP", "1")
        else:
            monkeypatch.delenv("SDSO_BA_HOST_LOOP", raising=False)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 70, C.byref(W)))
        s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 70, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
        out[host] = (s, i, r, o)
    monkeypatch.delenv("SDSO_BA_HOST_LOOP", raising=False)
    (s1, i1, r1, o1), (s0, i0, r0, o0) = out[0], out[1]
    assert o1.iterations == o0.iterations == oo.iterations and o1.resInA == o0.resInA
    assert np.array_equal(r1, r0)
    assert np.abs(s1 - s0).max() <= 2e-6 and np.abs(i1 - i0).max() <= 2e-5, (np.abs(s1 - s0).max(), np.abs(i1 - i0).max())
    assert abs(o1.lastEnergy - o0.lastEnergy) <= 1e-5 * o0.lastEnergy
    assert np.abs(s1 - so).max() <= 2e-4 and helpers.idepths_close(i1, io, 2e-4)     # same accept / reject sequence as the oracle
    # the same window twice in a batch: sdso_ba_batch_optimize takes the gated flow for every member
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 70, C.byref(W)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 71, C.byref(W)))
    ids = np.array([70, 71], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 2, abi.ip(ids)))
    res = (abi.BAOptResult * 2)()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize(gpu_ctx.h, 6, res))
    for k in range(2):
        s = np.zeros((nf, 10))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, 70 + k, abi.dp(s), None, None))
        assert res[k].iterations == o1.iterations and np.array_equal(s, s1)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize_begin(gpu_ctx.h, 1))
    assert gpu_ctx.L.sdso_ba_batch_step(gpu_ctx.h) != 0                              # the piecewise driver is the accepted-step flow only
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 70))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 71))


def test_resident_loop_degenerate_windows(gpu_ctx, oracle, monkeypatch):
    """Window shapes at the edge — two keyframes (mnumOptIts becomes 15), a keyframe that hosts no point, points without residuals, no
    points at all (the quantile falls back to 12*12*8, the break test divides 0 by 0 like the reference) — through the device loop:
    same iteration counts as the oracle and as the library's host loop, nothing faults."""
    def strip(win, keep_pts=None, drop_res=None, empty=False):
        win = dict(win)
        if keep_pts is not None:
            idx = np.nonzero(keep_pts)[0]
            remap = -np.ones(win["np"], np.int64); remap[idx] = np.arange(len(idx))
            rk = keep_pts[win["res_point"]]
            for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
                win[k] = win[k][idx]
            win["res_point"] = remap[win["res_point"][rk]].astype(np.int32); win["res_target"] = win["res_target"][rk]; win["res_state"] = win["res_state"][rk]
            win["np"], win["nr"] = len(idx), int(rk.sum())
        if drop_res is not None:
            for k in ("res_point", "res_target", "res_state"):
                win[k] = win[k][~drop_res]
            win["nr"] = int((~drop_res).sum())
        if empty:
            for k in ("u", "v", "idepth", "idepth_zero", "host", "hasDepthPrior", "color", "weights", "res_point", "res_target", "res_state"):
                win[k] = win[k][:0]
            win["np"] = win["nr"] = 0
        return win
    w2 = synth.ba_window(w=320, h=240, nf=2, pts_per_kf=40, seed=3071)
    w3 = synth.ba_window(w=320, h=240, nf=4, pts_per_kf=30, seed=3072)
    w4 = synth.ba_window(w=320, h=240, nf=4, pts_per_kf=30, seed=3073)
    cases = [("nf2", w2), ("empty_host", strip(w3, keep_pts=w3["host"] != 1)),
             ("points_without_residuals", strip(w4, drop_res=np.isin(w4["res_point"], np.arange(0, w4["np"], 3)))),
             ("no_points", strip(w2, empty=True))]
    for k, (name, win) in enumerate(cases):
        nf, npts, nr = win["nf"], win["np"], win["nr"]
        so, io, ro, oo = _oracle_opt(oracle, win, 6)
        for f in range(nf):
            gpu_ctx.upload_pyramid(580 + f, win["pyrs"][f][:1])
        W, keep = abi.make_ba_window(win, frame_slots=[580 + f for f in range(nf)])
        res = {}
        for host in (0, 1):
            if host:
                monkeypatch.setenv("SDSO_BA_HOST_LOOP", "1")
            else:
                monkeypatch.delenv("SDSO_BA_HOST_LOOP", raising=False)
            gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 75, C.byref(W)))
            s, i, r, o = np.zeros((nf, 10)), np.zeros(max(npts, 1), np.float32), np.zeros(max(nr, 1), np.uint8), abi.BAOptResult()
            gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 75, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
            res[host] = (s, i[:npts], r[:nr], o.iterations, o.resInA)
        monkeypatch.delenv("SDSO_BA_HOST_LOOP", raising=False)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 75))
        assert res[0][3] == res[1][3] == oo.iterations, name
        assert res[0][4] == res[1][4] and helpers.counts_close(res[0][4], oo.resInA, win["nr"]), name
        assert np.isfinite(res[0][0]).all() and np.abs(res[0][0] - res[1][0]).max() <= 2e-6, name
        assert np.abs(res[0][0] - so).max() <= 2e-4, name
        if npts:
            assert np.abs(res[0][1] - res[1][1]).max() <= 2e-5 and helpers.idepths_close(res[0][1], io, 2e-4), name
        if nr:
            assert (res[0][2] != ro).sum() <= 2, name


def test_batch_of_one_window_per_cu_takes_the_auto_selected_forms(gpu_ctx, oracle):
    """From one window per CU on (256 on MI355X) the library switches forms BY ITSELF: the Schur kernel to a wave per host frame (launch_sc_and_folds:
    2 nf nwin > 10 CUs) and the tail kernel takes the points' back-substitution and step too (TAIL_RESUB: nwin >= CUs).  The other parity
    tests reach those forms through the SDSO_BA_SC_WPH / SDSO_BA_TAIL_RESUB overrides only (round-5 advisor finding); this one reaches them
    on the DEFAULT path: 264 copies of one small 8-keyframe window as ONE batch, no override, every copy against the oracle's loop on that
    window and bit-identical to the single-window call (which takes the workgroup-per-host Schur form and the separate points kernel)."""
    import os
    if any(k in os.environ for k in ("SDSO_BA_SC_WPH", "SDSO_BA_TAIL_RESUB", "SDSO_BA_TAIL")):
        pytest.skip("this test is about the DEFAULT path (tests/test_variants_gpu.py runs this file under overrides)")
    win = synth.ba_window(w=640, h=480, nf=8, pts_per_kf=40, seed=3301)
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    so, io, ro, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_optimize(h, 6, abi.dp(so), abi.fp(io), abi.bp(ro), C.byref(oo))
    oracle.orc_ba_destroy(h)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 299, C.byref(W)))
    ss, is_, rs, os_ = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 299, 6, abi.dp(ss), abi.fp(is_), abi.bp(rs), C.byref(os_)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 299))
    nwin = 264
    ids = np.arange(300, 300 + nwin, dtype=np.int32)
    for wid in ids:
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, int(wid), C.byref(W)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, nwin, abi.ip(ids)))
    res = (abi.BAOptResult * nwin)()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize(gpu_ctx.h, 6, res))
    assert os_.iterations == oo.iterations
    scale = np.array([0.5, 0.5, 0.5, 1.0, 1.0, 1.0, 10.0, 1000.0])
    for k in (0, 1, 127, 255, 256, nwin - 1):
        s, i, r = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, int(ids[k]), abi.dp(s), abi.fp(i), abi.bp(r)))
        assert res[k].iterations == oo.iterations
        d = np.abs(s - so)[:, :8] * scale
        # (320 points: the scale direction is weakly held and carries the CPU float path's own summation noise — the spread-relative bar
        #  of tests/test_ba_gpu.py::test_optimize_full_gn_loop's small window, as a fixed number)
        assert d[:, :6].max() <= 2e-4 and d[:, 6].max() <= 2e-5 and d[:, 7].max() <= 2e-3, (k, d.max(axis=0))
        assert (r != ro).sum() <= 2
        # the batch's forms against the single call's: the per-point and per-residual arithmetic is the same statement for statement, the
        # cross-point sums run in f64 in either form -> the same window to rounding of the packed floats
        assert np.abs(s - ss).max() <= 1e-6 and np.abs(i - is_).max() <= 1e-5 and (r != rs).sum() <= 2, (k, np.abs(s - ss).max())
    for wid in ids:
        gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, int(wid)))


def test_cu_partitioned_ctx_gives_the_same_windows(oracle):
    """sdso_ctx_partition_cus: the Schur accumulation and the fused tail kernel of a batch's GN iterations on an aux stream that owns 64 CUs,
    the linearisation on the other 192 (hipExtStreamCreateWithCUMask; measured as a schedule in profiles/r06_cumask_ab.txt and NOT adopted
    by bench.py: it loses).  A schedule must not change a bit: the same batch through the same split calls on a plain ctx and on a
    partitioned one ends on identical states, idepths and residual states."""
    win = synth.ba_window(w=640, h=480, nf=6, pts_per_kf=80, seed=3401)
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    out = []
    for part in (0, 64):
        ctx = abi.Context(0)
        try:
            if part:
                ctx.check(ctx.L.sdso_ctx_partition_cus(ctx.h, part, 32))
            for f in range(nf):
                ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
            W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)])
            ids = np.arange(500, 504, dtype=np.int32)
            for wid in ids:
                ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, int(wid), C.byref(W)))
            ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, len(ids), abi.ip(ids)))
            ctx.check(ctx.L.sdso_ba_batch_optimize_begin(ctx.h, 1))
            for it in range(5):
                ctx.check(ctx.L.sdso_ba_batch_linearize(ctx.h))
                ctx.check(ctx.L.sdso_ba_batch_schur(ctx.h))
                ctx.check(ctx.L.sdso_ba_batch_solve_step(ctx.h, 0.1 * 0.25 ** it, 1 if it >= 2 else 0))
            res = (abi.BAOptResult * len(ids))()
            ctx.check(ctx.L.sdso_ba_batch_optimize_end(ctx.h, res))
            s, i, r = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8)
            ctx.check(ctx.L.sdso_ba_get_state(ctx.h, 503, abi.dp(s), abi.fp(i), abi.bp(r)))
            out.append((s, i, r, res[3].iterations, res[3].lastEnergy))
        finally:
            ctx.close()
    a, b = out
    assert a[3] == b[3] and a[3] >= 2 and a[4] == b[4]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.abs(a[0]).max() > 0
