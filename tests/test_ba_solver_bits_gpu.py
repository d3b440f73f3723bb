"""The four bits of setting_solverMode that live OUTSIDE solveSystemF (round 6; rounds 1-5 refused them at upload), at BASELINE
configs[2] (8 keyframes x 2000 points) against the oracle, through the C-ABI:

  SOLVER_STEPMOMENTUM             FullSystem::optimize, src/FullSystem/FullSystemOptimize.cpp:933-948 (incDirChange, stepsize in [0.25, 2])
  SOLVER_MOMENTUM                 backupState(backupLastStep) :311-345, doStepFromBackup :225-251 (step + 0.5f * step_backup)
  SOLVER_ORTHOGONALIZE_POINTMARG  EnergyFunctional::marginalizePointsF, src/OptimizationBackend/EnergyFunctional.cpp:711-723 (only without frame 0)
  SOLVER_ORTHOGONALIZE_FULL       :730-731

Same bars as the default branch (tests/test_ba_gpu.py): same iteration counts, pose states <= 2e-5 (fixed), idepths, residual states up to
threshold flips; HM / bM <= 1e-4 whitened.  The device-resident loop and the host-driven loop (SDSO_BA_HOST_LOOP=1) give the same window."""
import ctypes as C

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth

pytestmark = pytest.mark.gpu

FIX_LAMBDA, ORTH_X_LATER, MOMENTUM, STEPMOMENTUM, ORTH_POINTMARG, ORTH_FULL = 128, 2048, 512, 1024, 4, 8
DEFAULT = FIX_LAMBDA | ORTH_X_LATER                                                     # settings.cpp:51
_STATE_SCALE = np.array([0.5, 0.5, 0.5, 1.0, 1.0, 1.0, 10.0, 1000.0])


@pytest.fixture(scope="module")
def win_c3():
    return synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)      # configs[2]


@pytest.fixture(scope="module")
def win_small():
    return synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001)


def _upload(ctx, win, wid=3, slot0=40):
    for f in range(win["nf"]):
        ctx.upload_pyramid(slot0 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[slot0 + f for f in range(win["nf"])], dI_list=[p[0] for p in win["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
    return W, keep


def _oracle_loop(oracle, W, win, its, acc64=False):
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    oracle.orc_set_acc64(1 if acc64 else 0)
    try:
        return _oracle_loop_body(oracle, W, win, its)
    finally:
        oracle.orc_set_acc64(0)


def _oracle_loop_body(oracle, W, win, its):
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    h = oracle.orc_ba_create(C.byref(W))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    oracle.orc_ba_optimize(h, its, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o))
    x = np.zeros((its + 16, n))
    nx = oracle.orc_ba_get_x_trace(h, abi.dp(x), its + 16)
    st = np.zeros(its + 16, np.float32)
    ns = oracle.orc_ba_get_step_trace(h, st.ctypes.data_as(C.POINTER(C.c_float)), its + 16)
    ne, nw = np.zeros(nr, np.float32), np.zeros(nr, np.float32)
    oracle.orc_ba_get_linearization(h, None, None, abi.fp(ne), abi.fp(nw), None, None)     # of the closing linearizeAll(true)
    oracle.orc_ba_destroy(h)
    assert nx == ns == o.iterations
    return s, i, r, o, x[:nx], st[:ns], ne, nw


def _device_loop(ctx, W, win, its, wid=3):
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, wid, its, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
    ne, nw = np.zeros(nr, np.float32), np.zeros(nr, np.float32)
    ctx.check(ctx.L.sdso_ba_get_linearization(ctx.h, wid, None, None, abi.fp(ne), abi.fp(nw), None, None))
    return s, i, r, o, ne, nw


def _compare_loops(win, dev, orc, pose_bar=2e-5, truth=None):
    """truth: the same loop with the oracle's accumulator sums in f64.  The device carries its sums in f64 too (ba_kernels.hip ACC_MODE 1), so
    against the truth the fixed bar holds by itself; against the float oracle the distance may be the float oracle's own distance from the
    truth (the momentum modes feed half of every step into the next one, and the float path's summation noise with it)."""
    sg, ig, rg, og = dev[:4]
    so, io, ro, oo = orc[:4]
    assert og.iterations == oo.iterations, (og.iterations, oo.iterations)
    d = np.abs(sg - so)[:, :8] * _STATE_SCALE
    if truth is not None:
        dt = (np.abs(sg - truth[0])[:, :8] * _STATE_SCALE)[:, :6].max()
        dc = (np.abs(so - truth[0])[:, :8] * _STATE_SCALE)[:, :6].max()
        flipped = np.nonzero(rg != truth[2])[0]
        print("pose states after the loop: device-truth %.2e, cpu_f32-truth %.2e, device-cpu_f32 %.2e; residuals whose final state differs from the truth's: %s"
              % (dt, dc, d[:, :6].max(), flipped.tolist()))
        assert og.iterations == truth[3].iterations
        assert len(flipped) <= max(2, win["nr"] // 2000)
        for j in flipped:
            # a residual on its outlier threshold (Residuals.cpp:303-311) kept on one side, clamped on the other: the kept energy is the
            # clamp value to 1e-3.  One such residual moves the next solve by more than all summation noise together (measured with the
            # momentum bits at configs[2]: the update after the flip sits 4e-5 off, every one before it < 5e-6: tests/diag/momentum_trace.py)
            if {int(rg[j]), int(truth[2][j])} == {0, 2}:
                e_in, th = (dev[5][j], truth[6][j]) if rg[j] == 0 else (truth[7][j], dev[4][j])
                assert abs(float(e_in) - float(th)) <= 1e-3 * float(th), (j, e_in, th)
        if len(flipped):
            pose_bar = 2.5 * pose_bar
        assert dt <= pose_bar, (dt, dc)
        pose_bar = max(pose_bar, dc + dt)
    assert d[:, :6].max() <= pose_bar, d[:, :6].max()
    assert d[:, 6].max() <= 2e-5 and d[:, 7].max() <= 2e-3, d[:, 6:].max(axis=0)
    di = np.abs(ig.astype(np.float64) - io)
    assert np.percentile(di, 99) <= 5e-5 and di.max() <= 2e-3, (np.percentile(di, 99), di.max())
    assert (rg != ro).sum() <= max(2, win["nr"] // 2000)
    assert abs(og.lastEnergy - oo.lastEnergy) <= 1e-4 * oo.lastEnergy
    assert helpers.counts_close(og.resInA, oo.resInA, win["nr"])


@pytest.mark.parametrize("bits", [STEPMOMENTUM, MOMENTUM, MOMENTUM | STEPMOMENTUM])
def test_momentum_bits_full_gn_loop_c3(gpu_ctx, oracle, win_c3, bits, monkeypatch):
    """FullSystem::optimize with SOLVER_STEPMOMENTUM / SOLVER_MOMENTUM at configs[2].  The bit really acts (the oracle's stepsize leaves 1, its
    final state differs from the default mode's); the device-resident loop follows the oracle within the default branch's bars; the
    host-driven loop of the same library gives the resident loop's window."""
    win = dict(win_c3)
    win["solverMode"] = DEFAULT | bits
    W, keep = _upload(gpu_ctx, win)
    orc = _oracle_loop(oracle, W, win, 6)
    plain = dict(win_c3); plain["solverMode"] = DEFAULT
    Wp, keep_p = abi.make_ba_window(plain, frame_slots=[40 + f for f in range(plain["nf"])], dI_list=[p[0] for p in plain["pyrs"]])
    ref = _oracle_loop(oracle, Wp, plain, 6)
    if bits & STEPMOMENTUM:
        assert np.abs(orc[5] - 1.0).max() > 0.05, orc[5]               # the step size moved
        assert orc[5].min() >= 0.25 and orc[5].max() <= 2.0
    acted = np.abs((orc[0] - ref[0])[:, :6]).max()
    assert acted > 1e-5 or orc[3].iterations != ref[3].iterations, acted                   # a different trajectory than the default mode's
    dev = _device_loop(gpu_ctx, W, win, 6)
    _compare_loops(win, dev, orc, truth=_oracle_loop(oracle, W, win, 6, acc64=True))
    monkeypatch.setenv("SDSO_BA_HOST_LOOP", "1")
    host = _device_loop(gpu_ctx, W, win, 6)
    monkeypatch.delenv("SDSO_BA_HOST_LOOP")
    assert host[3].iterations == dev[3].iterations
    # (measured: the two loops end on the same bits — the host loop's frames / stepsize arithmetic is the kernels' statement for statement)
    assert np.abs(host[0] - dev[0]).max() <= 1e-12 and np.abs(host[1] - dev[1]).max() <= 1e-9, (np.abs(host[0] - dev[0]).max(), np.abs(host[1] - dev[1]).max())
    assert np.array_equal(host[2], dev[2])
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 3))


@pytest.mark.parametrize("bits", [STEPMOMENTUM, MOMENTUM])
def test_momentum_bits_energy_gated_and_batched(gpu_ctx, oracle, win_small, bits):
    """The same bits in the energy-gated flow (setting_forceAceptStep = false: rejected steps restore the backup, the kept step and previousX
    stay those of the rejected solve, FullSystemOptimize.cpp:929-990) and through the batch entry points (two windows, one loop)."""
    win = dict(synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001, idepth_noise=0.3, state_noise=1e-2))
    win["solverMode"] = ORTH_X_LATER | bits                             # (no FIX_LAMBDA: the gate's lambda schedule is live)
    win["forceAcceptStep"] = 0
    W, keep = _upload(gpu_ctx, win)
    orc = _oracle_loop(oracle, W, win, 6)
    dev = _device_loop(gpu_ctx, W, win, 6)
    assert dev[3].iterations == orc[3].iterations
    assert np.abs(dev[0] - orc[0]).max() <= 2e-4 and helpers.idepths_close(dev[1], orc[1], 2e-4)
    assert abs(dev[3].lastEnergy - orc[3].lastEnergy) <= 1e-3 * orc[3].lastEnergy
    # batch: the accepted-step flow of two windows in one resident loop == the single calls
    wa = dict(win_small); wa["solverMode"] = DEFAULT | bits
    wb = dict(synth.ba_window(w=640, h=480, nf=5, pts_per_kf=100, seed=3011)); wb["solverMode"] = DEFAULT | bits
    singles = []
    Ws = []
    for k, w in enumerate((wa, wb)):
        Wk, kk = _upload(gpu_ctx, w, wid=20 + k, slot0=60 + 8 * k)
        Ws.append((Wk, kk))
        singles.append(_device_loop(gpu_ctx, Wk, w, 6, wid=20 + k))
        _compare_loops(w, singles[-1], _oracle_loop(oracle, Wk, w, 6))
        gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 20 + k, C.byref(Wk)))
    ids = np.array([20, 21], np.int32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_create(gpu_ctx.h, 2, abi.ip(ids)))
    res = (abi.BAOptResult * 2)()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_batch_optimize(gpu_ctx.h, 6, res))
    for k, w in enumerate((wa, wb)):
        s, i, r = np.zeros((w["nf"], 10)), np.zeros(w["np"], np.float32), np.zeros(w["nr"], np.uint8)
        gpu_ctx.check(gpu_ctx.L.sdso_ba_get_state(gpu_ctx.h, 20 + k, abi.dp(s), abi.fp(i), abi.bp(r)))
        assert res[k].iterations == singles[k][3].iterations
        assert np.abs(s - singles[k][0]).max() <= 1e-9 and np.abs(i - singles[k][1]).max() <= 1e-7 and np.array_equal(r, singles[k][2])
    for k in range(2):
        gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 20 + k))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 3))


@pytest.mark.parametrize("bits,first_id", [(ORTH_POINTMARG, 3), (ORTH_POINTMARG, 0), (ORTH_FULL, 3), (ORTH_FULL, 0), (ORTH_POINTMARG | ORTH_FULL, 3)])
def test_marginalize_points_orthogonalised_c3(gpu_ctx, oracle, win_c3, bits, first_id):
    """EnergyFunctional::marginalizePointsF with SOLVER_ORTHOGONALIZE_POINTMARG (H, b of the marginalised points projected off the gauge —
    only when frame 0 has left the window) and SOLVER_ORTHOGONALIZE_FULL (the whole prior projected) at configs[2], on a window that
    already carries a prior as large as the points' own H (the projector is the one of the window's current evaluation points: FullSystem.cpp:1453
    refreshes the nullspaces right before marginalizePointsF)."""
    win = dict(win_c3)
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    win["solverMode"] = DEFAULT | bits
    win["frameID"] = (np.arange(nf) + first_id).astype(np.int32)
    A = np.random.RandomState(5).normal(size=(n, 6))
    win["HM"] = (A @ A.T) * 1e9                                          # as large as the marginalised points' H, and not gauge-free
    win["bM"] = np.random.RandomState(6).normal(size=n) * 1e6
    W, keep = _upload(gpu_ctx, win)
    h = oracle.orc_ba_create(C.byref(W))
    flag = (win["host"] <= 1).astype(np.uint8)                           # the points of the two oldest keyframes

    def prepare(hh, device):
        # one regular iteration's linearisation first, as FullSystem does; both sides at the uploaded state (after a whole optimize the two
        # float paths sit 1e-5 apart — along the gauge much more when nothing fixes it — and the prior would carry that, not the projection)
        oracle.orc_ba_linearize(hh, None); oracle.orc_ba_apply_res(hh); oracle.orc_ba_accumulate(hh)
        if device:
            gpu_ctx.check(gpu_ctx.L.sdso_ba_linearize(gpu_ctx.h, 3, None))
            gpu_ctx.check(gpu_ctx.L.sdso_ba_apply_res(gpu_ctx.h, 3))
            gpu_ctx.check(gpu_ctx.L.sdso_ba_accumulate(gpu_ctx.h, 3))
    prepare(h, True)
    HMo, bMo, HMg, bMg = np.zeros((n, n)), np.zeros(n), np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(h, abi.bp(flag), abi.dp(HMo), abi.dp(bMo))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_marginalize_points(gpu_ctx.h, 3, abi.bp(flag), abi.dp(HMg), abi.dp(bMg)))
    # the plain statement on the same window, for "the bit acts"
    plain = dict(win); plain["solverMode"] = DEFAULT
    Wp, kp = abi.make_ba_window(plain, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in plain["pyrs"]])
    hp = oracle.orc_ba_create(C.byref(Wp))
    prepare(hp, False)
    HMp, bMp = np.zeros((n, n)), np.zeros(n)
    oracle.orc_ba_marginalize_points(hp, abi.bp(flag), abi.dp(HMp), abi.dp(bMp))
    oracle.orc_ba_destroy(hp)
    oracle.orc_ba_destroy(h)
    d = np.sqrt(np.abs(np.diag(HMp))) + 1e-30
    dif = np.abs((HMo - HMp) / np.outer(d, d)).max()
    if bits & ORTH_FULL:
        assert dif > 1e-3, dif                                          # the prior lost its gauge part
    elif first_id == 0:
        assert np.array_equal(HMo, HMp)                                 # POINTMARG with frame 0 in the window is the plain statement
    else:
        assert not np.array_equal(HMo, HMp)                             # (the points' H is gauge-free up to float: the projection moves it by rounding)
    assert np.abs((HMg - HMo) / np.outer(d, d)).max() <= 1e-4, np.abs((HMg - HMo) / np.outer(d, d)).max()
    assert np.abs((bMg - bMo) / d).max() <= 1e-4 * max(1.0, np.abs(bMo / d).max())
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 3))
