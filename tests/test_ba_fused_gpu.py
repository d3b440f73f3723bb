"""GPU parity of the BENCHMARKED BA path against the oracle: `sdso_ba_batch_create` over several 8-keyframe / 2000-point windows of
BASELINE configs[2] shape (1232x368, distinct pyramids, 4x2-tiled images, cooperative tap gather) -> `sdso_ba_batch_accumulate`
(k_ba_lin_fused = linearize + applyRes + accumulateAF in one kernel, k_ba_sc_host, folds) -> `sdso_ba_batch_solve`.

Everything the fused kernel writes is read back per window and compared with the oracle's linearizeAll + applyRes + accumulate +
solveSystemF on the same window: the RawResidualJacobian records it materialises (EFResidual::J of applied residuals,
PointFrameResidual::J of the others), residual states / energies, JpJdF and the per-point terms bit-exact; the packed accumulators
<= 3e-5 of their block maximum; x <= 2e-4 in the whitened metric.  Variant: Jacobians kept in registers (materialize = 0)."""
import ctypes as C
import os

import numpy as np
import pytest

from sdso_amd import abi
import synth
from test_ba_gpu import _check_accum

SDSO_ERR_STATE = -4   # include/sdso_abi.h:36

pytestmark = pytest.mark.gpu

NWIN = 3


@pytest.fixture(scope="module")
def bench_windows():
    # window 0 is the bench window (seed 3001); the others are different trajectories / point sets, so chunk lists, residual
    # counts and pyramids all differ inside one launch
    wins = [synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001 + 7 * k) for k in range(NWIN)]
    # window 1 has been through EnergyFunctional::dropResidual (EnergyFunctional.cpp:524-533): its residualsAll lists are no longer in
    # target order — the per-point sums below must stay bit-exact on it (round-3 verdict, Missing #2)
    import helpers
    wins[1], _ = helpers.drop_residuals(wins[1], seed=23, drop_frac=0.2)
    return wins


def _oracle_iteration(oracle, win, W):
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    h = oracle.orc_ba_create(C.byref(W))
    o = dict(Jn=np.zeros((nr, 74), np.float32), ns=np.zeros(nr, np.uint8), ne=np.zeros(nr, np.float32), nw=np.zeros(nr, np.float32),
             Je=np.zeros((nr, 74), np.float32), st=np.zeros(nr, np.uint8), act=np.zeros(nr, np.uint8), jp=np.zeros((nr, 8), np.float32),
             acc=np.zeros(abi.accum_floats(nf), np.float32), x=np.zeros(n), H=np.zeros((n, n)), step=np.zeros(npts, np.float32))
    o["pt"] = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    oracle.orc_ba_linearize(h, None)
    oracle.orc_ba_get_linearization(h, abi.fp(o["Jn"]), abi.bp(o["ns"]), abi.fp(o["ne"]), abi.fp(o["nw"]), None, None)
    oracle.orc_ba_apply_res(h)
    oracle.orc_ba_get_ef_jacobians(h, abi.fp(o["Je"]))
    oracle.orc_ba_get_residual_state(h, abi.bp(o["st"]), abi.bp(o["act"]), abi.fp(o["jp"]))
    oracle.orc_ba_solve(h, 0, 1e-5, abi.dp(o["x"]), abi.dp(o["H"]), None, None, None)      # accumulates, stitches, solves, resubstitutes
    oracle.orc_ba_get_accumulators(h, abi.fp(o["acc"]))
    oracle.orc_ba_get_point_terms(h, *[abi.fp(a) for a in o["pt"]])
    oracle.orc_ba_get_point_steps(h, abi.fp(o["step"]))
    oracle.orc_ba_destroy(h)
    return o


def _gpu_readback(ctx, win, wid, materialize):
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    g = dict(Jn=np.zeros((nr, 74), np.float32), ns=np.zeros(nr, np.uint8), ne=np.zeros(nr, np.float32), nw=np.zeros(nr, np.float32),
             Je=np.zeros((nr, 74), np.float32), st=np.zeros(nr, np.uint8), act=np.zeros(nr, np.uint8), jp=np.zeros((nr, 8), np.float32),
             acc=np.zeros(abi.accum_floats(nf), np.float32), step=np.zeros(npts, np.float32))
    g["pt"] = [np.zeros(npts, np.float32) for _ in range(4)] + [np.zeros(npts * 4, np.float32)]
    ctx.check(ctx.L.sdso_ba_get_linearization(ctx.h, wid, abi.fp(g["Jn"]) if materialize else None, abi.bp(g["ns"]), abi.fp(g["ne"]), abi.fp(g["nw"]), None, None))
    if materialize:
        ctx.check(ctx.L.sdso_ba_get_ef_jacobians(ctx.h, wid, abi.fp(g["Je"])))
    ctx.check(ctx.L.sdso_ba_get_residual_state(ctx.h, wid, abi.bp(g["st"]), abi.bp(g["act"]), abi.fp(g["jp"])))
    ctx.check(ctx.L.sdso_ba_get_accumulators(ctx.h, wid, abi.fp(g["acc"])))
    ctx.check(ctx.L.sdso_ba_get_point_terms(ctx.h, wid, *[abi.fp(a) for a in g["pt"]]))
    ctx.check(ctx.L.sdso_ba_get_point_steps(ctx.h, wid, abi.fp(g["step"])))
    return g


@pytest.mark.parametrize("variant", ["materialize", "registers"])
def test_fused_batch_matches_oracle_at_bench_config(gpu_ctx, oracle, bench_windows, variant):
    ctx = gpu_ctx
    materialize = "registers" not in variant
    ids, Ws = [], []
    for k, win in enumerate(bench_windows):
        slots = [700 + 10 * k + f for f in range(win["nf"])]
        for f in range(win["nf"]):
            ctx.upload_pyramid(slots[f], win["pyrs"][f][:1])
        W, keep = abi.make_ba_window(win, frame_slots=slots, dI_list=[p[0] for p in win["pyrs"]])
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 70 + k, C.byref(W)))
        ids.append(70 + k); Ws.append((W, keep))
    ids = np.array(ids, np.int32)
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, len(ids), abi.ip(ids)))
    ctx.check(ctx.L.sdso_ba_batch_set_materialize(ctx.h, int(materialize)))
    ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
    ctx.check(ctx.L.sdso_ba_batch_solve(ctx.h, 1e-5, 0))
    xb = np.zeros((len(ids), 68))
    ctx.check(ctx.L.sdso_ba_batch_get_x(ctx.h, abi.dp(xb)))
    for k, win in enumerate(bench_windows):
        o = _oracle_iteration(oracle, win, Ws[k][0])
        g = _gpu_readback(ctx, win, 70 + k, materialize)
        nr = win["nr"]
        assert (o["ns"] == 0).sum() > 0.4 * nr                                       # the comparison is not vacuous
        # linearize: decisions and energies of every residual
        assert np.array_equal(o["ns"], g["ns"]) and np.array_equal(o["ne"], g["ne"]) and np.array_equal(o["nw"], g["nw"])
        # applyRes: states, isActive, JpJdF
        assert np.array_equal(o["st"], g["st"]) and np.array_equal(o["act"], g["act"])
        act = o["act"] == 1
        assert np.array_equal(o["jp"][act], g["jp"][act])
        if materialize:
            # the 296-byte records the kernel streams out: applied residuals hold theirs as EFResidual::J (takeDataF swapped it in),
            # freshly linearized but not applied ones (OUTLIER) keep it as PointFrameResidual::J; OOB residuals write none
            assert np.array_equal(o["Je"][act], g["Je"][act])
            outl = o["ns"] == 2
            assert outl.sum() > 0
            assert np.array_equal(o["Jn"][outl], g["Jn"][outl])
        # accumulateAF / SCF: per-point terms bit-exact, packed accumulators to float-order tolerance
        for a, b in zip(o["pt"], g["pt"]):
            assert np.array_equal(a, b)
        _check_accum(o["acc"], g["acc"], win["nf"])
        # stitch + solve + resubstitute
        d = np.sqrt(np.abs(np.diag(o["H"]))) + 1e-30
        assert np.abs((xb[k] - o["x"]) * d).max() <= 2e-4 * max(1.0, np.abs(o["x"] * d).max())
        assert np.abs(g["step"] - o["step"]).max() <= 2e-4 * max(np.abs(o["step"]).max(), 1e-6)
    for k in ids:
        ctx.check(ctx.L.sdso_ba_release_window(ctx.h, int(k)))


def test_batch_lifetime(gpu_ctx, oracle):
    """A window that leaves a batch (released, re-uploaded, or dropped by a second batch_create) gets its own accumulator block back:
    per-window calls afterwards must neither fault nor change result (freed batch block / stale descriptor snapshot)."""
    ctx = gpu_ctx
    win = synth.ba_window(w=640, h=480, nf=5, pts_per_kf=60, seed=3021)
    slots = [760 + f for f in range(win["nf"])]
    for f in range(win["nf"]):
        ctx.upload_pyramid(slots[f], win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=slots)
    n = 8 * win["nf"] + 4

    def single(wid):
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
        ctx.check(ctx.L.sdso_ba_linearize(ctx.h, wid, None))
        ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, wid))
        ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, wid))
        x = np.zeros(n)
        ctx.check(ctx.L.sdso_ba_solve(ctx.h, wid, 0, 0.1, abi.dp(x), None, None, None, None))
        return x

    ref = single(80)
    for wid in (80, 81, 82):
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, 3, abi.ip(np.array([80, 81, 82], np.int32))))
    ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
    # second batch leaves window 82 out: it must be usable on its own again
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, 2, abi.ip(np.array([80, 81], np.int32))))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 82, None))
    ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 82))
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 82))
    x = np.zeros(n)
    ctx.check(ctx.L.sdso_ba_solve(ctx.h, 82, 0, 0.1, abi.dp(x), None, None, None, None))
    assert np.array_equal(x, ref)
    # releasing a member dissolves the batch instead of leaving a stale snapshot behind
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 81))
    assert ctx.L.sdso_ba_batch_accumulate(ctx.h) == SDSO_ERR_STATE                # "no batch"
    assert np.array_equal(single(80), ref)
    # a batch naming an unknown window is refused and registers nothing
    assert ctx.L.sdso_ba_batch_create(ctx.h, 2, abi.ip(np.array([80, 999], np.int32))) != 0
    assert ctx.L.sdso_ba_batch_accumulate(ctx.h) != 0
    for wid in (80, 82):
        ctx.check(ctx.L.sdso_ba_release_window(ctx.h, wid))
