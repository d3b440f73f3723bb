"""The device BA against the f64-ACCUMULATOR oracle (orc_set_acc64: every AccumulatorApprox / Schur sum of the restatement carried in
double — the float summation order of the CPU path and of the GPU path both disappear from the reference value).

The float oracle the other BA tests compare with is itself one sample of float-accumulation noise; against it the bars are
3e-5 (accumulators) / 1e-4 (stitched H) / 2e-4 (x).  Against the truth they can be stated as what they are:
  * packed accumulators: device max relative error <= 3e-6 per bin block, and its rms no worse than 2x the float CPU path's
    (measured on MI355X: device 1.3e-6 max / 1e-8..1e-7 rms, CPU float 1.8e-6 max / 1.2e-8..1.7e-7 rms — the MFMA block
    reductions are tree sums, the CPU adds sequentially);
  * stitched H (whitened): <= 1e-6 (device 1.5e-7, CPU float 2.8e-7);
  * x of one GN iteration (whitened by the truth system, relative to its largest step): <= 6e-5.  Device 0.6..2.5e-5, CPU float
    0.3..3.4e-5: neither path is consistently closer — the system is near-singular along the scale gauge and amplifies either noise;
  * states / idepths after the whole GN loop: <= 1e-4 / 5e-5 from the truth loop with the same iteration count (device <= 4.9e-5 /
    2.4e-5, CPU float <= 4.8e-5 / 1.1e-5; the idepth maximum sits on single weakly observed points)."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi, synth

pytestmark = pytest.mark.gpu


def _cases():
    c = {"small": synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001),
         "c3": synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001),
         "c3b": synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3008),
         "noisy": synth.ba_window(w=640, h=480, nf=6, pts_per_kf=150, seed=3017, idepth_noise=0.3, state_noise=1e-2)}
    a = dict(synth.ba_window(w=640, h=480, nf=4, pts_per_kf=80, seed=3081))
    a["affineOptModeA"] = a["affineOptModeB"] = -1.0
    c["aff_fixed"] = a
    return c


CASES = _cases()
SECTIONS = lambda nf: (("topA", nf * nf, 91), ("topL", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8), ("Hcc", 1, 16), ("bc", 1, 4))  # noqa: E731


def _oracle_iteration(oracle, W, win, acc64):
    """linearize + applyRes + accumulate + solve on the oracle; acc64: the truth mode (accumulators returned as doubles)"""
    nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
    oracle.orc_set_acc64(1 if acc64 else 0)
    try:
        h = oracle.orc_ba_create(C.byref(W))
        oracle.orc_ba_linearize(h, None)
        oracle.orc_ba_apply_res(h)
        oracle.orc_ba_accumulate(h)
        na = abi.accum_floats(nf)
        if acc64:
            acc = np.zeros(na)
            oracle.orc_ba_get_accumulators_f64(h, abi.dp(acc))
        else:
            a32 = np.zeros(na, np.float32)
            oracle.orc_ba_get_accumulators(h, abi.fp(a32))
            acc = a32.astype(np.float64)
        x, H, st = np.zeros(n), np.zeros((n, n)), np.zeros(npts, np.float32)
        oracle.orc_ba_solve(h, 0, 1e-5, abi.dp(x), abi.dp(H), None, None, None)
        oracle.orc_ba_get_point_steps(h, abi.fp(st))
        oracle.orc_ba_destroy(h)
    finally:
        oracle.orc_set_acc64(0)
    return acc, x, H, st


@pytest.mark.parametrize("which", list(CASES))
@pytest.mark.parametrize("path", ["single", "fused_batch"])
def test_one_iteration_against_f64_truth(gpu_ctx, oracle, which, path):
    win = CASES[which]
    nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    acc64, x64, H64, st64 = _oracle_iteration(oracle, W, win, True)
    acc32, x32, H32, st32 = _oracle_iteration(oracle, W, win, False)
    L = gpu_ctx.L
    gpu_ctx.check(L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    ag = np.zeros(abi.accum_floats(nf), np.float32)
    xg, Hg, sg = np.zeros(n), np.zeros((n, n)), np.zeros(npts, np.float32)
    if path == "single":
        gpu_ctx.check(L.sdso_ba_linearize(gpu_ctx.h, 3, None))
        gpu_ctx.check(L.sdso_ba_apply_res(gpu_ctx.h, 3))
        gpu_ctx.check(L.sdso_ba_accumulate(gpu_ctx.h, 3))
        gpu_ctx.check(L.sdso_ba_get_accumulators(gpu_ctx.h, 3, abi.fp(ag)))
        gpu_ctx.check(L.sdso_ba_solve(gpu_ctx.h, 3, 0, 1e-5, abi.dp(xg), abi.dp(Hg), None, None, None))
    else:
        ids = np.array([3], np.int32)
        gpu_ctx.check(L.sdso_ba_batch_create(gpu_ctx.h, 1, abi.ip(ids)))
        gpu_ctx.check(L.sdso_ba_batch_accumulate(gpu_ctx.h))
        gpu_ctx.check(L.sdso_ba_get_accumulators(gpu_ctx.h, 3, abi.fp(ag)))
        gpu_ctx.check(L.sdso_ba_solve(gpu_ctx.h, 3, 0, 1e-5, abi.dp(xg), abi.dp(Hg), None, None, None))     # stitch + solve of the batch's accumulators
    gpu_ctx.check(L.sdso_ba_get_point_steps(gpu_ctx.h, 3, abi.fp(sg)))
    gpu_ctx.check(L.sdso_ba_release_window(gpu_ctx.h, 3))

    o0 = 0
    for name, cnt, w in SECTIONS(nf):
        T = acc64[o0:o0 + cnt * w].reshape(-1, w)
        m = np.maximum(np.abs(T).max(axis=1, keepdims=True), 1e-30)
        eg = (ag[o0:o0 + cnt * w].reshape(-1, w).astype(np.float64) - T) / m
        ec = (acc32[o0:o0 + cnt * w].reshape(-1, w) - T) / m
        live = np.abs(T).max(axis=1) > 0
        if live.any():
            assert np.abs(eg[live]).max() <= 3e-6, (name, np.abs(eg[live]).max())
            rg, rc = np.sqrt((eg[live] ** 2).mean()), np.sqrt((ec[live] ** 2).mean())
            assert rg <= 2.0 * rc + 2e-8, (name, rg, rc)                   # the device sums are no noisier than the CPU float sums
        assert not np.abs(ag[o0:o0 + cnt * w].reshape(-1, w)[~live]).any()  # bins empty in the truth are empty on the device
        o0 += cnt * w
    assert np.array_equal(ag[o0:o0 + 2].astype(np.float64), acc64[o0:o0 + 2])      # residual counts
    d = np.sqrt(np.abs(np.diag(H64))) + 1e-30
    sc = max(1.0, np.abs(x64 * d).max())
    assert np.abs((Hg - H64) / np.outer(d, d)).max() <= 1e-6
    eg, ec = np.abs((xg - x64) * d).max() / sc, np.abs((x32 - x64) * d).max() / sc
    assert eg <= 6e-5, (eg, ec)
    assert ec <= 6e-5, (eg, ec)                                             # the same bar holds for the CPU float path: it is the noise floor, not slack
    assert np.abs(sg - st64).max() <= 2e-4 * max(np.abs(st64).max(), 1e-6)


@pytest.mark.parametrize("which", ["small", "c3", "aff_fixed"])
def test_gn_loop_against_f64_truth(gpu_ctx, oracle, which):
    win = CASES[which]
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    oracle.orc_set_acc64(1)
    try:
        h = oracle.orc_ba_create(C.byref(W))
        s64, i64, r64, o64 = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h, 6, abi.dp(s64), abi.fp(i64), abi.bp(r64), C.byref(o64))
        oracle.orc_ba_destroy(h)
    finally:
        oracle.orc_set_acc64(0)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 3))
    assert o.iterations == o64.iterations
    assert np.abs(s - s64).max() <= 1e-4, np.abs(s - s64).max()
    assert np.abs(i.astype(np.float64) - i64).max() <= 5e-5, np.abs(i.astype(np.float64) - i64).max()
    assert (r != r64).sum() <= max(2, nr // 2000)
    assert abs(o.lastEnergy - o64.lastEnergy) <= 1e-4 * o64.lastEnergy
