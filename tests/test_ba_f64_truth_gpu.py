"""The device BA against the f64-ACCUMULATOR oracle (orc_set_acc64: every AccumulatorApprox / Schur sum of the restatement carried in
double — the float summation order of the CPU path and of the GPU path both disappear from the reference value).

The float oracle the other BA tests compare with is itself one sample of float-accumulation noise; against it the bars are
3e-5 (accumulators) / 1e-4 (stitched H) / 2e-4 (x).  Against the truth they can be stated as what they are:
  * packed accumulators: device max relative error <= 3e-7 per bin block and its rms below the float CPU path's (round 6: the device
    carries these sums in f64 on the matrix cores and rounds ONCE to the packed block's floats — measured on MI355X, tests/diag/
    acc_errors.py -> profiles/r06_acc_errors.txt: device 1.4e-8..2.3e-7 max / 3e-9..6e-8 rms, CPU float 5e-8..1.8e-6 max / 1.2e-8..1.8e-7 rms;
    rounds 1-5 with fp32 MFMA chains: 1.3e-6 max);
  * stitched H (whitened): <= 1e-7 (device 2.4e-8..3.8e-8, CPU float 1.5e-7..2.8e-7);
  * x of one GN iteration (whitened by the truth system, relative to its largest step): device <= 1e-5 (measured 0.3..5.1e-6: what the
    one float rounding of the packed block leaves), CPU float <= 6e-5 (1.2e-6..3.4e-5 — the system is near-singular along the scale
    gauge and amplifies the float sums' noise);
  * states / idepths after the whole GN loop: <= 1e-4 / 5e-5 from the truth loop with the same iteration count (device <= 4.9e-5 /
    2.4e-5, CPU float <= 4.8e-5 / 1.1e-5; the idepth maximum sits on single weakly observed points)."""
import ctypes as C

import numpy as np
import pytest

from sdso_amd import abi
import synth

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
            assert np.abs(eg[live]).max() <= 3e-7, (name, np.abs(eg[live]).max())
            rg, rc = np.sqrt((eg[live] ** 2).mean()), np.sqrt((ec[live] ** 2).mean())
            assert rg <= rc, (name, rg, rc)                                # f64 sums rounded once: closer than the CPU float sums, section by section
        assert not np.abs(ag[o0:o0 + cnt * w].reshape(-1, w)[~live]).any()  # bins empty in the truth are empty on the device
        o0 += cnt * w
    assert np.array_equal(ag[o0:o0 + 2].astype(np.float64), acc64[o0:o0 + 2])      # residual counts
    d = np.sqrt(np.abs(np.diag(H64))) + 1e-30
    sc = max(1.0, np.abs(x64 * d).max())
    assert np.abs((Hg - H64) / np.outer(d, d)).max() <= 1e-7
    eg, ec = np.abs((xg - x64) * d).max() / sc, np.abs((x32 - x64) * d).max() / sc
    assert eg <= 1e-5, (eg, ec)
    assert ec <= 6e-5, (eg, ec)                                             # the CPU float path: its summation order's noise, amplified by the gauge
    assert np.abs(sg - st64).max() <= 2e-4 * max(np.abs(st64).max(), 1e-6)


def _gauge_free(win, ds):
    """The component of a frame-state difference ds (nf x 10) outside the span of the window's seven gauge directions (global rigid
    motion + monocular scale; tests/test_oracle_ba.py::gauge_vectors, the vectors FrameHessian::setStateZero differentiates,
    HessianBlocks.cpp:78-123 / EnergyFunctional.cpp:775-835), least-squares projection in state coordinates."""
    from test_oracle_ba import gauge_vectors
    nf = win["nf"]
    N = gauge_vectors(win)                                   # (4 + 8 nf) x 7, calibration rows zero
    v = np.zeros(4 + 8 * nf)
    for f in range(nf):
        v[4 + 8 * f:12 + 8 * f] = ds[f, :8]
    c, *_ = np.linalg.lstsq(N, v, rcond=None)
    return v - N @ c, N @ c


@pytest.mark.parametrize("which", list(CASES))
def test_gn_loop_against_f64_truth(gpu_ctx, oracle, which):
    """All five windows: the loop's final states against the f64-accumulator truth.

    What the distance IS (round 4, tests/diag/truth_spread.py over 24 windows, profiles/r04_truth_spread.txt): the float accumulators put
    an absolute noise floor dx = H^-1 db under the last steps of the loop (b is a sum with cancellation: its float error does not shrink
    with |b|), and a residual sitting on its outlier threshold flips on one side only.  Per window the distance to the truth therefore
    scatters between 1e-6 and 2e-4 — for the device AND for the CPU float path, uncorrelated (device median 2.3e-5, CPU 2.4e-5; device
    worse on 11 of 24 windows, better on 13).  A per-window bar can only be the top of that scatter (3e-4 states); the statement about the
    kernels is the distribution test below.  Round 3's suspicion that the fused tail kernel adds noise of its own was half right: its
    `S2 = S1^T` shortcut assumed accD(i,j,k) == accD(i,k,j)^T, which the MFMA tiles only satisfied up to rounding; the Schur kernel now
    writes the lower tiles as mirror images (exact, as in the reference) and the two tails agree to 1e-9 on every window.
    Every residual whose final state differs from the truth sits at its threshold: the energy one side keeps is within 1e-3 of the value
    the other side clamps to."""
    win = CASES[which]
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    for f in range(nf):
        gpu_ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    ns64, ne64, nw64 = np.zeros(nr, np.uint8), np.zeros(nr, np.float32), np.zeros(nr, np.float32)
    P64, d64 = abi.make_post_state(nf, npts, nr)
    oracle.orc_set_acc64(1)
    try:
        h = oracle.orc_ba_create(C.byref(W))
        s64, i64, r64, o64 = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h, 6, abi.dp(s64), abi.fp(i64), abi.bp(r64), C.byref(o64))
        oracle.orc_ba_get_linearization(h, None, abi.bp(ns64), abi.fp(ne64), abi.fp(nw64), None, None)     # of the final linearizeAll(true)
        oracle.orc_ba_get_post_state(h, C.byref(P64))
        oracle.orc_ba_destroy(h)
    finally:
        oracle.orc_set_acc64(0)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_upload_window(gpu_ctx.h, 3, C.byref(W)))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    gpu_ctx.check(gpu_ctx.L.sdso_ba_optimize(gpu_ctx.h, 3, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
    ns, ne, nw = np.zeros(nr, np.uint8), np.zeros(nr, np.float32), np.zeros(nr, np.float32)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_linearization(gpu_ctx.h, 3, None, abi.bp(ns), abi.fp(ne), abi.fp(nw), None, None))
    P, d = abi.make_post_state(nf, npts, nr)
    gpu_ctx.check(gpu_ctx.L.sdso_ba_get_post_state(gpu_ctx.h, 3, C.byref(P)))
    gpu_ctx.check(gpu_ctx.L.sdso_ba_release_window(gpu_ctx.h, 3))
    assert o.iterations == o64.iterations
    raw = np.abs(s - s64).max()
    assert raw <= 3e-4, raw
    free, gauge = _gauge_free(win, s - s64)
    assert np.abs(free).max() <= raw * (1 + 1e-9) + 1e-12, (np.abs(free).max(), np.abs(gauge).max(), raw)
    # the calibration (CalibHessian::value, unscaled: fx fy cx cy / 50) and the newest frame's evaluation point travel with the states
    assert np.abs(np.array(P.calib_value[:]) - np.array(P64.calib_value[:])).max() <= 1e-5
    assert np.abs(d["evalPT"][nf - 1] - d64["evalPT"][nf - 1]).max() <= 3e-4
    di = np.abs(i.astype(np.float64) - i64)
    # idepths: the maximum sits on single weakly observed points (either path: the CPU float path reaches 9e-4 on such points); the bulk
    # is what the bar is about
    assert np.percentile(di, 99) <= 5e-5 and di.max() <= 2e-3, (np.percentile(di, 99), di.max())
    flipped = np.nonzero(r != r64)[0]
    assert len(flipped) <= max(2, nr // 2000)
    for j in flipped:
        # IN on one side, OUTLIER on the other (the energy test of Residuals.cpp:303-311): the kept energy against the clamp value
        if {int(r[j]), int(r64[j])} == {0, 2}:
            e_in, th = (nw[j], ne64[j]) if r[j] == 0 else (nw64[j], ne[j])
            assert abs(float(e_in) - float(th)) <= 1e-3 * float(th), (j, e_in, th)
    assert abs(o.lastEnergy - o64.lastEnergy) <= 1e-4 * o64.lastEnergy


def test_device_noise_is_the_cpu_float_noise(gpu_ctx, oracle):
    """The statement about the kernels: over 18 windows (4 .. 8 keyframes, two image sizes, two of them with points initialised 30 % off; the
    first 18 of tests/diag/truth_spread.py's 24 — the suite has to fit the GPU box's time limit, the diagnostic runs them all) the device is
    NO FARTHER from the f64-accumulator truth than the CPU float path — the reference's own arithmetic with its own summation order.

    Rounds 1-5 summed in fp32 MFMA chains and the two float paths scattered around the truth alike (profiles/r05_truth_updates.txt: first
    update 1.13e-5 / 1.31e-5 median, but 6.8e-5 / 3.1e-5 maximum: the round-5 verdict's asymmetry).  From round 6 the cross-residual and
    cross-point sums run in f64 on the matrix cores (csrc/ba_kernels.hip ACC_MODE 1) and what is left is the solve's own floor.  Measured on
    MI355X over the 24 windows (profiles/r06_truth_updates.txt, profiles/r06_acc_modes.txt): first update median 1.39e-6 against the CPU's
    1.31e-5, maximum 2.39e-5 against 3.09e-5; device farther than the CPU + 5e-6 in 3 of 141 updates, the CPU farther than the device + 5e-6
    in 30; final states median 7.5e-6 against 2.4e-5, device worse on 3 windows of 24; the two maxima (1.17e-4 / 1.12e-4) are one window
    (w06_noisy) whose distance is residuals flipping on their outlier threshold, the same on both paths."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag"))
    import truth_spread
    rows, upd = [], []
    for name, win in truth_spread.windows(18):
        tr = {}
        dev, cpu, its = truth_spread.loop_distances(gpu_ctx, oracle, win, traces=tr)
        assert its[0] == its[2], (name, its)                                 # the device takes the truth's number of iterations
        rows.append((dev, cpu))
        upd.append((name, tr["dev"], tr["cpu"]))
    # The pose UPDATE of every Gauss-Newton iteration against the truth's, in the units the pose moves in (x * SCALE, translation and
    # rotation entries of every frame).  Iteration 0 starts from the same state on all three paths: there the statement is about the
    # accumulation and the solve alone.
    us = truth_spread.summarize_updates(upd)
    bad = [(name, it, d, c) for name, dv, cv in upd for it, (d, c) in enumerate(zip(dv, cv)) if d > c + 5e-6]
    msg = "%s; device farther than the CPU float path + 5e-6 (window, iteration, device, cpu): %s" % (us, bad)
    # the round-5 verdict's bars, as stated there
    assert us["first_dev_max"] <= us["first_cpu_max"] + 5e-6, msg
    assert us["first_dev_median"] <= 0.5 * us["first_cpu_median"], msg                        # measured 0.11x
    first_within = sum(1 for _, dv, _ in upd if dv[0] <= 1e-5)
    assert first_within >= len(upd) - 2, (first_within, msg)                                  # north_star's 1e-5, against the truth, on iteration 0
    assert us["dev_median"] <= us["cpu_median"] + 2e-7, msg
    assert us["dev_farther_by_5e6"] <= max(4, us["n"] // 20) and us["dev_farther_by_5e6"] <= us["cpu_farther_by_5e6"], msg
    assert us["dev_max"] <= us["cpu_max"] + 1e-5 and us["cpu_max"] <= 2e-4, msg
    sm = truth_spread.summarize(rows)
    assert sm["dev_median"] <= sm["cpu_median"], sm
    assert sm["dev_mean"] <= sm["cpu_mean"], sm
    assert sm["dev_max"] <= sm["cpu_max"] + 1e-5 and sm["cpu_max"] <= 3e-4, sm              # (both maxima: the threshold flips of one noisy window)
    assert sm["dev_worse"] <= len(rows) // 3, sm                                            # measured: 3 of 18
