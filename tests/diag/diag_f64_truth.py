"""Diagnostic (GPU box): distance of the device path and of the float CPU path from the f64-accumulator truth:
(1) packed accumulators after one linearize+apply+accumulate, (2) states / idepths after the GN loop, raw and modulo the scale gauge."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import pyoracle
from sdso_amd import abi
import synth
import test_oracle_ba as T
orc = pyoracle.load()
ctx = abi.Context(0)
cases = {"small": (synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001), 6),
         "c3": (synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001), 6)}
a = dict(synth.ba_window(w=640, h=480, nf=4, pts_per_kf=80, seed=3081)); a["affineOptModeA"] = a["affineOptModeB"] = -1.0
cases["aff_fixed"] = (a, 4)
a = dict(synth.ba_window(w=640, h=480, nf=4, pts_per_kf=80, seed=3081)); a["affineOptModeA"] = a["affineOptModeB"] = 0.0
cases["aff_free"] = (a, 4)


def gauge_fit(win, s, i, s_ref, i_ref):
    """remove the best common scale change between (s, i) and (s_ref, i_ref): translations t -> (1+e) t, idepth -> idepth (1-e)"""
    nf = win["nf"]
    g = np.zeros((nf, 10))
    for f in range(nf):
        R, t = T.frame_pose(win, f, s_ref[f])
        g[f, :3] = t / 0.5
    d = (s - s_ref)
    e = (g[:, :3] * d[:, :3]).sum() / max((g[:, :3] ** 2).sum(), 1e-30)
    return np.abs(d - e * g).max(), np.abs((i - i_ref) + e * i_ref).max(), e


for name, (win, its) in cases.items():
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    for f in range(nf):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    # (1) accumulators
    na = abi.accum_floats(nf)
    a32, a64 = np.zeros(na, np.float32), np.zeros(na)
    for mode in (0, 1):                      # the f64 shadow sums are built only while the mode is on
        orc.orc_set_acc64(mode)
        h = orc.orc_ba_create(C.byref(W))
        orc.orc_ba_linearize(h, None); orc.orc_ba_apply_res(h); orc.orc_ba_accumulate(h)
        if mode:
            orc.orc_ba_get_accumulators_f64(h, abi.dp(a64))
        else:
            orc.orc_ba_get_accumulators(h, abi.fp(a32))
        orc.orc_ba_destroy(h)
    orc.orc_set_acc64(0)
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 3, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 3)); ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 3))
    ag = np.zeros(na, np.float32)
    ctx.check(ctx.L.sdso_ba_get_accumulators(ctx.h, 3, abi.fp(ag)))
    o0 = 0
    for nm, cnt, w in (("topA", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8)):
        if nm == "accD":
            o0 = 2 * nf * nf * 91
        D = a64[o0:o0 + cnt * w].reshape(-1, w)
        m = np.maximum(np.abs(D).max(axis=1, keepdims=True), 1e-30)
        eg = (ag[o0:o0 + cnt * w].reshape(-1, w) - D) / m
        ec = (a32[o0:o0 + cnt * w].reshape(-1, w) - D) / m
        print(name, nm, "gpu max %.2e rms %.2e | cpu32 max %.2e rms %.2e" % (np.abs(eg).max(), np.sqrt((eg ** 2).mean()), np.abs(ec).max(), np.sqrt((ec ** 2).mean())))
        o0 += cnt * w
    # (2) GN loop
    out = {}
    for mode in ("f32", "f64"):
        orc.orc_set_acc64(1 if mode == "f64" else 0)
        h = orc.orc_ba_create(C.byref(W))
        s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        orc.orc_ba_optimize(h, its, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o))
        orc.orc_ba_destroy(h)
        out[mode] = (s, i, r, o.iterations)
    orc.orc_set_acc64(0)
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 3, its, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
    s64, i64 = out["f64"][0], out["f64"][1].astype(np.float64)
    print(name, "its", o.iterations, out["f32"][3], out["f64"][3],
          "| state: gpu-f64 %.3g cpu32-f64 %.3g" % (np.abs(s - s64).max(), np.abs(out["f32"][0] - s64).max()),
          "| idepth: gpu-f64 %.3g cpu32-f64 %.3g" % (np.abs(i - i64).max(), np.abs(out["f32"][1] - i64).max()))
    print(name, "  modulo scale: gpu (state %.3g idepth %.3g e %.3g)  cpu32 (state %.3g idepth %.3g e %.3g)" %
          (gauge_fit(win, s, i.astype(np.float64), s64, i64) + gauge_fit(win, out["f32"][0], out["f32"][1].astype(np.float64), s64, i64)), flush=True)
    dg = np.abs(s - s64); print("   gpu worst state entry", np.unravel_index(dg.argmax(), dg.shape), "per-col max", dg.max(axis=0)[:8])
