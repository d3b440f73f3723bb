"""Diagnostic (GPU box): one GN iteration's x (device, float oracle) against the f64-accumulator oracle, whitened by the f64 system."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import pyoracle
from sdso_amd import abi
import synth
orc = pyoracle.load()
ctx = abi.Context(0)
cases = {"small": synth.ba_window(w=640, h=480, nf=5, pts_per_kf=120, seed=3001),
         "c3": synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001),
         "c3b": synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3008),
         "noisy": synth.ba_window(w=640, h=480, nf=6, pts_per_kf=150, seed=3017, idepth_noise=0.3, state_noise=1e-2)}
for name, win in cases.items():
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    for f in range(nf):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    xs, Hs, steps = {}, {}, {}
    for mode in ("f32", "f64"):
        orc.orc_set_acc64(1 if mode == "f64" else 0)
        h = orc.orc_ba_create(C.byref(W))
        orc.orc_ba_linearize(h, None); orc.orc_ba_apply_res(h)
        x, H, st = np.zeros(n), np.zeros((n, n)), np.zeros(npts, np.float32)
        orc.orc_ba_solve(h, 0, 1e-5, abi.dp(x), abi.dp(H), None, None, None)
        orc.orc_ba_get_point_steps(h, abi.fp(st))
        orc.orc_ba_destroy(h)
        xs[mode], Hs[mode], steps[mode] = x, H, st
    orc.orc_set_acc64(0)
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 3, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 3)); ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 3))
    xg, Hg, sg = np.zeros(n), np.zeros((n, n)), np.zeros(npts, np.float32)
    ctx.check(ctx.L.sdso_ba_solve(ctx.h, 3, 0, 1e-5, abi.dp(xg), abi.dp(Hg), None, None, None))
    ctx.check(ctx.L.sdso_ba_get_point_steps(ctx.h, 3, abi.fp(sg)))
    d = np.sqrt(np.abs(np.diag(Hs["f64"]))) + 1e-30
    sc = max(1.0, np.abs(xs["f64"] * d).max())
    print(name, "x whitened vs f64: gpu %.3e cpu32 %.3e (scale %.3g) | H: gpu %.3e cpu32 %.3e | point steps: gpu %.3e cpu32 %.3e of %.3g"
          % (np.abs((xg - xs["f64"]) * d).max() / sc, np.abs((xs["f32"] - xs["f64"]) * d).max() / sc, sc,
             np.abs((Hg - Hs["f64"]) / np.outer(d, d)).max(), np.abs((Hs["f32"] - Hs["f64"]) / np.outer(d, d)).max(),
             np.abs(sg - steps["f64"]).max(), np.abs(steps["f32"] - steps["f64"]).max(), np.abs(steps["f64"]).max()), flush=True)
