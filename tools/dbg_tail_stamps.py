"""Diagnostic for a -DSDSO_TAIL_STAMPS build of libsdso_hip.so (make EXTRA=-DSDSO_TAIL_STAMPS): k_ba_tail then returns the cycle
counts of its phases in x[0..7] instead of the solution (single-window sdso_ba_solve: flags HS | RESUB, folded sums)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth

ctx = abi.Context(0)
win = dict(synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3021))
for f in range(8):
    ctx.upload_pyramid(760 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[760 + f for f in range(8)])
names = ("stage", "S1", "tiles", "SVecI+zeroAs+order", "assemble", "factor+solve", "x+xAd", "sums")
for rep in range(4):
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 80, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 80, None))
    ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 80))
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 80))
    x = np.zeros(68)
    ctx.check(ctx.L.sdso_ba_solve(ctx.h, 80, 0, 0.0, abi.dp(x), None, None, None, None))
    print("s_memtime ticks: " + "  ".join("%s %d" % (nm, v) for nm, v in zip(names, x[:8])) + "  | sum %d" % x[:8].sum() + "  | wave 0's tile jobs: diagonal %d, 7 off-diagonal %d, frame-calibration %d, calibration %d" % tuple(x[8:12]))
