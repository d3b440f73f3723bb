"""Diagnostic for a -DSDSO_SOLVE_STAMPS build of libsdso_hip.so (make EXTRA=-DSDSO_SOLVE_STAMPS): k_ba_solve then returns the
cycle counts of its phases in x[0..8] instead of the solution."""
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
for rep in range(3):
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 80, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 80, None))
    ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 80))
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 80))
    x = np.zeros(68)
    ctx.check(ctx.L.sdso_ba_solve(ctx.h, 80, 0, 0.0, abi.dp(x), None, None, None, None))
    print("s_memtime ticks: assemble+scale %d factorise %d triangular %d tail %d | per-phase sums over the pivots: pivot search %d exchange %d w %d dot products %d division %d"
          % tuple(x[:9]))
