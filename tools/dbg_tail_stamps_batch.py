"""Diagnostic for a -DSDSO_TAIL_STAMPS build (tools/mk_variant.sh tailst -DSDSO_TAIL_STAMPS; SDSO_LIB_PATH=ab_libs/tailst.so): the phases of
k_ba_tail when 256 windows run at once (one workgroup per CU), against the single-window figures of tools/dbg_tail_stamps.py."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth  # noqa: E402

ctx = abi.Context(0)
win = dict(synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3021))
for f in range(8):
    ctx.upload_pyramid(760 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[760 + f for f in range(8)])
names = ("stage", "S1", "tiles", "SVecI+zeroAs+order", "assemble", "factor+solve", "x+xAd", "sums")
for nwin in (1, 64, 256):
    ids = np.arange(100, 100 + nwin, dtype=np.int32)
    for i in ids:
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, int(i), C.byref(W)))
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, nwin, abi.ip(ids)))
    for rep in range(2):
        ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
        ctx.check(ctx.L.sdso_ba_batch_solve(ctx.h, 1e-5, 0))
        x = np.zeros((nwin, 68))
        ctx.check(ctx.L.sdso_ba_batch_get_x(ctx.h, abi.dp(x)))
    m = x[:, :8].mean(axis=0)
    if hasattr(ctx.L, "sdso_dbg_tail_stamps"):     # the loop's form: solve + step in one launch (TAIL_STEP), the phases incl. the step part
        ctx.check(ctx.L.sdso_ba_batch_optimize_begin(ctx.h, 0))
        for it in range(3):
            ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
            ctx.check(ctx.L.sdso_ba_batch_solve_step(ctx.h, 1e-1 * 0.25 ** it, 0))
        ctx.check(ctx.L.sdso_ba_batch_optimize_end(ctx.h, None))
        st = np.zeros((nwin, 12))
        ctx.L.sdso_dbg_tail_stamps.argtypes = [C.c_void_p, C.c_int]
        assert ctx.L.sdso_dbg_tail_stamps(st.ctypes.data_as(C.c_void_p), nwin) == 0
        ms = st.mean(axis=0)
        print("%3d windows, in the loop (third iteration): " % nwin + "  ".join("%s %d" % (nm, v) for nm, v in zip(names + ("step part (P7)",), ms[:9]))
              + "  | sum %d" % ms[:9].sum(), flush=True)
    print("%3d windows, mean s_memtime ticks per phase: " % nwin + "  ".join("%s %d" % (nm, v) for nm, v in zip(names, m)) + "  | sum %d (max over windows %d)" % (m.sum(), x[:, :8].sum(axis=1).max()), flush=True)
ctx.close()
