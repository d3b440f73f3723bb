"""Wall time of FullSystem::optimize through the library: one 8KF/2000-point window (device-resident loop vs SDSO_BA_HOST_LOOP=1,
upload included / excluded) and a batch of windows through sdso_ba_batch_optimize."""
import ctypes as C
import os
os.environ.setdefault("SDSO_DEBUG_ENV", "1")   # the library reads its A/B switches only behind this gate
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth

ctx = abi.Context(0)
win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
nf, npts, nr = win["nf"], win["np"], win["nr"]
for f in range(nf):
    ctx.upload_pyramid(700 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[700 + f for f in range(nf)])
s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
for mode in ("resident", "host"):
    if mode == "host":
        os.environ["SDSO_BA_HOST_LOOP"] = "1"
    else:
        os.environ.pop("SDSO_BA_HOST_LOOP", None)
    t_up, t_opt = [], []
    for rep in range(12):
        t0 = time.perf_counter()
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 80, C.byref(W)))
        t1 = time.perf_counter()
        ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 80, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
        t2 = time.perf_counter()
        t_up.append(t1 - t0); t_opt.append(t2 - t1)
    print("%-8s upload %.3f ms  optimize %.3f ms (median of 10 after 2 warm-up; %d GN iterations, %d residuals)"
          % (mode, np.median(t_up[2:]) * 1e3, np.median(t_opt[2:]) * 1e3, o.iterations, nr))
os.environ.pop("SDSO_BA_HOST_LOOP", None)
for nwin in (8, 32, 128):
    ids = []
    rs = np.random.RandomState(1)
    for k in range(nwin):
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 100 + k, C.byref(W)))
        ids.append(100 + k)
    ids = np.array(ids, np.int32)
    ts = []
    for rep in range(5):
        for k in range(nwin):
            ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 100 + k, C.byref(W)))
        ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, nwin, abi.ip(ids)))
        res = (abi.BAOptResult * nwin)()
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(ctx.L.sdso_ba_batch_optimize(ctx.h, 6, res))
        ts.append(time.perf_counter() - t0)
    print("batch of %3d windows: sdso_ba_batch_optimize %.3f ms = %.1f us per window (%d iterations each)" % (nwin, np.median(ts[1:]) * 1e3, np.median(ts[1:]) * 1e6 / nwin, res[0].iterations))
