"""One warm sdso_ba_optimize (device-resident loop) on the 8KF / 2000-point window, for a rocprofv3 --kernel-trace timeline."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi
import synth
ctx = abi.Context(0)
win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
nf, npts, nr = win["nf"], win["np"], win["nr"]
for f in range(8):
    ctx.upload_pyramid(700 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[700 + f for f in range(8)])
s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
for rep in range(3):
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 80, C.byref(W)))
    t0 = time.perf_counter()
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 80, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
    print("optimize %d: %.1f us, %d iterations" % (rep, (time.perf_counter() - t0) * 1e6, o.iterations), file=sys.stderr)
