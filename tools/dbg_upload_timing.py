"""Phase times of sdso_ba_upload_window (SDSO_BA_UPLOAD_TIMING=1) on the 8KF / 2000-point bench window."""
import ctypes as C, os, sys
os.environ.setdefault("SDSO_DEBUG_ENV", "1")   # the library reads its A/B switches only behind this gate
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi
import synth
ctx = abi.Context(0)
win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
for f in range(8):
    ctx.upload_pyramid(700 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[700 + f for f in range(8)])
for rep in range(4):
    if rep == 3:
        os.environ["SDSO_BA_UPLOAD_TIMING"] = "1"
    import time
    t0 = time.perf_counter()
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 80, C.byref(W)))
    print("call %d: %.1f us" % (rep, (time.perf_counter() - t0) * 1e6), file=sys.stderr)
