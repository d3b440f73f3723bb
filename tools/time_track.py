"""Wall-clock latency of ONE sdso_track_newest_coarse call (1232x368, 2000 points), median of 200 calls after warm-up."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi
import synth
import helpers
ctx = abi.Context(0)
prob = synth.tracker_problem(w=1232, h=368, npts=2000, seed=2002)
ctx.upload_pyramid(2, prob["pyr_new"]); ctx.set_ref(1, prob["pc"])
prm = helpers.track_params(prob)
ts = []
for rep in range(230):
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); o = abi.TrackResult()
    t0 = time.perf_counter()
    ctx.check(ctx.L.sdso_track_newest_coarse(ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(o)))
    ts.append(time.perf_counter() - t0)
ts = np.array(ts[30:]) * 1e3
print("trackNewestCoarse: evaluations %d  median %.3f ms  min %.3f  p90 %.3f  (good %d)" % (o.evaluations, np.median(ts), ts.min(), np.percentile(ts, 90), o.good))
