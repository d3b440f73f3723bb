"""Diagnostic for a -DSDSO_LM_STAMPS build: cycle counts of k_track_lm's phases (thread 0) come back in the result fields."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi, synth
import helpers
ctx = abi.Context(0)
prob = synth.tracker_problem(w=1232, h=368, npts=2000, seed=2002)
ctx.upload_pyramid(2, prob["pyr_new"]); ctx.set_ref(1, prob["pc"])
prm = helpers.track_params(prob)
for rep in range(3):
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); o = abi.TrackResult()
    ctx.check(ctx.L.sdso_track_newest_coarse(ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(o)))
    print("evaluations", o.evaluations, "cycles: fill_eval %d accumulate %d reduce %d step %d" % (o.lastFlowIndicators[0], o.lastFlowIndicators[1], o.lastFlowIndicators[2], o.lastResiduals[4]))
