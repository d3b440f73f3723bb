"""Diagnostic for a -DSDSO_LM_STAMPS build: cycle counts of k_track_lm's phases (thread 0) come back in the result fields."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi
import synth
import helpers
ctx = abi.Context(0)
NP = int(os.environ.get("NPTS", "2000"))
prob = synth.tracker_problem(w=1232, h=368, npts=NP, seed=2002)
ctx.upload_pyramid(2, prob["pyr_new"]); ctx.set_ref(1, prob["pc"])
prm = helpers.track_params(prob)
for rep in range(3):
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); o = abi.TrackResult()
    ctx.check(ctx.L.sdso_track_newest_coarse(ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(o)))
    names = ["fill_eval+barrier", "acc: EV + pc loads", "acc: project + taps", "acc: products", "wave reduce", "barrier (slowest wave)", "cross-wave sum", "step: finalize H b res", "step: consume_pre", "step: copy + LDLT", "step: propose_post", "closing barrier"]
    st = [T.R[k] for k in range(9)] + [T.t[k] for k in range(3)]
    print("evaluations", o.evaluations, "cycles per evaluation: " + "  ".join("%s %d" % (nm, v / o.evaluations) for nm, v in zip(names, st)), " | total per evaluation %d" % (sum(st) / o.evaluations))
