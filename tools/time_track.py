"""Single-call latency of sdso_track_newest_coarse (1232x368, 2000 points) under the environment it is started with (SDSO_TRK_LM_SOLO,
SDSO_TRK_LM_CLUSTER), and of 8 / 64 hypotheses in one launch.  Prints one line."""
import ctypes as C, os, sys, time
os.environ.setdefault("SDSO_DEBUG_ENV", "1")   # the library reads its A/B switches only behind this gate
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi
import synth, helpers
ctx = abi.Context(0)
prob = synth.tracker_problem(w=1232, h=368, npts=int(os.environ.get("NPTS", "2000")), seed=2002)
ctx.upload_pyramid(2, prob["pyr_new"]); ctx.set_ref(1, prob["pc"])
prm = helpers.track_params(prob)
def trk():
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); o = abi.TrackResult()
    ctx.check(ctx.L.sdso_track_newest_coarse(ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(o)))
    return o, T
for _ in range(3): trk()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); o, T = trk(); ts.append((time.perf_counter() - t0) * 1e3)
line = "solo=%s cluster=%s: trackNewestCoarse median %.4f ms min %.4f (evaluations %d, iterations %s, t %s)" % (
    os.environ.get("SDSO_TRK_LM_SOLO", "default"), os.environ.get("SDSO_TRK_LM_CLUSTER", "default"), float(np.median(ts)), min(ts), o.evaluations, list(o.iterations), np.round(T.Rt()[1], 6).tolist())
ctx.check(ctx.L.sdso_prof_reset(ctx.h)); ctx.check(ctx.L.sdso_prof_enable(ctx.h, 1))
for _ in range(10): trk()
kms, kn = ctx.prof_read("k_track_lm")
ctx.check(ctx.L.sdso_prof_enable(ctx.h, 0))
line += " [kernel alone %.4f ms (HIP events, %d launches)]" % (kms / max(kn, 1), kn)
for nh in (8, 64):
    rs = np.random.RandomState(3)
    prms = (abi.TrackParams * nh)(*[prm for _ in range(nh)])
    def many():
        Ts = (abi.SE3 * nh)(*[abi.SE3.from_Rt(*synth.se3_exp(rs.normal(0, [0.01, 0.01, 0.05, 0.002, 0.002, 0.002]))) for _ in range(nh)])
        affs = (abi.Aff * nh)(*[abi.Aff(0, 0) for _ in range(nh)])
        outs = (abi.TrackResult * nh)()
        refs = np.full(nh, 1, np.int32); frames = np.full(nh, 2, np.int32)
        t0 = time.perf_counter()
        ctx.check(ctx.L.sdso_track_newest_coarse_batch(ctx.h, nh, abi.ip(refs), abi.ip(frames), prms, Ts, affs, outs))
        return (time.perf_counter() - t0) * 1e3
    many()
    line += " | %d hypotheses %.4f ms" % (nh, float(np.median([many() for _ in range(7)])))
print(line, flush=True)
ctx.close()
