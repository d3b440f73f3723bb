#!/usr/bin/env python3
"""Single-problem latencies of the drop-in calls (what one FullSystem call costs): trackNewestCoarse and optimize, GPU vs oracle."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi
import synth
import pyoracle, helpers

orc = pyoracle.load(fast=True)
ctx = abi.Context(0)
out = {}
prob = synth.tracker_problem(w=1232, h=368, npts=2000, seed=2002)
ctx.upload_pyramid(2, prob["pyr_new"]); ctx.set_ref(1, prob["pc"])
prm = helpers.track_params(prob)
def trk():
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); o = abi.TrackResult()
    ctx.check(ctx.L.sdso_track_newest_coarse(ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(o)))
    return o
trk(); t0 = time.perf_counter()
for _ in range(10): o = trk()
g = (time.perf_counter() - t0) / 10 * 1e3
t0 = time.perf_counter(); helpers.oracle_track(orc, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0)); c = (time.perf_counter() - t0) * 1e3
out["trackNewestCoarse_1232x368_2000pts"] = {"gpu_ms": g, "cpu_oracle_ms": c, "evaluations": o.evaluations, "point_evals": o.point_evals}
def trk_g2o():
    T = abi.SE3.from_Rt(np.eye(3), np.zeros(3)); aff = abi.Aff(0, 0); o = abi.TrackResult()
    ctx.check(ctx.L.sdso_g2o_track_newest_coarse(ctx.h, 1, 2, C.byref(prm), C.byref(T), C.byref(aff), C.byref(o)))
    return o
trk_g2o(); t0 = time.perf_counter()
for _ in range(10): o = trk_g2o()
g = (time.perf_counter() - t0) / 10 * 1e3
t0 = time.perf_counter(); helpers.oracle_track(orc, prob, prm, (np.eye(3), np.zeros(3)), (0.0, 0.0), fn="orc_g2o_track_newest_coarse"); c = (time.perf_counter() - t0) * 1e3
out["trackNewestCoarse_fork_live_g2o_factors"] = {"gpu_ms": g, "cpu_oracle_ms": c, "evaluations": o.evaluations, "point_evals": o.point_evals}
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
        return (time.perf_counter() - t0) * 1e3, sum(o.point_evals for o in outs), sum(o.good for o in outs)
    many(); r = [many() for _ in range(5)]
    g = float(np.mean([x[0] for x in r]))
    out["trackNewestCoarse_%d_hypotheses_lockstep" % nh] = {"gpu_ms": g, "gpu_ms_per_hypothesis": g / nh, "point_evals": int(r[0][1]), "good": int(r[0][2])}
win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
nf, npts, nr = win["nf"], win["np"], win["nr"]
for f in range(nf): ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
st, idp, rs, oo = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
def opt():
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 3, 6, abi.dp(st), abi.fp(idp), abi.bp(rs), C.byref(oo)))
opt(); t0 = time.perf_counter()
for _ in range(5): ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
up = (time.perf_counter() - t0) / 5 * 1e3
t0 = time.perf_counter()
for _ in range(5): opt()
g = (time.perf_counter() - t0) / 5 * 1e3
h = orc.orc_ba_create(C.byref(W)); t0 = time.perf_counter(); orc.orc_ba_optimize(h, 6, abi.dp(st), abi.fp(idp), abi.bp(rs), C.byref(oo)); c = (time.perf_counter() - t0) * 1e3; orc.orc_ba_destroy(h)
out["optimize_8kf_2000pts_%dres_upload_plus_%dits" % (nr, oo.iterations)] = {"gpu_ms": g, "of_which_window_upload_ms": up, "cpu_oracle_ms": c}
# the fork-live window edge: one computeError + linearizeOplus of every active residual of that window
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
pair_R = np.zeros((nf * nf, 9), np.float32); pair_t = np.zeros((nf * nf, 3), np.float32); pair_ab = np.zeros((nf * nf, 2), np.float32)
for h_ in range(nf):
    for t_ in range(nf):
        T = synth.se3_mul(win["poses"][t_], synth.se3_inv(win["poses"][h_]))
        pair_R[h_ * nf + t_] = T[0].astype(np.float32).ravel(); pair_t[h_ * nf + t_] = T[1].astype(np.float32)
        a = np.exp(win["affs"][t_][0] - win["affs"][h_][0]); pair_ab[h_ * nf + t_] = (a, win["affs"][t_][1] - a * win["affs"][h_][1])
rp = win["res_point"]
d = dict(nf=nf, nr=nr, w=win["w"], h=win["h"], pair_R=pair_R, pair_t=pair_t, pair_ab=pair_ab, host_b0=np.array([win["affs"][k][1] for k in range(nf)]),
         frameEnergyTH=win["frameEnergyTH"], cam=[float(x) for x in win["K"]], host=win["host"][rp], target=win["res_target"], u=win["u"][rp], v=win["v"][rp],
         idepth=win["idepth"][rp].astype(np.float64), color=win["color"][rp], weights=win["weights"][rp])
S, keep2 = cases.g2o_lba_struct(d, frame_slots=[40 + f for f in range(nf)], dI=[p[0] for p in win["pyrs"]])
e = np.zeros((nr, 8)); Jl = np.zeros((nr, 8, 13)); stl = np.zeros(nr, np.uint8); en = np.zeros((nr, 2), np.float32)
cpt = np.zeros((nr, 3), np.float32); ih = np.zeros(nr, np.float32); lv = np.zeros(nr, np.uint8)
args = (abi.dp(e), abi.dp(Jl), abi.bp(stl), abi.fp(en), abi.fp(cpt), abi.fp(ih), abi.bp(lv))
ctx.check(ctx.L.sdso_prof_enable(ctx.h, 1))
ctx.check(ctx.L.sdso_g2o_lba_eval(ctx.h, C.byref(S), *args)); ctx.check(ctx.L.sdso_prof_reset(ctx.h)); t0 = time.perf_counter()
for _ in range(5): ctx.check(ctx.L.sdso_g2o_lba_eval(ctx.h, C.byref(S), *args))
g = (time.perf_counter() - t0) / 5 * 1e3
kms, kn = ctx.prof_read("k_g2o_lba_eval")
t0 = time.perf_counter(); orc.orc_g2o_lba_eval(C.byref(S), *args); c = (time.perf_counter() - t0) * 1e3
out["g2o_window_edges_%dres_eval_with_host_copies" % nr] = {"gpu_ms": g, "kernel_ms": kms / max(kn, 1), "cpu_oracle_ms": c}
ctx.close()
print(json.dumps(out, indent=1))
