#!/usr/bin/env python3
"""Wall-clock of the widened rows through the host-buffer C-ABI calls, next to the oracle on one host core.
These calls include the H2D/D2H copies of their arguments (they run once per keyframe, not in the GN inner loop), so the numbers
are call latencies, not kernel roofline figures.  Usage on the GPU box:  python3 tests/diag/bench_next_rows.py > gpurun_out/next_rows.json"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth
import pyoracle  # noqa: E402  (reported CPU baseline only)
import test_stereo as ts  # noqa: E402


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    orc = pyoracle.load(fast=True)
    ctx = abi.Context(0)
    out = {}
    W, H = 1232, 368
    prob = synth.tracker_problem(w=W, h=H, npts=2000, seed=2002)
    pyr = prob["pyr_ref"]
    img = np.ascontiguousarray(pyr[0][..., 0])
    # N2 makeImages
    g = timeit(lambda: ctx.check(ctx.L.sdso_make_pyramid(ctx.h, 5, W, H, abi.fp(img))), 20)
    lv = [np.zeros((H >> l, W >> l, 3), np.float32) for l in range(prob["levels"])]
    ptr = (abi.c_float_p * len(lv))(*[abi.fp(a) for a in lv])
    c = timeit(lambda: orc.orc_make_images(abi.fp(img), W, H, prob["levels"], ptr), 5)
    out["N2_makeImages_1232x368_5lvl"] = {"gpu_ms_incl_upload": g, "cpu_oracle_ms": c}
    # N1 makeCoarseDepthL0 (2000 points)
    u, v, idp = prob["points"]
    u = u.astype(np.int32); v = v.astype(np.int32); idp = idp.astype(np.float32); wgt = np.ones(len(u), np.float32)
    pcn = np.zeros(8, np.int32)
    g = timeit(lambda: ctx.check(ctx.L.sdso_track_make_ref(ctx.h, 31, 5, len(u), abi.ip(u), abi.ip(v), abi.fp(idp), abi.fp(wgt), abi.ip(pcn))), 20)
    t0 = time.perf_counter(); synth.make_pc(u, v, idp, wgt, pyr); c = (time.perf_counter() - t0) * 1e3
    out["N1_makeCoarseDepthL0_2000pts"] = {"gpu_ms": g, "cpu_numpy_restatement_ms": c, "pc_n": [int(x) for x in pcn[:prob["levels"]]]}
    # S4 stereo match + N3 traceOn (20k points)
    pr = synth.stereo_problem(w=W, h=H, npts=20000, seed=4001)
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    ctx.upload_pyramid(80, [left]); ctx.upload_pyramid(81, [right])
    n = len(pr["u"])
    K = np.array(pr["K"], np.float32); bl = float(pr["calib"]["baseline"])
    M = abi.StereoMatch(); M.n = n; M.u = abi.fp(pr["u"]); M.v = abi.fp(pr["v"])
    o = dict(status_fwd=np.zeros(n, np.uint8), status_back=np.zeros(n, np.uint8), idepth_stereo=np.zeros(n, np.float32), back_uv=np.zeros((n, 2), np.float32))
    for k, a in o.items():
        setattr(M, k, abi.bp(a) if a.dtype == np.uint8 else abi.fp(a))
    g = timeit(lambda: ctx.check(ctx.L.sdso_stereo_match_batch(ctx.h, 80, 81, abi.fp(K), bl, 1, C.byref(M))), 10)

    def cpu_match():
        co, wo, go, eo = ts._oracle_init(orc, pr, left, pr["u"], pr["v"])
        P, d = abi.make_trace_points(n, pr["u"], pr["v"], co, wo, go, eo)
        st = ts._oracle_trace(orc, pr, right, P, 1)
        good = np.nonzero(st == 0)[0]
        ub, vb = d["lastTraceUV"][good, 0].copy(), d["lastTraceUV"][good, 1].copy()
        c2, w2, g2, e2 = ts._oracle_init(orc, pr, right, ub, vb)
        Pb, db = abi.make_trace_points(len(good), ub, vb, c2, w2, g2, e2)
        ts._oracle_trace(orc, pr, left, Pb, 0)
    t0 = time.perf_counter(); cpu_match(); c = (time.perf_counter() - t0) * 1e3
    out["S4_stereoMatch_LRL_20000pts"] = {"gpu_ms": g, "cpu_oracle_ms": c, "good_fwd": int((o["status_fwd"] == 0).sum())}
    probt, ut, vt, idt, G = ts._trace_on_case(n=20000, seed=2041)
    host = np.ascontiguousarray(probt["pyr_ref"][0]); new = np.ascontiguousarray(probt["pyr_new"][0])
    ctx.upload_pyramid(85, [new])
    col, wg, gH, eth = ts._oracle_init(orc, dict(w=640, h=480), host, ut, vt)
    pg = np.zeros(len(ut), np.int32)

    def run_on(which):
        P, d = abi.make_trace_points(len(ut), ut, vt, col, wg, gH, eth)
        st = np.zeros(len(ut), np.uint8)
        if which == "gpu":
            ctx.check(ctx.L.sdso_trace_on_batch(ctx.h, 85, 1, C.byref(G), abi.ip(pg), C.byref(P), abi.bp(st)))
        else:
            orc.orc_trace_on_batch(abi.fp(new), 640, 480, 1, C.byref(G), abi.ip(pg), C.byref(P), abi.bp(st))
    g = timeit(lambda: run_on("gpu"), 10)
    t0 = time.perf_counter(); run_on("cpu"); c = (time.perf_counter() - t0) * 1e3
    out["N3_traceOn_%dpts_640x480" % len(ut)] = {"gpu_ms": g, "cpu_oracle_ms": c}
    # N3 activation
    d = ts._activation_case(orc, nf=8, per_host=400, seed=3051)
    for f in range(8):
        ctx.upload_pyramid(60 + f, d["win"]["pyrs"][f][:1])
    A, keep = ts._activate_struct(d, frame_slots=[60 + f for f in range(8)], dI=[p[0] for p in d["win"]["pyrs"]])
    na = A.n
    so = np.zeros(na, np.int8); io = np.zeros(na, np.float32); ro = np.zeros((na, 8), np.uint8)
    g = timeit(lambda: ctx.check(ctx.L.sdso_activate_points_batch(ctx.h, C.byref(A), so.ctypes.data_as(C.POINTER(C.c_int8)), abi.fp(io), abi.bp(ro))), 10)
    t0 = time.perf_counter(); orc.orc_activate_points(C.byref(A), so.ctypes.data_as(C.POINTER(C.c_int8)), abi.fp(io), abi.bp(ro)); c = (time.perf_counter() - t0) * 1e3
    out["N3_activate_%dpts_8kf" % na] = {"gpu_ms": g, "cpu_oracle_ms": c, "activated": int((so == 1).sum())}
    # N4 pixel selector
    ctx.upload_pyramid(91, pyr)
    m = np.zeros((H, W), np.float32)

    def sel_gpu():
        p = C.c_int(3); nn = C.c_int(0)
        ctx.check(ctx.L.sdso_pixel_select(ctx.h, 91, 2000.0, 1, 1.0, C.byref(p), abi.fp(m), C.byref(nn)))
        return nn.value
    g = timeit(sel_gpu, 10)
    keepl = [np.ascontiguousarray(pyr[l]) for l in range(3)]
    ptrs = (abi.c_float_p * 3)(*[abi.fp(a) for a in keepl])

    def sel_cpu():
        p = C.c_int(3)
        return orc.orc_pixel_select(ptrs, W, H, 2000.0, 1, 1.0, C.byref(p), abi.fp(m))
    c = timeit(sel_cpu, 3)
    out["N4_pixelSelector_1232x368_density2000"] = {"gpu_ms": g, "cpu_oracle_ms": c, "selected": sel_gpu()}
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
