"""Shared helpers for the parity tests: marshal synthetic problems into the ABI structures."""
import ctypes as C

import numpy as np

from sdso_amd import abi


def track_params(prob, coarsest=None, ref_aff=(0.0, 0.0), exposure=(1.0, 1.0), max_its=(10, 20, 50, 50, 50)):
    p = abi.TrackParams()
    L = prob["levels"]
    p.levels = L
    for l in range(L):
        p.w[l] = prob["pyr_ref"][l].shape[1]
        p.h[l] = prob["pyr_ref"][l].shape[0]
        p.fx[l], p.fy[l], p.cx[l], p.cy[l] = prob["fx"][l], prob["fy"][l], prob["cx"][l], prob["cy"][l]
    p.ref_exposure, p.new_exposure = exposure
    p.ref_aff_g2l = abi.Aff(ref_aff[0], ref_aff[1])
    p.coarsestLvl = (min(L, 5) - 1) if coarsest is None else coarsest
    for i in range(5):
        p.minResForAbort[i] = float("nan")
        p.maxIterations[i] = max_its[i]
    p.coarseCutoffTH = 20.0
    p.huberTH = 9.0
    p.affineOptModeA = 1e12
    p.affineOptModeB = 1e8
    return p


def oracle_track(L, prob, prm, T0, aff0):
    n = prob["levels"]
    pc = prob["pc"]
    keep = []

    def ptrs(key):
        arr = (abi.c_float_p * n)()
        for l in range(n):
            a = np.ascontiguousarray(pc[l][key], np.float32)
            keep.append(a)
            arr[l] = abi.fp(a)
        return arr

    pcn = (C.c_int * n)(*[len(pc[l]["u"]) for l in range(n)])
    dI = (abi.c_float_p * n)()
    for l in range(n):
        a = np.ascontiguousarray(prob["pyr_new"][l], np.float32)
        keep.append(a)
        dI[l] = abi.fp(a)
    T = abi.SE3.from_Rt(*T0)
    aff = abi.Aff(*aff0)
    out = abi.TrackResult()
    L.orc_track_newest_coarse(pcn, ptrs("u"), ptrs("v"), ptrs("idepth"), ptrs("color"), dI, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out))
    return T, aff, out


def oracle_eval(L, pc_l, dI_l, ev, want_mask=True):
    n = len(pc_l["u"])
    H = np.zeros(64); b = np.zeros(8); res = np.zeros(6); nw = C.c_int(0)
    mask = np.zeros(max(n, 1), np.uint8)
    u, v, idp, col = [np.ascontiguousarray(pc_l[k], np.float32) for k in ("u", "v", "idepth", "color")]
    img = np.ascontiguousarray(dI_l, np.float32)
    L.orc_track_calc_res_gs(n, abi.fp(u), abi.fp(v), abi.fp(idp), abi.fp(col), abi.fp(img), C.byref(ev), abi.dp(H), abi.dp(b),
                            abi.dp(res), C.byref(nw), abi.bp(mask), None, 0)
    return H.reshape(8, 8), b, res, nw.value, mask[:n]


def permuted_window(win, seed):
    """The same BA window with the points shuffled inside every host group (residuals follow their
    points).  Mathematically identical problem; only the order of the float sums changes — used to
    measure the reference arithmetic's own sensitivity to summation order."""
    rs = np.random.RandomState(seed)
    order = np.concatenate([rs.permutation(np.nonzero(win["host"] == h)[0]) for h in range(win["nf"])])
    w2 = dict(win)
    for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
        w2[k] = win[k][order]
    rp, rt = [], []
    starts = np.searchsorted(win["res_point"], np.arange(win["np"]), side="left")
    ends = np.searchsorted(win["res_point"], np.arange(win["np"]), side="right")
    for p_new, p_old in enumerate(order):
        rp += [p_new] * int(ends[p_old] - starts[p_old])
        rt += list(win["res_target"][starts[p_old]:ends[p_old]])
    w2["res_point"] = np.array(rp, np.int32)
    w2["res_target"] = np.array(rt, np.int32)
    return w2, order
