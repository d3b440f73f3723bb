"""Diagnostic: GPU-vs-oracle state difference after sdso_ba_optimize next to the oracle's own
sensitivity to the order of its float sums (points permuted inside each host group)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "stereo-dso-g2o_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from sdso_amd import abi
import synth
import pyoracle

def permuted(win, seed):
    rs = np.random.RandomState(seed)
    order = np.concatenate([rs.permutation(np.nonzero(win["host"] == h)[0]) for h in range(win["nf"])])
    w2 = dict(win)
    for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
        w2[k] = win[k][order]
    newidx = np.empty(win["np"], np.int64); newidx[order] = np.arange(win["np"])
    rp, rt = [], []
    for p_new, p_old in enumerate(order):
        m = win["res_point"] == p_old
        rp += [p_new] * int(m.sum()); rt += list(win["res_target"][m])
    w2["res_point"] = np.array(rp, np.int32); w2["res_target"] = np.array(rt, np.int32)
    return w2, order

def run_oracle(L, win):
    W, keep = abi.make_ba_window(win, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
    h = L.orc_ba_create(C.byref(W))
    st = np.zeros((win["nf"], 10)); idp = np.zeros(win["np"], np.float32); rs = np.zeros(win["nr"], np.uint8); out = abi.BAOptResult()
    L.orc_ba_optimize(h, 6, abi.dp(st), abi.fp(idp), abi.bp(rs), C.byref(out))
    L.orc_ba_destroy(h)
    return st, idp, rs, out

L = pyoracle.load()
ctx = abi.Context(0)
for name, kw in (("small", dict(w=640, h=480, nf=5, pts_per_kf=120, seed=3001)), ("c3", dict(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001))):
    win = synth.ba_window(**kw)
    so, io, ro, oo = run_oracle(L, win)
    w2, order = permuted(win, 5)
    sp, ip, rp, op = run_oracle(L, w2)
    for f in range(win["nf"]):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(win["nf"])])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    sg = np.zeros_like(so); ig = np.zeros_like(io); rg = np.zeros_like(ro); og = abi.BAOptResult()
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 3, 6, abi.dp(sg), abi.fp(ig), abi.bp(rg), C.byref(og)))
    print(name, "its", oo.iterations, og.iterations, "E", oo.lastEnergy, og.lastEnergy, op.lastEnergy)
    print("  |gpu-oracle| per frame max:", np.abs(sg - so).max(axis=1))
    print("  |oracle(perm)-oracle| per frame max:", np.abs(sp - so).max(axis=1))
    print("  state magnitude per frame:", np.abs(so).max(axis=1))
    print("  idepth diff gpu:", np.abs(ig - io).max(), "perm:", np.abs(ip - io[order]).max(), "res mismatches gpu", (rg != ro).sum())
ctx.close()
