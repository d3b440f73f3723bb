"""Diagnostic: marginalise 10 % of a C3 window, then compare single-window and batch paths with the oracle."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth
import pyoracle  # noqa: E402

oracle = pyoracle.load()
ctx = abi.Context(0)
win = dict(synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001))
nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
for f in range(nf):
    ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
flag = (np.random.RandomState(77).uniform(size=npts) < 0.10).astype(np.uint8)


def prep(wid):
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, wid, C.byref(W)))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, wid, None))
    ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, wid))
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, wid))
    ctx.check(ctx.L.sdso_ba_marginalize_points(ctx.h, wid, abi.bp(flag), None, None))


h = oracle.orc_ba_create(C.byref(W))
oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
oracle.orc_ba_marginalize_points(h, abi.bp(flag), None, None)
oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
acc_o = np.zeros(abi.accum_floats(nf), np.float32)
oracle.orc_ba_get_accumulators(h, abi.fp(acc_o))
xo, Ho, bo = np.zeros(n), np.zeros((n, n)), np.zeros(n)
oracle.orc_ba_solve(h, 0, 0.1, abi.dp(xo), abi.dp(Ho), abi.dp(bo), None, None)
d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30


def report(tag, acc_g, xg, Hg=None, bg=None):
    o0 = 0
    s = []
    for name, cnt, w in (("topA", nf * nf, 91), ("topL", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8), ("Hcc", 1, 16), ("bc", 1, 4)):
        so, sg = acc_o[o0:o0 + cnt * w].reshape(-1, w), acc_g[o0:o0 + cnt * w].reshape(-1, w)
        sc = np.abs(so).max(axis=1, keepdims=True)
        s.append("%s %.1e" % (name, (np.abs(so - sg) / np.maximum(sc, 1e-30)).max()))
        o0 += cnt * w
    print(tag, " ".join(s), "nres", acc_o[o0:o0 + 2], acc_g[o0:o0 + 2])
    print(tag, "x whitened err %.3e of %.3e" % (np.abs((xg - xo) * d).max(), np.abs(xo * d).max()))
    if Hg is not None:
        print(tag, "H err %.3e  b err %.3e" % (np.abs((Hg - Ho) / np.outer(d, d)).max(), np.abs((bg - bo) / d).max()))


# single path
prep(3)
ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 3, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 3)); ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 3))
acc_g = np.zeros_like(acc_o); ctx.check(ctx.L.sdso_ba_get_accumulators(ctx.h, 3, abi.fp(acc_g)))
xg, Hg, bg = np.zeros(n), np.zeros((n, n)), np.zeros(n)
ctx.check(ctx.L.sdso_ba_solve(ctx.h, 3, 0, 0.1, abi.dp(xg), abi.dp(Hg), abi.dp(bg), None, None))
report("single", acc_g, xg, Hg, bg)
# batch path
prep(4)
ids = np.array([4], np.int32)
ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, 1, abi.ip(ids)))
for mat in (1, 1, 0):
    ctx.check(ctx.L.sdso_ba_batch_set_materialize(ctx.h, mat))
    ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
    acc_g = np.zeros_like(acc_o); ctx.check(ctx.L.sdso_ba_get_accumulators(ctx.h, 4, abi.fp(acc_g)))
    ctx.check(ctx.L.sdso_ba_batch_solve(ctx.h, 0.1, 0))
    xb = np.zeros(n); ctx.check(ctx.L.sdso_ba_batch_get_x(ctx.h, abi.dp(xb)))
    report("batch mat=%d" % mat, acc_g, xb)
    print("   batch vs single x: %.3e" % np.abs((xb - xg) * d).max())
    # the same accumulators through the single-window solve entry
    x2, H2, b2 = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    ctx.check(ctx.L.sdso_ba_solve(ctx.h, 4, 0, 0.1, abi.dp(x2), abi.dp(H2), abi.dp(b2), None, None))
    report("   sdso_ba_solve on the batch accumulators", acc_g, x2, H2, b2)
