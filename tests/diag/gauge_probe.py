"""Where does the difference between the device GN loop and the f64-accumulator truth live?  Prints, per window, the raw frame-state
difference, its part outside the seven gauge directions, and its part in the subspace the final system determines well (eigenvalues of
the whitened reduced Hessian above a threshold) — for the device and for the CPU float path."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import pyoracle  # noqa: E402
from test_ba_f64_truth_gpu import CASES, _gauge_free  # noqa: E402

oracle = pyoracle.load()
ctx = abi.Context(0)
for which, win in CASES.items():
    nf, npts, nr, n = win["nf"], win["np"], win["nr"], 8 * win["nf"] + 4
    for f in range(nf):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    res = {}
    for mode in ("f64", "f32"):
        oracle.orc_set_acc64(1 if mode == "f64" else 0)
        h = oracle.orc_ba_create(C.byref(W))
        s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
        oracle.orc_ba_optimize(h, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o))
        x, H = np.zeros(n), np.zeros((n, n))
        if mode == "f64":
            oracle.orc_ba_accumulate(h)
            oracle.orc_ba_solve(h, 0, 1e-5, abi.dp(x), abi.dp(H), None, None, None)
        oracle.orc_ba_destroy(h)
        res[mode] = (s, i, H)
    oracle.orc_set_acc64(0)
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    s, i, r, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), np.zeros(nr, np.uint8), abi.BAOptResult()
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 3, 6, abi.dp(s), abi.fp(i), abi.bp(r), C.byref(o)))
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))
    s64, i64, H = res["f64"]
    d = np.sqrt(np.abs(np.diag(H))) + 1e-30
    lam, V = np.linalg.eigh(H / np.outer(d, d))
    print("== %s  (n %d)  whitened eigenvalues: min %.2e  max %.2e  #<1e-3: %d  #<1e-4: %d" % (which, n, lam.min(), lam.max(), (lam < 1e-3).sum(), (lam < 1e-4).sum()))
    for name, ss in (("device", s), ("cpu-f32", res["f32"][0])):
        ds = ss - s64
        free, gauge = _gauge_free(win, ds)
        v = np.zeros(n)
        for f in range(nf):
            v[4 + 8 * f:12 + 8 * f] = ds[f, :8]
        line = "  %-8s raw %.2e  gauge-free %.2e" % (name, np.abs(ds).max(), np.abs(free).max())
        for th in (1e-2, 1e-3, 1e-4):
            Vs = V[:, lam >= th]
            strong = (Vs @ (Vs.T @ (v * d))) / d
            line += "  strong(lam>=%.0e) %.2e" % (th, np.abs(strong).max())
        print(line)
