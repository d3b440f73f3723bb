import ctypes as C, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import pyoracle
from sdso_amd import abi
import synth
orc = pyoracle.load()
ctx = abi.Context(0)
for (w, h, nf, ppk) in ((640, 480, 5, 60), (1232, 368, 8, 250)):
    win = dict(synth.ba_window(w=w, h=h, nf=nf, pts_per_kf=ppk, seed=3021))
    win["solverMode"] = 64   # USE_GN: lambda = 0 -> A = lastHS exactly
    slots = [760 + f for f in range(nf)]
    for f in range(nf):
        ctx.upload_pyramid(slots[f], win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=slots)
    n = 8 * nf + 4
    xs = []
    for rep in range(3):
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 80 + rep, C.byref(W)))
        ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 80 + rep, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 80 + rep)); ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 80 + rep))
        x = np.zeros(n); H = np.zeros((n, n)); b = np.zeros(n)
        ctx.check(ctx.L.sdso_ba_solve(ctx.h, 80 + rep, 0, 0.0, abi.dp(x), abi.dp(H), abi.dp(b), None, None))
        xs.append(x.copy())
    sv = 1.0 / np.sqrt(np.diag(H) + 10)
    As = np.ascontiguousarray(H * np.outer(sv, sv)); bs = np.ascontiguousarray(sv * b)
    y = np.zeros(n)
    orc.orc_ldlt_solve(n, abi.dp(As), abi.dp(bs), abi.dp(y))
    xr = sv * y
    d = np.sqrt(np.abs(np.diag(H)))
    print(nf, "determinism:", np.abs(xs[0] - xs[1]).max(), np.abs(xs[0] - xs[2]).max(), "| vs oracle ldlt on the same H,b: abs", np.abs(xs[0] - xr).max(),
          "whitened", np.abs((xs[0] - xr) * d).max() / max(1, np.abs(xr * d).max()), "sym err", np.abs(H - H.T).max() / np.abs(H).max())
    print("   residual |H x - b| / |b| gpu %.3e oracle %.3e" % (np.linalg.norm(H @ xs[0] - b) / np.linalg.norm(b), np.linalg.norm(H @ xr - b) / np.linalg.norm(b)))
