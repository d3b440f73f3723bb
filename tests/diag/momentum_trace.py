"""Per-iteration pose updates of the GN loop with SOLVER_MOMENTUM / SOLVER_STEPMOMENTUM at configs[2]: device and CPU float path against the
f64-accumulator truth (python tests/diag/momentum_trace.py [bits])."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth  # noqa: E402
import pyoracle  # noqa: E402
import test_ba_solver_bits_gpu as T  # noqa: E402

if __name__ == "__main__":
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    oracle = pyoracle.load()
    ctx = abi.Context(0)
    win = dict(synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001))
    win["solverMode"] = (int(sys.argv[2]) if len(sys.argv) > 2 else T.DEFAULT) | bits
    nf, n = win["nf"], 8 * win["nf"] + 4
    W, keep = T._upload(ctx, win)
    NIT = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    o32 = T._oracle_loop(oracle, W, win, NIT)
    o64 = T._oracle_loop(oracle, W, win, NIT, acc64=True)
    ids = np.array([3], np.int32)
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, 1, abi.ip(ids)))
    ctx.check(ctx.L.sdso_ba_batch_optimize_begin(ctx.h, 1))
    xg = np.zeros((6, n))
    for it in range(NIT):
        ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
        ctx.check(ctx.L.sdso_ba_batch_solve_step(ctx.h, 0.1 * 0.25 ** it, 1 if it >= 2 else 0))
        ctx.check(ctx.L.sdso_ba_batch_get_x(ctx.h, abi.dp(xg[it:it + 1])))
    og = (abi.BAOptResult * 1)()
    ctx.check(ctx.L.sdso_ba_batch_optimize_end(ctx.h, og))
    sg = np.zeros((nf, 10)); ig = np.zeros(win["np"], np.float32); rg = np.zeros(win["nr"], np.uint8)
    ctx.check(ctx.L.sdso_ba_get_state(ctx.h, 3, abi.dp(sg), abi.fp(ig), abi.bp(rg)))
    print("iterations dev/cpu/truth", og[0].iterations, o32[3].iterations, o64[3].iterations, "step trace cpu", o32[5], "truth", o64[5])
    for it in range(min(og[0].iterations, len(o64[4]))):
        dd = np.abs((xg[it, 4:] - o64[4][it, 4:]).reshape(nf, 8) * T._STATE_SCALE)[:, :6].max()
        dc = np.abs((o32[4][it, 4:] - o64[4][it, 4:]).reshape(nf, 8) * T._STATE_SCALE)[:, :6].max()
        print("it %d  pose update vs truth: device %.2e  cpu_f32 %.2e   |x| %.2e" % (it, dd, dc, np.abs(o64[4][it]).max()))
    it = min(3, og[0].iterations - 1)
    np.set_printoptions(precision=2, linewidth=200)
    print("it %d: (x_dev - x_truth) * SCALE per frame:\n" % it, (xg[it, 4:] - o64[4][it, 4:]).reshape(nf, 8) * T._STATE_SCALE, "\n calib", xg[it, :4] - o64[4][it, :4])
    print("it %d: x_truth * SCALE per frame:\n" % it, o64[4][it, 4:].reshape(nf, 8) * T._STATE_SCALE)
    print("final states: device-truth %.2e  cpu-truth %.2e" % ((np.abs(sg - o64[0])[:, :8] * T._STATE_SCALE)[:, :6].max(), (np.abs(o32[0] - o64[0])[:, :8] * T._STATE_SCALE)[:, :6].max()))
    print("state diff (dev - truth) * SCALE:\n", (sg - o64[0])[:, :8] * T._STATE_SCALE)
    di = ig.astype(np.float64) - o64[1]
    print("idepth diff: max %.2e  mean %.2e  rel-mean %.2e" % (np.abs(di).max(), di.mean(), (di / o64[1]).mean()))
    print("residual states differing: device-truth %d  cpu-truth %d" % ((rg != o64[2]).sum(), (o32[2] != o64[2]).sum()))
    ctx.close()
