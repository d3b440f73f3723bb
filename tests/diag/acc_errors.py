"""Per section of the packed accumulator block: the device's and the CPU float path's distance from the f64-accumulator truth after ONE
linearize + applyRes + accumulate (max and rms of the error relative to the bin block's largest entry), and the whitened x error — the
numbers the bars of tests/test_ba_f64_truth_gpu.py::test_one_iteration_against_f64_truth are set from.
  python tests/diag/acc_errors.py > gpurun_out/acc_errors.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import pyoracle  # noqa: E402
import test_ba_f64_truth_gpu as T  # noqa: E402

if __name__ == "__main__":
    oracle = pyoracle.load()
    ctx = abi.Context(0)
    L = ctx.L
    for which, win in T.CASES.items():
        nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
        for f in range(nf):
            ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
        W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
        acc64, x64, H64, st64 = T._oracle_iteration(oracle, W, win, True)
        acc32, x32, H32, st32 = T._oracle_iteration(oracle, W, win, False)
        ctx.check(L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
        ag = np.zeros(abi.accum_floats(nf), np.float32)
        xg, Hg = np.zeros(n), np.zeros((n, n))
        ids = np.array([3], np.int32)
        ctx.check(L.sdso_ba_batch_create(ctx.h, 1, abi.ip(ids)))
        ctx.check(L.sdso_ba_batch_accumulate(ctx.h))
        ctx.check(L.sdso_ba_get_accumulators(ctx.h, 3, abi.fp(ag)))
        ctx.check(L.sdso_ba_solve(ctx.h, 3, 0, 1e-5, abi.dp(xg), abi.dp(Hg), None, None, None))
        ctx.check(L.sdso_ba_release_window(ctx.h, 3))
        o0 = 0
        for name, cnt, w in T.SECTIONS(nf):
            Tt = acc64[o0:o0 + cnt * w].reshape(-1, w)
            m = np.maximum(np.abs(Tt).max(axis=1, keepdims=True), 1e-30)
            eg = (ag[o0:o0 + cnt * w].reshape(-1, w).astype(np.float64) - Tt) / m
            ec = (acc32[o0:o0 + cnt * w].reshape(-1, w) - Tt) / m
            live = np.abs(Tt).max(axis=1) > 0
            if live.any():
                print("%-10s %-6s device max %.2e rms %.2e | cpu-f32 max %.2e rms %.2e" % (which, name, np.abs(eg[live]).max(), np.sqrt((eg[live] ** 2).mean()),
                                                                                         np.abs(ec[live]).max(), np.sqrt((ec[live] ** 2).mean())))
            o0 += cnt * w
        d = np.sqrt(np.abs(np.diag(H64))) + 1e-30
        sc = max(1.0, np.abs(x64 * d).max())
        print("%-10s H whitened: device %.2e cpu %.2e | x whitened: device %.2e cpu %.2e" % (which, np.abs((Hg - H64) / np.outer(d, d)).max(), np.abs((H32 - H64) / np.outer(d, d)).max(),
                                                                                            np.abs((xg - x64) * d).max() / sc, np.abs((x32 - x64) * d).max() / sc), flush=True)
    ctx.close()
