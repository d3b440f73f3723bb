"""Diagnostic (GPU box): where does the device's first Gauss-Newton update pick up its distance from the f64-accumulator truth?
For a few windows at their uploaded state: the packed accumulators of the device, of the CPU float path and of the truth (rounded to
float once) are each pushed through the SAME device stitch + solve (sdso_ba_set_accumulators -> sdso_ba_solve), and so is the device's
block with ONE section at a time replaced by the truth's.  Distances are the pose entries of x in the units the pose moves in.
  python tests/diag/x_noise_blocks.py [n_windows]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests", os.path.join("tests", "diag")):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import pyoracle  # noqa: E402
import truth_spread  # noqa: E402

SC = truth_spread.STATE_SCALE


def pose_err(x, xr, nf):
    return float(np.abs((x[4:] - xr[4:]).reshape(nf, 8) * SC)[:, :6].max())


def main():
    n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    oracle = pyoracle.load()
    ctx = abi.Context(0)
    for name, win in truth_spread.windows(n_win):
        nf, npts, n = win["nf"], win["np"], 8 * win["nf"] + 4
        for f in range(nf):
            ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
        W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
        na = abi.accum_floats(nf)
        accs, xs = {}, {}
        for mode in ("f64", "f32"):
            oracle.orc_set_acc64(1 if mode == "f64" else 0)
            try:
                h = oracle.orc_ba_create(C.byref(W))
                oracle.orc_ba_linearize(h, None); oracle.orc_ba_apply_res(h); oracle.orc_ba_accumulate(h)
                if mode == "f64":
                    a = np.zeros(na); oracle.orc_ba_get_accumulators_f64(h, abi.dp(a))
                else:
                    a32 = np.zeros(na, np.float32); oracle.orc_ba_get_accumulators(h, abi.fp(a32)); a = a32.astype(np.float64)
                x = np.zeros(n)
                oracle.orc_ba_solve(h, 0, 1e-5, abi.dp(x), None, None, None, None)
                oracle.orc_ba_destroy(h)
            finally:
                oracle.orc_set_acc64(0)
            accs[mode], xs[mode] = a, x
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
        ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 3, None)); ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 3)); ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 3))
        ag = np.zeros(na, np.float32)
        ctx.check(ctx.L.sdso_ba_get_accumulators(ctx.h, 3, abi.fp(ag)))

        def solve_with(block):
            b = np.ascontiguousarray(block, np.float32)
            ctx.check(ctx.L.sdso_ba_set_accumulators(ctx.h, 3, abi.fp(b)))
            x = np.zeros(n)
            ctx.check(ctx.L.sdso_ba_solve(ctx.h, 3, 0, 1e-5, abi.dp(x), None, None, None, None))
            return x
        xt = solve_with(accs["f64"])                       # the truth's sums through the device's stitch + solve
        x_dev, x_cpu = solve_with(ag), solve_with(accs["f32"])
        secs, o0 = [], 0
        for sname, cnt, w in (("topA", nf * nf, 91), ("topL", nf * nf, 91), ("accD", nf ** 3, 64), ("accE", nf * nf, 32), ("accEB", nf * nf, 8), ("Hcc", 1, 16), ("bc", 1, 4)):
            secs.append((sname, o0, o0 + cnt * w)); o0 += cnt * w
        line = "%-14s solve(truth sums) vs oracle truth x %.2e | device sums %.2e  cpu-f32 sums %.2e | device with ONE section from the truth:" % (
            name, pose_err(xt, xs["f64"], nf), pose_err(x_dev, xt, nf), pose_err(x_cpu, xt, nf))
        for sname, lo, hi in secs:
            if sname == "topL":
                continue
            b = ag.astype(np.float64).copy(); b[lo:hi] = accs["f64"][lo:hi]
            line += "  %s %.2e" % (sname, pose_err(solve_with(b), xt, nf))
        # and the b columns of the top sums alone (entries 12 of the 13 x 13 blocks: packed indices of column r)
        print(line, flush=True)
        ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))
    ctx.close()


if __name__ == "__main__":
    main()
