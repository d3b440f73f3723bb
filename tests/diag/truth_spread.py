"""Distance of the GN loop's final frame states / idepths from the f64-ACCUMULATOR truth, over MANY windows: the device against the CPU
float path.  One window is one sample of float-accumulation noise carried through six iterations (a residual sitting on its outlier
threshold flips on either side), so a statement about a kernel needs a distribution, not three windows (round-3 verdict, Weak #2).
Prints one line per window and the summary the test asserts on (tests/test_ba_f64_truth_gpu.py::test_device_noise_is_the_cpu_float_noise).
  python tests/diag/truth_spread.py [n_windows]     (SDSO_BA_TAIL=0: the round-2 tail kernels)"""
import ctypes as C
import os
os.environ.setdefault("SDSO_DEBUG_ENV", "1")   # the library reads its A/B switches only behind this gate
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth
import pyoracle  # noqa: E402


def windows(n):
    """The n seeded windows, rendered on a few host threads (numpy releases the GIL; same arrays as one after the other: every window has
    its own RandomState)"""
    from concurrent.futures import ThreadPoolExecutor
    shapes = [(640, 480, 5, 120), (640, 480, 4, 80), (640, 480, 6, 150), (1232, 368, 8, 250), (640, 480, 7, 100), (1232, 368, 8, 120)]

    def gen(k):
        w, h, nf, ppk = shapes[k % len(shapes)]
        kw = dict(idepth_noise=0.3, state_noise=1e-2) if k % 7 == 6 else {}
        return "w%02d_nf%d%s" % (k, nf, "_noisy" if kw else ""), synth.ba_window(w=w, h=h, nf=nf, pts_per_kf=ppk, seed=5001 + 13 * k, **kw)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for item in ex.map(gen, range(n)):
            yield item


STATE_SCALE = np.array([0.5, 0.5, 0.5, 1.0, 1.0, 1.0, 10.0, 1000.0])   # state_scaled = SCALE * state (HessianBlocks.h:54-61)


def device_x_trace(ctx, wid, n, its):
    """lastX of every iteration of the resident loop, one iteration at a time on a one-window batch (the kernels of sdso_ba_optimize)"""
    ids = np.array([wid], np.int32)
    ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, 1, abi.ip(ids)))
    ctx.check(ctx.L.sdso_ba_batch_optimize_begin(ctx.h, 1))
    xg = np.zeros((its, n))
    for it in range(its):
        ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
        ctx.check(ctx.L.sdso_ba_batch_solve_step(ctx.h, 0.1 * 0.25 ** it, 1 if it >= 2 else 0))
        ctx.check(ctx.L.sdso_ba_batch_get_x(ctx.h, abi.dp(xg[it:it + 1])))
    og = (abi.BAOptResult * 1)()
    ctx.check(ctx.L.sdso_ba_batch_optimize_end(ctx.h, og))
    return xg, og[0].iterations


def pose_update_errors(x, xref, nf, its):
    """per iteration: max over frames of |x - xref| * SCALE on the six pose entries (the units the pose moves in)"""
    return [float(np.abs((x[it, 4:] - xref[it, 4:]).reshape(nf, 8) * STATE_SCALE)[:, :6].max()) for it in range(its)]


def loop_distances(ctx, oracle, win, its=6, traces=None):
    """(device - truth, cpu_f32 - truth) as (max |state|, max |idepth|) pairs, and the iteration counts.  traces: a dict that receives the
    per-iteration distance of the pose updates from the truth's ("dev", "cpu": lists over the iterations all three loops ran)"""
    nf, npts, nr = win["nf"], win["np"], win["nr"]
    for f in range(nf):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    res = {}
    for mode in ("f64", "f32"):
        oracle.orc_set_acc64(1 if mode == "f64" else 0)
        try:
            h = oracle.orc_ba_create(C.byref(W))
            s, i, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), abi.BAOptResult()
            oracle.orc_ba_optimize(h, its, abi.dp(s), abi.fp(i), None, C.byref(o))
            xt = np.zeros((8, 8 * nf + 4))
            oracle.orc_ba_get_x_trace(h, abi.dp(xt), 8)
            oracle.orc_ba_destroy(h)
        finally:
            oracle.orc_set_acc64(0)
        res[mode] = (s, i, o.iterations, xt)
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    s, i, o = np.zeros((nf, 10)), np.zeros(npts, np.float32), abi.BAOptResult()
    ctx.check(ctx.L.sdso_ba_optimize(ctx.h, 3, its, abi.dp(s), abi.fp(i), None, C.byref(o)))
    if traces is not None:
        ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
        xg, itg = device_x_trace(ctx, 3, 8 * nf + 4, its if nf >= 4 else 15)
        common = min(itg, res["f32"][2], res["f64"][2], its)
        traces["dev"] = pose_update_errors(xg, res["f64"][3], nf, common)
        traces["cpu"] = pose_update_errors(res["f32"][3], res["f64"][3], nf, common)
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))
    s64, i64, it64 = res["f64"][:3]
    dev = (np.abs(s - s64).max(), np.abs(i.astype(np.float64) - i64).max())
    cpu = (np.abs(res["f32"][0] - s64).max(), np.abs(res["f32"][1].astype(np.float64) - i64).max())
    return dev, cpu, (o.iterations, res["f32"][2], it64)


def summarize(rows):
    dev = np.array([r[0][0] for r in rows]); cpu = np.array([r[1][0] for r in rows])
    return dict(n=len(rows), dev_median=float(np.median(dev)), cpu_median=float(np.median(cpu)), dev_max=float(dev.max()), cpu_max=float(cpu.max()),
                dev_mean=float(dev.mean()), cpu_mean=float(cpu.mean()), dev_worse=int((dev > cpu).sum()))


def summarize_updates(upd):
    """upd: (name, device errors per iteration, cpu errors per iteration).  The per-iteration distance of the pose UPDATE from the truth's."""
    dev = np.array([d for _, dv, _ in upd for d in dv]); cpu = np.array([c for _, _, cv in upd for c in cv])
    d0 = np.array([dv[0] for _, dv, _ in upd]); c0 = np.array([cv[0] for _, _, cv in upd])
    return dict(n=int(len(dev)), dev_median=float(np.median(dev)), cpu_median=float(np.median(cpu)), dev_max=float(dev.max()), cpu_max=float(cpu.max()),
                dev_farther_by_5e6=int((dev > cpu + 5e-6).sum()), cpu_farther_by_5e6=int((cpu > dev + 5e-6).sum()),
                first_dev_median=float(np.median(d0)), first_cpu_median=float(np.median(c0)), first_dev_max=float(d0.max()), first_cpu_max=float(c0.max()),
                first_dev_farther=int((d0 > c0).sum()), first_n=int(len(d0)))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    oracle = pyoracle.load()
    ctx = abi.Context(0)
    rows = []
    print("# tail = %s" % ("round-2 kernels (SDSO_BA_TAIL=0)" if os.environ.get("SDSO_BA_TAIL") == "0" else "k_ba_tail (default)"))
    upd = []
    for name, win in windows(n):
        tr = {}
        dev, cpu, its = loop_distances(ctx, oracle, win, traces=tr)
        rows.append((dev, cpu))
        upd.append((name, tr["dev"], tr["cpu"]))
        print("%-16s its dev/cpu/truth %d/%d/%d   states: device %.2e  cpu-f32 %.2e   idepths: device %.2e  cpu-f32 %.2e" % (name, its[0], its[1], its[2], dev[0], cpu[0], dev[1], cpu[1]), flush=True)
        print("%-16s pose updates vs the truth's, per iteration   device %s   cpu-f32 %s" % ("", " ".join("%.1e" % v for v in tr["dev"]), " ".join("%.1e" % v for v in tr["cpu"])), flush=True)
    print("summary", summarize(rows))
    print("updates", summarize_updates(upd))
    ctx.close()
