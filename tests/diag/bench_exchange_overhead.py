"""What the SHARDED step costs beside the single-rank one, measured on ONE GPU: a 1-rank RCCL communicator with the exchange path forced
(SDSO_OPT_FORCE_EXCHANGE=1) makes sdso_ba_batch_optimize run accumulate -> ncclAllReduce (or ncclReduceScatter + solve + ncclAllGather of
x) -> tail without the fused step -> k_ba_resub_step -> k_ba_opt_pack -> ncclAllGather -> k_ba_opt_step, i.e. every kernel and every RCCL
call of the N > 1 step with nothing on the wire.  The difference to the plain run is the fixed overhead DESIGN.md §5's budgets add per
iteration.   python tests/diag/bench_exchange_overhead.py [nwin]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth


def run(mode, nwin, its=6, reps=5):
    if mode == "plain":
        os.environ.pop("SDSO_OPT_FORCE_EXCHANGE", None)
    else:
        os.environ["SDSO_OPT_FORCE_EXCHANGE"] = "1"
    ctx = abi.Context(0)
    if mode != "plain":
        uid = (C.c_ubyte * 128)()
        assert ctx.L.sdso_comm_unique_id(uid) == 0
        ctx.check(ctx.L.sdso_comm_init(ctx.h, 1, 0, uid))
    win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001)
    nf = win["nf"]
    for f in range(nf):
        ctx.upload_pyramid(100 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[100 + f for f in range(nf)])
    ids = np.arange(1, nwin + 1, dtype=np.int32)
    ts = []
    for r in range(reps + 1):
        for k in ids:
            ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, int(k), C.byref(W)))
        ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, nwin, abi.ip(ids)))
        ctx.check(ctx.L.sdso_ba_batch_exchange_mode(ctx.h, 1 if mode == "scatter" else 0))
        ctx.check(ctx.L.sdso_ba_batch_optimize_begin(ctx.h, 0))           # (no early stop: every window takes all its steps)
        ctx.L.sdso_ctx_sync(ctx.h)
        t0 = time.perf_counter()
        for it in range(its):
            ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
            if mode != "plain":
                ctx.check(ctx.L.sdso_ba_allreduce(ctx.h))
            ctx.check(ctx.L.sdso_ba_batch_solve_step(ctx.h, 1e-5, 0))
        ctx.L.sdso_ctx_sync(ctx.h)
        t1 = time.perf_counter()
        res = (abi.BAOptResult * nwin)()
        ctx.check(ctx.L.sdso_ba_batch_optimize_end(ctx.h, res))
        if r:
            ts.append((t1 - t0) / its * 1e3)
    ctx.close()
    return float(np.median(ts))


if __name__ == "__main__":
    nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    out = {"windows": nwin}
    for mode in ("plain", "allreduce", "scatter"):
        out[mode + "_ms_per_iteration"] = run(mode, nwin)
    out["allreduce_overhead_us"] = 1e3 * (out["allreduce_ms_per_iteration"] - out["plain_ms_per_iteration"])
    out["scatter_overhead_us"] = 1e3 * (out["scatter_ms_per_iteration"] - out["plain_ms_per_iteration"])
    print(json.dumps(out, indent=1))
