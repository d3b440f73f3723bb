"""Windowed-BA workload of bench.py (BASELINE configs[2] at N=1, configs[4]-style sharding at N>1).

One step = one DSO-native Gauss-Newton iteration of EnergyFunctional for `batch` independent
8-keyframe windows: linearizeAll + applyRes + accumulateAF/LF/SCF (+ RCCL all-reduce of the packed
accumulators when the points of every window are sharded over N ranks) + stitch + solveSystemF +
resubstituteF.  Per-GPU work is fixed (2000 points ~ 12.5k point-residuals per window and rank), so
the global window grows with N (weak scaling); the all-reduce payload is batch x 161 KiB."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


class _DevBlob:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 3}


class BAWorkload:
    name = "windowed_ba_8kf_2kpts_per_gpu_kitti1232x368"
    kernel = "k_ba_lin_fused"
    unit = "point-residuals/s"
    bytes_per_unit = 760.0  # 80 B point + 8x4 taps x 12 B + 296 B RawResidualJacobian written

    def __init__(self, ctx, args, rank, world):
        import torch
        from sdso_amd import abi, synth
        self.ctx, self.abi, self.world, self.torch = ctx, abi, world, torch
        t0 = time.time()
        nwin = args.batch or 128   # enough independent windows that the working set is >> the 256 MB MALL
        win = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001, point_seed=(3001 + 131 * rank) if world > 1 else None)
        self.win = win
        nf = win["nf"]
        rs = np.random.RandomState(11)
        ids = []
        shared = os.environ.get("SDSO_BA_SHARED_IMAGES") == "1"   # experiment: all windows read the same 8 pyramids (cache-resident)
        for k in range(nwin):
            # distinct HBM-resident pyramids per window (content: the rendered keyframes + a little noise)
            for f in range(nf if (k == 0 or not shared) else 0):
                img = win["pyrs"][f][0][..., 0]
                if k:
                    img = np.clip(img + rs.uniform(-0.5, 0.5, img.shape).astype(np.float32), 0, 255).astype(np.float32)
                ctx.check(ctx.L.sdso_make_pyramid(ctx.h, 1000 + k * nf + f, 1232, 368, abi.fp(np.ascontiguousarray(img, np.float32))))
            W, keep = abi.make_ba_window(win, frame_slots=[1000 + (0 if shared else k) * nf + f for f in range(nf)])
            ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 100 + k, C.byref(W)))
            ids.append(100 + k)
        self.ids = np.array(ids, np.int32)
        ctx.check(ctx.L.sdso_ba_batch_create(ctx.h, nwin, abi.ip(self.ids)))
        self.nwin = nwin
        self.materialize = 0 if os.environ.get("SDSO_BA_NO_J") == "1" else 1
        ctx.check(ctx.L.sdso_ba_batch_set_materialize(ctx.h, self.materialize))
        if not self.materialize:
            self.bytes_per_unit = 464.0  # SURVEY §8d figure without the Jacobian record
        self.units_per_step = nwin * win["nr"]
        ptr, nfl = C.c_void_p(), C.c_long(0)
        ctx.check(ctx.L.sdso_ba_batch_accum_dev(ctx.h, C.byref(ptr), C.byref(nfl)))
        self.accum = None
        if world > 1:
            self.accum = torch.as_tensor(_DevBlob(ptr.value, nfl.value), device="cuda")
            self.stream = torch.cuda.ExternalStream(ctx.L.sdso_ctx_stream(ctx.h))
        self.config = {"workload": self.name, "windows_per_step": nwin, "keyframes": nf, "points_per_window_per_gpu": win["np"],
                       "residuals_per_window_per_gpu": win["nr"], "jacobians_materialized": bool(self.materialize), "allreduce_floats": int(nfl.value) if world > 1 else 0,
                       "parallelism": ("points sharded over %d ranks, 1 RCCL all-reduce of the packed accumulators per iteration" % world) if world > 1 else "single GPU"}
        print("[rank %d] BA setup %.1fs: %d windows x %d residuals" % (rank, time.time() - t0, nwin, win["nr"]), file=sys.stderr, flush=True)

    def step(self):
        ctx = self.ctx
        ctx.check(ctx.L.sdso_ba_batch_accumulate(ctx.h))
        if self.accum is not None:
            import torch.distributed as dist
            with self.torch.cuda.stream(self.stream):
                dist.all_reduce(self.accum, op=dist.ReduceOp.SUM)
        ctx.check(ctx.L.sdso_ba_batch_solve(ctx.h, 1e-5, 0))

    def verify(self):
        x = np.zeros((self.nwin, 68))
        self.ctx.check(self.ctx.L.sdso_ba_batch_get_x(self.ctx.h, self.abi.dp(x)))
        assert np.isfinite(x).all() and np.abs(x).max() > 0
        out = {"ba_window_iters_per_s_per_gpu": None, "max_abs_x": float(np.abs(x).max())}
        out["jacobians_materialized"] = bool(self.materialize)
        for k in ("k_ba_lin_fused", "k_ba_sc"):
            ms, n = self.ctx.prof_read(k)
            out[k + "_avg_ms"] = ms / max(n, 1)
        return out

    def cpu_baseline(self, budget_s=15.0):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # cpu_baseline leg only
        abi = self.abi
        orc = pyoracle.load(fast=True)
        win = self.win
        W, keep = abi.make_ba_window(win, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
        h = orc.orc_ba_create(C.byref(W))
        x = np.zeros(68)
        t0, its = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s:
            orc.orc_ba_linearize(h, None)
            orc.orc_ba_apply_res(h)
            orc.orc_ba_solve(h, 0, 1e-5, abi.dp(x), None, None, None, None)
            its += 1
        dt = time.perf_counter() - t0
        orc.orc_ba_destroy(h)
        return {"value": its * win["nr"] / dt, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": "%d GN iterations of one 8KF/2000-point window (%d residuals) in %.1f s, oracle -O3 -march=native, 1 thread"
                          % (its, win["nr"], dt), "ba_iters_per_s": its / dt}
