"""Windowed-BA workload of bench.py (BASELINE configs[2] at N=1, configs[4] sharding at N>1).

One step = one DSO-native Gauss-Newton iteration of EnergyFunctional for `batch` independent
8-keyframe windows: linearizeAll + applyRes + accumulateAF/LF/SCF (+ ONE RCCL all-reduce of the packed
accumulators, issued by the library itself: sdso_ba_allreduce, when the points of every window are
sharded over N ranks) + stitch + solveSystemF + resubstituteF.

Multi-GPU: every window of the batch is ONE global window whose allPoints are cut into N contiguous
ranges (sdso_amd/dist.py), so after the all-reduce every rank solves the system of the unsharded
window; rank 0 checks that against a single-GPU, un-fused solve of the same global window
(`extra.sharded_x_whitened_err`).
  --scaling weak  (default) 2000 points per window AND RANK: the global window has 2000 N points
                  (BASELINE configs[2] at N=1), per-GPU work fixed;
  --scaling strong          BASELINE configs[4]: 8 keyframes x 8000 points in total, cut 1/2/4/8 ways.
The all-reduce payload is batch x 184 KiB.

SDSO_BA_GROUPS stream groups, each on its own sdso_ctx (= its own HIP stream); default 1 on one rank (see the
comment at `ngroups`), 2 on several: the bandwidth-bound linearisations of the groups are chained by events
(A, B, A, B, ...), so the rest of one group's iteration (Schur accumulation, fused tail kernel, points' step, and the
all-reduce at N>1) runs underneath the linearisation of the other group."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


class _DevBlob:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 3}


class _Group:
    pass


class BAWorkload:
    name = "windowed_ba_8kf_2kpts_per_gpu_kitti1232x368"
    kernel = "k_ba_lin_fused"
    unit = "point-residuals/s"
    bytes_per_unit = 760.0  # 80 B point + 8x4 taps x 12 B + 296 B RawResidualJacobian written

    def __init__(self, ctx, args, rank, world, device=0):
        import torch
        from sdso_amd import abi
        import synth
        self.ctx, self.abi, self.world, self.torch = ctx, abi, world, torch
        t0 = time.time()
        from sdso_amd import dist as sdist
        self.scaling = getattr(args, "scaling", "weak")
        strong = self.scaling == "strong"
        # enough independent windows that the working set is >> the 256 MB MALL.  256 since the end of round 3 (128 before): the
        # serial kernels behind the linearisation are one workgroup per window (k_ba_tail: 87 us for 128 windows on 256 CUs, 95 us for 256)
        # or rounds of resident workgroups, so a batch that fills the chip costs 4.2 instead of 4.6-4.7 us per window-iteration; the
        # linearisation itself scales linearly.  --batch 128 reproduces the earlier rounds' step (profiles/r03_bench_ba_128_windows.json).
        nwin = args.batch or (32 if strong else 256)
        # stream groups.  One rank: ONE group — every kernel of the step runs alone on the chip (the linearisation at its clean-stream
        # 0.40 of the HBM peak) and the step is lin + Schur + tail + points in sequence; two chained groups reach the same step time
        # with the linearisation slowed by the other group's tail (0.35), three unchained groups are 6 % faster but their overlapping
        # linearisation launches make the per-launch duration meaningless (profiles/r03_ab_groups.txt).  Several ranks: two chained
        # groups, so that the all-reduce of one group travels under the linearisation of the other.
        ngroups = max(1, min(int(os.environ.get("SDSO_BA_GROUPS", "1" if world == 1 else "2")), nwin))
        if strong:
            self.name = "windowed_ba_8kf_8kpts_sharded"
        # the GLOBAL window (identical on every rank) and this rank's share of its points: its slice of EVERY host keyframe's points
        # (sdso_amd/dist.py::shard_points, mode per_host — all nf Schur workgroups of a window stay busy on every rank)
        self.win_global = synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=1000 if strong else 250 * world, seed=3001)
        win = self.win_global if world == 1 else sdist.shard_window(self.win_global, rank, world)[0]
        self.win = win
        self.rank = rank
        nf = win["nf"]
        rs = np.random.RandomState(11)
        self.materialize = 0 if os.environ.get("SDSO_BA_NO_J") == "1" else 1
        if not self.materialize:
            self.bytes_per_unit = 464.0  # SURVEY §8d figure without the Jacobian record
        shared = os.environ.get("SDSO_BA_SHARED_IMAGES") == "1"   # experiment: all windows read the same 8 pyramids (cache-resident)
        self.groups = []
        nfl_total = 0
        for g in range(ngroups):
            G = _Group()
            G.ctx = ctx if g == 0 else abi.Context(device)
            # SDSO_BA_CUMASK="k[,stride]": k CUs of every group's ctx for its Schur + tail kernels (aux stream), the rest for the linearisations
            # (sdso_ctx_partition_cus) — with two or more chained groups the tail of one runs beside the linearisation of the next
            cm = os.environ.get("SDSO_BA_CUMASK")
            if cm:
                kk = [int(v) for v in cm.split(",")]
                G.ctx.check(G.ctx.L.sdso_ctx_partition_cus(G.ctx.h, kk[0], kk[1] if len(kk) > 1 else 0))
            lo, hi = g * nwin // ngroups, (g + 1) * nwin // ngroups
            ids = []
            for k in range(lo, hi):
                # distinct HBM-resident pyramids per window (content: the rendered keyframes + a little noise)
                first = (k == lo)
                for f in range(nf if (first or not shared) else 0):
                    img = win["pyrs"][f][0][..., 0]
                    if k:
                        img = np.clip(img + rs.uniform(-0.5, 0.5, img.shape).astype(np.float32), 0, 255).astype(np.float32)
                    G.ctx.check(G.ctx.L.sdso_make_pyramid(G.ctx.h, 1000 + k * nf + f, 1232, 368, abi.fp(np.ascontiguousarray(img, np.float32))))
                W, keep = abi.make_ba_window(win, frame_slots=[1000 + (lo if shared else k) * nf + f for f in range(nf)])
                G.ctx.check(G.ctx.L.sdso_ba_upload_window(G.ctx.h, 100 + k, C.byref(W)))
                ids.append(100 + k)
            G.ids = np.array(ids, np.int32)
            G.nwin = len(ids)
            G.ctx.check(G.ctx.L.sdso_ba_batch_create(G.ctx.h, G.nwin, abi.ip(G.ids)))
            G.ctx.check(G.ctx.L.sdso_ba_batch_set_materialize(G.ctx.h, self.materialize))
            # SDSO_BA_EXCHANGE=scatter: reduce-scatter by window + all-gather of x instead of the all-reduce (include/sdso_abi.h:
            # sdso_ba_batch_exchange_mode) — half the bytes per xGMI link; same results.  Applies when the group's windows divide by N.
            self.exchange_mode = 1 if (world > 1 and os.environ.get("SDSO_BA_EXCHANGE", "allreduce") == "scatter") else 0
            G.ctx.check(G.ctx.L.sdso_ba_batch_exchange_mode(G.ctx.h, self.exchange_mode))
            # (the all-reduce payload; sdso_ba_batch_accum_dev is NOT called here: handing the block's address out makes every
            #  accumulate fold eagerly, and the single-rank step leaves the folds to the fused tail kernel)
            nfl_total += int(G.ctx.L.sdso_ba_accum_floats(nf)) * G.nwin
            G.stream = torch.cuda.ExternalStream(G.ctx.L.sdso_ctx_stream(G.ctx.h))
            G.ev = torch.cuda.Event()
            self.groups.append(G)
        # communicator: RCCL inside the library (backend nccl).  Every multi-rank mode runs the SAME step — library all-reduce, fused
        # tail kernel, pack / all-gather / k_ba_opt_step, states advancing — or the run fails: there is no reduced-work fallback.
        #   nccl (default)  sdso_comm_init: the library opens its own RCCL communicator (one per stream group)
        #   when some rank cannot load RCCL inside the library (agreed on BEFORE any collective init, from a local probe): the library's
        #                   host transport (sdso_comm_init_host) with torch.distributed underneath — slower, same work; config.exchange says so
        #   gloo_lib        rehearsal on a box with fewer GPUs than ranks: host transport over gloo
        backend = os.environ.get("SDSO_DIST_BACKEND", "nccl")
        self.lib_comm = world > 1
        if world > 1:
            import torch.distributed as dist
            use_host = backend == "gloo_lib"
            why = ""
            if not use_host:
                probe = np.zeros(128, np.uint8)      # local and non-collective: dlopen of librccl + ncclGetUniqueId
                ok = 1 if ctx.L.sdso_comm_unique_id(probe.ctypes.data_as(C.c_void_p)) == 0 else 0
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) != 1:
                    use_host, why = True, "librccl not loadable inside libsdso_hip.so on some rank"
            def _host_transport(why):
                on_gpu = dist.get_backend() == "nccl"

                def _ar(user, buf, n):
                    a = np.ctypeslib.as_array(buf, shape=(n,))
                    t = torch.from_numpy(a)
                    if on_gpu:
                        t = t.cuda()
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    if on_gpu:
                        a[:] = t.cpu().numpy()
                    return 0

                def _ag(user, send, recv, n):
                    sd = torch.from_numpy(np.ctypeslib.as_array(send, shape=(n,)).copy())
                    if on_gpu:
                        sd = sd.cuda()
                    outl = [torch.empty(n, dtype=torch.float32, device=sd.device) for _ in range(world)]
                    dist.all_gather(outl, sd)
                    r = np.ctypeslib.as_array(recv, shape=(n * world,))
                    for k, o in enumerate(outl):
                        r[k * n:(k + 1) * n] = o.cpu().numpy()
                    return 0
                self._cbs = (abi.HOST_ALLREDUCE_FN(_ar), abi.HOST_ALLGATHER_FN(_ag))
                for G in self.groups:
                    G.ctx.check(G.ctx.L.sdso_comm_init_host(G.ctx.h, world, rank, self._cbs[0], self._cbs[1], None))
                self.exchange = "sdso_ba_allreduce over the library's host transport (torch.distributed %s underneath)%s" % (
                    dist.get_backend(), ("; " + why) if why else "; rehearsal")

            if use_host:
                _host_transport(why)
            else:
                # one communicator PER stream group: the groups' collectives run on different streams and interleave, and each group issues
                # its own sequence (all-reduce, all-gather) in program order on every rank.  ncclCommInitRank is collective: every rank
                # makes every call.  When it fails on some rank after the agreement above, every rank drops what it opened and the run
                # goes through the host transport: the SAME step (library all-reduce, fused tail, all-gather, states advancing), slower,
                # and config.exchange says which rank's RCCL said what — never a step that does less work.
                fails = 0
                for G in self.groups:
                    uid = np.zeros(128, np.uint8)
                    if rank == 0:
                        G.ctx.check(G.ctx.L.sdso_comm_unique_id(uid.ctypes.data_as(C.c_void_p)))
                    t = torch.from_numpy(uid).cuda()
                    dist.broadcast(t, src=0)
                    uid = t.cpu().numpy().copy()
                    if G.ctx.L.sdso_comm_init(G.ctx.h, world, rank, uid.ctypes.data_as(C.c_void_p)) != 0:
                        fails += 1
                flag = torch.tensor([fails], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()) != 0:
                    msg = ctx.L.sdso_last_error(ctx.h)
                    print("[rank %d] sdso_comm_init failed on some rank (this rank: %s): host transport instead" % (rank, msg.decode() if (msg and fails) else "ok"), file=sys.stderr, flush=True)
                    for G in self.groups:
                        G.ctx.L.sdso_comm_destroy(G.ctx.h)
                    _host_transport("sdso_comm_init (ncclCommInitRank inside libsdso_hip.so) failed on some rank")
                else:
                    self.exchange = "sdso_ba_allreduce (RCCL communicator owned by libsdso_hip.so)"
        self.nwin = nwin
        self.units_per_step = nwin * win["nr"]
        # what the LIBRARY's communicator says (sdso_comm_info), not what the launcher's environment says: 0 without a communicator
        nr_, rk_ = C.c_int(0), C.c_int(-1)
        ctx.check(ctx.L.sdso_comm_info(ctx.h, C.byref(nr_), C.byref(rk_)))
        self.comm_ranks = int(nr_.value)
        self.exchange_floats = nfl_total if world > 1 else 0
        self.config = {"workload": self.name, "windows_per_step": nwin, "keyframes": nf, "points_per_window_per_gpu": win["np"],
                       "residuals_per_window_per_gpu": win["nr"], "points_per_global_window": self.win_global["np"],
                       "jacobians_materialized": bool(self.materialize), "stream_groups": ngroups,
                       "cu_partition": os.environ.get("SDSO_BA_CUMASK") or None,
                       "allreduce_floats": nfl_total if world > 1 else 0, "exchange": getattr(self, "exchange", None),
                       "rccl_ranks": self.comm_ranks, "launcher_world_size": world,
                       "exchange_shape": ("reduce-scatter by window + all-gather of x" if getattr(self, "exchange_mode", 0) else "all-reduce, solve on every rank") if world > 1 else None,
                       "parallelism": ("the points of every host keyframe of every window cut %d ways (one slice of every host per rank), 1 RCCL all-reduce (sdso_ba_allreduce) of the packed accumulators per group and iteration" % world) if world > 1 else "single GPU"}
        # correctness at the initial state (one iteration, before the timed loop moves the states): see _verify_initial
        self.initial_check = self._verify_initial()
        # the timed step advances REAL state: the device-resident GN loop (sdso_ba_batch_optimize_begin / sdso_ba_batch_step) takes the step
        # the solver produced, rebuilds the tables and moves the newest frame's energy threshold, every step, on every window.
        # SDSO_BA_BENCH_STATIC=1: the round-1 behaviour (re-linearise the same state every step).
        self.advance = os.environ.get("SDSO_BA_BENCH_STATIC") != "1"
        if self.advance:
            for G in self.groups:
                G.ctx.check(G.ctx.L.sdso_ba_batch_optimize_begin(G.ctx.h, 0))
        self.config["state_advances"] = bool(self.advance)
        self.max_abs_x_steps = []
        print("[rank %d] BA setup %.1fs: %d windows x %d residuals in %d stream group(s)" % (rank, time.time() - t0, nwin, win["nr"], ngroups), file=sys.stderr, flush=True)

    # bench.py drives profiling / synchronisation through these so that every group's ctx is covered
    def prof_reset(self):
        for G in self.groups:
            G.ctx.check(G.ctx.L.sdso_prof_reset(G.ctx.h))

    def prof_enable(self, on):
        for G in self.groups:
            G.ctx.check(G.ctx.L.sdso_prof_enable(G.ctx.h, int(on)))

    def prof_read(self, kernel):
        ms, n = 0.0, 0
        for G in self.groups:
            a, b = G.ctx.prof_read(kernel)
            ms += a; n += b
        return ms, n

    def sync(self):
        for G in self.groups:
            G.ctx.sync()

    def close(self):
        for G in self.groups[1:]:
            G.ctx.close()

    def step(self):
        gs = self.groups
        prev = gs[-1]
        chain = len(gs) > 1 and os.environ.get("SDSO_BA_NOCHAIN") != "1"
        chain_lin = chain and os.environ.get("SDSO_BA_CHAIN", "lin") == "lin"
        for G in gs:
            if chain:
                G.stream.wait_event(prev.ev)          # the bandwidth-bound linearisations run one after the other ...
            if chain_lin:
                # ... and everything enqueued after the event overlaps the next group's linearisation: the Schur accumulation (which the
                # library puts on the ctx's side stream), the fused tail kernel (launched ahead of it on the main stream, so that its
                # workgroups take CUs before the next linearisation floods the chip, and waiting in-kernel for the Schur kernel's
                # signal), the points' step.  (SDSO_BA_CHAIN=acc records the event after the Schur kernel, as rounds 1-2 did.)
                G.ctx.check(G.ctx.L.sdso_ba_batch_linearize(G.ctx.h))
                G.ev.record(G.stream)
                G.ctx.check(G.ctx.L.sdso_ba_batch_schur(G.ctx.h))
            else:
                G.ctx.check(G.ctx.L.sdso_ba_batch_accumulate(G.ctx.h))
                if chain:
                    G.ev.record(G.stream)
            if self.lib_comm:
                G.ctx.check(G.ctx.L.sdso_ba_allreduce(G.ctx.h))        # RCCL over xGMI, enqueued on the ctx stream by the library
            if self.advance:
                # solveSystemF + resubstitute + doStepFromBackup + tables + setNewFrameEnergyTH: ONE launch of the fused tail kernel per group
                # (+ pack / all-gather / k_ba_opt_step when sharded)
                G.ctx.check(G.ctx.L.sdso_ba_batch_solve_step(G.ctx.h, 1e-5, 0))
            else:
                G.ctx.check(G.ctx.L.sdso_ba_batch_solve(G.ctx.h, 1e-5, 0))
            prev = G

    def _one_iteration(self):
        gs = self.groups
        for G in gs:
            G.ctx.check(G.ctx.L.sdso_ba_batch_accumulate(G.ctx.h))
            if self.lib_comm:
                G.ctx.check(G.ctx.L.sdso_ba_allreduce(G.ctx.h))
            G.ctx.check(G.ctx.L.sdso_ba_batch_solve(G.ctx.h, 1e-5, 0))

    def _verify_initial(self):
        """One GN iteration at the uploaded state: x of every window is finite; window 0's x (fused batch path, summed over the ranks
        when sharded) equals the x of the UNSHARDED global window solved through the un-fused single-window entry points on this GPU
        (2e-4 in the whitened metric, the bar of tests/test_ba_gpu.py)."""
        abi = self.abi
        self._one_iteration()
        out = {}
        mx, x0 = 0.0, None
        for G in self.groups:
            x = np.zeros((G.nwin, 68))
            G.ctx.check(G.ctx.L.sdso_ba_batch_get_x(G.ctx.h, abi.dp(x)))
            assert np.isfinite(x).all() and np.abs(x).max() > 0
            mx = max(mx, float(np.abs(x).max()))
            if x0 is None:
                x0 = x[0].copy()
        out["max_abs_x_initial"] = mx
        if self.rank == 0:
            g0, wg, nf = self.groups[0], self.win_global, self.win_global["nf"]
            W, keep = abi.make_ba_window(wg, frame_slots=[1000 + f for f in range(nf)])          # the pyramids of window 0
            g0.ctx.check(g0.ctx.L.sdso_ba_upload_window(g0.ctx.h, 9000, C.byref(W)))
            g0.ctx.check(g0.ctx.L.sdso_ba_linearize(g0.ctx.h, 9000, None))
            g0.ctx.check(g0.ctx.L.sdso_ba_apply_res(g0.ctx.h, 9000))
            g0.ctx.check(g0.ctx.L.sdso_ba_accumulate(g0.ctx.h, 9000))
            xr, Hr = np.zeros(68), np.zeros((68, 68))
            g0.ctx.check(g0.ctx.L.sdso_ba_solve(g0.ctx.h, 9000, 0, 1e-5, abi.dp(xr), abi.dp(Hr), None, None, None))
            g0.ctx.check(g0.ctx.L.sdso_ba_release_window(g0.ctx.h, 9000))
            d = np.sqrt(np.abs(np.diag(Hr))) + 1e-30
            err = float(np.abs((x0 - xr) * d).max() / max(1.0, np.abs(xr * d).max()))
            out["sharded_x_whitened_err" if self.world > 1 else "fused_vs_unfused_x_whitened_err"] = err
            if self.world == 1:
                out["sharded_x_whitened_err"] = 0.0      # (one rank holds the whole window: the key exists on every N)
            assert err <= 2e-4, "x of window 0 differs from the unsharded / un-fused reference solve: %g" % err
        return out

    def verify(self):
        """After the timed steps: x of every window still finite; with state_advances the Gauss-Newton steps must have shrunk
        (max |x| of the last step against the first iteration's) — the loop really moved the states towards the optimum."""
        abi = self.abi
        out = {"jacobians_materialized": bool(self.materialize)}
        out.update(self.initial_check)
        mx = 0.0
        for G in self.groups:
            x = np.zeros((G.nwin, 68))
            G.ctx.check(G.ctx.L.sdso_ba_batch_get_x(G.ctx.h, abi.dp(x)))
            assert np.isfinite(x).all()
            mx = max(mx, float(np.abs(x).max()))
        out["max_abs_x"] = mx
        if self.advance:
            assert mx < out["max_abs_x_initial"], "the GN steps did not shrink: %g -> %g" % (out["max_abs_x_initial"], mx)
        # the secondary kernels, bracketed in a few extra steps AFTER the timed loop (level-2 profiling: the timed steps bracket the
        # dominant kernel only — two event records per bracket and stream are queue time the step would otherwise pay for nothing)
        ms, n = self.prof_read("k_ba_lin_fused")
        out["k_ba_lin_fused_avg_ms"] = ms / max(n, 1)
        if os.environ.get("SDSO_BENCH_SECONDARY", "1") == "1":
            for G in self.groups:
                G.ctx.check(G.ctx.L.sdso_prof_enable(G.ctx.h, 2))
            names = ("k_ba_sc", "k_ba_tail", "k_ba_resub", "sdso_ba_allreduce")
            base = {k: self.prof_read(k) for k in names}
            nextra = 5
            for _ in range(nextra):
                self.step()
            self.sync()
            for k in names[:3]:
                ms, n = self.prof_read(k)
                # (no launch of its own: from one window per CU on the points' back-substitution and step run inside k_ba_tail — TAIL_RESUB)
                out[k + "_avg_ms"] = (ms - base[k][0]) / (n - base[k][1]) if n > base[k][1] else None
            if out.get("k_ba_resub_avg_ms") is None:
                out["k_ba_resub_note"] = "fused into k_ba_tail (TAIL_RESUB: one tail workgroup per CU or more)"
            # the exchange of the packed accumulators: bytes this rank hands to the collective per step (all stream groups) and the HIP-event
            # time of sdso_ba_allreduce on the ctx streams per step; zeros on one rank (no communicator, nothing is exchanged) — the same
            # keys on every N, so that an N = 1 line of a scaling series reads like the others
            ms, n = self.prof_read("sdso_ba_allreduce")
            out["exchange_bytes_per_step"] = 4 * self.exchange_floats
            out["exchange_ms_per_step"] = (ms - base["sdso_ba_allreduce"][0]) / nextra if self.world > 1 else 0.0
            out["exchange_calls_per_step"] = (n - base["sdso_ba_allreduce"][1]) / nextra if self.world > 1 else 0
            for G in self.groups:
                G.ctx.check(G.ctx.L.sdso_prof_enable(G.ctx.h, 0))
            # the Schur kernel against ITS roofline (24 B per point + 32 B per residual of it, i.e. what the reference's addPoint
            # reads): per launch of this rank's windows
            if out.get("k_ba_sc_avg_ms", 0) > 0:
                nwin = sum(G.nwin for G in self.groups)
                alg = (24.0 * self.win["np"] + 32.0 * self.win["nr"]) * nwin / max(len(self.groups), 1)
                gbps = alg / (out["k_ba_sc_avg_ms"] * 1e-3) / 1e9
                out["k_ba_sc_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_launch": alg, "achieved": gbps, "peak": 8000.0, "unit": "GB/s",
                                           "frac": gbps / 8000.0, "note": "PMC traffic per launch: profiles/rNN_ba_rocprof_summary.txt (k_ba_sc_host: 2*FETCH + WRITE)"}
        # The second unit of SURVEY §8d, measured the same way in a few extra steps: the records stay in registers (464 B per point-residual
        # instead of 760: nothing downstream of the linearisation reads RawResidualJacobian — the library's own resident loop,
        # sdso_ba_optimize, runs this way and materialises them in its closing linearizeAll only).  The headline above keeps the 760-B unit.
        if self.materialize and self.advance and os.environ.get("SDSO_BENCH_SECONDARY", "1") == "1":
            torch = self.torch
            for G in self.groups:
                G.ctx.check(G.ctx.L.sdso_ba_batch_set_materialize(G.ctx.h, 0))
                G.ctx.check(G.ctx.L.sdso_prof_enable(G.ctx.h, 1))
            for _ in range(3):
                self.step()
            self.sync()
            base = self.prof_read("k_ba_lin_fused")
            nst = 20
            t0 = time.perf_counter()
            for _ in range(nst):
                self.step()
            self.sync(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ms, n = self.prof_read("k_ba_lin_fused")
            kavg = (ms - base[0]) / max(n - base[1], 1)
            per_launch = self.units_per_step * nst / max(n - base[1], 1)
            ach = per_launch * 464.0 / (kavg * 1e-3) / 1e9 if kavg > 0 else 0.0
            out["without_jacobian_records"] = {"algorithmic_bytes_per_unit": 464.0, "ms_per_step": dt / nst * 1e3, "value": self.units_per_step * nst / dt,
                                               "kernel_avg_ms": kavg, "achieved_GBps": ach, "frac": ach / 8000.0, "steps": nst,
                                               "note": "single-rank wall clock of %d extra steps after the timed region" % nst}
            for G in self.groups:
                G.ctx.check(G.ctx.L.sdso_ba_batch_set_materialize(G.ctx.h, 1))
                G.ctx.check(G.ctx.L.sdso_prof_enable(G.ctx.h, 0))
        return out

    def cpu_baseline(self, warmup=5, reps=50):
        """SURVEY §8d protocol: the oracle port (-O3 -march=native) on one window, 5 warm-up + 50 timed GN iterations per leg,
        median / p10 / p90; legs = 1 thread, 6 threads (the reference's NUM_THREADS, util/NumType.h:38; IndexThreadReduce work
        stealing with chunks of 50 points and per-thread accumulator copies, EnergyFunctional.cpp:212-269), 16, 64 and all host cores
        (the reference's condition-variable pool stops scaling long before a 256-thread host is full; the legs show where).
        `value` is the fastest leg's median rate, `cores` the threads that leg used."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # cpu_baseline leg only
        abi = self.abi
        # the library is rebuilt with the reference's own flags (-O3 -march=native, CMakeLists.txt:83-84) ON THE HOST THAT TIMES IT when a
        # compiler is there; otherwise the shipped build (explicit -march=x86-64-v3: built in another container) is timed, and says so
        npath, nflags = pyoracle.build_native()
        if npath:
            orc, flags, built = pyoracle.load(path=npath), nflags, "on the timed host"
        else:
            orc, flags, built = pyoracle.load(fast=True), pyoracle.FAST_FLAGS_SHIPPED, "in the build container (%s)" % nflags
        win = self.win
        W, keep = abi.make_ba_window(win, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
        nproc = os.cpu_count() or 1
        allowed = None
        try:
            allowed = sorted(os.sched_getaffinity(0))
            nproc = len(allowed)
        except (AttributeError, OSError):
            pass
        model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        legs = []
        for nt in sorted({1, 6, min(16, nproc), min(64, nproc), nproc}):
            h = orc.orc_ba_create(C.byref(W))
            pinned = orc.orc_ba_set_threads(h, nt)            # the Reducer's workers are bound to distinct cores
            if nt == 1 and allowed:
                try:
                    os.sched_setaffinity(0, {allowed[0]}); pinned = 1     # the single-thread leg runs on the calling thread
                except OSError:
                    pass
            x = np.zeros(68)
            ts = []
            for it in range(warmup + reps):
                t0 = time.perf_counter()
                orc.orc_ba_linearize(h, None)
                orc.orc_ba_apply_res(h)
                orc.orc_ba_solve(h, 0, 1e-5, abi.dp(x), None, None, None, None)
                ts.append(time.perf_counter() - t0)
            orc.orc_ba_destroy(h)
            if nt == 1 and allowed:
                try:
                    os.sched_setaffinity(0, set(allowed))
                except OSError:
                    pass
            t = np.array(ts[warmup:])
            legs.append({"threads": nt, "pinned_workers": int(pinned), "median_ms": float(np.median(t) * 1e3), "p10_ms": float(np.percentile(t, 10) * 1e3),
                         "p90_ms": float(np.percentile(t, 90) * 1e3), "point_residuals_per_s": float(win["nr"] / np.median(t)),
                         "ba_iters_per_s": float(1.0 / np.median(t))})
        best = max(legs, key=lambda l: l["point_residuals_per_s"])
        ref6 = next((l for l in legs if l["threads"] == 6), None)    # the reference's NUM_THREADS (src/util/NumType.h:38)
        return {"value": best["point_residuals_per_s"], "unit": self.unit, "cores": best["threads"], "kind": "port",
                "sample": "%d warm-up + %d timed GN iterations (linearizeAll + applyRes + accumulate A/L/SC + stitch + solve + resubstitute) of one "
                          "%dKF/%d-point window (%d residuals) per leg; median of the fastest leg; workers pinned to distinct cores"
                          % (warmup, reps, win["nf"], win["np"], win["nr"]),
                "flags": flags, "built": built, "reference_num_threads_leg": ref6,
                "cpu_model": model, "nproc": nproc, "legs": legs, "ba_iters_per_s": best["ba_iters_per_s"]}
