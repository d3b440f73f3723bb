#!/usr/bin/env python3
"""bench.py — throughput of the Stereo-DSO photometric-alignment hot path on MI355X.

A "step" is ONE pass of the hot path over ONE batch of synthetic KITTI-shaped input that is
already resident in HBM:
  --workload ba      (default) one windowed-BA Gauss-Newton iteration (linearize + accumulate A/L/SC
                     + stitch + solve + resubstitute) for a batch of independent 8-keyframe windows;
                     unit = point-residual (8-pixel patch, one (point,target) pair)   [BASELINE configs[2]/[4]]
  --workload tracker one fused calcRes+calcGSSSE evaluation at every pyramid level for a batch of
                     independent tracking problems; unit = template point               [BASELINE configs[1]]
  --workload trace   ImmaturePoint::traceStereo over a batch of stereo pairs; unit = point [configs[3]]

Contract (one JSON line on rank 0): see the task statement; plus "roofline" (dominant kernel,
HIP-event timed inside libsdso_hip.so on the stream the kernel runs on) and "cpu_baseline"
(the oracle port, -O3 -march=native, timed on a bounded sample on this box's host cores).
Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL).  BA shards the points of
every window across ranks and all-reduces the packed accumulators once per iteration (weak scaling:
per-GPU work fixed, windows scale with N); tracker / trace are replicas (no collective).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stereo-dso-g2o_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))     # synth.py: the synthetic-input generators (test / bench infrastructure)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
COPY_RATE_BPS = 4.84e12  # read + write rate of a plain device copy on this part (profiles/r02_copy_bw.json)
# PMC passes of this same command (tools/profile_round.sh + tools/make_traffic.py); the latest round's file
TRAFFIC_FILE = (sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json"))) or [os.path.join(ROOT, "profiles", "r01_traffic.json")])[-1]
TRAFFIC_KEYS = ("workload", "windows_per_step", "keyframes", "points_per_window_per_gpu", "residuals_per_window_per_gpu", "jacobians_materialized",
                "stream_groups", "state_advances", "points", "problems", "levels")


def pmc_traffic(workload, config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes; None when the passes were
    taken on another configuration (a counter pass cannot run inside the timed region)."""
    try:
        rec = json.load(open(TRAFFIC_FILE)).get(workload)
    except (OSError, ValueError):
        return None
    rc = (rec or {}).get("config") or {}
    if not rec or any(rc.get(k) != config.get(k) for k in TRAFFIC_KEYS if k in config or k in rc):
        return None
    return rec["traffic_bytes_per_launch"]


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------
class TrackerWorkload:
    """configs[1]: CoarseTracker full 5-level pyramid, KITTI 1232x368, ~2k active points."""
    name = "tracker_full_pyramid_kitti1232x368_2kpts"
    kernel = "k_track_eval"
    unit = "point-residuals/s"
    bytes_per_unit = 64.0  # 16 B template point + 4 taps x 12 B

    def __init__(self, ctx, args, rank):
        from sdso_amd import abi
        import synth
        self.ctx, self.abi = ctx, abi
        t0 = time.time()
        prob = synth.tracker_problem(w=1232, h=368, npts=2000, seed=2002 + rank)
        self.prob = prob
        L = prob["levels"]
        nframes = args.batch or 128
        ctx.set_ref(1, prob["pc"])
        base = np.ascontiguousarray(prob["pyr_new"][0][..., 0])
        rs = np.random.RandomState(77 + rank)
        for f in range(nframes):
            img = np.clip(base + rs.uniform(-1.0, 1.0, base.shape).astype(np.float32), 0, 255).astype(np.float32)
            ctx.check(ctx.L.sdso_make_pyramid(ctx.h, 100 + f, 1232, 368, abi.fp(img)))
        from sdso_amd import params
        prm = params.track_params(prob)
        evs, refs, frames = [], [], []
        for f in range(nframes):
            for lvl in range(L):
                xi = np.array([0.02, -0.01, 0.35, 0.004, -0.006, 0.002]) + rs.normal(0, 2e-3, 6)
                T = abi.SE3.from_Rt(*synth.se3_exp(xi))
                ev = abi.TrackEval()
                ctx.L.sdso_track_make_eval(C.byref(prm), lvl, C.byref(T), C.byref(abi.Aff(0.02, 1.0)), 1.0, C.byref(ev))
                evs.append(ev); refs.append(1); frames.append(100 + f)
        self.nprob = len(evs)
        self.evs = (abi.TrackEval * self.nprob)(*evs)
        self.refs = np.array(refs, np.int32)
        self.frames = np.array(frames, np.int32)
        ctx.check(ctx.L.sdso_track_batch_prepare(ctx.h, self.nprob, abi.ip(self.refs), abi.ip(self.frames), self.evs))
        self.units_per_step = nframes * sum(len(p["u"]) for p in prob["pc"])
        self.config = {"workload": self.name, "problems_per_step": self.nprob, "frames": nframes, "levels": L,
                       "points_per_level": [len(p["u"]) for p in prob["pc"]], "parallelism": "replicas"}
        log("tracker setup %.1fs, %d problems, %d points/step" % (time.time() - t0, self.nprob, self.units_per_step))

    def step(self):
        self.ctx.check(self.ctx.L.sdso_track_batch_enqueue(self.ctx.h))

    def verify(self):
        abi = self.abi
        n = self.nprob
        H = np.zeros((n, 64)); res = np.zeros((n, 6)); nw = np.zeros(n, np.int32)
        self.ctx.check(self.ctx.L.sdso_track_batch_fetch(self.ctx.h, abi.dp(H), None, abi.dp(res), abi.ip(nw)))
        assert np.isfinite(H).all() and (nw > 0).all()
        return {"mean_inliers_per_problem": float(nw.mean())}

    def cpu_baseline(self, budget_s=12.0):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # the only place bench.py touches oracle/: the reported CPU baseline
        abi = self.abi
        orc = pyoracle.load(fast=True)
        prob = self.prob
        pts, t0, reps = 0, time.perf_counter(), 0
        H = np.zeros(64); b = np.zeros(8); res = np.zeros(6); nw = C.c_int(0)
        arrs = []
        for lvl in range(prob["levels"]):
            pc = prob["pc"][lvl]
            arrs.append([np.ascontiguousarray(pc[k], np.float32) for k in ("u", "v", "idepth", "color")] + [np.ascontiguousarray(prob["pyr_new"][lvl], np.float32)])
        while time.perf_counter() - t0 < budget_s:
            for lvl in range(prob["levels"]):
                u, v, i, c, img = arrs[lvl]
                orc.orc_track_calc_res_gs(len(u), abi.fp(u), abi.fp(v), abi.fp(i), abi.fp(c), abi.fp(img), C.byref(self.evs[lvl]),
                                          abi.dp(H), abi.dp(b), abi.dp(res), C.byref(nw), None, None, 0)
                pts += len(u)
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": pts / dt, "unit": self.unit, "cores": 1, "kind": "port",
                "sample": "%d full-pyramid evaluations of one problem (%d points) in %.1f s, oracle -O3 -march=native" % (reps, pts, dt)}


class TraceWorkload:
    """configs[3]: ImmaturePoint::traceStereo, ~20k fresh immature points on a KITTI-shaped stereo pair."""
    name = "trace_stereo_kitti1232x368_20kpts"
    kernel = "k_trace_stereo"
    unit = "points/s"
    # 100 B in + 32 B out per point + every touched image pixel once (whole L0 image, 12 B/px)
    bytes_per_unit = 132.0 + 1232 * 368 * 12.0 / 20000.0

    def __init__(self, ctx, args, rank):
        from sdso_amd import abi
        import synth
        self.ctx, self.abi = ctx, abi
        pr = synth.stereo_problem(w=1232, h=368, npts=args.batch or 20000, seed=4001 + rank)
        self.pr = pr
        left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
        ctx.upload_pyramid(80, [left]); ctx.upload_pyramid(81, [right])
        n = len(pr["u"])
        col, wgt, gH, eth = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
        ctx.check(ctx.L.sdso_immature_init_batch(ctx.h, 80, n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(col), abi.fp(wgt), abi.fp(gH), abi.fp(eth)))
        self.P, self.d = abi.make_trace_points(n, pr["u"], pr["v"], col, wgt, gH, eth)
        self.K = np.array(pr["K"], np.float32)
        ctx.check(ctx.L.sdso_trace_stereo_prepare(ctx.h, 81, abi.fp(self.K), float(pr["calib"]["baseline"]), 1, C.byref(self.P)))
        self.units_per_step = n
        self.config = {"workload": self.name, "points": n, "search_steps_per_point": 45, "parallelism": "replicas"}
        self.init = (col, wgt, gH, eth)

    def step(self):
        self.ctx.check(self.ctx.L.sdso_trace_stereo_enqueue(self.ctx.h))

    def verify(self):
        st = np.zeros(self.P.n, np.uint8)
        self.ctx.check(self.ctx.L.sdso_trace_stereo_fetch(self.ctx.h, C.byref(self.P), self.abi.bp(st)))
        # taps of one launch: a point that reaches the discrete search (status GOOD or OUTLIER; OOB / SKIPPED / BADCONDITION leave before
        # it) samples numSteps x 8 pattern pixels x 4 bilinear taps of the 4-byte plane, then <= 3 GN passes x 8 x 4 taps.  Fresh
        # immature points search the full maxPixSearch = 0.027 (w + h) = 43.2 px -> numSteps = 45 (ImmaturePoint.cpp:238-258).
        searched = int(((st == 0) | (st == 2)).sum())
        self.taps_per_step = searched * (45 * 32 + 3 * 32)
        return {"good_fraction": float((st == 0).mean()), "searched_points": searched, "taps_per_launch": self.taps_per_step}

    def cpu_baseline(self, budget_s=6.0):
        """1 thread and 6 threads (the reference traces points serially — 6 = its NUM_THREADS — here the point range is
        cut into 6 contiguous slices traced by 6 host threads, the oracle call releases the GIL)."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle  # cpu_baseline leg only
        from concurrent.futures import ThreadPoolExecutor
        abi, pr = self.abi, self.pr
        orc = pyoracle.load(fast=True)
        right = np.ascontiguousarray(pr["pyr_r"][0])
        n = len(pr["u"])
        legs = []
        for nt in (1, 6):
            cuts = [n * k // nt for k in range(nt + 1)]

            def one(k):
                lo, hi = cuts[k], cuts[k + 1]
                P, d = abi.make_trace_points(hi - lo, pr["u"][lo:hi], pr["v"][lo:hi], *[a[lo:hi] for a in self.init])
                st = np.zeros(hi - lo, np.uint8)
                orc.orc_trace_stereo_batch(abi.fp(right), pr["w"], pr["h"], abi.fp(self.K), float(pr["calib"]["baseline"]), 1, C.byref(P), abi.bp(st))
            t0, pts = time.perf_counter(), 0
            with ThreadPoolExecutor(nt) as ex:
                while time.perf_counter() - t0 < budget_s:
                    list(ex.map(one, range(nt)))
                    pts += n
            dt = time.perf_counter() - t0
            legs.append({"threads": nt, "points_per_s": pts / dt})
        best = max(legs, key=lambda l: l["points_per_s"])
        return {"value": best["points_per_s"], "unit": self.unit, "cores": best["threads"], "kind": "port", "legs": legs,
                "sample": "%.0f s per leg of 20k-point batches, oracle -O3 -march=native" % budget_s}


class MatchWorkload:
    """The L->R->L pattern every caller of traceStereo runs (FullSystem::stereoMatch FullSystem.cpp:581-613, makeCoarseDepthL0
    CoarseTracker.cpp:295-347) through the BOUNDARY call sdso_stereo_match_batch: host point arrays in, statuses / idepths / uvs out —
    ImmaturePoint ctor, trace, ctor at lastTraceUV, trace back, chained on the device.  The step includes the call's H2D / D2H copies
    (it is the boundary's rate, not a kernel's); unit = matched point (two traces, two constructors)."""
    name = "stereo_match_lrl_kitti1232x368"
    kernel = "k_trace_stereo"
    unit = "points/s"
    bytes_per_unit = 2 * (132.0 + 1232 * 368 * 12.0 / 20000.0)

    def __init__(self, ctx, args, rank):
        from sdso_amd import abi
        import synth
        self.ctx, self.abi = ctx, abi
        pr = synth.stereo_problem(w=1232, h=368, npts=args.batch or 20000, seed=4001 + rank)
        ctx.upload_pyramid(80, [np.ascontiguousarray(pr["pyr_l"][0])]); ctx.upload_pyramid(81, [np.ascontiguousarray(pr["pyr_r"][0])])
        n = len(pr["u"])
        self.K = np.array(pr["K"], np.float32); self.bl = float(pr["calib"]["baseline"])
        self.u, self.v = np.ascontiguousarray(pr["u"], np.float32), np.ascontiguousarray(pr["v"], np.float32)
        self.out = dict(status_fwd=np.zeros(n, np.uint8), status_back=np.zeros(n, np.uint8), idepth_stereo=np.zeros(n, np.float32), back_uv=np.zeros((n, 2), np.float32))
        M = abi.StereoMatch()
        M.n = n; M.u = abi.fp(self.u); M.v = abi.fp(self.v)
        for k, a in self.out.items():
            setattr(M, k, abi.bp(a) if a.dtype == np.uint8 else abi.fp(a))
        self.M = M
        self.units_per_step = n
        self.config = {"workload": self.name, "points": n, "includes_host_copies": True, "parallelism": "replicas"}

    def step(self):
        self.ctx.check(self.ctx.L.sdso_stereo_match_batch(self.ctx.h, 80, 81, self.abi.fp(self.K), self.bl, 1, C.byref(self.M)))

    def verify(self):
        sf, sb = self.out["status_fwd"], self.out["status_back"]
        good = sf == 0
        with np.errstate(divide="ignore"):
            ok = good & (sb == 0) & (np.abs(self.u - self.out["back_uv"][:, 0]) < 1) & (1.0 / self.out["idepth_stereo"] > 0) & (1.0 / self.out["idepth_stereo"] < 70)
        assert ok.mean() > 0.3
        ms, n = self.ctx.prof_read(self.kernel)
        assert n > 0, "the match chain's trace launches were not timed: an empty roofline must not be printed next to a valid value"
        return {"forward_good_fraction": float(good.mean()), "accepted_fraction": float(ok.mean()), "trace_launches_per_step": 2}

    def cpu_baseline(self, budget_s=6.0):
        return {"value": None, "unit": self.unit, "cores": 0, "kind": "port", "sample": "not timed for this workload: see --workload trace"}


WORKLOADS = {"tracker": TrackerWorkload, "trace": TraceWorkload}
ALL_WORKLOADS = dict(WORKLOADS, match=MatchWorkload)


def side_run(cls, ctx, args, rank, steps=30, warmup=5):
    import torch

    class A:
        batch = 0
    wl = cls(ctx, A, rank)
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize(); ctx.sync()
    ctx.check(ctx.L.sdso_prof_reset(ctx.h)); ctx.check(ctx.L.sdso_prof_enable(ctx.h, 1))
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize(); ctx.sync()
    dt = time.perf_counter() - t0
    ctx.check(ctx.L.sdso_prof_enable(ctx.h, 0))
    kms, kl = ctx.prof_read(wl.kernel)
    wl.verify()
    avg = kms / max(kl, 1)
    ach = wl.units_per_step * steps / max(kl, 1) * wl.bytes_per_unit / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
    out = {"workload": wl.config["workload"], "value": wl.units_per_step * steps / dt, "unit": wl.unit, "ms_per_step": dt / steps * 1e3,
           "kernel": wl.kernel, "kernel_avg_ms": avg, "roofline_frac_hbm": ach / HBM_PEAK_GBS}
    if getattr(wl, "taps_per_step", 0) and avg > 0:
        # on-chip roofline of the tap gathers (MI355X_MICROARCH.md: L2 ~34.5 TB/s aggregate; LDS ds_read_b32 ~75 TB/s aggregate):
        # 4-byte taps per second against what the cache hierarchy / the LDS array could deliver
        taps = wl.taps_per_step / (avg * 1e-3)
        out["onchip"] = {"taps_per_s": taps, "tap_bytes_GBps": taps * 4 / 1e9, "frac_of_l2_rate": taps * 4 / 34.5e12, "frac_of_lds_b32_rate": taps * 4 / 75e12}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this same command line, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), started from a parent that has not
    imported torch or made a HIP call.  Rank 0's stdout (the ONE JSON line) is relayed; any child that fails fails the run."""
    import socket
    import subprocess
    backend = os.environ.get("SDSO_DIST_BACKEND", "nccl")
    if backend == "nccl":
        # one rank per device: counted without initialising HIP in this process (sysfs render nodes of the KFD topology)
        ndev = _count_gpus()
        if ndev is not None and ndev < n:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible on this node (SDSO_DIST_BACKEND=gloo_lib rehearses "
                             "N ranks on fewer devices through the library's host transport)" % (n, ndev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    # a rank that dies leaves the others in a rendezvous or a collective: end exactly the processes started here
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    rd.join(timeout=10)
    out0 = (buf[0] if buf else b"").decode()
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % ", ".join("rank %d rc %d" % b for b in bad), file=sys.stderr, flush=True)
        return 1
    if not any(l.startswith("{") for l in out0.splitlines()):
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr, flush=True)
        return 1
    return 0


def _count_gpus():
    """GPUs of this node from the KFD topology in sysfs (no HIP call: the launcher must stay free of GPU state)."""
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    try:
        n = 0
        for d in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(d):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        if n == 0:
            return None
        if vis is not None and vis.strip() != "":
            n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
        return n
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("SDSO_BENCH_WORKLOAD", "ba"), choices=["ba", "tracker", "trace", "match"])
    ap.add_argument("--batch", type=int, default=0, help="independent problems (frames / windows / pairs) per step and GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default=os.environ.get("SDSO_BENCH_SCALING", "weak"), choices=["weak", "strong"],
                    help="BA at --gpus N: weak = 2000 points per window and rank (default); strong = BASELINE configs[4], 8000 points per window cut N ways")
    args = ap.parse_args()

    # --gpus N is a statement about the run, not a hint: either this process IS one of N ranks (the driver's
    # `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, WORLD_SIZE = N), or it is asked for N ranks and
    # starts them itself — before torch or HIP is touched in this process, which stays a plain relay.  Anything else fails loudly:
    # a line with "n_gpus": 1 must never come out of a command that said --gpus 8.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s — start it as `python bench.py --gpus N` (it launches its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, env_world))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SDSO_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # backend nccl == RCCL over xGMI.  SDSO_DIST_BACKEND=gloo exists only to rehearse the multi-rank path on a box
        # with fewer GPUs than ranks (RCCL refuses two ranks on one device).
        backend = os.environ.get("SDSO_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo" if backend == "gloo_lib" else backend)   # gloo_lib: gloo under the library's host transport

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sdso_amd import abi
    ctx = abi.Context(local_rank)
    wl = ALL_WORKLOADS[args.workload](ctx, args, rank) if args.workload != "ba" else None
    if wl is None:
        from bench_ba import BAWorkload
        wl = BAWorkload(ctx, args, rank, world, device=local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()
        if hasattr(wl, "sync"):
            wl.sync()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    prof_reset = getattr(wl, "prof_reset", lambda: ctx.check(ctx.L.sdso_prof_reset(ctx.h)))
    prof_enable = getattr(wl, "prof_enable", lambda on: ctx.check(ctx.L.sdso_prof_enable(ctx.h, int(on))))
    prof_read = getattr(wl, "prof_read", ctx.prof_read)
    prof_reset()
    prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    t_enq = time.perf_counter() - t0     # host time to enqueue the timed steps (the device runs behind)
    barrier()
    dt = time.perf_counter() - t0
    prof_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        units = torch.tensor([float(wl.units_per_step)], dtype=torch.float64, device="cuda")
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
        units_per_step = float(units.item())
    else:
        units_per_step = float(wl.units_per_step)
    kms, klaunch = prof_read(wl.kernel)
    extra = wl.verify()

    if rank == 0:
        value = units_per_step * args.steps / dt
        per_launch_units = wl.units_per_step * args.steps / max(klaunch, 1)   # a group launch covers its share of the windows
        avg_ms = kms / max(klaunch, 1)
        achieved = per_launch_units * wl.bytes_per_unit / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "point-residuals/sec (8-pix patches) per GN iter; windowed-BA iters/sec, 8KF window",
            "value": value, "unit": wl.unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": getattr(wl, "scaling", "weak"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": wl.config,
            "roofline": {"bound": "hbm", "kernel": wl.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, wl.config),
                         "traffic_source": "replayed from %s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (a counter pass cannot run inside the timed region); null when the configuration differs" % os.path.relpath(TRAFFIC_FILE, ROOT),
                         "kernel_avg_ms": avg_ms, "launches": klaunch, "algorithmic_bytes_per_unit": wl.bytes_per_unit},
            "extra": extra,
        }
        if args.workload in ("trace", "match"):
            # SQ / cache counters of this kernel (profiles/r05_trace_*): taps are served on chip (L1 / L2 hit rates there), HBM sees the
            # image once and the point state; the kernel is bound by instruction issue and the latency chain of its phases
            out["roofline"]["bound_note"] = ("not an HBM-bound kernel at this size: `frac` is its compulsory bytes over the HBM peak; the tap rate against "
                                             "the on-chip rates is in extra.onchip / profiles/r05_trace_cache_summary.txt")
            if getattr(wl, "taps_per_step", 0) and avg_ms > 0:
                taps = wl.taps_per_step / (avg_ms * 1e-3)
                out["extra"]["onchip"] = {"taps_per_s": taps, "tap_bytes_GBps": taps * 4 / 1e9, "frac_of_l2_rate": taps * 4 / 34.5e12, "frac_of_lds_b32_rate": taps * 4 / 75e12}
        tr = out["roofline"]["traffic"]
        if tr and avg_ms > 0:
            # what the counters say about the bound: the kernel's REAL HBM traffic (128-byte lines of sparse 16-byte taps, the records) over
            # its duration, next to the rate a plain device copy reaches on this part (profiles/r02_copy_bw.json: 4.84 TB/s read + write)
            rate = tr / (avg_ms * 1e-3)
            out["roofline"]["traffic_rate_GBps"] = rate / 1e9
            if args.workload in ("ba", "tracker"):
                # HBM-bound kernels only (trace / match keep the note above: their taps are served on chip).  The wording follows the measured rate.
                rel = rate / COPY_RATE_BPS
                note = ("hbm: the launch moves %.2fx its algorithmic bytes (128-byte lines for sparse 16-byte taps) at %.2f TB/s = %.2f of the %.2f TB/s a "
                        "device copy reaches here (%s)" % (tr / (per_launch_units * wl.bytes_per_unit), rate / 1e12, rel, COPY_RATE_BPS / 1e12,
                                                          "at or above the copy rate" if rel >= 1.0 else "below the copy rate"))
                if args.workload == "ba":
                    # The memory system is the bound, through its miss LATENCY under the CUs' miss queues rather than through bytes per second:
                    # SQ counters (profiles/) show the waves waiting on memory, not on issue, and 13 % fewer lines fetched (four shifted tile
                    # grids, best grid per residual) left the kernel 1.5 % slower (profiles/r05_tile_grids_ab.txt)
                    note += ("; what keeps frac from the peak is that overfetch plus the latency of the taps' misses under the CUs' miss queues: SQ counters "
                             "(profiles/) show the waves waiting on memory, not on issue, and fetching 13 % fewer lines did not shorten the kernel "
                             "(profiles/r05_tile_grids_ab.txt)")
                out["roofline"]["bound_note"] = note
        out["extra"]["host_enqueue_ms_per_step"] = t_enq / args.steps * 1e3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline()
            if out["cpu_baseline"].get("value"):
                out["extra"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if args.workload == "ba" and world == 1 and not args.no_cpu_baseline and os.environ.get("SDSO_BENCH_SKIP_OTHERS") != "1":
            # the two other units of BASELINE.json's metric family, measured the same way in short runs (informational; the
            # headline `value` / `roofline` above are the BA workload's)
            out["extra"]["other_workloads"] = {name: side_run(cls, ctx, args, rank) for name, cls in WORKLOADS.items()}
        if args.workload == "ba":
            out["extra"]["ba_window_iters_per_s"] = wl.nwin * args.steps / dt   # sharded ranks work on the SAME windows
        print(json.dumps(out), flush=True)
    if hasattr(wl, "close"):
        wl.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
