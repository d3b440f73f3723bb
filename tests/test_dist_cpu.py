"""N>1 path on CPU: two gloo ranks shard the points of one BA window, each accumulates its shard
(oracle arithmetic, CPU), ONE all-reduce(sum) of the packed accumulator block follows, and the
result must equal the unsharded accumulation — the same partition / packing / collective the
GPU path uses with RCCL (bench.py --gpus N, tests/test_dist_gpu.py)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, mode):
    import torch
    import torch.distributed as dist
    sys.path[:0] = [os.path.join(ROOT, "stereo-dso-g2o_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import pyoracle
    from sdso_amd import abi, dist as sdist
    import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = pyoracle.load()
    win = synth.ba_window(w=320, h=240, nf=4, pts_per_kf=60, seed=3021)
    sub, pidx, res_idx = sdist.shard_window(win, rank, world, mode)
    W, keep = abi.make_ba_window(sub, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
    h = L.orc_ba_create(C.byref(W))
    L.orc_ba_linearize(h, None)
    L.orc_ba_apply_res(h)
    L.orc_ba_accumulate(h)
    na = abi.accum_floats(win["nf"])
    acc = np.zeros(na, np.float32)
    L.orc_ba_get_accumulators(h, abi.fp(acc))
    t = sdist.allreduce_accumulators(acc)      # ONE collective per iteration
    # per-point quantities stay rank-local
    hdi = np.zeros(sub["np"], np.float32)
    L.orc_ba_get_point_terms(h, abi.fp(hdi), None, None, None, None)
    np.save(os.path.join(outdir, "acc_%d.npy" % rank), t.numpy())
    np.save(os.path.join(outdir, "hdi_%d.npy" % rank), hdi)
    np.save(os.path.join(outdir, "pidx_%d.npy" % rank), pidx)
    np.save(os.path.join(outdir, "nr_%d.npy" % rank), np.array([sub["nr"]]))
    L.orc_ba_destroy(h)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "per_host"), (3, "per_host"), (2, "contiguous"), (4, "per_host"), (8, "per_host")])
def test_sharded_accumulation_equals_unsharded(oracle, tmp_path, world, mode):
    sys.path[:0] = [os.path.join(ROOT, "stereo-dso-g2o_amd")]
    from sdso_amd import abi, dist as sdist
    import synth
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    win = synth.ba_window(w=320, h=240, nf=4, pts_per_kf=60, seed=3021)
    W, keep = abi.make_ba_window(win, frame_slots=list(range(win["nf"])), dI_list=[p[0] for p in win["pyrs"]])
    h = oracle.orc_ba_create(C.byref(W))
    oracle.orc_ba_linearize(h, None)
    oracle.orc_ba_apply_res(h)
    oracle.orc_ba_accumulate(h)
    na = abi.accum_floats(win["nf"])
    full = np.zeros(na, np.float32)
    oracle.orc_ba_get_accumulators(h, abi.fp(full))
    hdi_full = np.zeros(win["np"], np.float32)
    oracle.orc_ba_get_point_terms(h, abi.fp(hdi_full), None, None, None, None)
    oracle.orc_ba_destroy(h)
    accs = [np.load(tmp_path / ("acc_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(accs[0], accs[r])                 # every rank holds the same reduced block
    scale = np.abs(full).max()
    assert np.abs(accs[0] - full).max() <= 2e-6 * scale        # float sums in a different association
    assert accs[0][-2] == full[-2]                             # nresA: exact count
    pidx = [np.load(tmp_path / ("pidx_%d.npy" % r)) for r in range(world)]
    allp = np.sort(np.concatenate(pidx))
    assert np.array_equal(allp, np.arange(win["np"]))          # a partition of allPoints
    assert sum(int(np.load(tmp_path / ("nr_%d.npy" % r))[0]) for r in range(world)) == win["nr"]
    for r in range(world):
        assert np.all(np.diff(pidx[r]) > 0)                    # allPoints order inside a rank
        assert np.array_equal(np.load(tmp_path / ("hdi_%d.npy" % r)), hdi_full[pidx[r]])   # bit-exact, rank-local
        if mode == "per_host":                                 # every rank holds its share of EVERY host: balanced Schur workgroups
            counts = np.bincount(win["host"][pidx[r]], minlength=win["nf"])
            full_counts = np.bincount(win["host"], minlength=win["nf"])
            assert np.all(np.abs(counts - full_counts / world) <= 1)
    assert sdist.shard_ranges(10, 3) == [(0, 3), (3, 6), (6, 10)]



def _scatter_worker(rank, world, port, outdir, nwin):
    import torch
    import torch.distributed as dist
    sys.path[:0] = [os.path.join(ROOT, "stereo-dso-g2o_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import pyoracle
    from sdso_amd import abi, dist as sdist
    import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = pyoracle.load()
    nf = 4
    na = abi.accum_floats(nf)
    blocks = np.zeros((nwin, na), np.float32)
    for k in range(nwin):
        win = synth.ba_window(w=320, h=240, nf=nf, pts_per_kf=40, seed=3100 + k)
        sub = sdist.shard_window(win, rank, world)[0]
        W, keep = abi.make_ba_window(sub, frame_slots=list(range(nf)), dI_list=[p[0] for p in win["pyrs"]])
        h = L.orc_ba_create(C.byref(W))
        L.orc_ba_linearize(h, None)
        L.orc_ba_apply_res(h)
        L.orc_ba_accumulate(h)
        L.orc_ba_get_accumulators(h, abi.fp(blocks[k]))
        L.orc_ba_destroy(h)
    # reference shape: all-reduce, every rank looks at every window
    full = sdist.allreduce_accumulators(blocks.copy()).numpy()
    # the other shape: reduce-scatter by window, a per-window record from the owner, all-gather
    own = sdist.reduce_scatter_windows(blocks.copy(), rank, world).numpy()
    first, last = sdist.window_owner_range(nwin, rank, world)
    # stand-in for the solve: any deterministic function of a window's summed block (here: f64 sum, the exact residual count, 6 entries)
    record = lambda b: np.concatenate([[np.float32(b.astype(np.float64).sum()), b[-2]], b[:6]]).astype(np.float32)
    recs = sdist.allgather_window_records(np.stack([record(b) for b in own]), world).numpy()
    np.savez(os.path.join(outdir, "sc_%d.npz" % rank), own=own, full=full, recs=recs, want=np.stack([record(b) for b in full]), first=first, last=last)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nwin", [(2, 4), (3, 3)])
def test_reduce_scatter_by_window_then_allgather_equals_allreduce(tmp_path, world, nwin):
    """The data flow of the library's second exchange shape (include/sdso_abi.h: sdso_ba_batch_exchange_mode 1; csrc/comm.hip
    reduce_scatter_block + ba.hip opt_solve_step) on gloo: rank r ends with the summed blocks of windows [r*nwin/N, (r+1)*nwin/N) — the
    same bits the all-reduce leaves there — and after the all-gather every rank holds one record per window, in batch order, equal to
    what it would have computed from the all-reduced block itself."""
    mp.spawn(_scatter_worker, args=(world, _free_port(), str(tmp_path), nwin), nprocs=world, join=True)
    outs = [np.load(tmp_path / ("sc_%d.npz" % r)) for r in range(world)]
    covered = []
    for r, o in enumerate(outs):
        first, last = int(o["first"]), int(o["last"])
        covered += list(range(first, last))
        assert np.array_equal(o["own"], o["full"][first:last])          # the owner's sums are the all-reduce's
        assert np.array_equal(o["recs"], o["want"])                     # every rank, every window, batch order
        assert np.array_equal(o["recs"], outs[0]["recs"])
        assert o["full"][:, -2].min() > 0                               # (windows with residuals: the count entry is live)
    assert covered == list(range(nwin))                                 # every window has exactly one owner
