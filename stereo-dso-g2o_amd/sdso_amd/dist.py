"""Point sharding of a BA window across ranks.

Every accumulator of the windowed BA is a plain sum over points (AccumulatedTopHessian.cpp:131-156,
AccumulatedSCHessian.cpp:75-101) and the reference itself sums per-thread partial copies before the
stitch (AccumulatedTopHessian.cpp:299-308).  So a window is partitioned into contiguous allPoints
ranges, one per rank; a point's residuals stay with the point (Hdd/bd/Hcd/HdiF remain rank-local),
every rank holds all keyframe pyramids and frame states, and ONE all-reduce(sum) of the packed
accumulator block per Gauss-Newton iteration makes every rank stitch and solve the same system.
"""
import numpy as np

POINT_KEYS = ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior")


def shard_ranges(npts, world):
    """Contiguous, balanced [first, last) ranges over allPoints."""
    edges = [(npts * r) // world for r in range(world + 1)]
    return [(edges[r], edges[r + 1]) for r in range(world)]


def shard_window(win, rank, world):
    """The sub-window of `rank`: its points, their residuals, and the complete frame set."""
    first, last = shard_ranges(win["np"], world)[rank]
    sub = dict(win)
    for k in POINT_KEYS:
        sub[k] = np.ascontiguousarray(win[k][first:last])
    sel = (win["res_point"] >= first) & (win["res_point"] < last)
    sub["res_point"] = (win["res_point"][sel] - first).astype(np.int32)
    sub["res_target"] = np.ascontiguousarray(win["res_target"][sel])
    sub["res_state"] = np.ascontiguousarray(win["res_state"][sel])
    sub["np"] = int(last - first)
    sub["nr"] = int(sel.sum())
    return sub, (first, last), np.nonzero(sel)[0]


def allreduce_accumulators(packed, group=None):
    """Sum the packed accumulator block over ranks (torch.distributed; nccl == RCCL on ROCm, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    t = packed if isinstance(packed, torch.Tensor) else torch.from_numpy(packed)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
