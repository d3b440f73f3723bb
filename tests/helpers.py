"""Shared helpers for the parity tests: marshal synthetic problems into the ABI structures."""
import ctypes as C

import numpy as np

from sdso_amd import abi


from sdso_amd.params import track_params  # noqa: E402,F401  (one definition: the product's; the tests use it through this module)


def oracle_track(L, prob, prm, T0, aff0, fn="orc_track_newest_coarse"):
    n = prob["levels"]
    pc = prob["pc"]
    keep = []

    def ptrs(key):
        arr = (abi.c_float_p * n)()
        for l in range(n):
            a = np.ascontiguousarray(pc[l][key], np.float32)
            keep.append(a)
            arr[l] = abi.fp(a)
        return arr

    pcn = (C.c_int * n)(*[len(pc[l]["u"]) for l in range(n)])
    dI = (abi.c_float_p * n)()
    for l in range(n):
        a = np.ascontiguousarray(prob["pyr_new"][l], np.float32)
        keep.append(a)
        dI[l] = abi.fp(a)
    T = abi.SE3.from_Rt(*T0)
    aff = abi.Aff(*aff0)
    out = abi.TrackResult()
    getattr(L, fn)(pcn, ptrs("u"), ptrs("v"), ptrs("idepth"), ptrs("color"), dI, C.byref(prm), C.byref(T), C.byref(aff), C.byref(out))
    return T, aff, out


def oracle_eval(L, pc_l, dI_l, ev, want_mask=True):
    n = len(pc_l["u"])
    H = np.zeros(64); b = np.zeros(8); res = np.zeros(6); nw = C.c_int(0)
    mask = np.zeros(max(n, 1), np.uint8)
    u, v, idp, col = [np.ascontiguousarray(pc_l[k], np.float32) for k in ("u", "v", "idepth", "color")]
    img = np.ascontiguousarray(dI_l, np.float32)
    L.orc_track_calc_res_gs(n, abi.fp(u), abi.fp(v), abi.fp(idp), abi.fp(col), abi.fp(img), C.byref(ev), abi.dp(H), abi.dp(b),
                            abi.dp(res), C.byref(nw), abi.bp(mask), None, 0)
    return H.reshape(8, 8), b, res, nw.value, mask[:n]


def permuted_window(win, seed):
    """The same BA window with the points shuffled inside every host group (residuals follow their
    points).  Mathematically identical problem; only the order of the float sums changes — used to
    measure the reference arithmetic's own sensitivity to summation order."""
    rs = np.random.RandomState(seed)
    order = np.concatenate([rs.permutation(np.nonzero(win["host"] == h)[0]) for h in range(win["nf"])])
    w2 = dict(win)
    for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "host", "hasDepthPrior"):
        w2[k] = win[k][order]
    rp, rt = [], []
    starts = np.searchsorted(win["res_point"], np.arange(win["np"]), side="left")
    ends = np.searchsorted(win["res_point"], np.arange(win["np"]), side="right")
    for p_new, p_old in enumerate(order):
        rp += [p_new] * int(ends[p_old] - starts[p_old])
        rt += list(win["res_target"][starts[p_old]:ends[p_old]])
    w2["res_point"] = np.array(rp, np.int32)
    w2["res_target"] = np.array(rt, np.int32)
    return w2, order


def idepths_close(ig, io, bulk, worst=None):
    """Idepths after a whole GN loop against another implementation of the same loop: the bulk (99th percentile) within `bulk`; the
    maximum sits on single weakly observed points whose trajectory over the iterations amplifies any rounding difference (either side:
    the CPU float path reaches 9e-4 from the f64-accumulator truth on such points, profiles/r04_truth_spread.txt) and is only bounded
    (default 10 x bulk)."""
    d = np.abs(np.asarray(ig, np.float64) - np.asarray(io, np.float64))
    worst = 10.0 * bulk if worst is None else worst
    return bool(np.percentile(d, 99) <= bulk and d.max() <= worst)


def counts_close(a, b, nr):
    """resInA-like counts of two runs of the same loop: equal up to the residuals that sit on an outlier threshold in some iteration"""
    return abs(int(a) - int(b)) <= max(2, nr // 2000)


def drop_residuals(win, seed, drop_frac=0.25):
    """The window after EnergyFunctional::dropResidual removed a random part of every point's residuals
    (src/OptimizationBackend/EnergyFunctional.cpp:524-533: the LAST entry of residualsAll is swapped into the freed slot), the
    drops applied in activeResiduals order like FullSystem::linearizeAll does (FullSystemOptimize.cpp:176-195).  What is left of
    a point's list is no longer in target order — the shape every live window has.  Returns (window, kept original indices)."""
    rs = np.random.RandomState(seed)
    starts = np.searchsorted(win["res_point"], np.arange(win["np"]), side="left")
    ends = np.searchsorted(win["res_point"], np.arange(win["np"]), side="right")
    kept = []
    for p in range(win["np"]):
        ids = list(range(int(starts[p]), int(ends[p])))
        if len(ids) > 1:
            drop = [i for i in ids if rs.rand() < drop_frac]
            if len(drop) == len(ids):
                drop = drop[1:]
            lst = list(ids)
            for i in drop:                       # toRemove order = the point's original residual order
                k = lst.index(i)
                lst[k] = lst[-1]
                lst.pop()
            ids = lst
        kept += ids
    kept = np.array(kept, np.int64)
    w2 = dict(win)
    for k in ("res_point", "res_target", "res_state"):
        w2[k] = np.ascontiguousarray(win[k][kept])
    w2["nr"] = len(kept)
    return w2, kept


def drop_frame(win, idx):
    """The window after keyframe `idx` left it (EnergyFunctional::marginalizeFrame step 4, EnergyFunctional.cpp:631-641, once
    its points are gone): the frame, the points it hosted and every residual into it are removed, later frames move up.
    HM / bM are left to the caller (they change dimension)."""
    nf = win["nf"]
    keep_f = [f for f in range(nf) if f != idx]
    remap = {f: k for k, f in enumerate(keep_f)}
    keep_p = np.nonzero(win["host"] != idx)[0]
    pmap = -np.ones(win["np"], np.int64)
    pmap[keep_p] = np.arange(len(keep_p))
    keep_r = np.nonzero((pmap[win["res_point"]] >= 0) & (win["res_target"] != idx))[0]
    w2 = dict(win)
    w2["nf"] = nf - 1
    for k in ("evalPT", "state", "state_zero", "ab_exposure", "frameEnergyTH", "frameID"):
        w2[k] = np.ascontiguousarray(np.asarray(win[k])[keep_f])
    w2["pyrs"] = [win["pyrs"][f] for f in keep_f]
    for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "hasDepthPrior"):
        w2[k] = np.ascontiguousarray(win[k][keep_p])
    w2["host"] = np.array([remap[int(h)] for h in win["host"][keep_p]], np.int32)
    w2["res_point"] = pmap[win["res_point"][keep_r]].astype(np.int32)
    w2["res_target"] = np.array([remap[int(t)] for t in win["res_target"][keep_r]], np.int32)
    w2["res_state"] = np.ascontiguousarray(win["res_state"][keep_r])
    w2["np"], w2["nr"] = len(keep_p), len(keep_r)
    for k in ("numGoodResiduals", "maxRelBaseline", "res_isNew"):
        w2.pop(k, None)
    n2 = 8 * (nf - 1) + 4
    w2["HM"] = np.zeros((n2, n2)); w2["bM"] = np.zeros(n2)
    return w2


def apply_drops(res_lists, to_remove_ids):
    """dropResidual / deleteOut on per-point id lists (swap-with-last), in the given order."""
    out = [list(l) for l in res_lists]
    where = {i: p for p, l in enumerate(out) for i in l}
    for i in to_remove_ids:
        l = out[where[i]]
        k = l.index(i)
        l[k] = l[-1]
        l.pop()
    return out


def smoke_ba(ctx, orc):
    """One GN iteration of the windowed BA (linearize, applyRes, accumulate, solve) on a small window, vs the oracle."""
    import synth
    win = synth.ba_window(w=320, h=240, nf=4, pts_per_kf=60, seed=3031)
    nf, nr, n = win["nf"], win["nr"], 8 * win["nf"] + 4
    for f in range(nf):
        ctx.upload_pyramid(40 + f, win["pyrs"][f][:1])
    W, keep = abi.make_ba_window(win, frame_slots=[40 + f for f in range(nf)], dI_list=[p[0] for p in win["pyrs"]])
    ctx.check(ctx.L.sdso_ba_upload_window(ctx.h, 3, C.byref(W)))
    h = orc.orc_ba_create(C.byref(W))
    eo, eg = C.c_double(0), C.c_double(0)
    orc.orc_ba_linearize(h, C.byref(eo))
    ctx.check(ctx.L.sdso_ba_linearize(ctx.h, 3, C.byref(eg)))
    Jo, Jg = np.zeros((nr, 74), np.float32), np.zeros((nr, 74), np.float32)
    so, sg = np.zeros(nr, np.uint8), np.zeros(nr, np.uint8)
    orc.orc_ba_get_linearization(h, abi.fp(Jo), abi.bp(so), None, None, None, None)
    ctx.check(ctx.L.sdso_ba_get_linearization(ctx.h, 3, abi.fp(Jg), abi.bp(sg), None, None, None, None))
    assert np.array_equal(so, sg) and np.array_equal(Jo[so != 1], Jg[so != 1])
    orc.orc_ba_apply_res(h)
    ctx.check(ctx.L.sdso_ba_apply_res(ctx.h, 3))
    orc.orc_ba_accumulate(h)
    ctx.check(ctx.L.sdso_ba_accumulate(ctx.h, 3))
    xo, Ho, xg = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    orc.orc_ba_solve(h, 0, 0.1, abi.dp(xo), abi.dp(Ho), None, None, None)
    ctx.check(ctx.L.sdso_ba_solve(ctx.h, 3, 0, 0.1, abi.dp(xg), None, None, None, None))
    d = np.sqrt(np.abs(np.diag(Ho))) + 1e-30
    err = np.abs((xg - xo) * d).max() / max(1.0, np.abs(xo * d).max())
    assert err <= 2e-4, err
    orc.orc_ba_destroy(h)
    ctx.check(ctx.L.sdso_ba_release_window(ctx.h, 3))
    print("smoke BA ok: residuals", nr, "J bit-exact, whitened |dx|", err)


def smoke_stereo(ctx, orc):
    """ImmaturePoint ctor + traceStereo on a small stereo pair, bit-exact vs the oracle."""
    import synth
    pr = synth.stereo_problem(w=320, h=240, npts=400, seed=4031)
    left = np.ascontiguousarray(pr["pyr_l"][0]); right = np.ascontiguousarray(pr["pyr_r"][0])
    ctx.upload_pyramid(80, [left]); ctx.upload_pyramid(81, [right])
    n = len(pr["u"])
    co, wo, go, eo = np.zeros((n, 8), np.float32), np.zeros((n, 8), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    cg, wg, gg, eg = np.zeros_like(co), np.zeros_like(wo), np.zeros_like(go), np.zeros_like(eo)
    orc.orc_immature_init_batch(abi.fp(left), pr["w"], pr["h"], n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(co), abi.fp(wo), abi.fp(go), abi.fp(eo))
    ctx.check(ctx.L.sdso_immature_init_batch(ctx.h, 80, n, abi.fp(pr["u"]), abi.fp(pr["v"]), abi.fp(cg), abi.fp(wg), abi.fp(gg), abi.fp(eg)))
    assert np.array_equal(co, cg) and np.array_equal(eo, eg)
    K = np.array(pr["K"], np.float32); bl = float(pr["calib"]["baseline"])
    Po, do = abi.make_trace_points(n, pr["u"], pr["v"], co, wo, go, eo)
    Pg, dg = abi.make_trace_points(n, pr["u"], pr["v"], co, wo, go, eo)
    so, sg = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    orc.orc_trace_stereo_batch(abi.fp(right), pr["w"], pr["h"], abi.fp(K), bl, 1, C.byref(Po), abi.bp(so))
    ctx.check(ctx.L.sdso_trace_stereo_batch(ctx.h, 81, abi.fp(K), bl, 1, C.byref(Pg), abi.bp(sg)))
    assert np.array_equal(so, sg)
    for k in ("idepth_min_stereo", "idepth_max_stereo", "idepth_stereo", "lastTraceUV"):
        assert np.array_equal(do[k], dg[k], equal_nan=True), k
    print("smoke stereo ok: good", int((sg == 0).sum()), "of", n)


def gen_windows(specs):
    """synth.ba_window(**spec) for every spec, rendered on a few host threads (numpy releases the GIL; every window has its own
    RandomState, so the arrays are those of one call after the other)"""
    from concurrent.futures import ThreadPoolExecutor
    import synth
    specs = list(specs)
    if len(specs) <= 1:
        return [synth.ba_window(**s) for s in specs]
    with ThreadPoolExecutor(max_workers=min(4, len(specs))) as ex:
        return list(ex.map(lambda s: synth.ba_window(**s), specs))
