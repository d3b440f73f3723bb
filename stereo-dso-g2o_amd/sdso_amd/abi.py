"""ctypes view of include/sdso_abi.h: structure layouts and the loader for libsdso_hip.so.

There is no CPU fallback: load() raises if the HIP library has not been built, and Context()
raises if no MI355X/HIP device is usable.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDSO_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libsdso_hip.so")   # override: A/B experiments only

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_u8_p = C.POINTER(C.c_uint8)


class TrackEval(C.Structure):
    _fields_ = [("lvl", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("Ki", C.c_float * 9), ("RKi", C.c_float * 9), ("t", C.c_float * 3),
                ("affLL", C.c_float * 2), ("ref_b0", C.c_float), ("cutoffTH", C.c_float),
                ("huberTH", C.c_float)]


class SE3(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]

    @staticmethod
    def from_Rt(R, t):
        s = SE3()
        s.R[:] = np.asarray(R, np.float64).reshape(9).tolist()
        s.t[:] = np.asarray(t, np.float64).reshape(3).tolist()
        return s

    def Rt(self):
        return np.array(self.R[:]).reshape(3, 3), np.array(self.t[:])


class Aff(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double)]


class TrackParams(C.Structure):
    _fields_ = [("levels", C.c_int), ("w", C.c_int * 6), ("h", C.c_int * 6),
                ("fx", C.c_float * 6), ("fy", C.c_float * 6), ("cx", C.c_float * 6), ("cy", C.c_float * 6),
                ("ref_exposure", C.c_float), ("new_exposure", C.c_float), ("ref_aff_g2l", Aff),
                ("coarsestLvl", C.c_int), ("minResForAbort", C.c_double * 5),
                ("coarseCutoffTH", C.c_float), ("huberTH", C.c_float), ("maxIterations", C.c_int * 5),
                ("affineOptModeA", C.c_double), ("affineOptModeB", C.c_double)]


class TrackResult(C.Structure):
    _fields_ = [("good", C.c_int), ("lastResiduals", C.c_double * 5), ("lastFlowIndicators", C.c_double * 3),
                ("iterations", C.c_int * 5), ("evaluations", C.c_int), ("point_evals", C.c_longlong)]


class BAWindow(C.Structure):
    _fields_ = [("nf", C.c_int), ("np", C.c_int), ("nr", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("calib_value_scaled", C.c_double * 4), ("calib_value_zero", C.c_double * 4),
                ("evalPT", c_double_p), ("state", c_double_p), ("state_zero", c_double_p),
                ("ab_exposure", c_float_p), ("frameEnergyTH", c_float_p), ("frameID", c_int_p),
                ("frame_slot", c_int_p), ("dI", C.POINTER(c_float_p)),
                ("u", c_float_p), ("v", c_float_p), ("idepth", c_float_p), ("idepth_zero", c_float_p),
                ("color", c_float_p), ("weights", c_float_p), ("host", c_int_p), ("hasDepthPrior", c_u8_p),
                ("res_point", c_int_p), ("res_target", c_int_p), ("res_state", c_u8_p),
                ("HM", c_double_p), ("bM", c_double_p),
                ("solverMode", C.c_int), ("affineOptModeA", C.c_double), ("affineOptModeB", C.c_double),
                ("forceAcceptStep", C.c_int),
                ("maxRelBaseline", c_float_p), ("numGoodResiduals", c_int_p), ("res_isNew", c_u8_p)]


# sdso_comm_init_host's transport callbacks (include/sdso_abi.h)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_size_t)
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t)


class BAOptResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("lastEnergy", C.c_double), ("rmse", C.c_double), ("resInA", C.c_int)]


class BAPostState(C.Structure):
    """sdso_ba_post_state_t: what FullSystem::optimize leaves behind (FullSystemOptimize.cpp:52-87, :142-203, :997-1041)."""
    _fields_ = [("idepth", c_float_p), ("step", c_float_p), ("HdiF", c_float_p), ("bdSumF", c_float_p), ("idepth_hessian", c_float_p),
                ("maxRelBaseline", c_float_p), ("numGoodResiduals", c_int_p),
                ("state_state", c_u8_p), ("isActiveAndIsGoodNEW", c_u8_p), ("state_energy", c_float_p), ("centerProjectedTo", c_float_p),
                ("projectedTo", c_float_p), ("toRemove", c_u8_p),
                ("state", c_double_p), ("state_zero", c_double_p), ("evalPT", c_double_p), ("PRE_worldToCam", c_double_p),
                ("frame_step", c_double_p), ("frameEnergyTH", c_float_p),
                ("calib_value", C.c_double * 4), ("calib_value_scaled", C.c_double * 4), ("calib_step", C.c_double * 4),
                ("lastX", c_double_p), ("lastHS", c_double_p), ("lastbS", c_double_p),
                ("resInA", C.c_int), ("resInL", C.c_int), ("resInM", C.c_int), ("n_toRemove", C.c_int), ("result", BAOptResult)]


def make_post_state(nf, npts, nr, with_system=True):
    """A BAPostState with every array allocated; returns (struct, dict of numpy arrays)."""
    n = 8 * nf + 4
    d = dict(idepth=np.zeros(npts, np.float32), step=np.zeros(npts, np.float32), HdiF=np.zeros(npts, np.float32), bdSumF=np.zeros(npts, np.float32),
             idepth_hessian=np.zeros(npts, np.float32), maxRelBaseline=np.zeros(npts, np.float32), numGoodResiduals=np.zeros(npts, np.int32),
             state_state=np.zeros(nr, np.uint8), isActiveAndIsGoodNEW=np.zeros(nr, np.uint8), state_energy=np.zeros(nr, np.float32),
             centerProjectedTo=np.zeros((nr, 3), np.float32), projectedTo=np.zeros((nr, 16), np.float32), toRemove=np.zeros(nr, np.uint8),
             state=np.zeros((nf, 10)), state_zero=np.zeros((nf, 10)), evalPT=np.zeros((nf, 12)), PRE_worldToCam=np.zeros((nf, 12)),
             frame_step=np.zeros((nf, 10)), frameEnergyTH=np.zeros(nf, np.float32), lastX=np.zeros(n))
    if with_system:
        d["lastHS"] = np.zeros((n, n)); d["lastbS"] = np.zeros(n)
    P = BAPostState()
    for k, a in d.items():
        setattr(P, k, {np.dtype(np.float32): fp, np.dtype(np.float64): dp, np.dtype(np.int32): ip, np.dtype(np.uint8): bp}[a.dtype](a))
    return P, d


class TracePoints(C.Structure):
    _fields_ = [("n", C.c_int), ("u_stereo", c_float_p), ("v_stereo", c_float_p), ("idepth_min", c_float_p),
                ("idepth_min_stereo", c_float_p), ("idepth_max_stereo", c_float_p), ("idepth_stereo", c_float_p),
                ("color", c_float_p), ("weights", c_float_p), ("gradH", c_float_p), ("energyTH", c_float_p),
                ("quality", c_float_p), ("lastTraceStatus", c_u8_p), ("lastTraceUV", c_float_p),
                ("lastTracePixelInterval", c_float_p)]


class TraceGeom(C.Structure):
    _fields_ = [("KRKi", C.c_float * 9), ("Kt", C.c_float * 3), ("aff", C.c_float * 2)]


class Activate(C.Structure):
    _fields_ = [("nf", C.c_int), ("w", C.c_int), ("h", C.c_int), ("n", C.c_int), ("minObs", C.c_int), ("K", C.c_float * 4),
                ("pair_R", c_float_p), ("pair_t", c_float_p), ("pair_aff", c_float_p), ("frame_slot", c_int_p), ("dI", C.POINTER(c_float_p)),
                ("host", c_int_p), ("u", c_float_p), ("v", c_float_p), ("idepth_min", c_float_p), ("idepth_max", c_float_p),
                ("color", c_float_p), ("weights", c_float_p), ("energyTH", c_float_p)]


class StereoMatch(C.Structure):
    _fields_ = [("n", C.c_int), ("u", c_float_p), ("v", c_float_p), ("idepth_min_stereo", c_float_p), ("idepth_max_stereo", c_float_p),
                ("back_idepth_min_stereo", c_float_p), ("back_idepth_max_stereo", c_float_p), ("status_fwd", c_u8_p), ("status_back", c_u8_p),
                ("idepth_stereo", c_float_p), ("idepth_min_out", c_float_p), ("idepth_max_out", c_float_p), ("fwd_uv", c_float_p),
                ("back_uv", c_float_p)]


class G2oTrackEval(C.Structure):
    """sdso_g2o_track_eval_t: one EdgeSE3PosePhotoDSO evaluation point (fork-live tracker, dso_g2o_edge.cpp:395-500)."""
    _fields_ = [("lvl", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("Ki", C.c_float * 9), ("RKi", C.c_float * 9), ("t_cull", C.c_float * 3),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("ab", C.c_float * 2), ("b0", C.c_double),
                ("cutoffTH", C.c_float), ("huberTH", C.c_float)]


class G2oLba(C.Structure):
    """sdso_g2o_lba_t: EdgeLBASE3PosePhotoIdepthCamDSO batch (dso_g2o_edge.cpp:5-282)."""
    _fields_ = [("nf", C.c_int), ("nr", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("frame_slot", c_int_p), ("dI", C.POINTER(c_float_p)),
                ("pair_R", c_float_p), ("pair_t", c_float_p), ("pair_ab", c_float_p), ("host_b0", c_double_p),
                ("frameEnergyTH", c_float_p), ("cam", C.c_double * 4), ("host", c_int_p), ("target", c_int_p),
                ("u", c_float_p), ("v", c_float_p), ("idepth", c_double_p), ("color", c_float_p), ("weights", c_float_p)]


def fp(a):
    return a.ctypes.data_as(c_float_p)


def dp(a):
    return a.ctypes.data_as(c_double_p)


def ip(a):
    return a.ctypes.data_as(c_int_p)


def bp(a):
    return a.ctypes.data_as(c_u8_p)


def accum_floats(nf):
    return nf * nf * 91 * 2 + nf ** 3 * 64 + nf * nf * 32 + nf * nf * 8 + 16 + 4 + 2


def make_ba_window(win, frame_slots=None, dI_list=None):
    """Fill a BAWindow from the dict synth.ba_window() returns.  Keeps references alive in `keep`."""
    W = BAWindow()
    keep = []

    def arr(key, dt):
        a = np.ascontiguousarray(win[key], dtype=dt)
        keep.append(a)
        return a

    W.nf, W.np, W.nr, W.w, W.h = win["nf"], win["np"], win["nr"], win["w"], win["h"]
    W.calib_value_scaled[:] = list(win["calib_value_scaled"])
    W.calib_value_zero[:] = list(win["calib_value_zero"])
    W.evalPT = dp(arr("evalPT", np.float64))
    W.state = dp(arr("state", np.float64))
    W.state_zero = dp(arr("state_zero", np.float64))
    W.ab_exposure = fp(arr("ab_exposure", np.float32))
    W.frameEnergyTH = fp(arr("frameEnergyTH", np.float32))
    W.frameID = ip(arr("frameID", np.int32))
    if frame_slots is not None:
        fs = np.ascontiguousarray(frame_slots, np.int32)
        keep.append(fs)
        W.frame_slot = ip(fs)
    if dI_list is not None:
        ptrs = (c_float_p * len(dI_list))()
        for i, a in enumerate(dI_list):
            a = np.ascontiguousarray(a, np.float32)
            keep.append(a)
            ptrs[i] = fp(a)
        keep.append(ptrs)
        W.dI = C.cast(ptrs, C.POINTER(c_float_p))
    for key in ("u", "v", "idepth", "idepth_zero", "color", "weights"):
        setattr(W, key, fp(arr(key, np.float32)))
    W.host = ip(arr("host", np.int32))
    W.hasDepthPrior = bp(arr("hasDepthPrior", np.uint8))
    W.res_point = ip(arr("res_point", np.int32))
    W.res_target = ip(arr("res_target", np.int32))
    W.res_state = bp(arr("res_state", np.uint8))
    W.HM = dp(arr("HM", np.float64))
    W.bM = dp(arr("bM", np.float64))
    W.solverMode = int(win["solverMode"])
    W.affineOptModeA = float(win["affineOptModeA"])
    W.affineOptModeB = float(win["affineOptModeB"])
    W.forceAcceptStep = int(win["forceAcceptStep"])
    if win.get("maxRelBaseline") is not None:
        W.maxRelBaseline = fp(arr("maxRelBaseline", np.float32))
    if win.get("numGoodResiduals") is not None:
        W.numGoodResiduals = ip(arr("numGoodResiduals", np.int32))
    if win.get("res_isNew") is not None:
        W.res_isNew = bp(arr("res_isNew", np.uint8))
    return W, keep


def make_trace_points(n, u, v, color, weights, gradH, energyTH, idepth_min_stereo=None, idepth_max_stereo=None):
    """Fresh immature points (idepth_min=0, idepth_max=NaN; ImmaturePoint.cpp:34)."""
    d = dict(
        u_stereo=np.ascontiguousarray(u, np.float32).copy(), v_stereo=np.ascontiguousarray(v, np.float32).copy(),
        idepth_min=np.zeros(n, np.float32),
        idepth_min_stereo=(np.zeros(n, np.float32) if idepth_min_stereo is None else np.ascontiguousarray(idepth_min_stereo, np.float32).copy()),
        idepth_max_stereo=(np.full(n, np.nan, np.float32) if idepth_max_stereo is None else np.ascontiguousarray(idepth_max_stereo, np.float32).copy()),
        idepth_stereo=np.zeros(n, np.float32),
        color=np.ascontiguousarray(color, np.float32).copy(), weights=np.ascontiguousarray(weights, np.float32).copy(),
        gradH=np.ascontiguousarray(gradH, np.float32).copy(), energyTH=np.ascontiguousarray(energyTH, np.float32).copy(),
        quality=np.full(n, 10000, np.float32), lastTraceStatus=np.full(n, 5, np.uint8),
        lastTraceUV=np.zeros((n, 2), np.float32), lastTracePixelInterval=np.zeros(n, np.float32))
    P = TracePoints()
    P.n = n
    for k, a in d.items():
        setattr(P, k, bp(a) if a.dtype == np.uint8 else fp(a))
    return P, d


_lib = None


def load():
    """Load libsdso_hip.so (built by `make -C stereo-dso-g2o_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsdso_hip.so is not built (%s): run __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sdso_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.sdso_ctx_destroy.argtypes = [vp]
    L.sdso_ctx_destroy.restype = None
    L.sdso_last_error.argtypes = [vp]
    L.sdso_last_error.restype = C.c_char_p
    L.sdso_ctx_stream.argtypes = [vp]
    L.sdso_ctx_stream.restype = vp
    L.sdso_ctx_sync.argtypes = [vp]
    L.sdso_ctx_partition_cus.argtypes = [vp, C.c_int, C.c_int]
    L.sdso_prof_enable.argtypes = [vp, C.c_int]
    L.sdso_selftest_se3.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    L.sdso_prof_reset.argtypes = [vp]
    L.sdso_prof_read.argtypes = [vp, C.c_char_p, c_double_p, C.POINTER(C.c_long)]
    L.sdso_pyramid_levels.argtypes = [C.c_int, C.c_int]
    L.sdso_upload_pyramid.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_int_p, C.POINTER(c_float_p)]
    L.sdso_make_pyramid.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_float_p]
    L.sdso_set_gamma.argtypes = [vp, c_float_p]
    L.sdso_gamma_from_binv.argtypes = [c_float_p, c_float_p]
    L.sdso_download_pyramid_level.argtypes = [vp, C.c_int, C.c_int, c_float_p]
    L.sdso_download_abs_grad.argtypes = [vp, C.c_int, C.c_int, c_float_p]
    L.sdso_release_pyramid.argtypes = [vp, C.c_int]
    L.sdso_track_set_ref.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]
    L.sdso_track_release_ref.argtypes = [vp, C.c_int]
    L.sdso_track_make_eval.argtypes = [C.POINTER(TrackParams), C.c_int, C.POINTER(SE3), C.POINTER(Aff), C.c_float, C.POINTER(TrackEval)]
    L.sdso_track_calc_res_gs.argtypes = [vp, C.c_int, C.c_int, C.POINTER(TrackEval), c_double_p, c_double_p, c_double_p, c_int_p, c_u8_p]
    L.sdso_track_calc_res_gs_batch.argtypes = [vp, C.c_int, c_int_p, c_int_p, C.POINTER(TrackEval), c_double_p, c_double_p, c_double_p, c_int_p]
    L.sdso_track_batch_prepare.argtypes = [vp, C.c_int, c_int_p, c_int_p, C.POINTER(TrackEval)]
    L.sdso_track_batch_enqueue.argtypes = [vp]
    L.sdso_track_batch_fetch.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_int_p]
    L.sdso_track_newest_coarse.argtypes = [vp, C.c_int, C.c_int, C.POINTER(TrackParams), C.POINTER(SE3), C.POINTER(Aff), C.POINTER(TrackResult)]
    L.sdso_ba_upload_window.argtypes = [vp, C.c_int, C.POINTER(BAWindow)]
    L.sdso_ba_release_window.argtypes = [vp, C.c_int]
    L.sdso_ba_linearize.argtypes = [vp, C.c_int, c_double_p]
    L.sdso_ba_get_linearization.argtypes = [vp, C.c_int, c_float_p, c_u8_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.sdso_ba_apply_res.argtypes = [vp, C.c_int]
    L.sdso_ba_get_residual_state.argtypes = [vp, C.c_int, c_u8_p, c_u8_p, c_float_p]
    L.sdso_ba_get_ef_jacobians.argtypes = [vp, C.c_int, c_float_p]
    L.sdso_ba_accumulate.argtypes = [vp, C.c_int]
    L.sdso_ba_accum_floats.argtypes = [C.c_int]
    L.sdso_ba_accum_dev.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.sdso_ba_get_accumulators.argtypes = [vp, C.c_int, c_float_p]
    L.sdso_ba_set_accumulators.argtypes = [vp, C.c_int, c_float_p]
    L.sdso_ba_get_point_terms.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.sdso_ba_solve.argtypes = [vp, C.c_int, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.sdso_ba_get_point_steps.argtypes = [vp, C.c_int, c_float_p]
    L.sdso_ba_resubstitute.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_double_p]
    L.sdso_ba_optimize.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_float_p, c_u8_p, C.POINTER(BAOptResult)]
    L.sdso_ba_marginalize_points.argtypes = [vp, C.c_int, c_u8_p, c_double_p, c_double_p]
    L.sdso_ba_get_tables.argtypes = [vp, C.c_int, c_float_p, c_double_p, c_double_p, c_float_p]
    L.sdso_ba_keep_projections.argtypes = [vp, C.c_int, C.c_int]
    L.sdso_ba_batch_create.argtypes = [vp, C.c_int, c_int_p]
    L.sdso_ba_batch_accumulate.argtypes = [vp]
    L.sdso_ba_batch_linearize.argtypes = [vp]
    L.sdso_ba_batch_schur.argtypes = [vp]
    L.sdso_ba_batch_set_materialize.argtypes = [vp, C.c_int]
    L.sdso_ba_batch_solve.argtypes = [vp, C.c_double, C.c_int]
    L.sdso_ba_batch_accum_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_long)]
    L.sdso_ba_batch_get_x.argtypes = [vp, c_double_p]
    L.sdso_ba_batch_optimize.argtypes = [vp, C.c_int, C.POINTER(BAOptResult)]
    L.sdso_ba_batch_optimize_begin.argtypes = [vp, C.c_int]
    L.sdso_ba_batch_step.argtypes = [vp]
    L.sdso_ba_batch_solve_step.argtypes = [vp, C.c_double, C.c_int]
    L.sdso_ba_batch_optimize_end.argtypes = [vp, C.POINTER(BAOptResult)]
    L.sdso_ba_get_state.argtypes = [vp, C.c_int, c_double_p, c_float_p, c_u8_p]
    L.sdso_ba_get_post_state.argtypes = [vp, C.c_int, C.POINTER(BAPostState)]
    L.sdso_ba_batch_keep_system.argtypes = [vp, C.c_int]
    L.sdso_ba_get_stitched.argtypes = [vp, C.c_int] + [c_double_p] * 6
    L.sdso_ba_batch_exchange_mode.argtypes = [vp, C.c_int]
    L.sdso_ba_get_counts.argtypes = [vp, C.c_int, c_int_p, c_int_p, c_int_p]
    L.sdso_ba_calc_energies.argtypes = [vp, C.c_int, c_double_p, c_double_p]
    L.sdso_ba_marginalize_frame_dev.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_double_p]
    L.sdso_ba_adopt_prior.argtypes = [vp, C.c_int, C.c_int]
    L.sdso_ba_get_deltas.argtypes = [vp, C.c_int, c_float_p, c_double_p, c_double_p, c_float_p]
    L.sdso_comm_unique_id.argtypes = [vp]
    L.sdso_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.sdso_comm_attach.argtypes = [vp, vp]
    L.sdso_comm_init_host.argtypes = [vp, C.c_int, C.c_int, HOST_ALLREDUCE_FN, HOST_ALLGATHER_FN, vp]
    L.sdso_comm_info.argtypes = [vp, c_int_p, c_int_p]
    L.sdso_comm_destroy.argtypes = [vp]
    L.sdso_ba_allreduce.argtypes = [vp]
    L.sdso_ba_allreduce_window.argtypes = [vp, C.c_int]
    L.sdso_immature_init_batch.argtypes = [vp, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.sdso_trace_stereo_batch.argtypes = [vp, C.c_int, c_float_p, C.c_float, C.c_int, C.POINTER(TracePoints), c_u8_p]
    L.sdso_trace_stereo_prepare.argtypes = [vp, C.c_int, c_float_p, C.c_float, C.c_int, C.POINTER(TracePoints)]
    L.sdso_trace_stereo_enqueue.argtypes = [vp]
    L.sdso_trace_stereo_fetch.argtypes = [vp, C.POINTER(TracePoints), c_u8_p]
    L.sdso_track_newest_coarse_batch.argtypes = [vp, C.c_int, c_int_p, c_int_p, C.POINTER(TrackParams), C.POINTER(SE3), C.POINTER(Aff), C.POINTER(TrackResult)]
    L.sdso_track_make_ref.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_float_p, c_float_p, c_int_p]
    L.sdso_track_get_ref.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.sdso_trace_on_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(TraceGeom), c_int_p, C.POINTER(TracePoints), c_u8_p]
    L.sdso_pixel_select.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_float, c_int_p, c_float_p, c_int_p]
    L.sdso_pixel_selector_pattern.argtypes = [C.c_int, c_u8_p]
    L.sdso_ba_marginalize_frame.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.sdso_activate_points_batch.argtypes = [vp, C.POINTER(Activate), C.POINTER(C.c_int8), c_float_p, c_u8_p]
    L.sdso_stereo_match_batch.argtypes = [vp, C.c_int, C.c_int, c_float_p, C.c_float, C.c_int, C.POINTER(StereoMatch)]
    L.sdso_g2o_track_add_edges.argtypes = [vp, C.c_int, C.c_int, C.POINTER(G2oTrackEval), c_double_p, c_int_p, c_u8_p, c_float_p]
    L.sdso_g2o_track_linearize.argtypes = [vp, C.c_int, C.c_int, C.POINTER(G2oTrackEval), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.sdso_g2o_track_newest_coarse.argtypes = [vp, C.c_int, C.c_int, C.POINTER(TrackParams), C.POINTER(SE3), C.POINTER(Aff), C.POINTER(TrackResult)]
    L.sdso_g2o_lba_eval.argtypes = [vp, C.POINTER(G2oLba), c_double_p, c_double_p, c_u8_p, c_float_p, c_float_p, c_float_p, c_u8_p]
    L.sdso_trace_set_gn_mode.argtypes = [vp, C.c_int]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "sdso_ctx_create", "sdso_ctx_destroy", "sdso_last_error", "sdso_ctx_stream", "sdso_ctx_sync", "sdso_ctx_partition_cus",
    "sdso_prof_enable", "sdso_prof_reset", "sdso_prof_read", "sdso_selftest_se3",
    "sdso_pyramid_levels", "sdso_upload_pyramid", "sdso_make_pyramid", "sdso_set_gamma", "sdso_gamma_from_binv", "sdso_download_pyramid_level", "sdso_download_abs_grad",
    "sdso_release_pyramid", "sdso_track_set_ref", "sdso_track_release_ref", "sdso_track_make_eval",
    "sdso_track_calc_res_gs", "sdso_track_calc_res_gs_batch", "sdso_track_batch_prepare",
    "sdso_track_batch_enqueue", "sdso_track_batch_fetch", "sdso_track_newest_coarse",
    "sdso_ba_upload_window", "sdso_ba_release_window", "sdso_ba_linearize", "sdso_ba_get_linearization",
    "sdso_ba_apply_res", "sdso_ba_get_residual_state", "sdso_ba_get_ef_jacobians", "sdso_ba_accumulate", "sdso_ba_accum_floats",
    "sdso_ba_accum_dev", "sdso_ba_get_accumulators", "sdso_ba_set_accumulators", "sdso_ba_get_point_terms", "sdso_ba_solve",
    "sdso_ba_get_point_steps", "sdso_ba_optimize", "sdso_ba_marginalize_points", "sdso_ba_get_tables",
    "sdso_ba_keep_projections", "sdso_ba_batch_create", "sdso_ba_batch_accumulate", "sdso_ba_batch_solve",
    "sdso_ba_batch_accum_dev", "sdso_ba_batch_get_x", "sdso_ba_batch_set_materialize",
    "sdso_immature_init_batch", "sdso_trace_stereo_batch", "sdso_trace_stereo_prepare", "sdso_trace_stereo_enqueue",
    "sdso_trace_stereo_fetch", "sdso_stereo_match_batch", "sdso_activate_points_batch", "sdso_ba_marginalize_frame", "sdso_ba_batch_linearize", "sdso_ba_batch_schur", "sdso_pixel_select", "sdso_pixel_selector_pattern", "sdso_trace_on_batch", "sdso_track_make_ref", "sdso_track_newest_coarse_batch", "sdso_track_get_ref",
    "sdso_ba_get_post_state", "sdso_ba_batch_keep_system", "sdso_ba_get_stitched", "sdso_ba_batch_exchange_mode", "sdso_ba_resubstitute", "sdso_ba_get_counts", "sdso_ba_calc_energies", "sdso_ba_get_deltas", "sdso_ba_marginalize_frame_dev", "sdso_ba_adopt_prior",
    "sdso_ba_batch_optimize", "sdso_ba_batch_optimize_begin", "sdso_ba_batch_step", "sdso_ba_batch_solve_step", "sdso_ba_batch_optimize_end", "sdso_ba_get_state",
    "sdso_comm_unique_id", "sdso_comm_init", "sdso_comm_init_host", "sdso_comm_attach", "sdso_comm_info", "sdso_comm_destroy", "sdso_ba_allreduce", "sdso_ba_allreduce_window",
    "sdso_g2o_track_add_edges", "sdso_g2o_track_linearize", "sdso_g2o_track_newest_coarse", "sdso_g2o_lba_eval", "sdso_trace_set_gn_mode",
]


class SdsoError(RuntimeError):
    pass


class Context:
    """One GPU + one HIP stream (sdso_ctx)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.sdso_ctx_create(device, C.byref(h))
        if rc != 0:
            raise SdsoError("sdso_ctx_create(%d) failed with %d: no usable HIP device (no CPU fallback)" % (device, rc))
        self.h = h

    def check(self, rc):
        if rc != 0:
            msg = self.L.sdso_last_error(self.h)
            raise SdsoError("sdso call failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def close(self):
        if self.h:
            self.L.sdso_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self.check(self.L.sdso_ctx_sync(self.h))

    def prof_read(self, kernel):
        ms = C.c_double(0)
        n = C.c_long(0)
        self.check(self.L.sdso_prof_read(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def upload_pyramid(self, slot, pyr):
        n = len(pyr)
        ws = (C.c_int * n)(*[p.shape[1] for p in pyr])
        hs = (C.c_int * n)(*[p.shape[0] for p in pyr])
        arrs = [np.ascontiguousarray(p, np.float32) for p in pyr]
        ptrs = (c_float_p * n)(*[fp(a) for a in arrs])
        self.check(self.L.sdso_upload_pyramid(self.h, slot, n, ws, hs, ptrs))

    def set_ref(self, ref_slot, pc):
        for lvl, p in enumerate(pc):
            u, v, i, c = [np.ascontiguousarray(p[k], np.float32) for k in ("u", "v", "idepth", "color")]
            self.check(self.L.sdso_track_set_ref(self.h, ref_slot, lvl, len(u), fp(u), fp(v), fp(i), fp(c)))
