"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/liboracle.so (strict-IEEE CPU restatement, the parity checker) and
oracle/liboracle_fast.so (same sources, -O3 -march=native: bench.py's cpu_baseline "port").
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "stereo-dso-g2o_amd"))
from sdso_amd.abi import (G2oTrackEval, G2oLba, Activate, TraceGeom, TrackEval, SE3, Aff, TrackParams, TrackResult, BAWindow, BAOptResult, BAPostState, TracePoints,  # noqa: E402
                          c_float_p, c_double_p, c_int_p, c_u8_p)

_libs = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


FAST_FLAGS_SHIPPED = "-O3 -march=x86-64-v3 -mtune=generic"   # oracle/Makefile FAST_FLAGS


def build_native(out_dir="/tmp"):
    """The timed CPU baseline rebuilt ON THE HOST THAT TIMES IT (-O3 -march=native, the reference's own flags, CMakeLists.txt:83-84).
    Returns (path, flags), or (None, reason) when no compiler is there."""
    import shutil
    if not shutil.which(os.environ.get("CXX", "g++")):
        return None, "no g++ on the timed host"
    out = os.path.join(out_dir, "liboracle_native_%d.so" % os.getpid())
    flags = "-O3 -march=native"
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native", "FAST_FLAGS=" + flags, "FAST_OUT=" + out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    except (subprocess.SubprocessError, OSError) as e:
        return None, "native build failed: %r" % (e,)
    return out, flags


def load(fast=False, path=None):
    name = path or ("liboracle_fast.so" if fast else "liboracle.so")
    if name in _libs:
        return _libs[name]
    path = path or os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_se3_exp.argtypes = [c_double_p, C.POINTER(SE3)]
    L.orc_se3_exp.restype = None
    L.orc_se3_log.argtypes = [C.POINTER(SE3), c_double_p]
    L.orc_se3_log.restype = None
    L.orc_se3_adj.argtypes = [C.POINTER(SE3), c_double_p]
    L.orc_se3_adj.restype = None
    L.orc_se3_mul.argtypes = [C.POINTER(SE3), C.POINTER(SE3), C.POINTER(SE3)]
    L.orc_se3_mul.restype = None
    L.orc_se3_inv.argtypes = [C.POINTER(SE3), C.POINTER(SE3)]
    L.orc_se3_inv.restype = None
    L.orc_mat3f_inv.argtypes = [c_float_p, c_float_p]
    L.orc_mat3f_inv.restype = None
    L.orc_ldlt_solve.argtypes = [C.c_int, c_double_p, c_double_p, c_double_p]
    L.orc_pyramid_levels.argtypes = [C.c_int, C.c_int]
    L.orc_make_images.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, C.POINTER(c_float_p)]
    L.orc_make_images.restype = None
    L.orc_track_calc_res_gs.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, C.POINTER(TrackEval),
                                        c_double_p, c_double_p, c_double_p, c_int_p, c_u8_p, c_float_p, C.c_int]
    L.orc_track_make_eval.argtypes = [C.POINTER(TrackParams), C.c_int, C.POINTER(SE3), C.POINTER(Aff), C.c_float, C.POINTER(TrackEval)]
    L.orc_track_make_eval.restype = None
    L.orc_track_newest_coarse.argtypes = [c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p),
                                          C.POINTER(c_float_p), C.POINTER(TrackParams), C.POINTER(SE3), C.POINTER(Aff), C.POINTER(TrackResult)]
    L.orc_make_coarse_depth.argtypes = [C.c_int, c_int_p, c_int_p, C.POINTER(c_float_p), C.c_int, c_int_p, c_int_p, c_float_p, c_float_p, c_int_p,
                                        C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p)]
    L.orc_ba_get_stitched.argtypes = [vp] + [c_double_p] * 6
    L.orc_track_last_margins.argtypes = [c_double_p]
    L.orc_track_last_margins.restype = None
    L.orc_ba_create.argtypes = [C.POINTER(BAWindow)]
    L.orc_ba_create.restype = vp
    L.orc_ba_destroy.argtypes = [vp]
    L.orc_ba_destroy.restype = None
    L.orc_ba_set_threads.argtypes = [vp, C.c_int]
    L.orc_ba_linearize.argtypes = [vp, c_double_p]
    L.orc_ba_get_linearization.argtypes = [vp, c_float_p, c_u8_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.orc_ba_apply_res.argtypes = [vp]
    L.orc_ba_get_ef_jacobians.argtypes = [vp, c_float_p]
    L.orc_ba_get_residual_state.argtypes = [vp, c_u8_p, c_u8_p, c_float_p]
    L.orc_ba_accumulate.argtypes = [vp]
    L.orc_ba_accum_floats.argtypes = [C.c_int]
    L.orc_ba_get_accumulators.argtypes = [vp, c_float_p]
    L.orc_set_acc64.argtypes = [C.c_int]
    L.orc_set_acc64.restype = None
    L.orc_ba_get_accumulators_f64.argtypes = [vp, c_double_p]
    L.orc_ba_get_point_terms.argtypes = [vp, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.orc_ba_solve.argtypes = [vp, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.orc_ba_get_point_steps.argtypes = [vp, c_float_p]
    L.orc_ba_optimize.argtypes = [vp, C.c_int, c_double_p, c_float_p, c_u8_p, C.POINTER(BAOptResult)]
    L.orc_ba_marginalize_points.argtypes = [vp, c_u8_p, c_double_p, c_double_p]
    L.orc_ba_get_post_state.argtypes = [vp, C.POINTER(BAPostState)]
    L.orc_ba_get_x_trace.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    L.orc_ba_get_step_trace.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.orc_ba_calc_energies.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_ba_get_deltas.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.orc_ba_get_tables.argtypes = [vp, c_float_p, c_double_p, c_double_p, c_float_p]
    L.orc_immature_init_batch.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.orc_pixel_select.argtypes = [C.POINTER(c_float_p), C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, c_int_p, c_float_p]
    L.orc_set_gamma.argtypes = [c_float_p]
    L.orc_set_gamma.restype = None
    L.orc_gamma_from_binv.argtypes = [c_float_p, c_float_p]
    L.orc_gamma_from_binv.restype = None
    L.orc_selector_random_pattern.argtypes = [C.c_int, c_u8_p]
    L.orc_selector_random_pattern.restype = None
    L.orc_marginalize_frame.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.orc_activate_points.argtypes = [C.POINTER(Activate), C.POINTER(C.c_int8), c_float_p, c_u8_p]
    L.orc_trace_on_batch.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, C.POINTER(TraceGeom), c_int_p, C.POINTER(TracePoints), c_u8_p]
    L.orc_trace_stereo_batch.argtypes = [c_float_p, C.c_int, C.c_int, c_float_p, C.c_float, C.c_int, C.POINTER(TracePoints), c_u8_p]
    L.orc_trace_stereo_batch_gn.argtypes = [c_float_p, C.c_int, C.c_int, c_float_p, C.c_float, C.c_int, C.POINTER(TracePoints), c_u8_p, C.c_int]
    L.orc_g2o_track_add_edges.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, C.POINTER(G2oTrackEval), c_double_p, c_u8_p, c_float_p]
    L.orc_g2o_track_linearize.argtypes = [C.c_int, c_u8_p, c_float_p, c_float_p, c_float_p, C.POINTER(G2oTrackEval), c_double_p, c_double_p, c_double_p,
                                          c_double_p, c_double_p]
    L.orc_g2o_track_newest_coarse.argtypes = [c_int_p, C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p), C.POINTER(c_float_p),
                                              C.POINTER(c_float_p), C.POINTER(TrackParams), C.POINTER(SE3), C.POINTER(Aff), C.POINTER(TrackResult)]
    L.orc_g2o_lba_eval.argtypes = [C.POINTER(G2oLba), c_double_p, c_double_p, c_u8_p, c_float_p, c_float_p, c_float_p, c_u8_p]
    _libs[name] = L
    return L


def make_coarse_depth(L, u, v, new_idepth, weight, ref_pyr):
    """orc_make_coarse_depth on numpy arrays: a list (per level) of dicts u, v, idepth, color (float32), like the tracker template."""
    import numpy as np
    levels = len(ref_pyr)
    ws = (C.c_int * levels)(*[p.shape[1] for p in ref_pyr]); hs = (C.c_int * levels)(*[p.shape[0] for p in ref_pyr])
    imgs = [np.ascontiguousarray(p, np.float32) for p in ref_pyr]
    dI = (c_float_p * levels)(*[a.ctypes.data_as(c_float_p) for a in imgs])
    out = [[np.zeros(p.shape[0] * p.shape[1], np.float32) for p in ref_pyr] for _ in range(4)]
    ptrs = [(c_float_p * levels)(*[a.ctypes.data_as(c_float_p) for a in arrs]) for arrs in out]
    pc_n = (C.c_int * levels)()
    ui, vi = np.ascontiguousarray(u, np.int32), np.ascontiguousarray(v, np.int32)
    idp, wgt = np.ascontiguousarray(new_idepth, np.float32), np.ascontiguousarray(weight, np.float32)
    L.orc_make_coarse_depth(levels, ws, hs, dI, len(ui), ui.ctypes.data_as(c_int_p), vi.ctypes.data_as(c_int_p), idp.ctypes.data_as(c_float_p),
                            wgt.ctypes.data_as(c_float_p), pc_n, *ptrs)
    return [dict(u=out[0][l][:pc_n[l]].copy(), v=out[1][l][:pc_n[l]].copy(), idepth=out[2][l][:pc_n[l]].copy(), color=out[3][l][:pc_n[l]].copy())
            for l in range(levels)]
