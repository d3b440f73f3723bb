"""Default parameter blocks of the reference (settings.cpp / CoarseTracker.cpp) for the ABI structs."""
from . import abi


def track_params(prob, coarsest=None, ref_aff=(0.0, 0.0), exposure=(1.0, 1.0), max_its=(10, 20, 50, 50, 50)):
    """sdso_track_params_t for a synth.tracker_problem(): per-level intrinsics (makeK), DSO-native
    iteration budget {10,20,50,50,50}, setting_coarseCutoffTH=20, setting_huberTH=9,
    setting_affineOptModeA/B = 1e12/1e8, no abort thresholds (NaN, as on the first try)."""
    p = abi.TrackParams()
    L = prob["levels"]
    p.levels = L
    for l in range(L):
        p.w[l] = prob["pyr_ref"][l].shape[1]
        p.h[l] = prob["pyr_ref"][l].shape[0]
        p.fx[l], p.fy[l], p.cx[l], p.cy[l] = prob["fx"][l], prob["fy"][l], prob["cx"][l], prob["cy"][l]
    p.ref_exposure, p.new_exposure = exposure
    p.ref_aff_g2l = abi.Aff(ref_aff[0], ref_aff[1])
    p.coarsestLvl = (min(L, 5) - 1) if coarsest is None else coarsest
    for i in range(5):
        p.minResForAbort[i] = float("nan")
        p.maxIterations[i] = max_its[i]
    p.coarseCutoffTH = 20.0
    p.huberTH = 9.0
    p.affineOptModeA = 1e12
    p.affineOptModeB = 1e8
    return p
