"""CPU-side checks of the product library: it loads, exports every symbol include/sdso_abi.h
declares, fails loudly without a GPU, and its host-only helpers agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
from sdso_amd import abi
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "sdso_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdso_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = abi.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libsdso_hip.so does not export %s" % n
    assert sorted(abi.EXPORTED_SYMBOLS) == names


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.SdsoError):
        abi.Context(0)
    L = abi.load()
    assert L.sdso_ctx_sync(None) != 0
    assert L.sdso_ba_linearize(None, 0, None) != 0


def test_pyramid_levels_rule(oracle):
    L = abi.load()
    for w, h in ((1241, 376), (1232, 368), (640, 480), (64, 48), (4096, 2048)):
        assert L.sdso_pyramid_levels(w, h) == oracle.orc_pyramid_levels(w, h) == synth.pyramid_levels(w, h)


def test_make_eval_matches_oracle_bitwise(oracle):
    L = abi.load()
    prob = dict(levels=5, pyr_ref=[np.zeros((368 >> l, 1232 >> l, 3), np.float32) for l in range(5)])
    cal = synth.kitti_calib(1232, 368)
    prob["fx"], prob["fy"], prob["cx"], prob["cy"] = synth.level_intrinsics(cal["fx"], cal["fy"], cal["cx"], cal["cy"], 5)
    prm = helpers.track_params(prob, ref_aff=(0.03, -2.0), exposure=(0.8, 1.3))
    rs = np.random.RandomState(3)
    for lvl in range(5):
        T = abi.SE3.from_Rt(*synth.se3_exp(rs.normal(0, 0.05, 6)))
        aff = abi.Aff(rs.normal(0, 0.1), rs.normal(0, 5))
        a, b = abi.TrackEval(), abi.TrackEval()
        L.sdso_track_make_eval(C.byref(prm), lvl, C.byref(T), C.byref(aff), 2.0, C.byref(a))
        oracle.orc_track_make_eval(C.byref(prm), lvl, C.byref(T), C.byref(aff), 2.0, C.byref(b))
        assert bytes(a) == bytes(b)
        assert a.cutoffTH == 40.0 and a.w == 1232 >> lvl


def test_marginalize_frame_host_algebra(oracle):
    """EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:554-660) is ~70x70 double algebra that needs no GPU: the library
    function against the oracle and against the textbook Schur complement of the re-ordered system."""
    import numpy as np
    from sdso_amd import abi
    L = abi.load()
    rs = np.random.RandomState(7)
    for nf, idx in ((8, 0), (8, 3), (8, 7), (3, 1), (1, 0)):
        n = 8 * nf + 4
        A = rs.normal(0, 1, (n, 2 * n))
        scale = 10.0 ** rs.uniform(0, 6, n)                       # entries spanning many decades, like the real prior
        HM = (A @ A.T) * np.outer(scale, scale)
        bM = rs.normal(0, 1, n) * scale
        prior = np.where(rs.rand(8) < 0.5, 10.0 ** rs.uniform(2, 10, 8), 0.0)
        dprior = rs.normal(0, 1e-2, 8)
        m = n - 8
        Ho, bo, Hg, bg = np.zeros((m, m)), np.zeros(m), np.zeros((m, m)), np.zeros(m)
        assert oracle.orc_marginalize_frame(nf, idx, abi.dp(prior), abi.dp(dprior), abi.dp(HM), abi.dp(bM), abi.dp(Ho), abi.dp(bo)) == 0
        assert L.sdso_ba_marginalize_frame(nf, idx, abi.dp(prior), abi.dp(dprior), abi.dp(HM), abi.dp(bM), abi.dp(Hg), abi.dp(bg)) == 0
        d = np.sqrt(np.abs(np.diag(Ho))) + 1e-300
        assert np.abs((Hg - Ho) / np.outer(d, d)).max() <= 1e-9 and np.abs((bg - bo) / d).max() <= 1e-9 * max(1.0, np.abs(bo / d).max())
        keep = [i for i in range(n) if not (4 + 8 * idx <= i < 12 + 8 * idx)]
        drop = list(range(4 + 8 * idx, 12 + 8 * idx))
        D = HM[np.ix_(drop, drop)] + np.diag(prior)
        B = HM[np.ix_(keep, drop)]
        Href = HM[np.ix_(keep, keep)] - B @ np.linalg.solve(D, B.T)
        bref = bM[keep] - B @ np.linalg.solve(D, bM[drop] + prior * dprior)
        assert np.abs((Hg - Href) / np.outer(d, d)).max() <= 1e-6 and np.allclose(Hg, Hg.T)
        assert np.abs((bg - bref) / d).max() <= 1e-6 * max(1.0, np.abs(bref / d).max())
    assert L.sdso_ba_marginalize_frame(2, 5, abi.dp(prior), abi.dp(dprior), abi.dp(HM), abi.dp(bM), abi.dp(Hg), abi.dp(bg)) == -1


def test_selector_random_pattern_is_glibc_rand(oracle):
    """PixelSelector's randomPattern is rand() & 0xFF after srand(3141592) (PixelSelector2.cpp:43-44).  The library carries its
    own copy of glibc's generator (it must not reseed the process-wide one); the oracle calls srand/rand."""
    import numpy as np
    from sdso_amd import abi
    L = abi.load()
    n = 640 * 480
    a = np.zeros(n, np.uint8); b = np.zeros(n, np.uint8)
    oracle.orc_selector_random_pattern(n, abi.bp(a))
    assert L.sdso_pixel_selector_pattern(n, abi.bp(b)) == 0
    assert np.array_equal(a, b) and len(np.unique(a)) == 256


def test_bench_gpus_flag_is_honoured_or_refused():
    """`bench.py --gpus N` must never print a line for another N: with WORLD_SIZE set to something else it refuses before touching
    torch; with no launcher it starts N ranks itself, and when they cannot run (this container has no GPU) the command fails."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"WORLD_SIZE=4" in r.stderr and b"{" not in r.stdout
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], env=dict(env, SDSO_DIST_BACKEND="gloo_lib"),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode != 0 and b"rank(s) failed" in r.stderr and b"{" not in r.stdout
