"""resident loop vs host-driven loop vs oracle (f32 / f64 sums) at configs[2] with a momentum bit: final pose states."""
import ctypes as C
import os
os.environ.setdefault("SDSO_DEBUG_ENV", "1")   # the library reads its A/B switches only behind this gate
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("stereo-dso-g2o_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from sdso_amd import abi  # noqa: E402
import synth  # noqa: E402
import pyoracle  # noqa: E402
import test_ba_solver_bits_gpu as T  # noqa: E402

if __name__ == "__main__":
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    oracle = pyoracle.load()
    ctx = abi.Context(0)
    win = dict(synth.ba_window(w=1232, h=368, nf=8, pts_per_kf=250, seed=3001))
    win["solverMode"] = T.DEFAULT | bits
    W, keep = T._upload(ctx, win)
    o32 = T._oracle_loop(oracle, W, win, nit)
    o64 = T._oracle_loop(oracle, W, win, nit, acc64=True)
    dev = T._device_loop(ctx, W, win, nit)
    os.environ["SDSO_BA_HOST_LOOP"] = "1"
    host = T._device_loop(ctx, W, win, nit)
    del os.environ["SDSO_BA_HOST_LOOP"]
    f = lambda a, b: (np.abs(a - b)[:, :8] * T._STATE_SCALE)[:, :6].max()
    print("bits %d its %d/%d/%d/%d: resident-truth %.2e  host-truth %.2e  cpu32-truth %.2e  resident-host %.2e" % (bits, dev[3].iterations, host[3].iterations, o32[3].iterations, o64[3].iterations,
          f(dev[0], o64[0]), f(host[0], o64[0]), f(o32[0], o64[0]), f(dev[0], host[0])))
    ctx.close()
