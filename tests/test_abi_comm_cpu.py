"""CPU-side checks of the communicator entry points that need no GPU: a forced RCCL load failure must come back as an error code
(round-2 advisor finding: dlerror() was called twice and the second, NULL, result went into std::string — a crash instead of
SDSO_ERR_STATE)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_load_failure_is_an_error_code_not_a_crash():
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from sdso_amd import abi\n"
        "L = abi.load()\n"
        "buf = (C.c_ubyte * 128)()\n"
        "rc = L.sdso_comm_unique_id(buf)\n"
        "print('rc', rc)\n"
        "rc2 = L.sdso_comm_unique_id(buf)\n"          # the cached failure path
        "print('rc2', rc2)\n"
    ) % os.path.join(ROOT, "stereo-dso-g2o_amd")
    env = dict(os.environ, SDSO_RCCL_LIB="/nonexistent/librccl-forced-missing.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rc -4" in r.stdout and "rc2 -4" in r.stdout, r.stdout          # SDSO_ERR_STATE both times
