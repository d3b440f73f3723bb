import os
import sys

import pytest

# the library's A/B / diagnostic switches (SDSO_BA_*, SDSO_TRK_* ...) exist only behind this gate, read once per process (csrc/sdso_internal.h:
# dbg_env); the variant tests flip such switches, so the test processes — and the subprocesses they start — run with it on
os.environ.setdefault("SDSO_DEBUG_ENV", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-dso-g2o_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the libraries are build artefacts (git-ignored): build whatever is missing before collecting
    need = [os.path.join(ROOT, "stereo-dso-g2o_amd", "csrc", "libsdso_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "stereo-dso-g2o_amd", "host", "test_shim")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    return pyoracle.load()


@pytest.fixture(scope="session")
def gpu_ctx():
    from sdso_amd import abi
    ctx = abi.Context(0)
    yield ctx
    ctx.close()
