"""sdso_amd — Python binding (ctypes) of libsdso_hip.so (the tests' and the benchmark's way in; the synthetic-input generators live in tests/synth.py),
the MI355X implementation of Stereo-DSO's photometric alignment hot path."""
from . import abi  # noqa: F401
