"""sdso_amd — Python binding (ctypes) and synthetic-input generators for libsdso_hip.so,
the MI355X implementation of Stereo-DSO's photometric alignment hot path."""
from . import abi, synth  # noqa: F401
