#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the oracle:  python tests/golden/make_golden.py
(run from the repository root; needs oracle/liboracle.so, i.e. `make -C oracle`)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "stereo-dso-g2o_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), HERE):
    sys.path.insert(0, p)
import pyoracle  # noqa: E402
import cases  # noqa: E402

if __name__ == "__main__":
    orc = pyoracle.load()
    for name, fn in cases.CASES.items():
        out = fn(orc)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: np.asarray(v).shape for k, v in out.items()})
