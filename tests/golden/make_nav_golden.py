#!/usr/bin/env python3
"""Generates tests/golden/nav_ref_golden.npz: the small factors the GVINS window adds next to preintegration / reprojection (GnssFactor,
ImuErrorFactor, ImuPosePriorFactor, ImuMixPriorFactor), the Earth / attitude / GPS-time helpers and MISC::detectZeroVelocity, computed by
the REFERENCE's own headers (oracle/_ref/libref_nav.so).  Build container only:
    make -C oracle/ref_build && python tests/golden/make_nav_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nav_utils as nu  # noqa: E402

if __name__ == "__main__":
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_nav.so"))
    out = nu.evaluate(lib, "ref_")
    np.savez_compressed(nu.GOLDEN, **out)
    print(len(out), "arrays;", "zero-velocity decisions:", [int(out[f"zv_{i}"][0]) for i in range(len(nu.zero_velocity_cases()))])
