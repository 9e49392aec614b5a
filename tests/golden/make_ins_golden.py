#!/usr/bin/env python3
"""Generates tests/golden/ins_ref_golden.npz by running the REFERENCE's own misc.cc (oracle/_ref/libref_misc.so =
/root/reference/ic_gvins/ic_gvins/misc.{h,cc} compiled unmodified against the interface shims of oracle/ref_build/shim) through
the scenario of tests/ins_utils.run_all: INS mechanization series (normal / Earth / scale factors / timing jitter), bracket
search, camera pose prior, IMU series extraction and redo-mechanization.  Build container only:
    make -C oracle/ref_build && python tests/golden/make_ins_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ins_utils as iu  # noqa: E402

if __name__ == "__main__":
    out = iu.run_all(iu.RefMisc())
    np.savez_compressed(iu.GOLDEN, **out)
    print(len(out), "arrays ->", iu.GOLDEN, os.path.getsize(iu.GOLDEN), "bytes")
