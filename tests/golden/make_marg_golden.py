#!/usr/bin/env python3
"""Generates tests/golden/marg_ref_golden.npz by running the REFERENCE's marginalization pipeline
(oracle/_ref/libref_marg.so = /root/reference/.../factors/{residual_block_info,marginalization_info,marginalization_factor,
reprojection_factor}.h compiled unmodified against the interface shims in oracle/ref_build/shim) on the standard scenario of
tests/backend_utils.py.  Stored: order-/sign-independent quantities (see backend_utils.marginalize_canonical).
Run in the build container only (needs /root/reference):
    make -C oracle/ref_build && python tests/golden/make_marg_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import backend_utils as bu  # noqa: E402


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_marg.so"))
    c = bu.marginalize_canonical(lib)
    np.savez(os.path.join(ROOT, "tests", "golden", "marg_ref_golden.npz"), **c)
    print("m", c["m"], "r", c["r"], "cost", c["cost"], "|Hp|max", np.abs(c["Hp"]).max())


if __name__ == "__main__":
    main()
