#!/usr/bin/env python3
"""Generates tests/golden/reproj_ref_golden.npz by running the REFERENCE's ReprojectionFactor::Evaluate
(oracle/_ref/libref_reproj.so = /root/reference/.../factors/reprojection_factor.h compiled unmodified against the
Eigen-interface shim in oracle/ref_build/shim).  Run in the build container only (needs /root/reference):
    make -C oracle/ref_build && python tests/golden/make_reproj_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import reproj_data as rd  # noqa: E402


def ref_eval(lib, obs15, pi, pj, ext, rho, td):
    r, J = np.zeros(2), np.zeros(46)
    p = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    rc = lib.ref_reproj_eval_one(p(obs15), p(pi), p(pj), p(ext), C.c_double(rho), C.c_double(td), r.ctypes.data_as(C.c_void_p),
                                 J.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return r, J


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_reproj.so"))
    w = rd.make_window(40, 6, seed=123, pixel_noise=0.8)
    # non-unit quaternions (Ceres hands over whatever Plus() produced) and a non-zero td exercise every term
    poses = w["poses"].copy()
    poses[2, 3:] *= 1.0000003
    n = w["obs_soa"].shape[1]
    sel = np.arange(0, n, max(1, n // 64))[:64]
    R, J = [], []
    for k in sel:
        r, j = ref_eval(lib, w["obs_soa"][:, k], poses[w["idx_i"][k]], poses[w["idx_j"][k]], w["ext"], w["invdepth"][w["idx_lm"][k]], w["td"])
        R.append(r)
        J.append(j)
    np.savez(os.path.join(ROOT, "tests", "golden", "reproj_ref_golden.npz"), obs=w["obs_soa"][:, sel], idx_i=w["idx_i"][sel],
             idx_j=w["idx_j"][sel], idx_lm=w["idx_lm"][sel], poses=poses, ext=w["ext"], invdepth=w["invdepth"], td=w["td"],
             r=np.array(R), J=np.array(J))
    print("wrote", len(sel), "golden factor evaluations")


if __name__ == "__main__":
    main()
