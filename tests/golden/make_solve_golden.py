#!/usr/bin/env python3
"""Generates tests/golden/solve_ref_golden.npz: three sliding windows solved the way GVINS::gvinsOptimization solves them —
LM, chi-square removal of reprojection factors, LM — with the REFERENCE's own factor code (ReprojectionFactor, PoseParameterization,
ImuPosePriorFactor, ceres::HuberLoss, compiled unmodified into oracle/_ref/libref_gvins.so) and the Levenberg-Marquardt of
oracle/ref_build/shim/ceres/problem_shim.h (Ceres itself is absent: the solver is a restatement, written independently of the product's).
Build container only:   make -C oracle/ref_build && python tests/golden/make_solve_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import solve_utils as su  # noqa: E402

CASES = [(60, 6, 4), (300, 10, 10), (120, 8, 0)]  # landmarks, keyframes, gross outliers; seed = index
GOLDEN = os.path.join(ROOT, "tests", "golden", "solve_ref_golden.npz")


def reference_solve(ref, P, prior_weight=30.0, huber=1.0, iters1=6, iters2=18, chi2=5.991):
    s = P["start"]
    poses, ext, inv, td = s["poses"].copy(), s["ext"].copy(), s["invdepth"].copy(), np.array([s["td"]])
    n = P["obs"].shape[1]
    summ, act, obs = np.zeros(10), np.zeros(n, np.uint8), np.ascontiguousarray(P["obs"])
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    ii, jj, ll = i32(P["ii"]), i32(P["jj"]), i32(P["ll"])
    rc = ref.ref_window_solve(n, su._p(obs), su._p(ii), su._p(jj), su._p(ll), poses.shape[0], su._p(poses), su._p(ext), len(inv), su._p(inv), su._p(td),
                              su._p(np.ascontiguousarray(P["prior"])), C.c_double(prior_weight), C.c_double(huber), 0, 0, iters1, iters2, C.c_double(chi2),
                              su._p(summ), su._p(act))
    assert rc == 0
    return dict(poses=poses, ext=ext, invdepth=inv, td=td, summary=summ[:8], active=act)


if __name__ == "__main__":
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_gvins.so"))
    out = {}
    for seed, (nlm, nkf, nout) in enumerate(CASES):
        r = reference_solve(ref, su.make_problem(nlm, nkf, seed=seed, n_outliers=nout))
        for k, v in r.items():
            out[f"case{seed}_{k}"] = v
        print(seed, np.round(r["summary"], 4), "removed", int((r["active"] == 0).sum()))
    np.savez_compressed(GOLDEN, **out)
