#!/usr/bin/env python3
"""Profiling harness: 256 C2 windows through WindowSolverBatch, three repetitions (ICG_SOLVER_DEBUG=1 / ICG_ABI_DEBUG=1 print the phase split)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import harness as H  # noqa: E402
import solve_utils as su  # noqa: E402

hl = C.CDLL(H.HOST_LIB)
Pz = su.make_problem(300, 10, seed=4, n_outliers=10, perturb=0.2)
su.host_solve_batch(hl, [Pz] * 4)
for _ in range(3):
    res, ms = su.host_solve_batch(hl, [Pz] * 256)
print(ms)
