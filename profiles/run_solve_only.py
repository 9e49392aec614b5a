#!/usr/bin/env python3
"""Runs the window optimization of bench.py's `solve` block a few times (for rocprofv3 --kernel-trace --stats)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
import solve_utils as su  # noqa: E402

if __name__ == "__main__":
    P = su.make_problem(300, 10, seed=4, n_outliers=10, perturb=0.2)
    lib = C.CDLL(H.HOST_LIB)
    su.host_solve(lib, P)
    t = time.perf_counter()
    n = int(os.environ.get("N_SOLVES", "20"))
    ms = [su.host_solve(lib, P)["solve_ms"] for _ in range(n)]
    print("solve_ms median %.3f min %.3f wall/solve %.3f" % (sorted(ms)[n // 2], min(ms), (time.perf_counter() - t) / n * 1e3))
