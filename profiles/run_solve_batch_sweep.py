#!/usr/bin/env python3
"""256 C2 windows through WindowSolverBatch: solve wall time (ms) over repetitions, per host thread count (ICG_SOLVER_THREADS)."""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for threads in (sys.argv[2:] or ["0"]):
    if threads != "0":
        os.environ["ICG_SOLVER_THREADS"] = threads
    else:
        os.environ.pop("ICG_SOLVER_THREADS", None)
    ms = sorted(su.host_solve_batch(hl, [Pz] * 256)[1] for _ in range(reps))
    print("threads", threads, "cores", os.cpu_count(), "min %.2f median %.2f max %.2f ms -> %.0f windows/s (median)" % (ms[0], ms[len(ms) // 2], ms[-1], 256e3 / ms[len(ms) // 2]))
