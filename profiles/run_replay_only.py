#!/usr/bin/env python3
"""Profiling harness: the single-stream estimator replay of bench.py's `replay` block, three repetitions (ICG_GVINS_DEBUG=1 prints the phase split)."""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import harness as H  # noqa: E402
import gvins_checks as gvc  # noqa: E402
import gvins_data as gvd  # noqa: E402

root = tempfile.mkdtemp(prefix="replay_only_")
hostlib = C.CDLL(H.TOOLS_LIB)
seq = gvd.Sequence(hostlib)
files = seq.write(root)
for _ in range(3):
    S = gvc.run_replay(hostlib, files)
    print("x real time %.2f  wall %.3f s" % (S["data_seconds"] / S["wall_seconds"], S["wall_seconds"]), file=sys.stderr)
