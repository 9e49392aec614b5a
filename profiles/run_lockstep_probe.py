#!/usr/bin/env python3
"""Profiling harness: bench.py's `replay.lockstep64` block alone — 64 estimators in four lock-step groups of 16 (ICG_GVINS_DEBUG=1 prints every
estimator's phase clock when it goes: `... 2>&1 | grep gvins-phase` and sum by phase)."""
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

n_est = int(sys.argv[1]) if len(sys.argv) > 1 else 64
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 4
root = tempfile.mkdtemp(prefix="lockstep_probe_")
hostlib = C.CDLL(H.TOOLS_LIB)
seq = gvd.Sequence(hostlib)
files = seq.write(root)
gvc.run_replay(hostlib, files)  # (contexts, code paged in)
outs = [os.path.join(root, "l%d" % k) for k in range(n_est)]
S, wall, shared = gvc.run_replay_lockstep(hostlib, files, outs, groups=groups)
print("lockstep %d estimators / %d groups: %.2f x real time, wall %.3f s, solves %s, marg batches %s" %
      (n_est, groups, sum(x["data_seconds"] for x in S) / wall, wall, list(shared), list(gvc.lockstep_marg_counts(hostlib))))
