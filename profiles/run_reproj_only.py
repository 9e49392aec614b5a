#!/usr/bin/env python3
"""Profiling harness: the bench's `reproj` workload alone (256 C2 windows x 2 672 factors per k_reproj_eval launch, outputs resident),
for rocprofv3 --pmc passes of k_reproj_eval."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import icgvins  # noqa: E402
import reproj_data as rd  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
win = rd.make_window(300, 10, seed=0)
K, L = win["poses"].shape[0], win["invdepth"].shape[0]
obs = np.tile(win["obs_soa"], (1, reps))
ctx = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64, max_factors=obs.shape[1])
ctx.reproj_set_factors(obs, np.concatenate([win["idx_i"] + r * K for r in range(reps)]), np.concatenate([win["idx_j"] + r * K for r in range(reps)]),
                       np.concatenate([win["idx_lm"] + r * L for r in range(reps)]))
pR, iR = np.tile(win["poses"], (reps, 1)), np.tile(win["invdepth"], reps)
ctx.prof_enable(True)
for _ in range(8):
    ctx.reproj_eval_resident(pR, win["ext"], iR, win["td"], fetch=False)
n, ms = ctx.prof()["reproj_eval"]
print(f"{obs.shape[1]} factors per launch, kernel {ms / n * 1e3:.1f} us -> {obs.shape[1] * 516 / (ms / n * 1e-3) / 1e9:.0f} GB/s algorithmic")
ctx.close()
