#!/usr/bin/env python3
"""Profiling harness: one large k_lk_track_fb launch (S streams x N points) in isolation, for rocprofv3 PMC passes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import icgvins  # noqa: E402
import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 560
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w, h = 1280, 720
_lib = icgvins.load_library(os.environ["ICG_LK_LIB"]) if os.environ.get("ICG_LK_LIB") else None
c = icgvins.Context(w, h, n_slots=2 * S, max_batch=2, max_points=S * N + 16, lib=_lib)
c.set_camera(synth.CAM_1280)
base = synth.texture(w, h, seed=1)
for s in range(S):
    a = np.roll(base, 37 * s, axis=1)
    b = synth.shift_image(a, 6.5 + 0.1 * s, -2.25)
    c.preprocess([2 * s, 2 * s + 1], [a, b])
pts = np.concatenate([synth.random_points(N, w, h, 8, seed=10 + s) for s in range(S)])
prev = np.repeat(np.arange(S) * 2, N).astype(np.int32)
guess = pts + np.array([5.0, -1.5], np.float32)
c.prof_enable(True)
t0 = time.time()
for _ in range(reps):
    out, st = c.lk_track_fb(prev, prev + 1, pts, guess)
dt = (time.time() - t0) / reps
n, ms = c.prof()["lk_track_fb"]
import hashlib
print("output sha1", hashlib.sha1(out.tobytes() + st.tobytes()).hexdigest()[:16], "ICG_LK_PAIR", os.environ.get("ICG_LK_PAIR", "default"))
print(f"S={S} N={N}: {S*N} points, kernel {ms/n*1e3:.1f} us/launch ({S*N/(ms/n*1e-3)/1e6:.2f} Mpoints/s), call {dt*1e6:.0f} us, kept {st.mean():.3f}")
