#!/usr/bin/env python3
"""Isolated k_lk_track_fb launch with and without the template set-up cache (icg_lk_track_fb_reuse): S streams x N points, frames A -> B
fill the cache, frames B -> C are timed with hints (hits), with hints disabled (-1: misses, cache still written) and through the plain entry
point (no cache traffic).  Prints kernel time per launch of the B -> C call for the three modes and the output hashes (must be equal)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import icgvins  # noqa: E402
import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w, h = 1280, 720
c = icgvins.Context(w, h, n_slots=3 * S, max_batch=3, max_points=S * N + 16)
c.set_camera(synth.CAM_1280)
base = synth.texture(w, h, seed=1)
for s in range(S):
    a = np.roll(base, 37 * s, axis=1)
    b = synth.shift_image(a, 6.5 + 0.1 * s, -2.25)
    cc = synth.shift_image(b, 5.5 - 0.1 * s, 1.75)
    c.preprocess([3 * s, 3 * s + 1, 3 * s + 2], [a, b, cc])
pts = np.concatenate([synth.random_points(N, w, h, 8, seed=10 + s) for s in range(S)])
s0 = np.repeat(np.arange(S) * 3, N).astype(np.int32)
n = len(pts)
c.prof_enable(True)


def kernel_us():
    k, ms = c.prof()["lk_track_fb"]
    return k, ms


for mode in ("plain", "miss", "hit"):
    tot, cnt, sha = 0.0, 0, None
    for _ in range(reps):
        out1, st1 = c.lk_track_fb_reuse(s0, s0 + 1, pts, pts + np.float32([5.0, -1.5]))
        guess2 = out1 + np.float32([4.5, 1.5])
        k0, m0 = kernel_us()
        if mode == "plain":
            out2, st2 = c.lk_track_fb(s0 + 1, s0 + 2, out1, guess2)
        else:
            idx = np.arange(n, dtype=np.int32) if mode == "hit" else np.full(n, -1, np.int32)
            out2, st2 = c.lk_track_fb_reuse(s0 + 1, s0 + 2, out1, guess2, prev_index=idx)
        k1, m1 = kernel_us()
        tot += m1 - m0
        cnt += k1 - k0
        sha = hashlib.sha1(out2.tobytes() + st2.tobytes()).hexdigest()[:16]
    print(f"{mode:5s}: S={S} N={N} {n} points, B->C kernel {1e3 * tot / cnt:.1f} us/launch ({n / (tot / cnt * 1e-3) / 1e6:.2f} Mpoints/s), kept {st2.mean():.3f}, sha1 {sha}")
