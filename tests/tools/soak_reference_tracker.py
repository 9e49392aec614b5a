#!/usr/bin/env python3
"""Soak test (outside pytest): random scenarios — image size, length, feature budget, scene, histogram gate, blank frames that kill every
track, slow phases that force the time-out keyframe branch — through the REFERENCE's own tracker (oracle/_ref/libref_tracking.so, needs
/root/reference at build time) and through the product's host layer on the same oracle primitives, compared frame by frame exactly as the
committed goldens are (tests/ref_tracking_utils.py: track state, map-point ids, key-point float bits, candidate lists, window bookkeeping,
tracking.txt rows).

    python tests/tools/soak_reference_tracker.py run <seed> <n_scenarios>

Round 2: seeds 1 (12 scenarios), 2 (40) and 3 (40): no divergence.  Round 3 (track-table engine): seeds 11, 12 and 7 (40 each): no divergence.
Round 5 (final tree): seed 53 with ICG_TRACK_ENGINE=core and seed 54 with ICG_TRACK_ENGINE=device (30 scenarios each): no divergence."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import numpy as np  # noqa: E402

import ref_tracking_utils as rt  # noqa: E402
from stream_utils import ensure_oracle_host  # noqa: E402


def register(name, w, h, n, mf, stream, hist, blank, slow):
    rt.SCENARIOS[name] = (w, h, n, mf, stream, hist)
    if blank >= 0:
        rt.BLANK_FRAMES[name] = (blank, blank + 1, blank + 2)
        rt.BLANK_VALUE[name] = 90 if not hist else 235
    if slow >= 0:
        rt.SLOW_AFTER[name] = (slow, 0.02)


if sys.argv[1] == "ref":
    name = sys.argv[2]
    w, h, n, mf, stream, hist, blank, slow = [int(v) for v in sys.argv[3:11]]
    register(name, w, h, n, mf, stream, bool(hist), blank, slow)
    rt.CONFIG["check_hist"] = bool(hist)
    rt._current_scenario[0] = name
    rt.save(sys.argv[11], rt.run_reference(w, h, n, mf, stream))
else:
    rng = np.random.RandomState(int(sys.argv[2]))
    lib = ensure_oracle_host()
    nfail = 0
    for it in range(int(sys.argv[3])):
        w, h = (640, 480) if rng.rand() < 0.8 else (1280, 720)
        n = int(rng.randint(40, 140)) if w == 640 else int(rng.randint(20, 50))
        mf = int(rng.choice([60, 100, 150, 220]))
        stream = int(rng.randint(10, 10000))
        hist = int(rng.rand() < 0.3)
        blank = int(rng.randint(10, n - 8)) if rng.rand() < 0.3 else -1
        slow = int(rng.randint(8, n - 8)) if rng.rand() < 0.25 else -1
        name = f"soak_{sys.argv[2]}_{it}"
        register(name, w, h, n, mf, stream, bool(hist), blank, slow)
        out = tempfile.mktemp(suffix=".npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), "ref", name] + [str(v) for v in (w, h, n, mf, stream, hist, blank, slow)] + [out], check=True)
        try:
            rt.compare_scenario(lib, name, ref=rt.load(out))
            print("ok  ", name, w, h, n, mf, stream, hist, blank, slow, flush=True)
        except AssertionError as e:
            nfail += 1
            print("FAIL", name, w, h, n, mf, stream, hist, blank, slow, str(e)[:300], flush=True)
        os.unlink(out)
    print("failures:", nfail)
    sys.exit(1 if nfail else 0)
