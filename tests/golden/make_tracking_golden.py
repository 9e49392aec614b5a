#!/usr/bin/env python3
"""Generates tests/golden/tracking_ref_<scenario>.npz by running the REFERENCE's own tracker
(oracle/_ref/libref_tracking.so = /root/reference/.../tracking/*.{h,cc} compiled unmodified against the interface shims in
oracle/ref_build/shim, OpenCV entry points forwarded to the oracle primitives) on the harness's synthetic streams; one fresh
process per scenario because the reference's id factories are process-wide statics.  Per frame: track state, map-point ids of
the frame's features, distorted / undistorted key-point floats, window bookkeeping.  Run in the build container only:
    make -C oracle && make -C oracle/ref_build && python tests/golden/make_tracking_golden.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_tracking_utils as rt  # noqa: E402

if __name__ == "__main__":
    for name in rt.SCENARIOS:
        rt.run_scenario_in_subprocess(name, rt.golden_path(name))
        r = rt.load(rt.golden_path(name))
        print(name, "states", list(r["states"]), "last stats", list(r["stats"][-1]))
