#!/usr/bin/env python3
"""Inputs of the ThreadSanitizer drivers (tests/tsan/run.sh): 6 synthetic 640x480 streams x 12 frames + INS poses for frontend_groups.cc, and
one GNSS / IMU / camera sequence for the replay binary.  Rendered with the regular (uninstrumented) checker build."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
import gvins_data as gvd  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/icg_tsan"
os.makedirs(out, exist_ok=True)
lib = C.CDLL(os.path.join(ROOT, "oracle", "libicgvins_host_oracle.so"))
w, h, B, ring = 640, 480, 6, 12
cam = H.camera_for(w, h)
scene = H.SynthScene(lib, w, h, cam, tex_size=1024, threads=4)
np.stack([np.stack([scene.render(k, stream=s) for k in range(ring)]) for s in range(B)]).tofile(os.path.join(out, "fe_frames.bin"))
np.stack([np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for k in range(ring)]) for s in range(B)]).astype(np.float64).tofile(os.path.join(out, "fe_poses.bin"))
np.asarray(cam, np.float64).tofile(os.path.join(out, "fe_cam.bin"))
files = gvd.Sequence(lib).write(os.path.join(out, "seq"))
print(files["config"], files["imu"], files["gnss"], files["images"])
# 7. marg_batch.cc: one marginalization problem (the window of the back-end tests), flat binary: header of 4 int32 (factors, poses, landmarks, 0),
#    obs (15 x n, SoA), idx_i, idx_j, idx_lm (int32 each), poses (7 per pose), extrinsic (7), inverse depths, td
import marg_data as md  # noqa: E402
Pm = md.make_problem(n_lm=80, n_kf=6, seed=2)
wm = Pm["w"]
with open(os.path.join(out, "marg_problem.bin"), "wb") as f:
    n = Pm["obs"].shape[1]
    np.asarray([n, wm["poses"].shape[0], wm["invdepth"].shape[0], 0], np.int32).tofile(f)
    np.ascontiguousarray(Pm["obs"], np.float64).tofile(f)
    for k in ("ii", "jj", "ll"):
        np.ascontiguousarray(Pm[k], np.int32).tofile(f)
    np.ascontiguousarray(wm["poses"], np.float64).tofile(f)
    np.ascontiguousarray(wm["ext"], np.float64).tofile(f)
    np.ascontiguousarray(wm["invdepth"], np.float64).tofile(f)
    np.asarray([wm["td"]], np.float64).tofile(f)
