#!/usr/bin/env python3
"""Host-logic section timers of the track-table engine WITHOUT a GPU: the product host layer on the CPU restatement of the C ABI
(libicgvins_host_oracle.so: the device calls are slow, the host sections are the product's own code).  Thread CPU time per section, us per
frame, at the C2 configuration.   ICG_HOST_PROF=cpu python profiles/run_host_prof_cpu.py [streams] [frames]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
os.environ.setdefault("ICG_HOST_PROF", "cpu")
import numpy as np  # noqa: E402

import harness as H  # noqa: E402
from stream_utils import ensure_oracle_host  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
w, h = 1280, 720
cam = H.camera_for(w, h)
sb = H.StreamBatch(ensure_oracle_host(), B, w, h, cam, max_features=300, window=10, engine=os.environ.get("ICG_TRACK_ENGINE", "table"))
scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=4)
ring = [[scene.render(k, stream=s) for s in range(B)] for k in range(32)]
hp = np.zeros(64)
warm = 20
for k in range(warm + N):
    kk = k % 62
    kk = kk if kk < 32 else 62 - kk  # ping-pong like bench.py
    poses = np.stack([H.pose12(*scene.ins_pose(kk, stream=s)) for s in range(B)])
    if k == warm:
        sb.lib.icgh_hostprof(hp.ctypes.data_as(C.c_void_p), 32, None, 0, 1)
    sb.step([f.ctypes.data for f in ring[kk]], w, np.full(B, 100.0 + k / 20.0), poses)
nm = C.create_string_buffer(1024)
n = sb.lib.icgh_hostprof(hp.ctypes.data_as(C.c_void_p), 32, nm, 1024, 0)
for k, name in enumerate(nm.value.decode().split(";")[:n]):
    if hp[2 * k + 1] > 0:
        print(f"{name:18s} {1e6 * hp[2 * k] / (B * N):8.2f} us/frame   calls/frame {hp[2 * k + 1] / (B * N):.3f}")
print("states", [sb.stats(s)["landmarks"] for s in range(B)])
sb.close()
