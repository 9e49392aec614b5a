#!/usr/bin/env python3
"""One small pass through every layer of the product library, printed as digests (JSON on the last stdout line): tests/test_gpu_switches.py
runs it in a child process per environment and compares — every run-time switch that `getenv` reads in csrc/ and host/ is either a
diagnostic (prints, never changes a result) or a scheduling choice (waits, threads, launch grouping: placement only), so the digests of
any combination must equal those of the clean environment.

Workload: 4 streams x 14 frames at 640 x 480 through the tracker of $ICG_TRACK_ENGINE (two stream groups), the map -> optimizer -> map
refinement of the tracked windows, 6 windows through WindowSolverBatch, 5 windows through MarginalizationBatch."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import backend_utils as bu  # noqa: E402
import harness as H  # noqa: E402
import ins_utils as iu  # noqa: E402
import marg_data as md  # noqa: E402
import refine_checks as rc  # noqa: E402
import solve_utils as su  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


def main():
    out = {}
    w, h, ns, nframes = 640, 480, 4, 14
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(H.HOST_LIB, ns, w, h, cam, max_features=100, window=10, host_threads=2, groups=2)
    out["engine"] = sb.engine()
    scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=4)
    states = []
    for k in range(nframes):
        frames = [scene.render(k, stream=s) for s in range(ns)]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for s in range(ns)])
        states.append(sb.step([f.ctypes.data for f in frames], w, np.full(ns, 100.0 + k / 20.0), poses).copy())
    out["track_states"] = sha(np.stack(states))
    out["track_digests"] = [sb.stats(s)["digest"] for s in range(ns)]
    out7, kf = rc.refine(sb, iu.pose_b_c())
    out["refine"] = sha(out7, kf)
    out["dumps"] = hashlib.sha256("".join(sb.dump(s, 0) for s in range(ns)).encode()).hexdigest()[:24]
    sb.close()
    lib = C.CDLL(H.HOST_LIB)
    probs = [su.make_problem(40 + 10 * k, 5 + k % 3, seed=30 + k, n_outliers=3, perturb=0.3) for k in range(6)]
    res, _ = su.host_solve_batch(lib, probs)
    out["solve_batch"] = sha(*[r[key] for r in res for key in ("poses", "ext", "invdepth", "summary")])
    m = bu.backend_marginalize_batch(lib, md.make_problem(n_lm=80, n_kf=6, seed=2), 5, 0, 2)
    out["marg_batch"] = sha(m["Hp"], m["bp"], m["J0"], m["e0"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
