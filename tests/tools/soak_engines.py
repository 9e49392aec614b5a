#!/usr/bin/env python3
"""Soak test (outside pytest): the two engines of TrackingBatch — track table (default) and reference-shaped object graph — driven through
random scenarios WITH the map-writing entry points interleaved at random frames: window refinement (VisualWindow -> WindowSolver ->
write-back -> WindowCulling), outlier culling with random optimisation lists, and landmark moves (optimizer write-back stand-in).  On the
table engine those work on TableTracker::view() and are absorbed back; after every frame and every operation the canonical state dumps
(floats as bit patterns, landmark iteration order included) and the operations' outputs must be equal.  Oracle-backed (CPU).

    python tests/tools/soak_engines.py <seed> <n_scenarios> [engine_a,engine_b]

The engine pair defaults to table,object; round 4 adds table,core (the tracker core compiled for the host) and table,device (the
device-resident tracker's host side on the CPU backend of icg_tracker_*: every map-writing operation goes through download / view / absorb /
upload; ICG_TRACKER_LOG_DRAIN=30 makes the landmark-history drains frequent).
Round 3: seeds 1 (24 scenarios) and 2 (24): no divergence.  Round 4: seed 31 (30 scenarios, table,device, ICG_TRACKER_LOG_DRAIN=30) and seed 32 (30, table,core): no divergence.
Round 5 (tracker core after the stages' latency diet: fused compaction, parallel releases / triangulation list / window keeper, list ranking): seed 51
(30, table,device, ICG_TRACKER_LOG_DRAIN=30) and seed 52 (30, table,core): no divergence."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import numpy as np  # noqa: E402

import cull_utils as cu  # noqa: E402
import harness as H  # noqa: E402
import ins_utils as iu  # noqa: E402
import refine_checks as rc  # noqa: E402
from stream_utils import ensure_oracle_host  # noqa: E402


def first_difference(a, b):
    la, lb = a.splitlines(), b.splitlines()
    i = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), min(len(la), len(lb)))
    return i, la[i] if i < len(la) else "<end>", lb[i] if i < len(lb) else "<end>"


def scenario(lib, rng, tag, engines=("table", "object")):
    A, B = engines
    w, h = (640, 480) if rng.rand() < 0.85 else (1280, 720)
    n = int(rng.randint(30, 90)) if w == 640 else int(rng.randint(20, 32))
    mf = int(rng.choice([60, 100, 150]))
    window = int(rng.choice([5, 10]))
    stream = int(rng.randint(10, 10000))
    blank = int(rng.randint(10, max(11, n - 8))) if rng.rand() < 0.3 else -1
    cam = H.camera_for(w, h)
    sbs = {e: H.StreamBatch(lib, 1, w, h, cam, max_features=mf, window=window, engine=e) for e in engines}
    scene = H.SynthScene(sbs[A].lib, w, h, cam, tex_size=1024, threads=4)
    ops = []
    for k in range(n):
        img = scene.render(k, stream=stream)
        if blank >= 0 and blank <= k < blank + 3:
            img = np.full_like(img, 90)
        R, t = scene.ins_pose(k, stream=stream)
        pose = np.stack([H.pose12(R, t)])
        states = {e: int(sb.step([img.ctypes.data], w, [100.0 + k / 20.0], pose)[0]) for e, sb in sbs.items()}
        assert states[A] == states[B], (tag, k, states)
        op = None
        if k >= 10 and rng.rand() < 0.25:
            op = rng.choice(["refine", "cull", "move", "stats"])
        seed = int(rng.randint(1 << 30))
        outs = {}
        for e, sb in sbs.items():
            r = np.random.RandomState(seed)
            if op == "refine":
                if sb.stats(0)["landmarks"] < 20:
                    outs[e] = None
                    continue
                try:
                    out7, kf = rc.refine(sb, iu.pose_b_c(), iters1=int(r.randint(1, 5)), iters2=int(r.randint(1, 8)))
                    outs[e] = (out7.tobytes(), kf.tobytes())
                except AssertionError as ex:  # (both engines must refuse alike)
                    outs[e] = ("refused", str(ex)[:40])
            elif op in ("cull", "stats"):
                T = cu.landmark_table(sb, 0)
                lists = [T["id"][r.rand(len(T["id"])) < 0.8]]
                outs[e] = (T["id"].tobytes(), T["pos"].tobytes(), cu.run_culling(sb, 1 if op == "stats" else 0, lists).tobytes())
            elif op == "move":
                T = cu.landmark_table(sb, 0)
                if len(T["id"]) == 0:
                    outs[e] = None
                    continue
                sel = np.arange(len(T["id"]))[:: int(r.randint(2, 6))]
                newpos = np.ascontiguousarray(T["pos"][sel] + r.normal(0, 1, (len(sel), 3)) * r.choice([0.05, 0.5, 50.0], (len(sel), 1)))
                ids = np.ascontiguousarray(T["id"][sel])
                assert sb.lib.icgh_batch_set_landmark_pos(C.c_void_p(sb.h_), 0, len(sel), ids.ctypes.data_as(C.c_void_p), newpos.ctypes.data_as(C.c_void_p)) == 0
                outs[e] = (ids.tobytes(), newpos.tobytes())
        if op:
            ops.append(op)
            assert outs[A] == outs[B], (tag, k, op)
        a, b = sbs[A].dump(0, 0), sbs[B].dump(0, 0)
        assert a == b, (tag, k, op, first_difference(a, b))
    for sb in sbs.values():
        sb.close()
    return (w, h, n, mf, window, stream, blank), ops


if __name__ == "__main__":
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    engines = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("table", "object")
    rng = np.random.RandomState(seed)
    lib = ensure_oracle_host()
    fails = 0
    for it in range(count):
        tag = f"soak_{seed}_{it}"
        try:
            params, ops = scenario(lib, rng, tag, engines)
            print("ok  ", tag, *params, "ops:", ",".join(ops) or "-", flush=True)
        except AssertionError as e:
            fails += 1
            print("FAIL", tag, str(e)[:600], flush=True)
    print(f"{count - fails} of {count} scenarios without divergence")
    sys.exit(1 if fails else 0)
