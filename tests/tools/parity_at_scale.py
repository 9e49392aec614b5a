#!/usr/bin/env python3
"""Parity witness at the MEASURED configuration: many streams in many free-running stream groups through
`icgh_batch_run` with device-resident frames, ICG_WAIT_POLL (StreamGroups sets it for > 1 group) and GPU_MAX_HW_QUEUES=20 — the exact
path that produces bench.py's `value` — compared stream by stream with the SAME frames run through the oracle-backed host layer
(oracle/libicgvins_host_oracle.so: the reference's per-frame algorithm, tracking/tracking.cc:144-245, on the CPU restatement).

Run as a script (so that GPU_MAX_HW_QUEUES is set before the HIP runtime starts):

    python tests/tools/parity_at_scale.py --streams 64 --groups 16 --frames 16 --c2-streams 8

prints one JSON line {"ok": bool, "cases": [...]}; exit code 0 iff every digest, state series and feature list agrees.
The oracle is the checker here, never the thing measured."""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import harness as H  # noqa: E402


def render_streams(lib, w, h, cam, stream_ids, n_frames, threads=8, tex_size=1024):
    scene = H.SynthScene(lib, w, h, cam, tex_size=tex_size, threads=threads)
    frames = {s: [scene.render(k, stream=s) for k in range(n_frames)] for s in stream_ids}
    poses = {s: [H.pose12(*scene.ins_pose(k, stream=s)) for k in range(n_frames)] for s in stream_ids}
    return frames, poses


def run_side(lib_path, w, h, nfeat, window, stream_ids, frames, poses, order, groups, host_threads, on_device, hip=None, stamp0=1000.0):
    """`order`: frame index (into the rendered ring) of every step.  Returns per-stream dicts: digest, states, final features."""
    cam = H.camera_for(w, h)
    n = len(stream_ids)
    sb = H.StreamBatch(lib_path, n, w, h, cam, max_features=nfeat, window=window, host_threads=host_threads, groups=groups)
    dev_ptrs, keep = [], []
    ctxh = C.c_void_p(sb.ctx_handle(0)) if on_device else None

    def addr(img):
        if not on_device:
            keep.append(img)
            return img.ctypes.data
        p = C.c_void_p()
        assert hip.icg_dev_alloc(ctxh, C.c_size_t(img.nbytes), C.byref(p)) == 0
        assert hip.icg_dev_upload(ctxh, p, img.ctypes.data_as(C.c_void_p), C.c_size_t(img.nbytes)) == 0
        dev_ptrs.append(p)
        return p.value

    ring = {s: [addr(im) for im in frames[s]] for s in stream_ids}
    K = len(order)
    ptrs = [[ring[s][f] for s in stream_ids] for f in order]
    P = np.stack([np.stack([poses[s][f] for s in stream_ids]) for f in order])
    stamps = np.stack([np.full(n, stamp0 + j / 20.0) for j in range(K)])
    t0 = time.perf_counter()
    states = sb.run(ptrs, w, stamps, P, on_device=on_device)
    wall = time.perf_counter() - t0
    out = []
    for i in range(n):
        ids, px = sb.features(i)
        cur, ref = sb.candidates(i)
        st = sb.stats(i)
        out.append({"digest": st["digest"], "states": states[:, i].copy(), "ids": ids.copy(), "px": px.copy(), "cand_cur": cur, "cand_ref": ref,
                    "frames": st["frames"], "keyframes": st["keyframes"], "mappoints": st["mappoints_created"]})
    n_groups = sb.n_groups()
    sb.close()
    for p in dev_ptrs:
        hip.icg_dev_free(ctxh, p)
    return out, wall, n_groups


def compare(gpu, orc):
    bad = []
    for i, (g, o) in enumerate(zip(gpu, orc)):
        why = None
        if not np.array_equal(g["states"], o["states"]):
            why = "track states differ at step %d" % int(np.nonzero(g["states"] != o["states"])[0][0])
        elif g["digest"] != o["digest"]:
            why = "digest"
        elif not np.array_equal(g["ids"], o["ids"]):
            why = "map-point ids of the last frame"
        elif not np.array_equal(g["px"].view(np.uint32), o["px"].view(np.uint32)):
            why = "key-point bits of the last frame"
        elif not (np.array_equal(g["cand_cur"].view(np.uint32), o["cand_cur"].view(np.uint32))
                  and np.array_equal(g["cand_ref"].view(np.uint32), o["cand_ref"].view(np.uint32))):
            why = "candidate lists"
        if why:
            bad.append({"stream": i, "why": why})
    return bad


def case(hip, oracle_host, w, h, nfeat, window, stream_ids, n_frames, groups, oracle_threads):
    """hip None: CPU self-test of this tool (both sides on the oracle-backed host layer, grouped vs ungrouped)"""
    cam = H.camera_for(w, h)
    tools = C.CDLL(H.TOOLS_LIB if hip is not None else oracle_host)
    frames, poses = render_streams(tools, w, h, cam, stream_ids, n_frames, threads=min(16, os.cpu_count() or 1))
    order = list(range(n_frames))
    if hip is not None:
        gpu, wall_g, ng = run_side(H.HOST_LIB, w, h, nfeat, window, stream_ids, frames, poses, order, groups, 1, True, hip=hip)
    else:
        gpu, wall_g, ng = run_side(oracle_host, w, h, nfeat, window, stream_ids, frames, poses, order, groups, 1, False)
    orc, wall_o, _ = run_side(oracle_host, w, h, nfeat, window, stream_ids, frames, poses, order, 1, oracle_threads, False)
    bad = compare(gpu, orc)
    tracking = int(sum(int((g["states"] == 2).sum()) for g in gpu))
    return {"config": f"{w}x{h}/{nfeat}", "streams": len(stream_ids), "groups": ng, "frames": n_frames, "mismatches": bad,
            "tracking_states": tracking, "keyframes": int(sum(g["keyframes"] for g in gpu)), "mappoints": int(sum(g["mappoints"] for g in gpu)),
            "gpu_wall_s": round(wall_g, 3), "oracle_wall_s": round(wall_o, 3), "wait_mode": "ICG_WAIT_POLL (StreamGroups, > 1 group)",
            "frames_resident_in_hbm": True, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--groups", type=int, default=16)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--c2-streams", type=int, default=8)
    ap.add_argument("--c2-groups", type=int, default=4)
    ap.add_argument("--cpu-selftest", action="store_true", help="no GPU: both sides on the oracle-backed host layer (grouped vs one batch)")
    ap.add_argument("--oracle-threads", type=int, default=max(1, min(16, len(os.sched_getaffinity(0)))))
    args = ap.parse_args()
    import icgvins
    from stream_utils import ensure_oracle_host
    hip = None if args.cpu_selftest else icgvins.load_library()
    oracle_host = ensure_oracle_host()
    cases = [case(hip, oracle_host, 640, 480, 100, 10, list(range(args.streams)), args.frames, args.groups, args.oracle_threads)]
    if args.c2_streams > 0:
        cases.append(case(hip, oracle_host, 1280, 720, 300, 10, list(range(args.c2_streams)), args.frames, args.c2_groups, args.oracle_threads))
    ok = all(not c["mismatches"] for c in cases)
    print(json.dumps({"ok": ok, "cases": cases}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
