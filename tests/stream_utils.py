"""Shared helpers for the end-to-end stream tests."""
import os

import numpy as np

import harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_HOST = os.path.join(ROOT, "oracle", "libicgvins_host_oracle.so")


def ensure_oracle_host():
    import oracle_lib
    if not os.path.exists(ORACLE_HOST):
        oracle_lib.build()
    return ORACLE_HOST


def run_streams(lib_path, n_streams, w, h, n_frames, max_features, scene_frames=None, host_threads=1, stream_ids=None,
                window=10, groups=1):
    """Runs n_streams synthetic streams for n_frames; returns (per-frame records, stats, rendered frames)."""
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(lib_path, n_streams, w, h, cam, max_features=max_features, host_threads=host_threads, window=window,
                       groups=groups)
    stream_ids = list(range(n_streams)) if stream_ids is None else stream_ids
    if scene_frames is None:
        scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=4)
        scene_frames = {s: [scene.render(k, stream=s) for k in range(n_frames)] for s in stream_ids}
        scene_frames["scene"] = scene
    scene = scene_frames["scene"]
    records = []
    for k in range(n_frames):
        imgs = [scene_frames[s][k] for s in stream_ids]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for s in stream_ids])
        st = sb.step([im.ctypes.data for im in imgs], w, np.full(n_streams, 100.0 + k / 20.0), poses)
        rec = []
        for i in range(n_streams):
            ids, px = sb.features(i)
            rec.append((int(st[i]), ids.copy(), px.copy()))
        records.append(rec)
    stats = [sb.stats(i) for i in range(n_streams)]
    sb.close()
    return records, stats, scene_frames
