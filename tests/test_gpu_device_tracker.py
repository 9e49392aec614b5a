"""The device-resident tracker (icg_tracker_*, csrc/tracker.hip; ICG_TRACK_ENGINE=device) on the MI355X: the streams' state lives in HBM, the
stage kernels run the tracker core (host/track_core.h) between the segmented launches of the primitives, the host issues one chain of launches
per step and waits once.  Everything index-like must be the reference's: checked against the REFERENCE's own tracker (goldens), against the
oracle-backed table engine (states, ids, key-point bits, full state dumps downloaded from HBM) and through the map-writing entry points."""
import numpy as np
import pytest

import harness as H
from stream_utils import ensure_oracle_host, run_streams

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scenario", ["c1_640x480_100", "c2_1280x720_300", "c1_histgate", "c1_lost_and_reinit", "c1_lost_histgate", "c1_slow_second_new",
                                      "c4_1920x1080_500", "c1_bgr", "c1_long_160", "c2_long_60"])
def test_device_tracker_matches_reference_tracker_golden(scenario):
    """what the REFERENCE's own tracking.cc produced (tests/golden/tracking_ref_*.npz): track states, map-point ids, key-point float bits,
    candidate lists in list order, keyframe / window / landmark counts per frame — from blocks that never left the GPU except to be compared"""
    import ref_tracking_utils as rt
    rt.compare_scenario(H.HOST_LIB, scenario, engine="device")


@pytest.mark.parametrize("scenario", ["c1_640x480_100", "c1_lost_histgate", "c1_slow_second_new", "c2_long_60", "c4_1920x1080_500"])
def test_device_tracker_writes_the_reference_tracking_txt(scenario):
    """tracking.txt (reference tracking/tracking.cc:297-315): the keyframe decision's numbers travel in icg_tracker_result.log_* from the
    stage kernel on the MI355X, the executor writes the line — the reference tracker's own text in the deterministic columns"""
    import ref_tracking_utils as rt
    rt.compare_scenario(H.HOST_LIB, scenario, engine="device", with_log=True)


def _drive(lib, engine, w, h, nfeat, frames, poses, n_streams, dump_at, groups=1):
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(lib, n_streams, w, h, cam, max_features=nfeat, engine=engine, groups=groups)
    states, dumps = [], {}
    for k in range(len(frames[0])):
        st = sb.step([frames[s][k].ctypes.data for s in range(n_streams)], w, np.full(n_streams, 100.0 + k / 20.0), np.stack([poses[s][k] for s in range(n_streams)]))
        states.append([int(v) for v in st])
        if k in dump_at:
            dumps[k] = [sb.dump(s, 0) for s in range(n_streams)]
    stats = [sb.stats(s) for s in range(n_streams)]
    sb.close()
    return states, dumps, stats


@pytest.mark.parametrize("cfg", [(640, 480, 100, 36, 6), (1280, 720, 300, 20, 3)])
def test_device_tracker_state_dumps_equal_the_table_engine(cfg):
    """several streams in one tracker (one launch chain for all of them), full canonical state dumps — every live frame's rows in the
    reference container's order, every landmark with counters and observations, Map::landmarks_' iteration order, candidate lists — downloaded
    from HBM at several frames, against the oracle-backed table engine on the same frames"""
    import ctypes as C
    w, h, nfeat, n, ns = cfg
    cam = H.camera_for(w, h)
    scene = H.SynthScene(C.CDLL(ensure_oracle_host()), w, h, cam, tex_size=1024, threads=4)
    frames = [[scene.render(k, stream=40 + s) for k in range(n)] for s in range(ns)]
    poses = [[H.pose12(*scene.ins_pose(k, stream=40 + s)) for k in range(n)] for s in range(ns)]
    dump_at = {2, n // 2, n - 1}
    st_t, d_t, stats_t = _drive(ensure_oracle_host(), "table", w, h, nfeat, frames, poses, ns, dump_at)
    st_d, d_d, stats_d = _drive(H.HOST_LIB, "device", w, h, nfeat, frames, poses, ns, dump_at)
    assert st_t == st_d
    assert stats_t == stats_d
    for k in sorted(dump_at):
        for s in range(ns):
            if d_t[k][s] != d_d[k][s]:
                la, lb = d_t[k][s].splitlines(), d_d[k][s].splitlines()
                i = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), min(len(la), len(lb)))
                raise AssertionError(f"stream {s} after frame {k}, dump line {i}:\n table : {la[i] if i < len(la) else '<end>'}\n device: {lb[i] if i < len(lb) else '<end>'}")
    assert all(s["mappoints_created"] > 40 for s in stats_d)


def test_device_tracker_in_stream_groups_equals_oracle():
    """free-running groups, each with its own tracker and context: digests of every stream equal the oracle-backed host layer's"""
    w, h, nfeat, nframes, ns = 640, 480, 100, 14, 6
    rec_o, stats_o, frames = run_streams(ensure_oracle_host(), ns, w, h, nframes, nfeat)
    import os
    os.environ["ICG_TRACK_ENGINE"] = "device"
    try:
        rec_g, stats_g, _ = run_streams(H.HOST_LIB, ns, w, h, nframes, nfeat, scene_frames=frames, groups=3)
    finally:
        del os.environ["ICG_TRACK_ENGINE"]
    for s in range(ns):
        assert stats_o[s]["digest"] == stats_g[s]["digest"], s
        for k in range(nframes):
            assert rec_o[k][s][0] == rec_g[k][s][0]
            assert np.array_equal(rec_o[k][s][1], rec_g[k][s][1]) and np.array_equal(rec_o[k][s][2].view(np.uint32), rec_g[k][s][2].view(np.uint32))


def test_map_writers_on_the_device_tracker(oracle):
    """outlier culling on blocks in HBM: download -> object view -> absorb -> upload, then tracking continues on the uploaded blocks; outputs,
    per-observation flags and the states four frames later equal the object engine's (both on the HIP kernels)"""
    import cull_checks as cc
    a = cc.check_window_culling(H.HOST_LIB, oracle, engine="object")
    b = cc.check_window_culling(H.HOST_LIB, oracle, engine="device")
    assert a[0] == b[0] and a[1] == b[1]
    for x, y in zip(a[2], b[2]):
        assert x == y
