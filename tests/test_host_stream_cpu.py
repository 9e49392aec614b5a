"""CPU tests of the host layer (Tracking state machine, TrackingBatch, WindowKeeper) linked against the oracle ABI shim:
exercises exactly the host code that ships, with the CPU restatement standing in for the GPU."""
import numpy as np
import pytest

import harness as H
from stream_utils import ensure_oracle_host, run_streams


def test_single_stream_initialises_and_tracks():
    lib = ensure_oracle_host()
    rec, stats, _ = run_streams(lib, 1, 640, 480, 24, 100)
    states = [H.TRACK_STATES[r[0][0]] for r in rec]
    assert states[0] == "FIRST_FRAME"
    assert "INITIALIZING" in states and states[-1] == "TRACKING"
    first_tracking = states.index("TRACKING")
    assert 1 < first_tracking < 12
    assert all(s == "TRACKING" for s in states[first_tracking:])
    # map points were triangulated and keep being tracked
    assert stats[0]["mappoints_created"] > 50
    assert len(rec[-1][0][1]) > 30
    # ids are stable: features of consecutive frames share most map-point ids
    a, b = set(rec[-2][0][1].tolist()), set(rec[-1][0][1].tolist())
    assert len(a & b) > 0.5 * len(b)
    assert stats[0]["keyframes"] >= 3 and stats[0]["window_keyframes"] <= 11


def test_window_keeper_bounds_the_window():
    lib = ensure_oracle_host()
    rec, stats, _ = run_streams(lib, 1, 640, 480, 40, 100, window=3)
    assert stats[0]["keyframes"] > 4
    assert stats[0]["window_keyframes"] <= 4


def test_batch_equals_individual_streams():
    """Shard invariance: 3 streams in one lock-step batch == the same streams run one by one,
    with and without host threads."""
    lib = ensure_oracle_host()
    rec_b, stats_b, frames = run_streams(lib, 3, 640, 480, 14, 100)
    rec_t, stats_t, _ = run_streams(lib, 3, 640, 480, 14, 100, scene_frames=frames, host_threads=3)
    rec_g, stats_g, _ = run_streams(lib, 3, 640, 480, 14, 100, scene_frames=frames, groups=2)  # group threads
    for s in range(3):
        assert stats_g[s]["digest"] == stats_b[s]["digest"]
        rec_1, stats_1, _ = run_streams(lib, 1, 640, 480, 14, 100, scene_frames=frames, stream_ids=[s])
        assert stats_1[0]["digest"] == stats_b[s]["digest"] == stats_t[s]["digest"]
        for k in range(14):
            assert rec_1[k][0][0] == rec_b[k][s][0]
            assert np.array_equal(rec_1[k][0][1], rec_b[k][s][1])
            assert np.array_equal(rec_1[k][0][2].view(np.uint32), rec_b[k][s][2].view(np.uint32))
    # different streams see different imagery
    assert stats_b[0]["digest"] != stats_b[1]["digest"]


@pytest.mark.parametrize("scenario", ["c1_640x480_100", "c2_1280x720_300", "c1_histgate", "c1_lost_and_reinit", "c1_lost_histgate", "c1_slow_second_new", "c4_1920x1080_500", "c1_bgr", "c1_long_160", "c2_long_60"])
def test_host_layer_matches_reference_tracker_golden(scenario):
    """The product's host layer (icg::Tracking on the oracle primitives) reproduces, frame by frame and bit for bit, what the
    REFERENCE's own tracking.cc produced on the same primitives (tests/golden/tracking_ref_*.npz, generator
    tests/golden/make_tracking_golden.py): track states, map-point ids, key-point floats, window bookkeeping."""
    import ref_tracking_utils as rt
    rt.compare_scenario(ensure_oracle_host(), scenario)


def test_oracle_inner_threads_give_identical_results():
    """ICG_ORACLE_INNER_THREADS (bench.py's cpu_baseline_reference_decomposition leg: LK points and detection blocks in parallel inside one
    stream, the reference's own CPU decomposition) changes the schedule only: same tracker state, same digest"""
    import os

    import numpy as np

    import harness as H
    from stream_utils import ensure_oracle_host
    w, h = 640, 480
    cam = H.camera_for(w, h)

    def run(threads):
        if threads > 1:
            os.environ["ICG_ORACLE_INNER_THREADS"] = str(threads)
        try:
            sb = H.StreamBatch(ensure_oracle_host(), 1, w, h, cam, max_features=100, window=10)
            scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=4)
            for k in range(14):
                img = scene.render(k, stream=21)
                sb.step([img.ctypes.data], w, [100.0 + k / 20.0], np.stack([H.pose12(*scene.ins_pose(k, stream=21))]))
            out = (sb.dump(0, 0), sb.stats(0)["digest"])
            sb.close()
            return out
        finally:
            os.environ.pop("ICG_ORACLE_INNER_THREADS", None)

    assert run(1) == run(5)
