"""The two engines of TrackingBatch — the track table (track_table.h, the throughput path) and the reference-shaped object graph
(icg::Tracking + Map + WindowKeeper) — must hold the SAME state after every frame: tracker state, candidate lists, window, every live
frame's features in container order (ids, key-point bits, velocities), every landmark with position, reference frame, counters and
observation list.  Checked on the canonical text dumps (floats as bit patterns), oracle-backed (CPU).  Also: the table engine's B2 view
(materialize(): real icg::Map / Frame / Feature / MapPoint objects built from the table) dumps to the same text as the table itself, and
HashOrder reproduces std::unordered_map's iteration order."""
import ctypes as C

import numpy as np
import pytest

import harness as H
from stream_utils import ensure_oracle_host


def _drive(engine, w, h, nfeat, n_frames, frames, poses, blank=(), check_hist=False, window=10, dump_every=1):
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(ensure_oracle_host(), 1, w, h, cam, max_features=nfeat, window=window, engine=engine, check_hist=check_hist)
    assert sb.engine() == engine
    dumps, states = [], []
    for k in range(n_frames):
        img = frames[k]
        st = sb.step([img.ctypes.data], w, [100.0 + k / 20.0], poses[k])
        states.append(int(st[0]))
        if k % dump_every == 0 or k == n_frames - 1:
            dumps.append((k, sb.dump(0, 0), sb.dump(0, 1) if engine == "table" else None, sb.dump(0, 2) if engine == "table" else None))
    stats = sb.stats(0)
    sb.close()
    return states, dumps, stats


def _scene(w, h, n_frames, stream, blank=(), blank_value=90, slow_after=None):
    cam = H.camera_for(w, h)
    lib = C.CDLL(ensure_oracle_host())
    scene = H.SynthScene(lib, w, h, cam, tex_size=1024, threads=4)
    warp = (lambda k: float(k)) if slow_after is None else (lambda k: float(k) if k < slow_after[0] else slow_after[0] + (k - slow_after[0]) * slow_after[1])
    frames = [scene.render(warp(k), stream=stream) for k in range(n_frames)]
    for k in blank:
        frames[k] = np.full_like(frames[k], blank_value)
    poses = []
    for k in range(n_frames):
        R, t = scene.pose(warp(k), stream=stream)
        rng = np.random.RandomState(9000 + k)
        poses.append(H.pose12(R @ H._rot_yp(*rng.normal(0, np.deg2rad(0.1), 2)), t + rng.normal(0, 0.02, 3)))
    return frames, poses


CASES = {
    # name: (w, h, features, frames, stream, blank frames, histogram gate, slow phase)
    "c1_window_rolls": (640, 480, 100, 70, 11, (), False, None),
    "c1_lost_and_reinit": (640, 480, 100, 34, 12, (14, 15, 16), False, None),
    "c1_lost_histgate": (640, 480, 100, 34, 13, (14, 15, 16), True, None),
    "c1_slow_second_new": (640, 480, 100, 40, 14, (), False, (8, 0.02)),
    "c2": (1280, 720, 300, 16, 15, (), False, None),
    "c4_window15": (1920, 1080, 500, 44, 16, (), False, None),  # BASELINE.json configs[3]: 15-keyframe window
}
WINDOW = {"c4_window15": 15}


@pytest.mark.parametrize("name", list(CASES))
def test_table_engine_state_equals_object_engine(name):
    w, h, nfeat, n, stream, blank, hist, slow = CASES[name]
    frames, poses = _scene(w, h, n, stream, blank=blank, blank_value=235 if hist else 90, slow_after=slow)
    win = WINDOW.get(name, 10)
    st_t, d_t, stats_t = _drive("table", w, h, nfeat, n, frames, poses, check_hist=hist, window=win)
    st_o, d_o, stats_o = _drive("object", w, h, nfeat, n, frames, poses, check_hist=hist, window=win)
    assert st_t == st_o
    assert stats_t == stats_o
    for (k, full_t, map_t, mat_t), (_, full_o, _, _) in zip(d_t, d_o):
        if full_t != full_o:
            lt, lo = full_t.splitlines(), full_o.splitlines()
            first = next((i for i in range(min(len(lt), len(lo))) if lt[i] != lo[i]), min(len(lt), len(lo)))
            raise AssertionError(f"{name}: engines differ after frame {k}, dump line {first}:\n table : {lt[first] if first < len(lt) else '<end>'}\n"
                                 f" object: {lo[first] if first < len(lo) else '<end>'}")
        # B2 view: the materialized object graph says the same as the table
        if map_t != mat_t:
            lt, lo = map_t.splitlines(), mat_t.splitlines()
            first = next((i for i in range(min(len(lt), len(lo))) if lt[i] != lo[i]), min(len(lt), len(lo)))
            raise AssertionError(f"{name}: materialized view differs after frame {k}, line {first}:\n table       : {lt[first] if first < len(lt) else '<end>'}\n"
                                 f" materialized: {lo[first] if first < len(lo) else '<end>'}")
    assert 2 in st_t  # TRACK_TRACKING reached
    if blank:
        assert 4 in st_t  # TRACK_LOST exercised
    assert stats_t["mappoints_created"] > 40


def test_hash_order_equals_std_unordered_map():
    lib = C.CDLL(ensure_oracle_host())
    lib.icgh_hashorder_selftest.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
    for seed in range(24):
        for dense in (0, 1):
            assert lib.icgh_hashorder_selftest(seed, 900, 1 if seed < 4 else 41, dense) == 0, (seed, dense)


def _first_difference(a, b):
    la, lb = a.splitlines(), b.splitlines()
    i = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), min(len(la), len(lb)))
    return i, la[i] if i < len(la) else "<end>", lb[i] if i < len(lb) else "<end>"


def test_culling_through_the_object_view_equals_the_object_engine(oracle):
    """icgh_batch_culling on the table engine works on a lazily built object view (TableTracker::view) and writes the result back
    (absorb): outputs, per-observation flags (checked inside against the python restatement) and the tracker state four frames
    later are the object engine's, bit for bit"""
    import cull_checks as cc
    lib = ensure_oracle_host()
    a = cc.check_window_culling(lib, oracle, engine="object")
    b = cc.check_window_culling(lib, oracle, engine="table")
    assert a[0] == b[0] and a[1] == b[1]
    for s, (x, y) in enumerate(zip(a[2], b[2])):
        assert x == y, (s, _first_difference(x, y))


def test_refinement_through_the_object_view_equals_the_object_engine():
    """same for icgh_batch_refine_windows (VisualWindow gather -> WindowSolver -> write-back -> WindowCulling): keyframe poses,
    landmark positions, outlier flags and removals land in the track table exactly as they land in icg::Map"""
    import refine_checks as rc
    lib = ensure_oracle_host()
    ra, oa, fa = rc.check_refinement(lib, engine="object")
    rb, ob, fb = rc.check_refinement(lib, engine="table")
    assert np.array_equal(oa, ob), (oa, ob)
    assert ra == rb
    for s, (x, y) in enumerate(zip(fa, fb)):
        assert x == y, (s, _first_difference(x, y))


# ---- the tracker core (track_core.h): the stage bodies of the device-resident tracker, compiled for the host ------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_core_engine_state_equals_table_engine(name):
    """ICG_TRACK_ENGINE=core runs the SAME source the stage kernels of the device-resident tracker are compiled from (tracker core: flat
    per-stream block, fixed capacities) between the same batched device calls.  After every frame its state — imported back into the table
    members — must dump to the table engine's text: tracker scalars, candidate lists, window, every live frame's rows in container order,
    every landmark with counters and observation list, and Map::landmarks_' iteration order (replayed from the core's operation log).  The
    digest / statistics the core computes in its end-of-frame stage must equal the executor's."""
    w, h, nfeat, n, stream, blank, hist, slow = CASES[name]
    frames, poses = _scene(w, h, n, stream, blank=blank, blank_value=235 if hist else 90, slow_after=slow)
    win = WINDOW.get(name, 10)
    st_t, d_t, stats_t = _drive("table", w, h, nfeat, n, frames, poses, check_hist=hist, window=win)
    st_c, d_c, stats_c = _drive("core", w, h, nfeat, n, frames, poses, check_hist=hist, window=win)
    assert st_t == st_c
    for (k, full_t, _, _), (_, full_c, _, _) in zip(d_t, d_c):
        if full_t != full_c:
            i, a, b = _first_difference(full_t, full_c)
            raise AssertionError(f"{name}: core differs from the table after frame {k}, dump line {i}:\n table: {a}\n core : {b}")
    assert stats_t == stats_c


def test_map_writers_on_the_core_engine_equal_the_object_engine(oracle):
    """culling and window refinement on the core engine: the object view is built from the block's imported image, absorb() writes the image
    back into the block (exportCore), and the tracker continues on it — outputs and the states four frames later equal the object engine's"""
    import cull_checks as cc
    import refine_checks as rc
    lib = ensure_oracle_host()
    a = cc.check_window_culling(lib, oracle, engine="object")
    b = cc.check_window_culling(lib, oracle, engine="core")
    assert a[0] == b[0] and a[1] == b[1]
    for s, (x, y) in enumerate(zip(a[2], b[2])):
        assert x == y, (s, _first_difference(x, y))
    ra, oa, fa = rc.check_refinement(lib, engine="object")
    rb, ob, fb = rc.check_refinement(lib, engine="core")
    assert np.array_equal(oa, ob), (oa, ob)
    assert ra == rb
    for s, (x, y) in enumerate(zip(fa, fb)):
        assert x == y, (s, _first_difference(x, y))


# ---- the device engine's HOST side, on the CPU backend of the tracker ABI (oracle/abi_shim.cc) ----------------------------------------------
@pytest.mark.parametrize("name", ["c1_window_rolls", "c1_lost_histgate", "c2"])
def test_device_engine_host_side_equals_table_engine(name):
    """ICG_TRACK_ENGINE=device: TrackingBatch::stepDevice drives icg_tracker_* — here the shim's CPU backend, the same stage bodies between the
    oracle's primitives — and everything the host keeps for it: results -> statistics / digest, block download -> import -> dump on demand,
    the landmark-history cursor across downloads.  Same text as the table engine after every frame."""
    w, h, nfeat, n, stream, blank, hist, slow = CASES[name]
    frames, poses = _scene(w, h, n, stream, blank=blank, blank_value=235 if hist else 90, slow_after=slow)
    st_t, d_t, stats_t = _drive("table", w, h, nfeat, n, frames, poses, check_hist=hist)
    st_d, d_d, stats_d = _drive("device", w, h, nfeat, n, frames, poses, check_hist=hist, dump_every=3)
    assert st_t == st_d
    by_frame = {k: full for k, full, _, _ in d_t}
    for k, full_d, _, _ in d_d:
        if by_frame[k] != full_d:
            i, a, b = _first_difference(by_frame[k], full_d)
            raise AssertionError(f"{name}: device engine differs from the table after frame {k}, dump line {i}:\n table : {a}\n device: {b}")
    assert stats_t == stats_d


def test_map_writers_on_the_device_engine_equal_the_object_engine(oracle):
    """culling and window refinement with the blocks behind the tracker ABI: download -> view -> absorb -> export -> upload, then the tracker
    continues on the uploaded block"""
    import cull_checks as cc
    import refine_checks as rc
    lib = ensure_oracle_host()
    a = cc.check_window_culling(lib, oracle, engine="object")
    b = cc.check_window_culling(lib, oracle, engine="device")
    assert a[0] == b[0] and a[1] == b[1]
    for s, (x, y) in enumerate(zip(a[2], b[2])):
        assert x == y, (s, _first_difference(x, y))
    ra, oa, fa = rc.check_refinement(lib, engine="object")
    rb, ob, fb = rc.check_refinement(lib, engine="device")
    assert np.array_equal(oa, ob), (oa, ob)
    assert ra == rb
    for s, (x, y) in enumerate(zip(fa, fb)):
        assert x == y, (s, _first_difference(x, y))


def test_device_engine_landmark_history_drains(monkeypatch):
    """the landmark-container history of a block is fetched (log only) and replayed into the host's Map::landmarks_ twin whenever it passes
    the drain threshold — here every ~30 operations instead of every 4096: the iteration order in the dumps (line "O buckets=... order=...")
    must still be the table engine's after every frame, with downloads of the block interleaved at other frames"""
    monkeypatch.setenv("ICG_TRACKER_LOG_DRAIN", "30")
    w, h, nfeat, n, stream, blank, hist, slow = CASES["c1_window_rolls"]
    frames, poses = _scene(w, h, n, stream)
    st_t, d_t, stats_t = _drive("table", w, h, nfeat, n, frames, poses)
    st_d, d_d, stats_d = _drive("device", w, h, nfeat, n, frames, poses, dump_every=7)
    assert st_t == st_d and stats_t == stats_d
    by_frame = {k: full for k, full, _, _ in d_t}
    for k, full_d, _, _ in d_d:
        assert by_frame[k] == full_d, (k, _first_difference(by_frame[k], full_d))


def test_core_container_order_equals_std_unordered_map():
    """tc::order_extend — the batched, scratch-resident insertion the stage kernels run (fresh frames: all rows at once; triangulation: a few
    rows appended to an existing order) — against a real std::unordered_map, across every rehash threshold up to the row capacity"""
    lib = C.CDLL(ensure_oracle_host())
    lib.icgh_core_order_selftest.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
    for seed in range(12):
        for n_first, n_more, rounds in ((0, 1, 60), (240, 17, 20), (1, 0, 0), (13, 1, 30), (300, 50, 6), (639, 1, 1), (58, 1, 4), (127, 130, 3)):
            assert lib.icgh_core_order_selftest(seed, n_first, n_more, rounds) == 0, (seed, n_first, n_more, rounds)


@pytest.mark.parametrize("scenario", ["c1_640x480_100", "c1_lost_histgate", "c1_slow_second_new", "c2_long_60"])
def test_device_engine_writes_the_reference_trackers_log(scenario):
    """tracking.txt (tracking.cc:309-315) of the device-resident tracker: the keyframe decision's numbers travel in icg_tracker_result
    (log_valid, log_features, log_data) and the host executor writes the line — same text as the REFERENCE's own tracker wrote, in the six
    deterministic columns (here on the shim's CPU backend of icg_tracker_*)."""
    import ref_tracking_utils as rt
    if scenario not in rt.SCENARIOS:
        pytest.skip("no such golden scenario")
    rt.compare_scenario(ensure_oracle_host(), scenario, engine="device", with_log=True)
