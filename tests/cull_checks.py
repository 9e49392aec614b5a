"""Shared body of the host-level f3 check (oracle-backed on CPU, HIP-backed on the GPU)."""
import numpy as np

import cull_utils as cu


def check_window_culling(lib_path, oracle, n_streams=3, n_frames=16, engine="table"):
    """tracks a few synthetic streams, moves some landmarks (as an optimizer write-back would), then runs
    reprojectionStatistics and gvinsOutlierCulling for all streams in one call each and compares counters, the removed landmark
    set and the features flagged as outliers with the Python restatement working on the raw landmark-graph dump"""
    import harness as H
    w, h = 640, 480
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(lib_path, n_streams, w, h, cam, max_features=100, window=10, engine=engine)  # icg::Map itself, or a view of the track table
    scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=4)
    for k in range(n_frames):
        frames = [scene.render(k, stream=s) for s in range(n_streams)]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for s in range(n_streams)])
        sb.step([f.ctypes.data for f in frames], w, np.full(n_streams, 100.0 + k / 20.0), poses)
    tables = [cu.landmark_table(sb, s) for s in range(n_streams)]
    assert all(len(T["id"]) > 40 for T in tables), [len(T["id"]) for T in tables]
    # optimizer write-back stand-in: every 4th landmark moves by 0.05..2 m, every 9th far away
    import ctypes as C
    rng = np.random.RandomState(11)
    for s, T in enumerate(tables):
        sel = np.arange(len(T["id"]))[::4]
        newpos = T["pos"][sel] + rng.normal(0, 1, (len(sel), 3)) * rng.choice([0.05, 0.3, 2.0], (len(sel), 1))
        newpos[::3] += np.array([0.0, 0.0, 400.0])
        ids = np.ascontiguousarray(T["id"][sel])
        rc = sb.lib.icgh_batch_set_landmark_pos(C.c_void_p(sb.h_), s, len(sel), ids.ctypes.data_as(C.c_void_p),
                                                np.ascontiguousarray(newpos).ctypes.data_as(C.c_void_p))
        assert rc == 0
    tables = [cu.landmark_table(sb, s) for s in range(n_streams)]
    lists = [T["id"][T["id"] % 5 != 0] for T in tables]  # landmarks "in the optimization"
    # statistics first (read-only)
    stats = cu.run_culling(sb, 1, lists)
    for s, T in enumerate(tables):
        exp = cu.expected_statistics(oracle, cam, T, lists[s])
        assert int(stats[s][4]) == exp[4], (s, stats[s], exp)
        assert np.abs(stats[s][:4] - np.array(exp[:4])).max() < 1e-12 * max(1.0, exp[1]), (s, stats[s], exp)
    out = cu.run_culling(sb, 0, lists)
    total = np.zeros(5, int)
    for s, T in enumerate(tables):
        c, removed, flagged = cu.expected_culling(oracle, cam, T, lists[s])
        assert list(out[s]) == c, (s, list(out[s]), c)
        total += np.array(c)
        after = cu.landmark_table(sb, s)
        assert set(int(x) for x in after["id"]) == set(int(x) for x in T["id"]) - removed, s
        # features flagged: compare per surviving landmark (removed landmarks are gone from the dump)
        pos_of = {int(i): k for k, i in enumerate(after["id"])}
        for (lid, j) in flagged:
            if lid in pos_of:
                k = after["off"][pos_of[lid]] + j
                assert after["obs_flags"][k] & 2, (s, lid, j)
        n_flagged_after = sum(int(((after["obs_flags"][after["off"][k]:after["off"][k + 1]] & 2) != 0).sum()) for k in range(len(after["id"])))
        n_flagged_before = sum(int(((T["obs_flags"][T["off"][k]:T["off"][k + 1]] & 2) != 0).sum()) for k in range(len(T["id"]))
                               if int(T["id"][k]) in pos_of)
        assert n_flagged_after - n_flagged_before == sum(1 for (lid, j) in flagged if lid in pos_of), s
    assert total[0] > 5 and total[1] > 0 and total[2] > 0 and total[3] > 0 and total[4] > 0, total  # every branch was exercised
    # the tracker goes on with the culled map; the state it ends in is returned for engine-vs-engine comparisons
    for k in range(n_frames, n_frames + 4):
        frames = [scene.render(k, stream=s) for s in range(n_streams)]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for s in range(n_streams)])
        sb.step([f.ctypes.data for f in frames], w, np.full(n_streams, 100.0 + k / 20.0), poses)
    final = [sb.dump(s, 0) for s in range(n_streams)]
    sb.close()
    return [list(o) for o in out], [list(x) for x in stats], final
