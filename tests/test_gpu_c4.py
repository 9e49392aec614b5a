"""BASELINE config C4 shapes on the GPU: 1920x1080, 500 features (grid 10x5, quota 10, min distance 52), 15-KF window
(7000 factors), 15 IMU intervals x 40 samples — parity vs the oracle at full size."""
import numpy as np
import pytest

import harness as H
import preint_data as pd
import reproj_data as rd
import synth
from stream_utils import ensure_oracle_host, run_streams
from test_gpu_geometry import grid_for

pytestmark = pytest.mark.gpu


def test_c4_frontend_ops_bit_exact(oracle):
    import icgvins
    w, h = 1920, 1080
    cam = H.camera_for(w, h)
    c = icgvins.Context(w, h, n_slots=2, max_batch=2, max_points=4096)
    try:
        c.set_camera(cam)
        a = synth.texture(w, h, seed=300)
        b = synth.shift_image(a, 4.5, -3.25)
        c.preprocess([0, 1], [a, b])
        ca, cb = oracle.clahe(a), oracle.clahe(b)
        assert np.array_equal(c.download(0, 0), ca)
        lv = cb
        for l in range(1, c.levels()):
            lv = oracle.pyrdown(lv)
            assert np.array_equal(c.download(1, l), lv)
        pts = synth.random_points(500, w, h, 4, seed=301)
        got, st = c.lk_track_fb(0, 1, pts, pts + np.float32(3.0))
        exp, est = oracle.lk_track_fb(ca, cb, pts, pts + np.float32(3.0))
        assert np.array_equal(st, est) and np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        assert est.sum() > 300
        grid = grid_for(w, h, 500)
        assert grid == [10, 5, 192, 216, 52, 10]
        q = np.full(50, 10, np.int32)
        out, cnt, blk = c.detect([0], grid, [0, 0], np.zeros((0, 2)), q, 600)
        ep, eb = oracle.detect(ca, grid, np.zeros((0, 2)), q, 600)
        assert cnt[0] == len(ep) > 200
        assert np.array_equal(blk[0, :cnt[0]], eb) and np.array_equal(out[0, :cnt[0]].view(np.uint32), ep.view(np.uint32))
    finally:
        c.close()


def test_c4_stream_parity():
    w, h, nfeat, nframes = 1920, 1080, 500, 6
    rec_o, stats_o, frames = run_streams(ensure_oracle_host(), 1, w, h, nframes, nfeat, window=15)
    rec_g, stats_g, _ = run_streams(H.HOST_LIB, 1, w, h, nframes, nfeat, scene_frames=frames, window=15)
    for k in range(nframes):
        assert rec_o[k][0][0] == rec_g[k][0][0]
        assert np.array_equal(rec_o[k][0][1], rec_g[k][0][1])
        assert np.array_equal(rec_o[k][0][2].view(np.uint32), rec_g[k][0][2].view(np.uint32))
    assert stats_o[0]["digest"] == stats_g[0]["digest"]
    assert stats_g[0]["mappoints_created"] > 100


def test_c4_backend_shapes(oracle):
    import icgvins
    c = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64, max_factors=8192)
    try:
        win = rd.make_window(500, 15, seed=15)
        args = (win["obs_soa"], win["idx_i"], win["idx_j"], win["idx_lm"], win["poses"], win["ext"], win["invdepth"], win["td"])
        assert win["obs_soa"].shape[1] > 6000
        r, J = c.reproj_eval(*args)
        re_, Je = oracle.reproj_eval(*args)
        assert np.abs(r - re_).max() <= 1e-9 * max(1, np.abs(re_).max()) and np.abs(J - Je).max() <= 1e-9 * max(1, np.abs(Je).max())
        lens = [41] * 15  # 200 Hz, 0.2 s keyframe spacing
        imus = [pd.make_interval(n, seed=40 + i) for i, n in enumerate(lens)]
        states = [pd.state(p=(i, 0, 0)) for i in range(15)]
        off = np.cumsum([0] + lens).astype(np.int32)
        cur, delta, jac, cov, dt, pn = c.preint_batch(1, off, np.concatenate(imus), np.stack(states), pd.PARAMS)
        for i in range(15):
            e = oracle.preint_integrate(1, imus[i], states[i], pd.PARAMS)
            assert np.abs(cur[i] - e["cur"]).max() <= 1e-9 * np.abs(e["cur"]).max()
            assert np.abs(cov[i] - e["cov"]).max() <= 1e-8 * np.abs(e["cov"]).max()
    finally:
        c.close()
