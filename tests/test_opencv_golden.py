"""Oracle (and, on a GPU box, the HIP path) against golden vectors from REAL OpenCV (tests/golden/opencv_golden.npz, produced by
tests/golden/make_opencv_golden.py on any box that has cv2).  The build image has no OpenCV and no network, so until that file is
committed every test here SKIPS with that reason — the skip is the visible marker that the OpenCV boundary is still unpinned
(oracle/README.md "Deviations").  Bounds per primitive (why each is exact or toleranced) are in oracle/README.md."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(GOLDEN),
                                reason="tests/golden/opencv_golden.npz absent: run tests/golden/make_opencv_golden.py where cv2 is installed "
                                       "(OpenCV boundary parity UNPINNED until then)")

CONFIGS = ["c1", "c2", "c4"]


def test_manifest(G):
    """the file is complete and comes from a supported OpenCV (tests/golden/README.md lists keys, dtypes and shapes)"""
    major, minor = [int(v) for v in str(G["opencv_version"]).split(".")[:2]]
    assert (major, minor) >= (4, 5) and major < 5, str(G["opencv_version"])
    sizes = {"c1": (640, 480, 100), "c2": (1280, 720, 300), "c4": (1920, 1080, 500)}
    for tag, (w, h, n) in sizes.items():
        for key, dtype, shape in ((f"{tag}_clahe_a", np.uint8, (h, w)), (f"{tag}_clahe_b", np.uint8, (h, w)), (f"{tag}_pyr3", np.uint8, None),
                                  (f"{tag}_lk_prev", np.float32, (n, 2)), (f"{tag}_lk_next", np.float32, (n, 2)), (f"{tag}_lk_status", None, (n,)),
                                  (f"{tag}_lk_err", np.float32, (n,)), (f"{tag}_undist_out", np.float32, (n, 2)), (f"{tag}_det_grid", np.int32, (6,)),
                                  (f"{tag}_det_mask", np.uint8, (h, w)), (f"{tag}_det_pts", np.float32, None), (f"{tag}_det_block", np.int32, None),
                                  (f"{tag}_fm_p1", np.float32, None), (f"{tag}_fm_mask", np.uint8, None), (f"{tag}_fm_F", np.float64, (3, 3))):
            assert key in G.files, key
            if dtype is not None:
                assert G[key].dtype == dtype, (key, G[key].dtype)
            if shape is not None:
                assert G[key].shape == shape, (key, G[key].shape)


@pytest.fixture(scope="module")
def G():
    return np.load(GOLDEN)


def _inputs(tag):
    import synth
    w, h = {"c1": (640, 480), "c2": (1280, 720), "c4": (1920, 1080)}[tag]
    a = synth.texture(w, h, seed=21)
    return a, synth.shift_image(a, 3.25, -1.75), w, h


@pytest.mark.parametrize("tag", CONFIGS)
def test_clahe_bit_exact(oracle, G, tag):
    a, b, _, _ = _inputs(tag)
    assert np.array_equal(oracle.clahe(a), G[f"{tag}_clahe_a"])
    assert np.array_equal(oracle.clahe(b), G[f"{tag}_clahe_b"])


@pytest.mark.parametrize("tag", CONFIGS)
def test_pyramid_bit_exact(oracle, G, tag):
    img = G[f"{tag}_clahe_a"]
    assert np.array_equal(img, G[f"{tag}_pyr0"])
    for lvl in (1, 2, 3):
        img = oracle.pyrdown(img)
        assert np.array_equal(img, G[f"{tag}_pyr{lvl}"]), f"level {lvl}"


@pytest.mark.parametrize("tag", CONFIGS)
def test_lk_status_and_points(oracle, G, tag):
    """status identical; positions within 2e-3 px: OpenCV accumulates the 441-term window sums in float (SIMD lane order), the
    oracle in exact integers with one rounding — the two differ by float rounding of sums of ~1e6-magnitude terms"""
    ca, cb = G[f"{tag}_clahe_a"], G[f"{tag}_clahe_b"]
    nxt, st, err = oracle.lk_track(ca, cb, G[f"{tag}_lk_prev"], G[f"{tag}_lk_guess"])
    exp_st = G[f"{tag}_lk_status"].astype(np.uint8)
    flips = int((st != exp_st).sum())
    assert flips <= max(1, len(st) // 200), f"{flips} status flips (threshold decisions at minEig / window border)"
    ok = (st == 1) & (exp_st == 1)
    assert np.abs(nxt[ok] - G[f"{tag}_lk_next"][ok]).max() < 2e-3
    assert np.abs(err[ok] - G[f"{tag}_lk_err"][ok]).max() < 1e-3 * max(1.0, float(G[f"{tag}_lk_err"][ok].max()))


@pytest.mark.parametrize("tag", CONFIGS)
def test_undistort_points(oracle, G, tag):
    import harness as H  # noqa: F401  (camera_for lives in the package harness)
    w = {"c1": 640, "c2": 1280, "c4": 1920}[tag]
    h = {"c1": 480, "c2": 720, "c4": 1080}[tag]
    cam = H.camera_for(w, h)
    got = oracle.undistort(cam, G[f"{tag}_undist_in"])
    assert np.array_equal(got.view(np.uint32), G[f"{tag}_undist_out"].view(np.uint32))  # double math, one rounding to float


@pytest.mark.parametrize("tag", CONFIGS)
def test_gridded_detection(oracle, G, tag):
    """corner SET, ORDER and block assignment identical; sub-pixel coordinates within 1e-3 px (OpenCV's 8u->32f getRectSubPix runs
    an algebraically equal recurrence)"""
    cols, rows, bw, bh, quota, min_dist = [int(v) for v in G[f"{tag}_det_grid"]]
    ca = G[f"{tag}_clahe_a"]
    pts, blk = oracle.detect(ca, [cols, rows, bw, bh, min_dist, quota], G[f"{tag}_det_exist"], np.full(cols * rows, quota, np.int32),
                             cols * rows * quota)
    assert np.array_equal(blk, G[f"{tag}_det_block"])
    assert np.abs(pts - G[f"{tag}_det_pts"]).max() < 1e-3


@pytest.mark.parametrize("tag", CONFIGS)
def test_fm_ransac_mask(oracle, G, tag):
    """inlier mask identical: same RNG stream, same subset rejection, same root order, double scoring with float compare"""
    ok, mask, _, _ = oracle.fm_ransac(G[f"{tag}_fm_p1"], G[f"{tag}_fm_p2"], 1.5, 0.99)
    assert ok == 1
    assert np.array_equal(mask, G[f"{tag}_fm_mask"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CONFIGS)
def test_hip_path_against_real_opencv_on_the_gpu_box(G, tag):
    """Selected by `-m gpu`, so the GPU summary carries the marker too: while the golden is absent this SKIPS (visible as skipped with the
    reason above) — on the GPU lease there is no cv2, no wheel in /opt/wheelhouse, no pip index, no apt source and no network either
    (profiles/archive/r04_opencv_probe.txt, one gpurun call of round 4).  With the golden present the WHOLE HIP front-end is pinned at once, through the C ABI
    (VERDICT r4 item 7), with the bounds the oracle tests above state per primitive:
      CLAHE (tracking.cc:63,139) and the LK pyramid levels: bit-exact; calcOpticalFlowPyrLK (:385-393, 487-496): status identical up to
      threshold decisions, positions within 2e-3 px; undistortPoints (camera.cc:72-74): float bit patterns; goodFeaturesToTrack +
      cornerSubPix per block (:647-652): corner set, order and block assignment identical, coordinates within 1e-3 px;
      findFundamentalMat (:548): inlier mask identical — on the host-looped entry point and on the one-launch kernel of the device-resident tracker."""
    import harness as H
    import icgvins
    a, b, w, h = _inputs(tag)
    n = {"c1": 100, "c2": 300, "c4": 500}[tag]
    c = icgvins.Context(w, h, n_slots=2, max_batch=2, max_points=max(1024, 4 * n))
    try:
        c.set_camera(H.camera_for(w, h))
        c.preprocess([0, 1], [a, b])
        assert np.array_equal(c.download(0, 0), G[f"{tag}_clahe_a"]) and np.array_equal(c.download(1, 0), G[f"{tag}_clahe_b"])
        for lvl in (1, 2, 3):
            assert np.array_equal(c.download(0, lvl), G[f"{tag}_pyr{lvl}"]), f"pyramid level {lvl}"
        # F2
        prev, guess = G[f"{tag}_lk_prev"], G[f"{tag}_lk_guess"]
        zeros, ones = np.zeros(len(prev), np.int32), np.ones(len(prev), np.int32)
        nxt, st, err = c.lk_track(zeros, ones, prev, guess)
        exp_st = G[f"{tag}_lk_status"].astype(np.uint8)
        flips = int((st != exp_st).sum())
        assert flips <= max(1, len(st) // 200), f"{flips} LK status flips"
        ok = (st == 1) & (exp_st == 1)
        assert np.abs(nxt[ok] - G[f"{tag}_lk_next"][ok]).max() < 2e-3
        assert np.abs(err[ok] - G[f"{tag}_lk_err"][ok]).max() < 1e-3 * max(1.0, float(G[f"{tag}_lk_err"][ok].max()))
        # F4
        got = c.undistort(G[f"{tag}_undist_in"])
        assert np.array_equal(got.view(np.uint32), G[f"{tag}_undist_out"].view(np.uint32))
        # F7
        cols, rows, bw, bh, quota, min_dist = [int(v) for v in G[f"{tag}_det_grid"]]
        exist = np.asarray(G[f"{tag}_det_exist"], np.float32).reshape(-1, 2)
        cap = cols * rows * quota
        out, cnt, blk = c.detect([0], [cols, rows, bw, bh, min_dist, quota], [0, len(exist)], exist, np.full(cols * rows, quota, np.int32), cap)
        exp_blk = G[f"{tag}_det_block"]
        assert cnt[0] == len(exp_blk) and np.array_equal(blk[0, :cnt[0]], exp_blk)
        assert np.abs(out[0, :cnt[0]] - G[f"{tag}_det_pts"]).max() < 1e-3
        # F6
        p1, p2 = G[f"{tag}_fm_p1"], G[f"{tag}_fm_p2"]
        off = np.array([0, len(p1)], np.int32)
        assert np.array_equal(c.fm_ransac(off, p1, p2, 1.5, 0.99), G[f"{tag}_fm_mask"])
        assert np.array_equal(c.fm_ransac_device(off, p1, p2, 1.5, 0.99), G[f"{tag}_fm_mask"])
    finally:
        c.close()
