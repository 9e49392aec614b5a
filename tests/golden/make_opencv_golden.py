#!/usr/bin/env python3
"""Golden vectors of the five OpenCV primitives on the reference's hot path, from REAL OpenCV (run this where `cv2` exists):

    python tests/golden/make_opencv_golden.py            # writes tests/golden/opencv_golden.npz

The image this repo is built in has no OpenCV (no headers, no wheel, no network), so the file is NOT committed yet and
tests/test_opencv_golden.py skips — visibly — until someone runs this script on any box with opencv-python (>= 4.5) and commits the
result.  That closes SURVEY.md 8(c) for the front-end ("parity unpinned" at the OpenCV boundary).

Each primitive is called exactly as the reference calls it:
  CLAHE        cv::createCLAHE(3.0, Size(21,21))->apply                tracking/tracking.cc:63,139
  pyramid      cv::buildOpticalFlowPyramid (what calcOpticalFlowPyrLK builds internally, maxLevel 3, 21x21)
  LK           cv::calcOpticalFlowPyrLK(prev, cur, pts, guess, status, err, Size(21,21), 3,
               TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW)   tracking.cc:385-393, 487-496 (minEigThreshold default 1e-4)
  undistort    cv::undistortPoints(pts, pts, K, D, noArray(), K)        tracking/camera.cc:72-74
  GFTT+subpix  cv::goodFeaturesToTrack(block, out, quota, 0.01, min_dist, block_mask) + cv::cornerSubPix(block, out, Size(5,5),
               Size(-1,-1), TermCriteria(COUNT+EPS, 20, 0.01))          tracking.cc:647-652, per block ROI of :632-645
  RANSAC       cv::findFundamentalMat(p1, p2, FM_RANSAC, 1.5, 0.99, mask)    tracking.cc:548
Inputs come from tests/synth.py (deterministic integer texture), at the sizes of configs C1 / C2 / C4 of BASELINE.json.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

CONFIGS = {"c1": (640, 480, 100), "c2": (1280, 720, 300), "c4": (1920, 1080, 500)}
TRACK_BLOCK_SIZE = 200.0


def camera_for(w, h):
    s = w / 1280.0
    K = np.array([[787.1611861559479 * s, 0.0, w / 2.0], [0.0, 787.3928431375225 * s, h / 2.0], [0.0, 0.0, 1.0]])
    D = np.array([-0.0917403092279957, 0.08134715036932794, 0.00017620136958692255, 0.00016737385248865412, 0.0])
    return K, D


def grid_for(w, h, nfeat):
    """tracking.cc:66-85"""
    lround = lambda v: int(np.floor(v + 0.5))  # C lround for positive values (Python's round() is banker's rounding)
    cols, rows = lround(w / TRACK_BLOCK_SIZE), lround(h / TRACK_BLOCK_SIZE)
    bw, bh = w // cols, h // rows
    quota = lround(nfeat / float(cols * rows))
    min_dist = lround(TRACK_BLOCK_SIZE / np.sqrt(quota * 1.5))
    return cols, rows, bw, bh, quota, min_dist


def two_view_points(n, seed, w, h):
    """pixel correspondences of a random rigid scene + 20 % gross outliers (float32, as the tracker hands them over)"""
    rng = np.random.RandomState(seed)
    K, _ = camera_for(w, h)
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(8, 40, n)], 1)
    a = 0.04
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.8, 0.05, 0.1])
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    p1, p2 = x1[:, :2] / x1[:, 2:], x2[:, :2] / x2[:, 2:]
    p2 += rng.normal(0, 0.3, p2.shape)
    out = rng.rand(n) < 0.2
    p2[out] += rng.uniform(-60, 60, (out.sum(), 2))
    return p1.astype(np.float32), p2.astype(np.float32)


def expected_keys():
    """the keys tests/test_opencv_golden.py reads (per config tag) — printed by `--list-keys` and after a run, so a hand-off needs no second look"""
    per_tag = ["clahe_a", "clahe_b", "pyr0", "pyr1", "pyr2", "pyr3", "lk_prev", "lk_guess", "lk_next", "lk_status", "lk_err", "undist_in",
               "undist_out", "det_grid", "det_exist", "det_mask", "det_pts", "det_block", "fm_p1", "fm_p2", "fm_mask", "fm_F"]
    return ["opencv_version"] + [f"{tag}_{k}" for tag in CONFIGS for k in per_tag]


def main():
    if "--list-keys" in sys.argv:  # needs neither cv2 nor the inputs
        print("\n".join(expected_keys()))
        return
    try:
        import cv2
    except ImportError:
        sys.exit("make_opencv_golden.py needs only numpy and cv2 (opencv-python >= 4.5, < 5): `import cv2` failed on this box.  Run it where "
                 "OpenCV is installed and commit tests/golden/opencv_golden.npz; keys: python tests/golden/make_opencv_golden.py --list-keys")
    out = {"opencv_version": np.array(cv2.__version__)}
    for tag, (w, h, nfeat) in CONFIGS.items():
        a = synth.texture(w, h, seed=21)
        b = synth.shift_image(a, 3.25, -1.75)
        clahe = cv2.createCLAHE(3.0, (21, 21))
        ca, cb = clahe.apply(a), clahe.apply(b)
        out[f"{tag}_clahe_a"], out[f"{tag}_clahe_b"] = ca, cb
        _, pyr = cv2.buildOpticalFlowPyramid(ca, (21, 21), 3, withDerivatives=False)
        for lvl, im in enumerate(pyr):
            out[f"{tag}_pyr{lvl}"] = np.ascontiguousarray(im)
        pts = synth.random_points(nfeat, w, h, 12, seed=22).astype(np.float32)
        guess = (pts + np.float32([2.5, -1.0])).astype(np.float32)
        crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01)
        nxt, st, err = cv2.calcOpticalFlowPyrLK(ca, cb, pts.reshape(-1, 1, 2), guess.reshape(-1, 1, 2).copy(), winSize=(21, 21),
                                                maxLevel=3, criteria=crit, flags=cv2.OPTFLOW_USE_INITIAL_FLOW)
        out[f"{tag}_lk_prev"], out[f"{tag}_lk_guess"] = pts, guess
        out[f"{tag}_lk_next"], out[f"{tag}_lk_status"], out[f"{tag}_lk_err"] = nxt.reshape(-1, 2), st.ravel(), err.ravel()
        # intermediates (diagnosis only: a mismatch of the final positions is localised in one pass): the position every point has after
        # the coarse levels, obtained by stopping the pyramid early — calcOpticalFlowPyrLK(maxLevel = L) on the level-(3-L) images with the
        # guess scaled accordingly is what the full call does internally for its first L+1 levels
        for top in (3, 2, 1):  # track only levels 3..top on the images of level `top`
            sc = 1.0 / (1 << top)
            pa, pb = pyr[top], cv2.buildOpticalFlowPyramid(cb, (21, 21), 3, withDerivatives=False)[1][top]
            n_t, s_t, _ = cv2.calcOpticalFlowPyrLK(np.ascontiguousarray(pa), np.ascontiguousarray(pb), (pts * sc).reshape(-1, 1, 2).astype(np.float32),
                                                   (guess * sc).reshape(-1, 1, 2).astype(np.float32), winSize=(21, 21), maxLevel=3 - top,
                                                   criteria=crit, flags=cv2.OPTFLOW_USE_INITIAL_FLOW)
            out[f"{tag}_lk_partial_top{top}"] = n_t.reshape(-1, 2)
            out[f"{tag}_lk_partial_top{top}_status"] = s_t.ravel()
        out[f"{tag}_scharr_x"] = cv2.Scharr(ca, cv2.CV_16S, 1, 0)  # the derivative planes LK samples (lkpyramid.cpp calcSharrDeriv)
        out[f"{tag}_scharr_y"] = cv2.Scharr(ca, cv2.CV_16S, 0, 1)
        K, D = camera_for(w, h)
        und = cv2.undistortPoints(pts.reshape(-1, 1, 2), K, D, None, K).reshape(-1, 2)
        out[f"{tag}_undist_in"], out[f"{tag}_undist_out"] = pts, und.astype(np.float32)
        # gridded detection: mask discs at a few existing features, then GFTT + cornerSubPix per block ROI in block order
        cols, rows, bw, bh, quota, min_dist = grid_for(w, h, nfeat)
        exist = synth.random_points(max(4, nfeat // 10), w, h, 12, seed=23).astype(np.float32)
        mask = np.full((h, w), 255, np.uint8)
        for p in exist:
            cv2.circle(mask, (int(round(float(p[0]))), int(round(float(p[1])))), min_dist, 0, cv2.FILLED)
        det, det_blk, det_raw = [], [], []
        out[f"{tag}_mineig"] = cv2.cornerMinEigenVal(ca, 3, ksize=3)  # whole image; the per-block maps differ only at the ROI borders
        out[f"{tag}_sobel_x"] = cv2.Sobel(ca, cv2.CV_32F, 1, 0, ksize=3)
        out[f"{tag}_sobel_y"] = cv2.Sobel(ca, cv2.CV_32F, 0, 1, ksize=3)
        for k in range(cols * rows):
            c, r = k % cols, k // cols
            x0, y0, rw, rh = c * bw, r * bh, bw, bh
            if k != cols * rows - 1:
                rw, rh = rw - 5, rh - 5
            blk, blk_mask = ca[y0:y0 + rh, x0:x0 + rw], mask[y0:y0 + rh, x0:x0 + rw]
            corners = cv2.goodFeaturesToTrack(blk, quota, 0.01, min_dist, mask=blk_mask)
            if corners is None or len(corners) == 0:
                continue
            for q in corners.reshape(-1, 2):
                det_raw.append([q[0] + x0, q[1] + y0])  # integer corners before cornerSubPix
            if k == 0:
                out[f"{tag}_mineig_block0"] = cv2.cornerMinEigenVal(np.ascontiguousarray(blk), 3, ksize=3)  # what GFTT sees for block 0
            corners = cv2.cornerSubPix(blk, corners, (5, 5), (-1, -1), (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 20, 0.01))
            for q in corners.reshape(-1, 2):
                det.append([q[0] + x0, q[1] + y0])
                det_blk.append(k)
        out[f"{tag}_det_exist"] = exist
        out[f"{tag}_det_mask"] = mask
        out[f"{tag}_det_pts"] = np.array(det, np.float32).reshape(-1, 2)
        out[f"{tag}_det_block"] = np.array(det_blk, np.int32)
        out[f"{tag}_det_raw"] = np.array(det_raw, np.float32).reshape(-1, 2)
        out[f"{tag}_det_grid"] = np.array([cols, rows, bw, bh, quota, min_dist], np.int32)
        p1, p2 = two_view_points(max(40, nfeat // 2), 24, w, h)
        F, m = cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, 1.5, 0.99)
        out[f"{tag}_fm_p1"], out[f"{tag}_fm_p2"] = p1, p2
        out[f"{tag}_fm_mask"] = (m.ravel().astype(np.uint8) if m is not None else np.zeros(len(p1), np.uint8))
        out[f"{tag}_fm_F"] = F if F is not None else np.zeros((3, 3))
    path = os.path.join(HERE, "opencv_golden.npz")
    missing = [k for k in expected_keys() if k not in out]
    assert not missing, missing
    np.savez_compressed(path, **out)
    print("wrote", path, "with OpenCV", cv2.__version__, f"({len(out)} arrays; the {len(expected_keys())} the tests read are all present)")
    print("next: git add tests/golden/opencv_golden.npz && python -m pytest tests/test_opencv_golden.py -q      (CPU: the oracle; -m gpu: the HIP path)")


if __name__ == "__main__":
    main()
