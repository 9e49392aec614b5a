#!/usr/bin/env python3
"""Generates tests/golden/camera_ref_golden.npz from the REFERENCE's own Camera closed-form point maps
(tracking/camera.cc:76-157 inside oracle/_ref/libref_tracking.so): distortPoints, distortCameraPoint, pixel2cam, world2pixel,
reprojectionError.  (undistortPoints is cv::undistortPoints and therefore not a reference-side function.)
Run in the build container only:  make -C oracle && make -C oracle/ref_build && python tests/golden/make_camera_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
import harness as H  # noqa: E402


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_tracking.so"))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    w, h = 1280, 720
    cam = np.asarray(H.camera_for(w, h), np.float64)
    cam[4] = 0.3  # non-zero skew exercises every term
    rng = np.random.RandomState(4)
    n = 400
    pts = np.ascontiguousarray(rng.uniform([-40, -40], [w + 40, h + 40], (n, 2)), np.float32)
    dist = pts.copy()
    lib.ref_camera_distort_points(p(cam), w, h, n, p(dist))
    pc = np.zeros((n, 3))
    lib.ref_camera_pixel2cam(p(cam), w, h, n, p(pts), p(pc))
    pcs = np.ascontiguousarray(rng.uniform([-8, -5, 4], [8, 5, 60], (n, 3)))
    dcp = np.zeros((n, 2), np.float32)
    lib.ref_camera_distort_camera_points(p(cam), w, h, n, p(pcs), p(dcp))
    R = H._rot_yp(0.07, -0.04)
    pose12 = np.ascontiguousarray(np.concatenate([R.ravel(), [1.5, -0.4, 0.8]]))
    pw = np.ascontiguousarray(rng.uniform([-15, -8, 10], [15, 8, 70], (n, 3)))
    w2p = np.zeros((n, 2), np.float32)
    lib.ref_camera_world2pixel(p(cam), w, h, p(pose12), n, p(pw), p(w2p))
    err = np.zeros((n, 2))
    lib.ref_camera_reprojection_error(p(cam), w, h, p(pose12), n, p(pw), p(pts), p(err))
    # the tracker's INS-aided prediction (tracking.cc:367-378): world2pixel, then distortPoints
    pred = w2p.copy()
    lib.ref_camera_distort_points(p(cam), w, h, n, p(pred))
    np.savez(os.path.join(ROOT, "tests", "golden", "camera_ref_golden.npz"), cam=cam, w=w, h=h, pts=pts, distorted=dist, pixel2cam=pc, pcs=pcs,
             distort_camera=dcp, pose12=pose12, pw=pw, world2pixel=w2p, reproj_err=err, predicted=pred)
    print("ok", n)


if __name__ == "__main__":
    main()
