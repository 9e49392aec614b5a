"""Driver for the REFERENCE's own estimator (oracle/_ref/libref_gvins.so: ic_gvins.cc compiled unmodified on interface shims, see
oracle/ref_build/ref_gvins.cc) on the synthetic sequence of gvins_data.py — build container only; its results are committed as
tests/golden/gvins_ref_golden.npz by tests/golden/make_gvins_golden.py."""
import ctypes as C
import os

import numpy as np

import gvins_data as gd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_gvins.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "gvins_ref_golden.npz")
# scenario -> (golden file, gvins_data.Sequence.write keyword arguments, (first, last) index of blacked-out images or None)
SCENARIOS = {
    "default": (GOLDEN, {}, None),
    "earth_td": (os.path.join(ROOT, "tests", "golden", "gvins_ref_earth_td_golden.npz"), dict(estimate_td=True, with_earth=True), None),
    "loss": (os.path.join(ROOT, "tests", "golden", "gvins_ref_loss_golden.npz"), {}, (40, 50)),
    # a six-keyframe window (more marginalizations per second), extrinsic + time-delay estimation with the Earth-rotation variants on
    "small_window_calibration": (os.path.join(ROOT, "tests", "golden", "gvins_ref_small_window_golden.npz"),
                                 dict(optimize_windows_size=6, estimate_extrinsic=True, estimate_td=True, with_earth=True, track_max_features=150), None),
}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def read_inputs(files, w, h):
    imu = np.ascontiguousarray(np.loadtxt(files["imu"]))
    gn = np.loadtxt(files["gnss"])
    gn[:, 1:3] *= gd.D2R
    gn = np.ascontiguousarray(gn)
    stamps, imgs = [], []
    root = os.path.dirname(files["images"])
    for line in open(files["images"]):
        t, name = line.split()
        raw = open(os.path.join(root, name), "rb").read()
        imgs.append(np.frombuffer(raw[-w * h:], np.uint8))
        stamps.append(float(t))
    return imu, gn, np.ascontiguousarray(stamps), np.ascontiguousarray(np.stack(imgs))


def run_reference(files, out_dir, w, h, slowdown=2.0):
    """plays the files into the reference's GVINS (three threads, paced `slowdown` x slower than real time); returns its final state"""
    lib = C.CDLL(REF_SO)
    imu, gn, stamps, imgs = read_inputs(files, w, h)
    os.makedirs(out_dir, exist_ok=True)
    return lib.ref_gvins_run(files["config"].encode(), out_dir.encode(), len(imu), _p(imu), len(gn), _p(gn), len(stamps), _p(stamps), _p(imgs), w, h,
                             C.c_double(slowdown))
