"""Inputs and drivers for the navigation factors / Earth helpers of the GVINS estimator: the same seeded cases are
evaluated by the reference's own headers (oracle/_ref/libref_nav.so -> tests/golden/nav_ref_golden.npz) and by the host layer
(icgh_nav_factor / icgh_nav_helper / icgh_detect_zero_velocity)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "nav_ref_golden.npz")
D2R = np.pi / 180.0
SHAPES = {0: (3, 7), 1: (6, 9), 2: (6, 7), 3: (9, 9)}  # residuals x block size per factor kind


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _quat(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def factor_cases(seed=7, n=12):
    rng = np.random.RandomState(seed)
    cases = []
    for k in range(n):
        pose = np.concatenate([rng.normal(0, 30, 3), _quat(rng, 1)[0]])
        if k == 3:
            pose[3:] *= 1.3  # a non-unit quaternion: Eigen's toRotationMatrix / inverse() are used on raw coefficients
        cases.append((0, np.concatenate([pose[:3] + rng.normal(0, 0.2, 3), rng.uniform(0.01, 0.5, 3), rng.normal(0, 0.4, 3)]), pose))
        cases.append((1, np.zeros(1), rng.normal(0, [1, 1, 1, 1e-3, 1e-3, 1e-3, 0.05, 0.05, 0.05])))
        prior = np.concatenate([pose[:3] + rng.normal(0, 0.1, 3), _quat(rng, 1)[0] if k % 4 == 0 else pose[3:] + rng.normal(0, 0.01, 4)])
        cases.append((2, np.concatenate([prior, rng.uniform(0.01, 0.2, 6)]), pose))
        mix = rng.normal(0, [1, 1, 1, 1e-3, 1e-3, 1e-3, 0.05, 0.05, 0.05])
        cases.append((3, np.concatenate([mix + rng.normal(0, 0.01, 9), rng.uniform(1e-3, 0.3, 9)]), mix))
    return cases


def helper_cases(seed=11):
    rng = np.random.RandomState(seed)
    cases = []
    for lat, lon, h in [(30.5, 114.3, 20.0), (-33.9, 151.2, 120.0), (0.0, 0.0, 0.0), (78.2, 15.6, 450.0), (35.0, -120.0, -40.0)]:
        blh = np.array([lat * D2R, lon * D2R, h])
        cases.append((0, blh, None))
        for _ in range(3):
            local = rng.normal(0, [2000, 2000, 50])
            other = blh + np.array([rng.normal(0, 1e-4), rng.normal(0, 1e-4), rng.normal(0, 30)])
            cases += [(1, blh, other), (2, blh, local), (3, blh, local)]
    cases.append((3, np.zeros(3), np.array([12.0, -250.0, 3.0])))  # station = 0: the reference's effective preintegration setting (H9)
    for _ in range(8):
        cases.append((4, np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-np.pi, np.pi)]), None))
    for q in _quat(rng, 8):
        cases.append((5, q, None))
    cases.append((5, np.array([0.0, -np.sin(np.pi / 4), 0.0, np.cos(np.pi / 4)]), None))  # pitch +90 deg: the dcm(2,0) <= -0.999 branch
    cases.append((5, np.array([0.0, np.sin(np.pi / 4), 0.0, np.cos(np.pi / 4)]), None))   # pitch -90 deg: the >= 0.999 branch
    for t in (1664000000.25, 1500000000.0, 315964800.0 - 18 + 604800.0 * 3 + 12.5):
        cases.append((6, np.array([t, 0.0, 0.0]), None))
    return cases


def zero_velocity_cases(seed=5):
    rng = np.random.RandomState(seed)
    out = []
    for gyr, acc in [(2e-6, 1e-4), (2e-5, 1e-4), (2e-6, 1e-3), (8e-6, 4e-4), (1.2e-5, 2e-4)]:
        rows = np.concatenate([rng.normal([1e-6, -2e-6, 5e-7], gyr, (60, 3)), rng.normal([1e-4, 2e-4, -0.049], acc, (60, 3))], axis=1)
        out.append(np.ascontiguousarray(rows))
    return out


def evaluate(lib, prefix):
    """run every case through `lib` (functions <prefix>nav_factor / nav_helper / detect_zero_velocity) -> dict of arrays"""
    fn_factor, fn_helper, fn_zv = getattr(lib, prefix + "nav_factor"), getattr(lib, prefix + "nav_helper"), getattr(lib, prefix + "detect_zero_velocity")
    out = {}
    for i, (kind, aux, x) in enumerate(factor_cases()):
        nr, nb = SHAPES[kind]
        r, J = np.zeros(nr), np.zeros((nr, nb))
        aux, x = np.ascontiguousarray(aux, np.float64), np.ascontiguousarray(x, np.float64)
        assert fn_factor(kind, _p(aux), _p(x), _p(r), _p(J)) == 0
        r2 = np.zeros(nr)
        assert fn_factor(kind, _p(aux), _p(x), _p(r2), None) == 0 and np.array_equal(r, r2)
        out[f"factor_{i}_r"], out[f"factor_{i}_J"] = r, J
    for i, (what, a, b) in enumerate(helper_cases()):
        a = np.ascontiguousarray(np.concatenate([a, np.zeros(4)])[:4] if what == 5 else np.concatenate([a, np.zeros(3)])[:3], np.float64)
        b = None if b is None else np.ascontiguousarray(b, np.float64)
        o = np.zeros(4)
        assert fn_helper(what, _p(a), _p(b), _p(o)) == 0
        out[f"helper_{i}"] = o
    for i, rows in enumerate(zero_velocity_cases()):
        avg = np.zeros(6)
        z = fn_zv(len(rows), _p(rows), C.c_double(200.0), _p(avg))
        out[f"zv_{i}"] = np.concatenate([[float(z)], avg])
    return out
