"""Pins the oracle's ReprojectionFactor restatement against the REFERENCE's own code:
  * tests/golden/reproj_ref_golden.npz — outputs of /root/reference/.../factors/reprojection_factor.h compiled unmodified
    against the Eigen-interface shim (generator: tests/golden/make_reproj_golden.py), committed so the check runs anywhere;
  * oracle/_ref/libref_reproj.so directly, when it has been built in this container.
Tolerance 1e-12 relative (both FP64; only summation order may differ) vs north_star's 1e-5."""
import ctypes as C
import os

import numpy as np
import pytest

import reproj_data as rd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reproj_ref_golden.npz")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_reproj.so")


def test_oracle_matches_reference_golden(oracle):
    g = np.load(GOLDEN)
    r, J = oracle.reproj_eval(g["obs"], g["idx_i"], g["idx_j"], g["idx_lm"], g["poses"], g["ext"], g["invdepth"], float(g["td"]))
    assert np.abs(r - g["r"]).max() <= 1e-12 * max(1.0, np.abs(g["r"]).max())
    assert np.abs(J - g["J"]).max() <= 1e-12 * max(1.0, np.abs(g["J"]).max())
    assert np.abs(g["r"]).max() > 1e-3  # non-trivial residuals


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_matches_reference_library_on_fresh_inputs(oracle):
    lib = C.CDLL(REF_SO)
    w = rd.make_window(25, 5, seed=77, pixel_noise=1.5)
    p = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    for k in range(0, w["obs_soa"].shape[1], 3):
        o = np.ascontiguousarray(w["obs_soa"][:, k])
        pi, pj = w["poses"][w["idx_i"][k]], w["poses"][w["idx_j"][k]]
        rho = w["invdepth"][w["idx_lm"][k]]
        r_ref, J_ref = np.zeros(2), np.zeros(46)
        assert lib.ref_reproj_eval_one(p(o), p(pi), p(pj), p(w["ext"]), C.c_double(rho), C.c_double(w["td"]), p(r_ref), p(J_ref)) == 0
        r, J = oracle.reproj_eval_one(o, pi, pj, w["ext"], rho, w["td"])
        assert np.abs(r - r_ref).max() <= 1e-12 * max(1.0, np.abs(r_ref).max())
        assert np.abs(J - J_ref).max() <= 1e-12 * max(1.0, np.abs(J_ref).max())
    # rotation.h:72-76 convention check, incl. the zero vector
    for rv in ([0.1, -0.2, 0.3], [0, 0, 0], [1e-9, 0, 0]):
        q = np.zeros(4)
        lib.ref_rotvec2quaternion(p(np.array(rv, float)), p(q))
        assert np.allclose(q, rd.quat_from_rotvec(np.array(rv, float)), atol=1e-15)
