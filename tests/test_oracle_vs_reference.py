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


# ---- IMU preintegration (P1, P2): reference sources preintegration_{base,normal,earth}.{h,cc} + preintegration_factor.h ----
PREINT_GOLDEN = os.path.join(ROOT, "tests", "golden", "preint_ref_golden.npz")


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max())


def preint_golden_cases():
    g = np.load(PREINT_GOLDEN)
    for k in range(int(g["n_cases"])):
        c = {name: g[f"c{k}_{name}"] for name in ("variant", "imu", "s0", "cur", "delta", "jac", "cov", "dt", "iewn", "s1", "r", "J")}
        c["params"] = g["params"].copy()
        if int(c["variant"]) == 1:
            c["params"][6:9] = c["iewn"]  # the oracle takes the Earth rate explicitly (SURVEY.md hazard H9)
        yield k, c


def test_oracle_preintegration_matches_reference_golden(oracle):
    """Integration (state, 15x15 Jacobian, covariance) to 1e-12; factor residual/Jacobians to 1e-9 relative: they contain
    the inverse + Cholesky of the covariance, where the shim's and the oracle's factorizations round differently."""
    import preint_data as pd
    n = 0
    for k, c in preint_golden_cases():
        v = int(c["variant"])
        pre = oracle.preint_integrate(v, c["imu"], c["s0"], c["params"])
        assert _rel(pre["cur"], c["cur"]) < 1e-12, k
        assert _rel(pre["delta"], c["delta"]) < 1e-12, k
        assert _rel(pre["jac"], c["jac"]) < 1e-12, k
        assert _rel(pre["cov"], c["cov"]) < 1e-12, k
        assert abs(pre["dt"] - float(c["dt"])) < 1e-15, k
        pose0, mix0 = pd.split(c["s0"])
        pose1, mix1 = pd.split(c["s1"])
        r, J = oracle.preint_evaluate(v, pre, [0, 0, c["params"][5]], c["params"][6:9], pose0, mix0, pose1, mix1)
        assert _rel(r, c["r"]) < 1e-9, k
        Jr = c["J"]
        for a, b in zip(J, (Jr[:105].reshape(15, 7), Jr[105:240].reshape(15, 9), Jr[240:345].reshape(15, 7), Jr[345:].reshape(15, 9))):
            assert _rel(a, b) < 1e-9, k
        assert np.abs(c["r"]).max() > 1.0  # non-trivial whitened residuals
        n += 1
    assert n == 8


REF_PREINT_SO = os.path.join(ROOT, "oracle", "_ref", "libref_preint.so")


@pytest.mark.skipif(not os.path.exists(REF_PREINT_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_preintegration_matches_reference_library_on_fresh_inputs(oracle):
    import preint_data as pd
    lib = C.CDLL(REF_PREINT_SO)
    p = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    station = np.array([0.7, -1.2, 350.0])
    for seed in range(3):
        imu = pd.make_interval(n=30 + 7 * seed, seed=500 + seed, omega=(0.3, -0.1, 0.5))
        s0 = pd.state(p=(10.0 * seed, -3.0, 2.0), v=(12.0, -1.0, 0.3))
        for variant in (0, 1):
            cur, delta, jac, cov = np.zeros(16), np.zeros(16), np.zeros((15, 15)), np.zeros((15, 15))
            dt, iewn = C.c_double(), np.zeros(3)
            if variant == 0:
                lib.ref_preint_integrate(len(imu), p(imu), p(s0), p(pd.PARAMS), p(cur), p(delta), p(jac), p(cov), C.byref(dt))
            else:
                lib.ref_preint_integrate_earth(len(imu), p(imu), p(s0), p(pd.PARAMS), p(station), p(cur), p(delta), p(jac), p(cov),
                                               C.byref(dt), p(iewn))
            par = pd.PARAMS.copy()
            par[6:9] = iewn
            pre = oracle.preint_integrate(variant, imu, s0, par)
            for name, ref in (("cur", cur), ("delta", delta), ("jac", jac), ("cov", cov)):
                assert _rel(pre[name], ref) < 1e-12, (seed, variant, name)


# ---- marginalization (R2, M1-M4): reference sources factors/{residual_block_info,marginalization_info,marginalization_factor}.h
REF_MARG_SO = os.path.join(ROOT, "oracle", "_ref", "libref_marg.so")


@pytest.mark.skipif(not os.path.exists(REF_MARG_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_marginalization_matches_reference_library(oracle):
    """The same scenario check the product's host layer has to pass against the oracle (Schur complement, linearization
    identities, MarginalizationFactor residuals at a perturbed point), run on the REFERENCE's own pipeline."""
    import backend_utils as bu
    bu.check_marginalization(C.CDLL(REF_MARG_SO), oracle)


@pytest.mark.skipif(not os.path.exists(REF_MARG_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_marginalization_golden_is_current():
    import backend_utils as bu
    bu.check_marginalization_golden(C.CDLL(REF_MARG_SO), os.path.join(ROOT, "tests", "golden", "marg_ref_golden.npz"))


# ---- the front-end around the OpenCV calls: reference tracking/*.{h,cc} compiled against shims ------------------------------
REF_TRACKING_SO = os.path.join(ROOT, "oracle", "_ref", "libref_tracking.so")


@pytest.mark.skipif(not os.path.exists(REF_TRACKING_SO), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("scenario", ["c1_640x480_100", "c1_bgr", "c1_lost_histgate"])
def test_tracking_golden_is_current(tmp_path, scenario):
    """Re-runs the reference tracker (fresh process: its id factories are process-wide statics) and checks the committed
    golden file of the scenario is what it produces (states, ids, key points, window bookkeeping, tracking.txt rows)."""
    import ref_tracking_utils as rt
    out = str(tmp_path / "t.npz")
    rt.run_scenario_in_subprocess(scenario, out)
    a, b = rt.load(out), rt.load(rt.golden_path(scenario))
    assert np.array_equal(a["states"], b["states"]) and np.array_equal(a["stats"], b["stats"])
    assert list(a["log"]) == list(b["log"])
    for k in range(len(a["states"])):
        assert np.array_equal(a["ids"][k], b["ids"][k]) and np.array_equal(a["px"][k].view(np.uint32), b["px"][k].view(np.uint32))


# ---- camera closed forms: reference tracking/camera.cc:76-157 -----------------------------------------------------------------
CAMERA_GOLDEN = os.path.join(ROOT, "tests", "golden", "camera_ref_golden.npz")


def _same_f32(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def test_oracle_camera_matches_reference_golden(oracle):
    """distortPoints, pixel2cam, world2pixel and the tracker's INS-aided prediction (world2pixel + distortPoints) of the
    REFERENCE's Camera class: float outputs bit-identical, double outputs equal."""
    g = np.load(CAMERA_GOLDEN)
    cam = g["cam"]
    assert _same_f32(oracle.distort(cam, g["pts"]), g["distorted"])
    assert np.array_equal(oracle.pixel2cam(cam, g["pts"]), g["pixel2cam"])
    w2p = oracle.world2pixel(cam, g["pose12"], g["pw"])
    assert _same_f32(w2p, g["world2pixel"])
    assert _same_f32(oracle.distort(cam, w2p), g["predicted"])
