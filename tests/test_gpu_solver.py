"""GPU parity for SURVEY.md §8 row f1: the device-side landmark elimination (k_reproj_normal in the Schur layout,
k_schur_reduce, k_schur_backsub, k_reproj_cost) and icg::WindowSolver on the HIP library, against the same independent numpy
restatements as the CPU suite (dense elimination 1e-9, dense LM optimum 1e-7).  FP64 atomics make the assembly order free, so
the comparison is by tolerance, not by bits."""
import ctypes as C

import numpy as np
import pytest

import schur_checks as sc
import solve_utils as su

pytestmark = pytest.mark.gpu


def test_schur_entry_points_on_gpu(oracle):
    import icgvins
    ctx = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64)
    sc.check_schur(ctx, oracle)
    sc.check_schur_any_factor_order(ctx, oracle)
    with pytest.raises(icgvins.IcgError):  # a pose column outside the reduced system
        ctx.reproj_schur(6, np.array([0, 6, 12, 18, 24, 30, 36], np.int32), -1, -1)
    ctx.close()


def test_backsub_without_system_fails():
    import icgvins
    ctx = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64)
    with pytest.raises(icgvins.IcgError):
        ctx.reproj_backsub(10, np.zeros(10), 5)
    ctx.close()


@pytest.mark.parametrize("cfg", [dict(seed=0, n_outliers=8), dict(seed=2, n_outliers=5, ext_const=True, td_const=True),
                                 dict(seed=4, n_outliers=10, n_lm=300, n_kf=10)])
def test_window_solver_matches_dense_lm_on_gpu(oracle, cfg):
    import harness as H
    cfg = dict(cfg)
    P = su.make_problem(cfg.pop("n_lm", 60), cfg.pop("n_kf", 6), seed=cfg.pop("seed"), n_outliers=cfg.pop("n_outliers"))
    h = su.host_solve(C.CDLL(H.HOST_LIB), P, **cfg)
    d = su.dense_solve(oracle, P, **cfg)
    assert np.array_equal(h["summary"][3:], d["summary"][3:]), (h["summary"], d["summary"])
    assert np.abs(h["summary"][:3] - d["summary"][:3]).max() < 1e-7 * max(1.0, d["summary"][0])
    assert np.array_equal(h["active"], d["active"])
    for k in ("poses", "ext", "invdepth"):
        assert np.abs(h[k] - d[k]).max() < 1e-7, k
    assert np.abs(h["poses"][:, :3] - P["truth"]["poses"][:, :3]).max() < 0.01


def test_window_solver_visual_inertial_window_on_gpu(oracle):
    """preintegration + reprojection + priors in one solve on the HIP library: same optimum as the same host layer on the oracle
    shim (1e-6), back at the IMU-consistent truth"""
    import harness as H
    import vio_data as vd
    from stream_utils import ensure_oracle_host
    W = vd.make_vio_window(oracle)
    s, inv0 = vd.perturbed_start(W)
    st, inv, summ = vd.host_solve_vio(C.CDLL(H.HOST_LIB), W, s, inv0)
    st_c, inv_c, summ_c = vd.host_solve_vio(C.CDLL(ensure_oracle_host()), W, s, inv0)
    assert np.array_equal(summ[2:], summ_c[2:]), (summ, summ_c)
    assert np.abs(st - st_c).max() < 1e-6 and np.abs(inv - inv_c).max() < 1e-6
    assert np.abs(st[:, :3] - W["states"][:, :3]).max() < 5e-3
    assert np.abs(st[:, 7:10] - W["states"][:, 7:10]).max() < 1e-2


def test_map_to_optimizer_to_map_on_tracked_windows_on_gpu():
    import harness as H
    import refine_checks as rc
    rc.check_refinement(H.HOST_LIB)


def test_schur_windows_entry_points_on_gpu():
    import icgvins
    sc.check_schur_windows(lambda: icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64))


def test_window_solver_batch_equals_single_solvers_on_gpu():
    import harness as H
    from test_host_solver_cpu import _batch_problems
    lib = C.CDLL(H.HOST_LIB)
    probs = _batch_problems()
    res, _ = su.host_solve_batch(lib, probs)
    for k, P in enumerate(probs):
        h = su.host_solve(lib, P)
        assert np.array_equal(res[k]["summary"][3:], h["summary"][3:]), (k, res[k]["summary"], h["summary"])
        for key in ("poses", "ext", "invdepth"):
            assert np.abs(res[k][key] - h[key]).max() < 1e-7, (k, key)


def test_window_solver_batch_device_side_reduced_solve_on_gpu(monkeypatch):
    """the batched Cholesky of k_chol_solve_w (ICG_SOLVER_DEVICE_CHOLESKY=1) against the default path (host factorization): same accepted /
    rejected steps and removals, optima within 1e-7"""
    import harness as H
    from test_host_solver_cpu import _batch_problems
    lib = C.CDLL(H.HOST_LIB)
    probs = _batch_problems()
    ref, _ = su.host_solve_batch(lib, probs)
    monkeypatch.setenv("ICG_SOLVER_DEVICE_CHOLESKY", "1")
    dev, _ = su.host_solve_batch(lib, probs)
    for k in range(len(probs)):
        assert np.array_equal(dev[k]["summary"][3:], ref[k]["summary"][3:]), (k, dev[k]["summary"], ref[k]["summary"])
        for key in ("poses", "ext", "invdepth"):
            assert np.abs(dev[k][key] - ref[k][key]).max() < 1e-7, (k, key)
