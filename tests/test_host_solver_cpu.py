"""SURVEY.md §8 row f1 on the CPU: the Schur entry points of the C ABI (here: the oracle shim) and icg::WindowSolver of the host
layer against independent numpy restatements (dense elimination, dense Levenberg-Marquardt without any Schur complement).
Parity status of this row: unpinned (Ceres is absent) — the checks are algebraic."""
import ctypes as C

import numpy as np
import pytest

import schur_checks as sc
import solve_utils as su


@pytest.fixture(scope="module")
def host_lib():
    from stream_utils import ensure_oracle_host
    return ensure_oracle_host()


def test_schur_entry_points_on_oracle_shim(host_lib, oracle):
    import icgvins
    ctx = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64, lib=icgvins.load_library(host_lib))
    sc.check_schur(ctx, oracle)
    sc.check_schur_any_factor_order(ctx, oracle, refuses_same_block=False)
    ctx.close()


@pytest.mark.parametrize("cfg", [dict(seed=0, n_outliers=8), dict(seed=1, n_outliers=0, chi2=-1.0, iters1=12),
                                 dict(seed=2, n_outliers=5, ext_const=True, td_const=True), dict(seed=3, n_outliers=3, huber=0.0, td_const=True)])
def test_window_solver_matches_dense_lm(host_lib, oracle, cfg):
    """same accepted / rejected step counts, same chi-square culling decisions, same optimum as a dense LM that never forms a
    Schur complement; the optimum recovers the true poses from a 10 cm / 0.6 deg perturbation"""
    cfg = dict(cfg)
    P = su.make_problem(60, 6, seed=cfg.pop("seed"), n_outliers=cfg.pop("n_outliers"))
    h = su.host_solve(C.CDLL(host_lib), P, **cfg)
    d = su.dense_solve(oracle, P, **cfg)
    assert np.array_equal(h["summary"][3:], d["summary"][3:]), (h["summary"], d["summary"])
    assert np.abs(h["summary"][:3] - d["summary"][:3]).max() < 1e-8 * max(1.0, d["summary"][0])
    assert np.array_equal(h["active"], d["active"])
    if cfg.get("chi2", 5.991) > 0:
        assert list(np.nonzero(h["active"] == 0)[0]) == list(P["outliers"]) or set(P["outliers"]) <= set(np.nonzero(h["active"] == 0)[0])
    for k in ("poses", "ext", "invdepth"):
        assert np.abs(h[k] - d[k]).max() < 1e-8, k
    assert abs(h["td"] - d["td"]) < 1e-8
    assert np.abs(h["poses"][:, :3] - P["truth"]["poses"][:, :3]).max() < 0.01
    assert h["summary"][2] < 0.05 * h["summary"][0]


def test_window_solver_visual_inertial_window(host_lib, oracle):
    """the factor mix of the real window — preintegration factors (4 blocks, 15 residuals, device P1 + host P2), reprojection factors
    on the same pose blocks (device), pose + velocity/bias priors on the first state: from a 10 cm / 0.6 deg / 0.2 m/s perturbation
    the solver returns to the IMU-consistent truth"""
    import vio_data as vd
    W = vd.make_vio_window(oracle)
    lib = C.CDLL(host_lib)
    _, _, at_truth = vd.host_solve_vio(lib, W, W["states"], W["invdepth"], iters=1)
    s, inv0 = vd.perturbed_start(W)
    st, inv, summ = vd.host_solve_vio(lib, W, s, inv0)
    assert summ[0] > 1e5 * summ[1] and summ[1] < 1.05 * at_truth[0]
    assert np.abs(st[:, :3] - W["states"][:, :3]).max() < 5e-3
    assert np.abs(st[:, 7:10] - W["states"][:, 7:10]).max() < 1e-2
    assert np.abs(st[:, 10:] - W["states"][:, 10:]).max() < 2e-3
    assert np.abs(inv / W["invdepth"] - 1).max() < 0.05


def test_map_to_optimizer_to_map_on_tracked_windows(host_lib):
    """VisualWindow (addReprojectionParameters / addReprojectionFactors / updateParametersFromOptimizer of GVINS) + WindowSolver +
    WindowCulling on the maps the tracker built from noisy INS priors"""
    import refine_checks as rc
    rc.check_refinement(host_lib)


def test_schur_windows_entry_points_on_oracle_shim(host_lib):
    import icgvins
    lib = icgvins.load_library(host_lib)
    sc.check_schur_windows(lambda: icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64, lib=lib))


def _batch_problems():
    return [su.make_problem(60, 6, seed=0, n_outliers=8), su.make_problem(40, 5, seed=1, n_outliers=0), su.make_problem(80, 7, seed=2, n_outliers=5),
            su.make_problem(30, 4, seed=3, n_outliers=3, perturb=0.2), su.make_problem(60, 6, seed=5, n_outliers=4, perturb=2.5)]


def test_window_solver_batch_equals_single_solvers(host_lib):
    """five windows of different size and difficulty (one with rejected steps, one that stops early) optimized in lock-step by
    WindowSolverBatch: per window the same accepted / rejected steps, the same chi-square removals and the same optimum as a
    WindowSolver of its own"""
    lib = C.CDLL(host_lib)
    probs = _batch_problems()
    res, _ = su.host_solve_batch(lib, probs)
    kinds = set()
    for k, P in enumerate(probs):
        h = su.host_solve(lib, P)
        assert np.array_equal(res[k]["summary"][3:], h["summary"][3:]), (k, res[k]["summary"], h["summary"])
        assert np.abs(res[k]["summary"][:3] - h["summary"][:3]).max() < 1e-8 * max(1.0, h["summary"][0])
        for key in ("poses", "ext", "invdepth"):
            assert np.abs(res[k][key] - h[key]).max() < 1e-8, (k, key)
        kinds.add((h["summary"][4] + h["summary"][6] > 0, h["summary"][5] < 3))
    assert (True, False) in kinds or (True, True) in kinds  # at least one window had a rejected step


def test_window_solver_batch_device_side_reduced_solve(host_lib, monkeypatch):
    """ICG_SOLVER_DEVICE_CHOLESKY=1 (icg_reproj_schur_windows_resident / _set_host_part_windows / _solve_backsub_windows: the reduced systems
    stay behind the ABI, the host factors' part goes up as packed lower triangles, a batched Cholesky solves them): the same step sequences,
    removals and optima as the default path (reduced systems factored by the host)"""
    lib = C.CDLL(host_lib)
    probs = _batch_problems()
    ref, _ = su.host_solve_batch(lib, probs)
    monkeypatch.setenv("ICG_SOLVER_DEVICE_CHOLESKY", "1")
    dev, _ = su.host_solve_batch(lib, probs)
    for k in range(len(probs)):
        assert np.array_equal(dev[k]["summary"][3:], ref[k]["summary"][3:]), (k, dev[k]["summary"], ref[k]["summary"])
        assert np.abs(dev[k]["summary"][:3] - ref[k]["summary"][:3]).max() < 1e-8 * max(1.0, ref[k]["summary"][0])
        for key in ("poses", "ext", "invdepth"):
            assert np.abs(dev[k][key] - ref[k][key]).max() < 1e-8, (k, key)


def test_window_solver_matches_reference_factors_with_independent_lm(host_lib):
    """icg::WindowSolver (landmark elimination + LM of the product, here on the CPU shim) against tests/golden/solve_ref_golden.npz: the same
    three windows solved with the REFERENCE's own factor code and an independently written LM (oracle/ref_build/shim/ceres/problem_shim.h;
    generator tests/golden/make_solve_golden.py).  Same costs, the same accepted / rejected step counts in both solves, the same factors
    removed by the chi-square test, the same optimum."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solve_ref_golden.npz"))
    lib = C.CDLL(host_lib)
    for seed, (nlm, nkf, nout) in enumerate([(60, 6, 4), (300, 10, 10), (120, 8, 0)]):
        P = su.make_problem(nlm, nkf, seed=seed, n_outliers=nout)
        H = su.host_solve(lib, P)
        e = lambda k: g[f"case{seed}_{k}"]
        assert np.array_equal(H["active"], e("active")) and int((e("active") == 0).sum()) >= nout
        assert np.array_equal(H["summary"][3:8], e("summary")[3:8])  # step counts of both solves, removed factors
        assert np.all(np.abs(H["summary"][:3] - e("summary")[:3]) <= 1e-7 * np.abs(e("summary")[:3]))  # initial / intermediate / final cost
        assert np.abs(H["poses"] - e("poses")).max() < 1e-8 and np.abs(H["ext"] - e("ext")).max() < 1e-6
        assert np.abs(H["invdepth"] - e("invdepth")).max() < 1e-7 and abs(H["td"] - float(e("td")[0])) < 1e-9
