"""Known-answer tests pinning the ORACLE's preintegration restatement (reference has no tests): closed forms for
constant inputs, integrate/evaluate consistency, finite-difference Jacobians under PoseParameterization::Plus."""
import numpy as np

import preint_data as pd
import reproj_data as rd


def test_constant_acceleration_closed_form(oracle):
    n, dt = 21, 0.005
    a = np.array([0.4, -0.3, 0.2])
    imu = np.zeros((n, 8))
    imu[:, 0] = np.arange(n) * dt
    imu[:, 1] = dt
    imu[:, 5:8] = a * dt
    s0 = pd.state(rv=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0))
    out = oracle.preint_integrate(0, imu, s0, pd.PARAMS)
    T = (n - 1) * dt
    g = np.array([0, 0, pd.PARAMS[5]])
    assert abs(out["dt"] - T) < 1e-15
    assert np.allclose(out["cur"][7:10], s0[7:10] + (a + g) * T, atol=1e-12)
    assert np.allclose(out["cur"][0:3], s0[0:3] + s0[7:10] * T + 0.5 * (a + g) * T * T, atol=1e-12)
    assert np.allclose(out["delta"][7:10], a * T, atol=1e-13) and np.allclose(out["delta"][0:3], 0.5 * a * T * T, atol=1e-13)
    # bias Jacobians of the preintegrated measurement: dv/dba = -T I, dp/dba = -T^2/2 I (first order, no rotation)
    assert np.allclose(out["jac"][3:6, 12:15], -T * np.eye(3), rtol=1e-4, atol=1e-12)  # (1 - dt/corr_time) decay ~1e-5
    # phi = I + F dt is forward Euler: dp/dba = -dt^2 * sum_k k = -T (T - dt) / 2
    assert np.allclose(out["jac"][0:3, 12:15], -0.5 * T * (T - dt) * np.eye(3), rtol=1e-4, atol=1e-12)
    # covariance is symmetric PSD and grows
    assert np.allclose(out["cov"], out["cov"].T, atol=1e-20) and np.all(np.linalg.eigvalsh(out["cov"]) > -1e-18)


def test_constant_rotation_closed_form(oracle):
    n, dt = 41, 0.005
    w = np.array([0.1, -0.2, 0.3])
    imu = np.zeros((n, 8))
    imu[:, 1] = dt
    imu[:, 2:5] = w * dt
    s0 = pd.state(rv=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0), v=(0, 0, 0))
    out = oracle.preint_integrate(0, imu, s0, pd.PARAMS)
    T = (n - 1) * dt
    exp = rd.quat_from_rotvec(w * T)
    assert np.allclose(out["delta"][3:7], exp, atol=1e-12)


def _perturb(pose, mix, blk, c, eps, which):
    pose, mix = pose.copy(), mix.copy()
    if which == "pose":
        d = np.zeros(6)
        d[c] = eps
        pose = rd.pose_plus(pose, d)
    else:
        mix[c] += eps
    return pose, mix


def test_evaluate_consistency_and_jacobians(oracle):
    for variant in (0, 1):
        imu = pd.make_interval(41, seed=variant)
        s0 = pd.state()
        pre = oracle.preint_integrate(variant, imu, s0, pd.PARAMS)
        g3 = [0, 0, pd.PARAMS[5]]
        iewn = pd.PARAMS[6:9]
        pose0, mix0 = pd.split(s0)
        pose1, mix1 = pd.split(pre["cur"])
        r, J = oracle.preint_evaluate(variant, pre, g3, iewn, pose0, mix0, pose1, mix1)
        # integrating from state0 lands on state1: whitened residual ~ 0 (Earth keeps O(w_ie^2) terms)
        assert np.abs(r).max() < (1e-6 if variant == 0 else 5e-2), (variant, np.abs(r).max())
        # finite differences around a perturbed state1 (non-zero residual)
        pose1 = rd.pose_plus(pose1, np.array([0.01, -0.02, 0.015, 0.002, -0.001, 0.003]))
        mix1 = mix1 + np.array([0.01, 0.02, -0.01, 1e-5, -2e-5, 1e-5, 1e-4, 2e-4, -1e-4])
        r0, J = oracle.preint_evaluate(variant, pre, g3, iewn, pose0, mix0, pose1, mix1)
        args = [pose0, mix0, pose1, mix1]
        for blk, (kind, ncol) in enumerate((("pose", 6), ("mix", 9), ("pose", 6), ("mix", 9))):
            scale = np.abs(J[blk]).max()
            for c in range(ncol):
                eps = 1e-6 if kind == "pose" or c < 3 else 1e-8
                ap, am = list(args), list(args)
                pi = blk - (blk % 2)
                pp, mp = _perturb(args[pi], args[pi + 1], blk, c, eps, kind)
                pm, mm = _perturb(args[pi], args[pi + 1], blk, c, -eps, kind)
                ap[pi], ap[pi + 1] = pp, mp
                am[pi], am[pi + 1] = pm, mm
                rp, _ = oracle.preint_evaluate(variant, pre, g3, iewn, *ap, want_jac=False)
                rm, _ = oracle.preint_evaluate(variant, pre, g3, iewn, *am, want_jac=False)
                fd = (rp - rm) / (2 * eps)
                assert np.allclose(fd, J[blk][:, c], rtol=2e-3, atol=2e-3 * scale), (variant, blk, c, np.abs(fd - J[blk][:, c]).max(), scale)
            if kind == "pose":
                assert np.all(J[blk][:, 6] == 0)
