"""Known-answer tests that pin the ORACLE's ReprojectionFactor restatement (reference has no tests of its own):
finite differences of the residual under PoseParameterization::Plus must reproduce the analytic Jacobians of
factors/reprojection_factor.h:98-143, and noise-free synthetic geometry must give ~zero residuals."""
import numpy as np

import reproj_data as rd


def test_zero_residual_on_exact_geometry(oracle):
    w = rd.make_window(60, 6, seed=3, pixel_noise=0.0)
    # remove the time-offset terms so the observation equals the exact projection
    obs = w["obs_soa"].copy()
    obs[6:12] = 0.0
    r, _ = oracle.reproj_eval(obs, w["idx_i"], w["idx_j"], w["idx_lm"], w["poses"], w["ext"], w["invdepth"], w["td"])
    assert np.abs(r).max() < 1e-9


def test_jacobians_match_finite_differences(oracle):
    w = rd.make_window(40, 6, seed=1)
    obs = w["obs_soa"]
    n = obs.shape[1]
    rng = np.random.RandomState(0)
    eps = 1e-6
    for k in rng.choice(n, 25, replace=False):
        o = obs[:, k]
        pi, pj = w["poses"][w["idx_i"][k]], w["poses"][w["idx_j"][k]]
        ext, rho, td = w["ext"], w["invdepth"][w["idx_lm"][k]], w["td"]
        r0, J = oracle.reproj_eval_one(o, pi, pj, ext, rho, td)
        Ji, Jj, Je = J[0:14].reshape(2, 7), J[14:28].reshape(2, 7), J[28:42].reshape(2, 7)
        Jr, Jt = J[42:44], J[44:46]
        assert np.all(Ji[:, 6] == 0) and np.all(Jj[:, 6] == 0) and np.all(Je[:, 6] == 0)
        for blk, (A, base) in enumerate(((Ji, pi), (Jj, pj), (Je, ext))):
            for c in range(6):
                d = np.zeros(6)
                d[c] = eps
                args = [pi, pj, ext]
                plus, minus = list(args), list(args)
                plus[blk] = rd.pose_plus(base, d)
                minus[blk] = rd.pose_plus(base, -d)
                rp, _ = oracle.reproj_eval_one(o, *plus, rho, td, want_jac=False)
                rm, _ = oracle.reproj_eval_one(o, *minus, rho, td, want_jac=False)
                fd = (rp - rm) / (2 * eps)
                assert np.allclose(fd, A[:, c], rtol=2e-5, atol=2e-4), (k, blk, c, fd, A[:, c])
        h = rho * 1e-6
        rp, _ = oracle.reproj_eval_one(o, pi, pj, ext, rho + h, td, want_jac=False)
        rm, _ = oracle.reproj_eval_one(o, pi, pj, ext, rho - h, td, want_jac=False)
        assert np.allclose((rp - rm) / (2 * h), Jr, rtol=1e-5, atol=1e-3)
        rp, _ = oracle.reproj_eval_one(o, pi, pj, ext, rho, td + eps, want_jac=False)
        rm, _ = oracle.reproj_eval_one(o, pi, pj, ext, rho, td - eps, want_jac=False)
        assert np.allclose((rp - rm) / (2 * eps), Jt, rtol=1e-5, atol=1e-4)


def test_huber_corrector_is_identity_inside_delta(oracle):
    w = rd.make_window(30, 5, seed=2, pixel_noise=0.1)
    args = (w["obs_soa"], w["idx_i"], w["idx_j"], w["idx_lm"], w["poses"], w["ext"], w["invdepth"], w["td"])
    r0, J0 = oracle.reproj_eval(*args)
    r1, J1 = oracle.reproj_eval(*args, huber=1e9)
    assert np.array_equal(r0, r1) and np.array_equal(J0, J1)
    # outside delta Huber has rho'' < 0, so the corrector takes its first branch (residual_block_info.h:69-71):
    # r' = sqrt(rho') r  ->  |r'| = sqrt(a |r|)
    a = 0.05
    r2, J2 = oracle.reproj_eval(*args, huber=a)
    s = (r0 ** 2).sum(1)
    big = s > a * a
    assert big.any()
    exp = np.sqrt(a * np.sqrt(s[big]))
    assert np.allclose(np.linalg.norm(r2[big], axis=1), exp, rtol=1e-12)
