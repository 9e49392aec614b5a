"""the small factors and Earth / attitude / GPS-time helpers of the GVINS estimator (host code: nav_factors.h,
earth.h, GVINS::detectZeroVelocity) against outputs of the REFERENCE's own headers (tests/golden/nav_ref_golden.npz, generated from
oracle/_ref/libref_nav.so by tests/golden/make_nav_golden.py).  Tolerances: residuals / Jacobians 1e-12 relative (the reference side runs
on the Eigen-interface shim: same expressions, association of a few 3x3 products may differ), helpers 1e-12 relative (same libm),
zero-velocity decisions exact."""
import ctypes as C

import numpy as np
import pytest

import nav_utils as nu


@pytest.fixture(scope="module")
def host():
    from stream_utils import ensure_oracle_host
    return C.CDLL(ensure_oracle_host())


def test_nav_factors_and_helpers_match_reference_golden(host):
    got = nu.evaluate(host, "icgh_")
    exp = np.load(nu.GOLDEN)
    assert set(got) == set(exp.files)
    worst = 0.0
    for k in exp.files:
        e, g = exp[k], got[k]
        if k.startswith("zv_"):
            assert g[0] == e[0], k  # the decision
        scale = max(1.0, float(np.abs(e).max()))
        err = float(np.abs(g - e).max()) / scale
        worst = max(worst, err)
        assert err < 1e-12, (k, err)
    assert worst < 1e-12


def test_nav_factor_jacobians_by_finite_differences(host):
    """independent of the reference: every analytic Jacobian under PoseParameterization::Plus (pose blocks) / plain addition"""
    import reproj_data as rd
    for i, (kind, aux, x) in enumerate(nu.factor_cases()):
        if kind in (0, 2) and abs(np.linalg.norm(x[3:7]) - 1) > 1e-6:
            continue  # the non-unit quaternion case pins the raw-coefficient arithmetic, it is not on the manifold
        nr, nb = nu.SHAPES[kind]
        aux, x = np.ascontiguousarray(aux, np.float64), np.ascontiguousarray(x, np.float64)

        def res(xx):
            r = np.zeros(nr)
            xx = np.ascontiguousarray(xx)
            assert host.icgh_nav_factor(kind, nu._p(aux), nu._p(xx), nu._p(r), None) == 0
            return r

        r0, J = np.zeros(nr), np.zeros((nr, nb))
        assert host.icgh_nav_factor(kind, nu._p(aux), nu._p(x), nu._p(r0), nu._p(J)) == 0
        nloc = 6 if nb == 7 else nb
        for c in range(nloc):
            h = 1e-6
            d = np.zeros(nloc)
            d[c] = h
            xp = rd.pose_plus(x, d) if nb == 7 else x + d
            xm = rd.pose_plus(x, -d) if nb == 7 else x - d
            fd = (res(xp) - res(xm)) / (2 * h)
            assert np.abs(fd - J[:, c]).max() < 1e-5 * max(1.0, np.abs(J[:, c]).max()), (i, kind, c)
        if nb == 7:
            assert np.all(J[:, 6] == 0)  # the 7th column of a pose Jacobian is zero (local parameterization with an identity top block)
