"""GPU parity: batched IMU preintegration (P1) vs the CPU oracle.  FP64 on both sides; the only differing primitive is
sin/cos (device libm vs glibc), so values agree to ~1e-12 relative — asserted at 1e-9 (north_star tolerance 1e-5)."""
import numpy as np
import pytest

import preint_data as pd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import icgvins
    c = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64)
    yield c
    c.close()


def _close(a, b, tol=1e-9):
    return np.abs(a - b).max() <= tol * max(1e-30, np.abs(b).max())


@pytest.mark.parametrize("variant", [0, 1])
def test_preint_batch_matches_oracle(oracle, ctx, variant):
    lens = [41, 17, 2, 101, 1, 60]  # C4-like intervals incl. the minimum (1 sample = empty interval)
    imus = [pd.make_interval(n, seed=10 + i) for i, n in enumerate(lens)]
    states = [pd.state(p=(i, -i, 0.5 * i), rv=(0.01 * i, -0.02, 0.1 * i + 0.05), v=(2.0, 0.1 * i, 0.0)) for i in range(len(lens))]
    offsets = np.cumsum([0] + lens).astype(np.int32)
    cur, delta, jac, cov, dt, pn = ctx.preint_batch(variant, offsets, np.concatenate(imus), np.stack(states), pd.PARAMS)
    for i, n in enumerate(lens):
        exp = oracle.preint_integrate(variant, imus[i], states[i], pd.PARAMS)
        assert _close(cur[i], exp["cur"]) and _close(delta[i], exp["delta"])
        assert abs(dt[i] - exp["dt"]) < 1e-12
        assert _close(jac[i], exp["jac"]) and _close(cov[i], exp["cov"], 1e-8)
        if variant == 1 and n > 1:
            assert _close(pn[offsets[i]:offsets[i] + n - 1], exp["pn"])


def test_preint_kernel_matches_reference_golden(ctx):
    """k_preint against outputs of the REFERENCE's own preintegration code (tests/golden/preint_ref_golden.npz, generator
    tests/golden/make_preint_golden.py): state, Jacobian and covariance of both variants, 1e-9 relative."""
    from test_oracle_vs_reference import preint_golden_cases
    for variant in (0, 1):
        cases = [c for _, c in preint_golden_cases() if int(c["variant"]) == variant]
        for c in cases:  # the Earth rate is a per-launch parameter: one case per launch
            offsets = np.array([0, len(c["imu"])], np.int32)
            cur, delta, jac, cov, dt, _ = ctx.preint_batch(variant, offsets, c["imu"], c["s0"][None, :], c["params"])
            assert _close(cur[0], c["cur"]) and _close(delta[0], c["delta"])
            assert abs(dt[0] - float(c["dt"])) < 1e-12
            assert _close(jac[0], c["jac"]) and _close(cov[0], c["cov"], 1e-8)
