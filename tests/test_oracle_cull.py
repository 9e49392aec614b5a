"""SURVEY.md §8 row f3 on the CPU: the oracle's per-observation arithmetic against outputs of the REFERENCE's own
Camera::reprojectionError / Tracking::isGoodToTrack (tests/golden/cull_ref_golden.npz), and icg::WindowCulling of the host layer
(oracle-backed) against a Python restatement of the reference's decision loops on the raw landmark graph."""
import os

import numpy as np
import pytest

import cull_utils as cu


def test_oracle_cull_arithmetic_matches_reference_golden(oracle):
    g = np.load(cu.GOLDEN)
    d = cu.make_observations()
    for scale, dscale in cu.SCALES:
        err, good = cu.oracle_eval(oracle, d, scale, dscale)
        assert np.array_equal(err, g[f"err_{scale}_{dscale}"])  # float pixel differences, double norm: bit-identical
        assert np.array_equal(good, g[f"good_{scale}_{dscale}"])


@pytest.mark.skipif(not os.path.exists(os.path.join(cu.ROOT, "oracle", "_ref", "libref_tracking.so")), reason="oracle/_ref not built")
def test_cull_golden_is_current(tmp_path):
    import subprocess
    import sys
    old = dict(np.load(cu.GOLDEN))
    bak = cu.GOLDEN + ".bak"
    os.replace(cu.GOLDEN, bak)
    try:
        subprocess.run([sys.executable, os.path.join(cu.ROOT, "tests", "golden", "make_cull_golden.py")], check=True, stdout=subprocess.DEVNULL)
        new = dict(np.load(cu.GOLDEN))
    finally:
        os.replace(bak, cu.GOLDEN)
    assert set(old) == set(new)
    for k in old:
        assert np.array_equal(old[k], new[k]), k


def test_host_window_culling_on_oracle(oracle):
    import cull_checks as cc
    from stream_utils import ensure_oracle_host
    cc.check_window_culling(ensure_oracle_host(), oracle)
