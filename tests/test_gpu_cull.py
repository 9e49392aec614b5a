"""GPU parity for SURVEY.md §8 row f3: k_reproj_error behind icg_reproj_error_batch against outputs of the REFERENCE's own
Camera::reprojectionError / Tracking::isGoodToTrack (tests/golden/cull_ref_golden.npz: errors to 1e-15 relative, gate decisions
exact) and icg::WindowCulling on the HIP library against the Python restatement of the reference's decision loops."""
import numpy as np
import pytest

import cull_utils as cu

pytestmark = pytest.mark.gpu


def test_reproj_error_batch_matches_reference_golden():
    import icgvins
    g = np.load(cu.GOLDEN)
    d = cu.make_observations()
    ctx = icgvins.Context(d["w"], d["h"], n_slots=1, max_batch=1, max_points=64)
    ctx.set_camera(d["cam"])
    for scale, dscale in cu.SCALES:
        err, good = ctx.reproj_error_batch(d["pose_idx"], d["lm_idx"], d["poses12"], d["pw"], d["pix"], cu.REPROJ_STD * scale, cu.NEAREST,
                                           cu.FARTHEST * dscale)
        e = g[f"err_{scale}_{dscale}"]
        assert np.abs(err - e).max() <= 1e-15 * np.abs(e).max()
        assert np.array_equal(good, g[f"good_{scale}_{dscale}"])
    with pytest.raises(icgvins.IcgError):
        ctx.reproj_error_batch(np.array([9], np.int32), np.array([0], np.int32), d["poses12"], d["pw"], d["pix"][:1], 1.0)
    ctx.close()


def test_host_window_culling_on_gpu(oracle):
    import cull_checks as cc
    import harness as H
    cc.check_window_culling(H.HOST_LIB, oracle)
