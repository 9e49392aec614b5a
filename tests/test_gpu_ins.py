"""GPU parity for SURVEY.md §8 row f4: k_ins_mechanize / k_ins_camera_pose behind the C ABI and icg::MISC of the host layer,
against outputs of the REFERENCE's own misc.cc (tests/golden/ins_ref_golden.npz) and against the CPU oracle on larger batches.
FP64 on both sides; the only differing primitives are sin/cos/atan2 (device math library vs glibc): every column of every state
must hold to 1e-12 of its magnitude after up to 160 sequential samples (north_star tolerance 1e-5); all index / bracket /
series decisions are exact."""
import ctypes as C

import numpy as np
import pytest

import ins_utils as iu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import icgvins
    c = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64)
    yield c
    c.close()


@pytest.fixture(scope="module")
def host():
    import harness as H
    return iu.HostMisc(C.CDLL(H.HOST_LIB))


def test_host_ins_matches_reference_golden_on_gpu(host):
    iu.compare(iu.run_all(host), dict(np.load(iu.GOLDEN)), tol_state=1e-12, tol_pose=1e-12)


def _colscale(e):
    return np.maximum(np.abs(e).reshape(-1, e.shape[-1]).max(axis=0), 1e-3)


def test_ins_mechanize_batch_matches_oracle(oracle, ctx):
    """256 streams of different lengths (incl. the empty and the one-sample series) in one launch, Earth + scale-factor terms"""
    orc = iu.OrcMisc(oracle.lib)
    rng = np.random.RandomState(3)
    lens = [int(x) for x in rng.randint(2, 120, 254)] + [1, 0]
    imus = [iu.make_imu(max(n, 1), seed=20 + i, jitter=(i % 2 == 0))[:n] for i, n in enumerate(lens)]
    s0 = [iu.make_state(2000.0, seed=i, scale=True) for i in range(len(lens))]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    for earth, scale in [(True, True), (False, False)]:
        cfg = iu.make_cfg(earth, scale)
        st, traj = ctx.ins_mechanize_batch(off, np.concatenate([i for i in imus if len(i)]), cfg, np.stack(s0))
        for i, n in enumerate(lens):
            if n == 0:
                assert np.array_equal(st[i], s0[i])
                continue
            e_st, e_tr = orc.mechanize(cfg, imus[i], s0[i])
            assert (np.abs(st[i] - e_st) / _colscale(e_st[None, :])).max() < 1e-12, i
            assert np.array_equal(traj[off[i]], s0[i])
            if n > 1:
                assert (np.abs(traj[off[i] + 1:off[i + 1]] - e_tr) / _colscale(e_tr)).max() < 1e-12, i


def test_ins_camera_pose_batch_matches_oracle(oracle, ctx):
    orc = iu.OrcMisc(oracle.lib)
    c, states = iu.window_for(orc, "earth_scale_jitter")
    t = c["imu"][:, 0]
    rng = np.random.RandomState(5)
    n = 300
    idx = rng.randint(1, len(t), n)
    frac = rng.uniform(0, 1, n)
    frac[:10] = 0.0  # exactly on the earlier node
    times = t[idx - 1] + frac * (t[idx] - t[idx - 1])
    times = np.minimum(times, np.nextafter(t[idx], -np.inf))
    br = np.concatenate([states[idx - 1][:, :8], states[idx][:, :8]], axis=1)
    interp = np.ones(n, np.int32)
    interp[-5:] = 0
    pbc = iu.pose_b_c()
    out = ctx.ins_camera_pose_batch(br, interp, pbc, times)
    for i in range(n):
        if interp[i]:
            e, f = orc.camera_pose(c["imu"], states, pbc, times[i])
            assert f == 1
        else:  # first state of the bracket as is == a one-entry window queried outside
            e, f = orc.camera_pose(c["imu"][idx[i] - 1:idx[i]], states[idx[i] - 1:idx[i]], pbc, times[i] + 100.0)
            assert f == 0
        assert np.abs(out[i] - e).max() < 1e-12 * max(1.0, np.abs(e).max()), i


def test_ins_invalid_arguments(ctx):
    import icgvins
    with pytest.raises(icgvins.IcgError):
        ctx.ins_mechanize_batch(np.array([0, 5, 3], np.int32), np.zeros((5, 8)), iu.make_cfg(), np.zeros((2, 23)))
