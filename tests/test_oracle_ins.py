"""SURVEY.md §8 row f4 on the CPU: the restatement (oracle/orc_ins.cc) and the product's host layer (icg::MISC, oracle-backed)
against outputs of the REFERENCE's own misc.cc (tests/golden/ins_ref_golden.npz, generator tests/golden/make_ins_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import ins_utils as iu


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(iu.GOLDEN))


@pytest.fixture(scope="module")
def host_oracle():
    from stream_utils import ensure_oracle_host
    return iu.HostMisc(C.CDLL(ensure_oracle_host()))


def test_oracle_ins_matches_reference_golden(oracle, golden):
    """mechanization series (normal / Earth / scale factors / jittered dt), bracket search, pose prior, IMU series extraction,
    redo-mechanization: decisions exact, states and poses to 1e-15 of the column magnitude (they come out bit-identical: same
    expression order, same libm)"""
    iu.compare(iu.run_all(iu.OrcMisc(oracle.lib)), golden, tol_state=1e-15, tol_pose=1e-15)


@pytest.mark.skipif(not os.path.exists(iu.REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_ins_golden_is_current(golden):
    out = iu.run_all(iu.RefMisc())
    assert set(out) == set(golden)
    for k in golden:
        assert np.array_equal(np.asarray(out[k]), golden[k]), k


def test_host_ins_matches_reference_golden(host_oracle, golden):
    """icg::MISC — window bookkeeping on the host, propagation behind the C ABI (here: the oracle shim)"""
    iu.compare(iu.run_all(host_oracle), golden, tol_state=1e-15, tol_pose=1e-15)


def test_host_ins_batches_equal_single_calls(host_oracle):
    """many streams in ONE mechanization / pose / redo call give what the per-stream calls give"""
    names = ["normal", "earth", "two_samples", "one_sample", "normal_jitter"]
    cases = [iu.mech_case(n) for n in names]
    cfg = iu.make_cfg(True, False)
    st, traj = host_oracle.mechanize_batch(cfg, [c["imu"] for c in cases], [c["s0"] for c in cases])
    for k, c in enumerate(cases):
        s1, t1 = host_oracle.mechanize(cfg, c["imu"], c["s0"])
        assert np.array_equal(st[k], s1) and np.array_equal(traj[k], t1), names[k]
    wins = [np.concatenate([c["s0"][None, :], t]) for c, t in zip(cases, traj)]
    big = [0, 1, 4]
    times = [cases[k]["imu"][len(cases[k]["imu"]) // 2, 0] + 0.001 for k in big]
    poses, found = host_oracle.camera_pose_batch([cases[k]["imu"] for k in big], [wins[k] for k in big], iu.pose_b_c(), times)
    for j, k in enumerate(big):
        p1, f1 = host_oracle.camera_pose(cases[k]["imu"], wins[k], iu.pose_b_c(), times[j])
        assert np.array_equal(poses[j], p1) and int(found[j]) == f1 == 1
    ups = [iu.redo_updates(cases[k], wins[k])[j] for j, k in enumerate(big)]
    res = host_oracle.redo_batch(cfg, ups, 10, [cases[k]["imu"] for k in big], [wins[k] for k in big])
    for j, k in enumerate(big):
        im, s = host_oracle.redo(cfg, ups[j], 10, cases[k]["imu"], wins[k])
        assert np.array_equal(res[j][0], im) and np.array_equal(res[j][1], s)


def test_ins_series_failures(host_oracle, oracle):
    """start or end outside the window: the reference only rejects the case where both are outside (and reads out of bounds when
    one is): both implementations refuse"""
    c = iu.mech_case("normal")
    t = c["imu"][:, 0]
    orc = iu.OrcMisc(oracle.lib)
    for a, b in [(t[0] - 1, t[-1] + 1), (t[0] - 1, t[10] + 0.001), (t[10] + 0.001, t[-1] + 1)]:
        assert host_oracle.imu_series(c["imu"], a, b) is None
        assert orc.imu_series(c["imu"], a, b) is None


def test_mechanization_constant_rate_closed_form(oracle):
    """no rotation, constant specific force, no Earth terms: v = v0 + (R f + g) t, p = p0 + v0 t + 0.5 (R f + g) t^2"""
    n, dt = 201, 0.005
    imu = np.zeros((n, 8))
    imu[:, 0] = 10.0 + dt * np.arange(n)
    imu[:, 1] = dt
    f = np.array([0.4, -0.3, -9.6])
    imu[:, 5:8] = f * dt
    s0 = iu.make_state(imu[0, 0], 9)
    s0[11:17] = 0
    cfg = iu.make_cfg(False, False)
    st, _ = iu.OrcMisc(oracle.lib).mechanize(cfg, imu, s0)
    x, y, z, w = s0[4:8]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    a = R @ f + cfg[:3]
    T = dt * (n - 1)
    assert np.abs(st[8:11] - (s0[8:11] + a * T)).max() < 1e-10
    assert np.abs(st[1:4] - (s0[1:4] + s0[8:11] * T + 0.5 * a * T * T)).max() < 1e-9
    assert np.abs(st[4:8] - s0[4:8]).max() < 1e-14 and st[0] == imu[-1, 0]
