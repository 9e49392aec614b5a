"""SURVEY.md §8 row f2 on the MI355X: the replay harness driving icg::GVINS on the HIP-backed host layer — tracker, INS series launches,
pose priors, preintegration, window solves with device-side landmark elimination, culling and marginalization assembly all run on the
device.  Same checks as test_host_gvins_cpu.py (truth of the synthetic sequence, file formats, determinism), plus agreement of the GPU
run with the oracle-backed run of the same files."""
import ctypes as C
import os

import numpy as np
import pytest

import gvins_checks as gc
import gvins_data as gd
import harness as H

pytestmark = pytest.mark.gpu


def test_replay_gnss_imu_camera_sequence_on_gpu(tmp_path):
    gc.check_replay(H.HOST_LIB, tmp_path, bitwise=True)


def test_replay_online_calibration_and_earth_rotation_on_gpu(tmp_path):
    gc.check_replay_calibration(H.HOST_LIB, tmp_path)


def test_replay_concurrent_estimators_on_gpu(tmp_path):
    """four camera streams' estimators side by side on one GPU (poll waits): each equals the stream run alone bit for bit, with an
    identical tracking log"""
    gc.check_replay_concurrent(H.HOST_LIB, tmp_path, n=4, bitwise=True, wait_poll_us=50)


def test_replay_lockstep_shared_window_solves_on_gpu(tmp_path):
    """four estimators in lock-step, every window solve of a tick through ONE batched launch sequence (k_reproj_eval, k_asm_*, k_schur_*_w):
    each stream equals the stream replayed alone bit for bit"""
    gc.check_replay_lockstep(H.HOST_LIB, tmp_path, n=4, bitwise=True, groups=2)


def test_replay_tracking_loss_and_reinitialization_on_gpu(tmp_path):
    gc.check_replay_tracking_loss(H.HOST_LIB, tmp_path)


@pytest.mark.parametrize("scenario", ["default", "earth_td", "loss"])
def test_estimator_on_gpu_against_reference_estimator_golden(tmp_path, scenario):
    import ref_gvins_utils as ru
    golden, kwargs, blank = ru.SCENARIOS[scenario]
    gc.check_against_reference_estimator(H.HOST_LIB, tmp_path, golden, kwargs, blank, pos_tol=0.10 if blank else 0.05)


def test_replay_gpu_agrees_with_oracle_backend(tmp_path):
    """the same files through the HIP-backed and the oracle-backed host layer: the front-end is bit-exact, the FP64 paths agree to rounding,
    so keyframe / landmark bookkeeping is identical and the trajectories agree to well below the estimator's accuracy"""
    from stream_utils import ensure_oracle_host
    gpu, cpu = C.CDLL(H.TOOLS_LIB), C.CDLL(ensure_oracle_host())
    seq = gd.Sequence(gpu)
    files = seq.write(str(tmp_path))
    Sg = gc.run_replay(gpu, files)
    tg = np.loadtxt(os.path.join(files["out"], "trajectory.csv"))
    Sc = gc.run_replay(cpu, files)
    tc = np.loadtxt(os.path.join(files["out"], "trajectory.csv"))
    for k in ("frames_tracked", "keyframes", "optimizations", "marginalizations", "lost", "final_state"):
        assert Sg[k] == Sc[k], k
    assert tg.shape == tc.shape and np.array_equal(tg[:, 0], tc[:, 0])
    assert np.abs(tg[:, 1:4] - tc[:, 1:4]).max() < 5e-3 and np.abs(tg[:, 4:8] - tc[:, 4:8]).max() < 1e-4
