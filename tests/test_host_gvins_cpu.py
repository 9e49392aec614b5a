"""SURVEY.md §8 row f2 on the CPU: the replay harness (IMU / GNSS text, PGM image list, gvins.yaml) driving icg::GVINS — the estimator
that calls the hot path from both sides — on the oracle-backed host layer, over a synthetic GNSS + IMU + camera sequence with known
truth (gvins_data.py).  Parity status: the window solver is unpinned (Ceres is absent); the anchors are the truth of the sequence, the
reference's file formats and determinism.  The small factors / Earth helpers the estimator adds are pinned in test_oracle_vs_reference.py."""
import ctypes as C

import numpy as np
import pytest

import gvins_checks as gc


@pytest.fixture(scope="module")
def host_lib():
    from stream_utils import ensure_oracle_host
    return ensure_oracle_host()


def test_replay_gnss_imu_camera_sequence(host_lib, tmp_path):
    gc.check_replay(host_lib, tmp_path)


def test_replay_online_calibration_and_earth_rotation(host_lib, tmp_path):
    gc.check_replay_calibration(host_lib, tmp_path)


def test_replay_time_window_and_gnss_outage(host_lib, tmp_path):
    gc.check_replay_window(host_lib, tmp_path)


def test_replay_concurrent_estimators_are_independent(host_lib, tmp_path):
    gc.check_replay_concurrent(host_lib, tmp_path, n=3)


def test_replay_lockstep_with_different_streams(host_lib, tmp_path):
    gc.check_replay_lockstep_different_streams(host_lib, tmp_path)


def test_replay_lockstep_shared_marginalizations(host_lib, tmp_path):
    gc.check_replay_lockstep_shared_marginalizations(host_lib, tmp_path)


def test_replay_lockstep_batches_wide_windows(host_lib, tmp_path):
    gc.check_replay_lockstep_wide_windows(host_lib, tmp_path)


def test_replay_tracking_loss_and_reinitialization(host_lib, tmp_path):
    gc.check_replay_tracking_loss(host_lib, tmp_path)


def test_replay_input_formats(host_lib, tmp_path):
    gc.check_replay_input_formats(host_lib, tmp_path)


@pytest.mark.parametrize("scenario", ["default", "earth_td", "loss", "small_window_calibration"])
def test_estimator_against_reference_estimator_golden(host_lib, tmp_path, scenario):
    """the reference's own ic_gvins.cc (oracle/_ref/libref_gvins.so, goldens made by tests/golden/make_gvins_golden.py) on the same files:
    the plain sequence; Earth rotation (INS + PreintegrationEarth with the reference's effective zero station, hazard H9) with time-delay
    estimation; half a second of black images (TRACK_LOST, empty keyframes, re-initialization)"""
    import ref_gvins_utils as ru
    golden, kwargs, blank = ru.SCENARIOS[scenario]
    r = gc.check_against_reference_estimator(host_lib, tmp_path, golden, kwargs, blank, n_keyframe_features_exact=7 if scenario == "small_window_calibration" else 10,
                                             pos_tol=0.05 if scenario in ("default", "earth_td") else 0.15)  # the reference's own runs of the other scenarios differ by 4-10 cm
    assert r["median_position_difference"] < 0.01, r


def test_icg_replay_command_line(host_lib, tmp_path):
    """the command-line front (tools/icg_replay_main.cc) linked on the oracle-backed host layer: one stream, two streams side by side and two
    streams as a lock-step group write identical trajectories"""
    import os
    import subprocess
    import gvins_data as gd
    exe = os.path.join(os.path.dirname(host_lib), "icg_replay_oracle")
    assert os.path.exists(exe), "make -C oracle builds it"
    files = gd.Sequence(C.CDLL(host_lib), duration=6.0).write(str(tmp_path))
    base = [exe, "--config", files["config"], "--imu", files["imu"], "--gnss", files["gnss"], "--images", files["images"]]
    r = subprocess.run(base + ["--output", str(tmp_path / "one")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "final state 4" in r.stdout, (r.stdout, r.stderr)
    one = (tmp_path / "one" / "trajectory.csv").read_bytes()
    r = subprocess.run(base + ["--output", str(tmp_path / "many"), "--streams", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "replayed 2 streams" in r.stdout, (r.stdout, r.stderr)
    r = subprocess.run(base + ["--output", str(tmp_path / "lock"), "--streams", "2", "--lockstep-groups", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "largest batch 2" in r.stdout, (r.stdout, r.stderr)
    for d in ("many", "lock"):
        for k in range(2):
            assert (tmp_path / d / ("stream%d" % k) / "trajectory.csv").read_bytes() == one, (d, k)
    r = subprocess.run([exe, "--config", files["config"]], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_replay_input_errors(host_lib, tmp_path):
    lib = C.CDLL(host_lib)
    err = C.create_string_buffer(512)
    summ = np.zeros(16)
    p = lambda s: str(s).encode()
    # missing configuration / IMU file: an error string, no crash
    rc = lib.icgh_replay_run(p(tmp_path / "none.yaml"), p(tmp_path), p(tmp_path / "imu.txt"), None, None, 0, C.c_double(0), C.c_double(0),
                             summ.ctypes.data_as(C.c_void_p), err, 512)
    assert rc != 0 and b"cannot open" in err.value
    # a configuration value that is not a number is refused (yaml-cpp would throw BadConversion), it is not read as 0
    cfg = tmp_path / "bad.yaml"
    cfg.write_text("outputpath: \"%s\"\ninitlength: 1\nimudatarate: fast\n" % tmp_path)
    (tmp_path / "imu.txt").write_text("100000.0 0 0 0 0 0 0\n100000.005 0 0 0 0 0 -0.049\n")
    rc = lib.icgh_replay_run(p(cfg), None, p(tmp_path / "imu.txt"), None, None, 0, C.c_double(0), C.c_double(0), summ.ctypes.data_as(C.c_void_p), err, 512)
    assert rc != 0 and b"not a number" in err.value, err.value
    # a truncated image is reported with its path
    img = tmp_path / "x.pgm"
    img.write_bytes(b"P5\n8 8\n255\n" + bytes(10))
    lib.icgh_replay_load_pnm.restype = C.c_int
    dims = np.zeros(3, np.int32)
    rc = lib.icgh_replay_load_pnm(p(img), dims.ctypes.data_as(C.c_void_p), None, 0, err, 512)
    assert rc != 0 and b"truncated" in err.value
    img.write_bytes(b"P6\n# comment\n2 1\n255\n" + bytes([1, 2, 3, 4, 5, 6]))
    out = np.zeros(6, np.uint8)
    rc = lib.icgh_replay_load_pnm(p(img), dims.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 6, err, 512)
    assert rc == 0 and list(dims) == [1, 2, 3] and list(out) == [3, 2, 1, 6, 5, 4]  # RGB -> BGR8, as the reference's colour input
