#!/usr/bin/env python3
"""Generates tests/golden/gvins_ref_*golden.npz (three scenarios: the plain sequence; Earth rotation + time-delay estimation; half a second of
black images = tracking loss and re-initialization): the result files of the REFERENCE's own estimator (GVINS of ic_gvins.cc, compiled unmodified into
oracle/_ref/libref_gvins.so — see oracle/ref_build/ref_gvins.cc for what the interface shims replace) on the synthetic GNSS + IMU + camera
sequence of tests/gvins_data.py, played into its three threads three times slower than real time.  The reference's output depends on thread
timing; this is one run of it.  Build container only:
    make -C oracle && make -C oracle/ref_build && python tests/golden/make_gvins_golden.py"""
import ctypes as C
import os
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gvins_data as gd  # noqa: E402
import ref_gvins_utils as ru  # noqa: E402
from stream_utils import ensure_oracle_host  # noqa: E402


def input_checksums(files):
    """crc32 of the generated input files: the golden is only meaningful for byte-identical inputs"""
    out = [zlib.crc32(open(files["imu"], "rb").read()), zlib.crc32(open(files["gnss"], "rb").read())]
    root = os.path.dirname(files["images"])
    names = [line.split()[1] for line in open(files["images"])]
    for name in (names[0], names[len(names) // 2], names[-1]):
        out.append(zlib.crc32(open(os.path.join(root, name), "rb").read()))
    return np.array(out, np.int64)


def blank_images(files, root, seq, first=40, last=50):
    names = [line.split()[1] for line in open(files["images"])]
    for name in names[first:last]:
        with open(os.path.join(root, "cam0", name), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (seq.w, seq.h) + bytes(seq.w * seq.h))


SCENARIOS = {name: (os.path.basename(path), kwargs, blank is not None) for name, (path, kwargs, blank) in ru.SCENARIOS.items()}

def one_run(name, root):
    """one run of the reference estimator on scenario `name` with its files under `root` (executed in a child process: the reference's threads
    signal each other without predicates, so a run can stall for good — the parent kills it after a time limit and tries again)"""
    golden, kwargs, blank = SCENARIOS[name]
    lib = C.CDLL(ensure_oracle_host())  # only its scene renderer is used here
    seq = gd.Sequence(lib)
    files = seq.write(root, **kwargs)
    if blank:
        blank_images(files, root, seq)
    out = os.path.join(root, "ref_out")
    # the scenario with more features and a short window keeps the shim's dense LM busy for longer: slower pacing, so that the optimizer still
    # finishes between two frames (the schedule the product's event loop follows)
    state = ru.run_reference(files, out, seq.w, seq.h, slowdown=8.0 if name == "small_window_calibration" else 3.0)
    load = lambda f: np.loadtxt(os.path.join(out, f))
    if not (state == 4 and len(load("statistics.txt")) >= 25 and len(load("trajectory.csv")) == 110):
        print(name, "incomplete: state", state, "statistics rows", len(load("statistics.txt")))
        return 1
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", golden), final_state=state, trajectory=load("trajectory.csv"), nav=load("gvins.nav"),
                        statistics=load("statistics.txt"), tracking=load("tracking.txt"), mappoints=load("mappoint.txt"), checksums=input_checksums(files),
                        extrinsic=(load("extrinsic.txt") if os.path.getsize(os.path.join(out, "extrinsic.txt")) else np.zeros((0, 8))),
                        imu_err=np.fromfile(os.path.join(out, "IMU_ERR.bin"), np.float64).reshape(-1, 8))
    print(name, "trajectory rows", len(load("trajectory.csv")), "statistics rows", len(load("statistics.txt")), "mappoints", len(load("mappoint.txt")))
    return 0


if __name__ == "__main__":
    import subprocess
    if len(sys.argv) == 4 and sys.argv[1] == "--worker":
        sys.exit(one_run(sys.argv[2], sys.argv[3]))
    for name in (sys.argv[1:] or list(SCENARIOS)):
        for attempt in range(5):
            root = tempfile.mkdtemp(prefix="gvins_golden_")
            try:
                rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", name, root], timeout=240).returncode
            except subprocess.TimeoutExpired:
                rc = -1
                print(name, "attempt", attempt, "stalled: killed")
            if rc == 0:
                break
        else:
            raise SystemExit("the reference estimator did not complete scenario " + name)
