"""Drives the REFERENCE's own tracker build (oracle/_ref/libref_tracking.so: tracking/*.{h,cc} compiled unmodified against
interface shims, OpenCV entry points forwarded to the oracle primitives) on the synthetic streams of the harness.

The reference keeps process-wide static id counters (frame.cc:38,46, mappoint.cc:47), so one process must run ONE tracker:
`python tests/ref_tracking_utils.py <out.npz> <w> <h> <n_frames> <max_features> <stream>` is the runner the tests (and the
golden generator) spawn."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ic-gvins_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_tracking.so")

CONFIG = dict(min_parallax=20.0, max_interval=0.5, check_hist=False, reproj_std=1.5, window=10)


def scene_and_frames(lib, w, h, n_frames, stream):
    import harness as H
    cam = H.camera_for(w, h)
    scene = H.SynthScene(lib, w, h, cam, tex_size=1024, threads=4)
    scene.vx *= SPEED.get(_current_scenario[0], 1.0)
    scene.vz *= SPEED.get(_current_scenario[0], 1.0)
    slow = SLOW_AFTER.get(_current_scenario[0])
    if slow:
        k0, f = slow
        warp = lambda k: float(k) if k < k0 else k0 + (k - k0) * f
        frames, poses = [], []
        for k in range(n_frames):
            frames.append(scene.render(warp(k), stream=stream))
            R, t = scene.pose(warp(k), stream=stream)
            rng = np.random.RandomState(7000 + k)
            poses.append(H.pose12(R @ H._rot_yp(*rng.normal(0, np.deg2rad(0.1), 2)), t + rng.normal(0, 0.02, 3)))
        stamps = [100.0 + k / 20.0 for k in range(n_frames)]
        return cam, frames, poses, stamps
    frames = [scene.render(k, stream=stream) for k in range(n_frames)]
    for k in BLANK_FRAMES.get(_current_scenario[0], ()):
        frames[k] = np.full_like(frames[k], BLANK_VALUE[_current_scenario[0]])
    if _current_scenario[0] in BGR_SCENARIOS:
        frames = [to_bgr(f) for f in frames]
    poses = [H.pose12(*scene.ins_pose(k, stream=stream)) for k in range(n_frames)]
    stamps = [100.0 + k / 20.0 for k in range(n_frames)]
    return cam, frames, poses, stamps


SCENARIOS = {  # name -> (w, h, n_frames, max_features, stream, check_hist)
    "c1_640x480_100": (640, 480, 40, 100, 0, False),
    "c2_1280x720_300": (1280, 720, 24, 300, 1, False),
    "c1_histgate": (640, 480, 16, 100, 2, True),
    # frames 12..14 are replaced by a featureless image: every track dies -> TRACK_LOST, reset, re-initialisation
    "c1_lost_and_reinit": (640, 480, 30, 100, 3, False),
    # same with the histogram gate on: the brightness jump makes the gate skip frames (TRACK_PASSED) first
    "c1_lost_histgate": (640, 480, 30, 100, 3, True),
    "c1_slow_second_new": (640, 480, 36, 100, 4, False),
    "c4_1920x1080_500": (1920, 1080, 10, 500, 5, False),
    # three-channel input: the BGR->gray conversion in front of everything else (tracking.cc:135-137, F1)
    "c1_bgr": (640, 480, 14, 100, 6, False),
    # long runs: many keyframe decisions, window roll-over several times, rare branches (edge features, quota corner cases)
    "c1_long_160": (640, 480, 160, 100, 7, False),
    "c2_long_60": (1280, 720, 60, 300, 8, True),
}
BGR_SCENARIOS = ("c1_bgr",)


def to_bgr(gray):
    """deterministic colourisation of a rendered gray frame: three different affine maps of the texture, so that the gray image
    the tracker sees is the fixed-point BGR mix (not any single channel)"""
    g = gray.astype(np.int32)
    b = np.clip(g + 20, 0, 255)
    gg = np.clip((g * 7) // 8 + 10, 0, 255)
    r = np.clip(255 - (g * 3) // 4, 0, 255) // 2 + g // 2
    return np.ascontiguousarray(np.stack([b, gg, r], axis=-1).astype(np.uint8))

# camera speed scale per scenario (1.0 = the harness default 16 m/s fly-by); slow motion never reaches the parallax threshold,
# so keyframes come from the time-out branch (KEYFRAME_REMOVE_SECOND_NEW, tracking.cc:283-286) and the window keeper drops them
SPEED = {}
# frame index -> scene time (in frames): normal fly-by for the first frames (initialisation succeeds), then almost stationary
SLOW_AFTER = {"c1_slow_second_new": (8, 0.02)}
BLANK_FRAMES = {"c1_lost_and_reinit": (12, 13, 14), "c1_lost_histgate": (12, 13, 14)}
BLANK_VALUE = {"c1_lost_and_reinit": 90, "c1_lost_histgate": 235}
_current_scenario = [None]


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", f"tracking_ref_{name}.npz")


def run_reference(w, h, n_frames, max_features, stream=0):
    """-> dict(states[n], ids (list of arrays), px (list of (m,2) float32 distorted keypoints), stats[n,8])"""
    from stream_utils import ensure_oracle_host
    synth_lib = C.CDLL(ensure_oracle_host())  # only for the renderer (icgs_*)
    cam, frames, poses, stamps = scene_and_frames(synth_lib, w, h, n_frames, stream)
    lib = C.CDLL(REF_SO)
    lib.ref_tracker_create.restype = C.c_void_p
    tmp = tempfile.mkdtemp(prefix="reftrk_")
    cfg = os.path.join(tmp, "track.yaml")
    with open(cfg, "w") as f:
        f.write(f"track_check_histogram: {'true' if CONFIG['check_hist'] else 'false'}\n"
                f"track_min_parallax: {CONFIG['min_parallax']}\ntrack_max_features: {max_features}\n"
                f"track_max_interval: {CONFIG['max_interval']}\nis_use_visualization: false\n"
                f"reprojection_error_std: {CONFIG['reproj_std']}\n")
    cam_a = np.asarray(cam, np.float64)
    T = C.c_void_p(lib.ref_tracker_create(cam_a.ctypes.data_as(C.c_void_p), w, h, cfg.encode(), tmp.encode(), CONFIG["window"]))
    states, ids_l, px_l, und_l, stats_l, cand_l = [], [], [], [], [], []
    cap = 4 * max_features + 64
    for k in range(n_frames):
        img = np.ascontiguousarray(frames[k])
        p = np.ascontiguousarray(poses[k], np.float64)
        ch = 3 if img.ndim == 3 else 1
        st = lib.ref_tracker_track(T, img.ctypes.data_as(C.c_void_p), w, h, w * ch, ch, C.c_double(stamps[k]), p.ctypes.data_as(C.c_void_p))
        ids, px4 = np.zeros(cap, np.uint64), np.zeros((cap, 4), np.float32)
        typ, vel = np.zeros(cap, np.int32), np.zeros((cap, 2), np.float64)
        n = lib.ref_tracker_features(T, cap, ids.ctypes.data_as(C.c_void_p), px4.ctypes.data_as(C.c_void_p),
                                     typ.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p))
        s8 = np.zeros(8, np.uint64)
        lib.ref_tracker_stats(T, s8.ctypes.data_as(C.c_void_p))
        cc, cr = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        nc = lib.ref_tracker_candidates(T, cap, cc.ctypes.data_as(C.c_void_p), cr.ctypes.data_as(C.c_void_p))
        assert nc >= 0
        cand_l.append(np.concatenate([cc[:nc], cr[:nc]], axis=1))
        states.append(st)
        ids_l.append(ids[:n].copy())
        px_l.append(px4[:n, :2].copy())
        und_l.append(px4[:n, 2:].copy())
        stats_l.append(s8.copy())
    lib.ref_tracker_destroy(T)
    # tracking.txt as the reference's own FileSaver wrote it (tracking.cc:309-315, fileio/filesaver.cc:51-66); the last column is
    # a wall-clock time cost and is dropped
    log = read_tracking_log(os.path.join(tmp, "tracking.txt"))
    return dict(states=np.array(states, np.int32), ids=ids_l, px=px_l, und=und_l, stats=np.stack(stats_l), cand=cand_l, log=log)


def read_tracking_log(path):
    """-> array of strings, one per keyframe decision: the first six columns (stamp, dt, parallax, relative translation, relative
    rotation, feature count) as written; the seventh (time cost in ms) is not deterministic"""
    rows = []
    if os.path.exists(path):
        with open(path) as f:
            for line in f:
                t = line.split()
                if t:
                    assert len(t) == 7, line
                    rows.append(" ".join(t[:6]))
    return np.array(rows, dtype=str)


def save(path, r):
    n = len(r["states"])
    counts = np.array([len(a) for a in r["ids"]], np.int32)
    np.savez(path, states=r["states"], counts=counts, ids=np.concatenate(r["ids"]) if n else np.zeros(0, np.uint64),
             px=np.concatenate(r["px"]) if n else np.zeros((0, 2), np.float32),
             und=np.concatenate(r["und"]) if n else np.zeros((0, 2), np.float32), stats=r["stats"],
             cand_counts=np.array([len(a) for a in r["cand"]], np.int32), cand=np.concatenate(r["cand"]) if n else np.zeros((0, 4), np.float32),
             log=r["log"])


def load(path):
    g = np.load(path)
    off = np.concatenate([[0], np.cumsum(g["counts"])])
    n = len(g["states"])
    coff = np.concatenate([[0], np.cumsum(g["cand_counts"])])
    return dict(cand=[g["cand"][coff[k]:coff[k + 1]] for k in range(n)], states=g["states"], ids=[g["ids"][off[k]:off[k + 1]] for k in range(n)], px=[g["px"][off[k]:off[k + 1]] for k in range(n)],
                und=[g["und"][off[k]:off[k + 1]] for k in range(n)], stats=g["stats"], log=g["log"])


def compare_scenario(lib_path, name, ref=None, engine=None, with_log=None):
    w, h, n, mf, stream, hist = SCENARIOS[name]
    old = CONFIG["check_hist"]
    CONFIG["check_hist"] = hist
    _current_scenario[0] = name
    try:
        compare_with_host(lib_path, ref if ref is not None else load(golden_path(name)), w, h, n, mf, stream, engine=engine, with_log=with_log)
    finally:
        CONFIG["check_hist"] = old
        _current_scenario[0] = None


def compare_with_host(lib_path, ref, w, h, n_frames, max_features, stream=0, engine=None, with_log=None):
    """Runs the product's host layer (lib_path: GPU-backed or oracle-backed) on the same frames and asserts frame-by-frame
    equality with the reference run: track state, map-point ids of the frame's features, distorted key-point float bits, the
    un-triangulated candidate lists (current and reference pixels, list order), and the window bookkeeping (keyframe count,
    window size, landmark count)."""
    import harness as H
    cam = H.camera_for(w, h)
    logdir = tempfile.mkdtemp(prefix="icgtrk_")
    # tracking.txt of the device-resident tracker: the decision's numbers come back with the step result (icg_tracker_result.log_*, round 4).
    # Compared on the CPU backend of the tracker ABI (tests/test_host_engines_cpu.py); the callers on the device do not ask for it yet.
    if with_log is None:
        with_log = engine != "device"
    if with_log:
        os.environ["ICG_TRACKING_LOG_DIR"] = logdir
    try:
        sb = H.StreamBatch(lib_path, 1, w, h, cam, max_features=max_features, window=CONFIG["window"], min_parallax=CONFIG["min_parallax"],
                           max_interval=CONFIG["max_interval"], check_hist=CONFIG["check_hist"], reproj_std=CONFIG["reproj_std"], engine=engine)
    finally:
        os.environ.pop("ICG_TRACKING_LOG_DIR", None)
    _, frames, poses, stamps = scene_and_frames(sb.lib, w, h, n_frames, stream)
    for k in range(n_frames):
        ch = 3 if frames[k].ndim == 3 else 1
        st = sb.step([frames[k].ctypes.data], w * ch, [stamps[k]], poses[k], channels=ch)
        ids, px = sb.features(0)
        assert int(st[0]) == int(ref["states"][k]), (k, int(st[0]), int(ref["states"][k]))
        assert np.array_equal(ids.astype(np.uint64), ref["ids"][k]), (k, len(ids), len(ref["ids"][k]))
        assert np.array_equal(px.view(np.uint32), ref["px"][k].view(np.uint32)), k
        s = sb.stats(0)
        assert s["keyframes"] == int(ref["stats"][k][1]) and s["window_keyframes"] == int(ref["stats"][k][5]), (k, s, ref["stats"][k])
        assert s["landmarks"] == int(ref["stats"][k][6]), (k, s, ref["stats"][k])
        cur, rf = sb.candidates(0)
        cand = np.concatenate([cur, rf], axis=1)
        assert cand.shape == ref["cand"][k].shape, (k, cand.shape, ref["cand"][k].shape)
        assert np.array_equal(cand.view(np.uint32), ref["cand"][k].view(np.uint32)), k
    sb.close()
    if not with_log:
        return
    # tracking.txt: same rows, same text in the six deterministic columns
    log = read_tracking_log(os.path.join(logdir, "stream0", "tracking.txt"))
    assert len(log) == len(ref["log"]), (len(log), len(ref["log"]))
    for a, b in zip(log, ref["log"]):
        assert a == str(b), (a, str(b))


def run_scenario_in_subprocess(name, out_path):
    import subprocess
    subprocess.run([sys.executable, os.path.abspath(__file__), out_path, name], check=True)


if __name__ == "__main__":
    out, name = sys.argv[1], sys.argv[2]
    w, h, n, mf, stream, hist = SCENARIOS[name]
    CONFIG["check_hist"] = hist
    _current_scenario[0] = name
    save(out, run_reference(w, h, n, mf, stream))
