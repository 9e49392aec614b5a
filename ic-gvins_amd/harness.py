"""Replay harness (plumbing for tests and bench.py): synthetic camera streams + a ctypes driver for the host
layer's multi-stream executor (libicgvins_host.so, C entry points in host/capi.cc).

The library path is a parameter: the product path loads ic-gvins_amd/libicgvins_host.so (HIP-backed, no fallback);
tests and bench.py's cpu_baseline leg may pass oracle/libicgvins_host_oracle.so (same host code linked on the CPU
restatement) — never the other way round.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "libicgvins_host.so")
# estimator port + replay harness + scene renderer (ic-gvins_amd/tools/): NOT part of the product libraries; it links on top of them,
# so a handle on it also resolves the product's icgh_* entry points
TOOLS_LIB = os.path.join(_HERE, "libicgvins_tools.so")


def tools_lib(host_lib_path):
    """library that carries icgs_* / icgh_replay_* / icgh_nav_* for a given host library: the tools library for the product's
    host layer, the library itself for the all-in-one oracle-backed checker build"""
    return TOOLS_LIB if os.path.abspath(host_lib_path) == os.path.abspath(HOST_LIB) else host_lib_path

TRACK_STATES = ["FIRST_FRAME", "INITIALIZING", "TRACKING", "PASSED", "LOST"]


def camera_for(width, height):
    """Reference intrinsics/distortion (config/gvins.yaml:65,69) with the principal point at the image centre and the
    focal length scaled with the image width relative to 1280."""
    s = width / 1280.0
    return [787.1611861559479 * s, 787.3928431375225 * s, width / 2.0, height / 2.0, 0.0, -0.0917403092279957,
            0.08134715036932794, 0.00017620136958692255, 0.00016737385248865412, 0.0]


def _rot_yp(yaw, pitch):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])       # about camera y (down)
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])       # about camera x (right)
    return Ry @ Rx


class SynthScene:
    """A slanted textured wall ~20 m in front of a camera that flies past it (x right, y down, z forward)."""

    def __init__(self, lib, width, height, cam10, tex_size=2048, seed=7, threads=8):
        if not hasattr(lib, "icgs_make_texture"):  # a handle on the product's host library: the renderer lives in the tools library
            lib = C.CDLL(TOOLS_LIB)
        self.lib, self.w, self.h, self.cam = lib, width, height, np.asarray(cam10, np.float64)
        self.tex_size, self.threads = tex_size, threads
        self.tex = np.zeros((tex_size, tex_size), np.uint8)
        lib.icgs_make_texture(tex_size, C.c_uint32(seed), self.tex.ctypes.data_as(C.c_void_p))
        self.rays = np.zeros((height, width, 2), np.float32)
        lib.icgs_ray_table(self.cam.ctypes.data_as(C.c_void_p), width, height, self.rays.ctypes.data_as(C.c_void_p))
        a = np.deg2rad(25.0)
        self.n = np.array([-np.sin(a), 0.0, np.cos(a)])
        self.plane = np.array([*self.n, 20.0 * np.cos(a)])
        self.e1 = np.array([np.cos(a), 0.0, np.sin(a)])
        self.e2 = np.array([0.0, 1.0, 0.0])
        self.texels_per_m = 40.0 * width / 1280.0
        self.vx, self.vz = 16.0, 1.0  # sideways / forward speed, m/s

    def pose(self, k, fps=20.0, stream=0):
        """True camera pose of frame k: (R camera->world 3x3, t 3)."""
        tau = k / fps
        ph = 0.37 * stream
        # streams start 7.3 m apart ALONG the wall (in-plane direction e1), so every stream sees the wall at the same
        # depth and only the texture patch differs (an offset along x would push stream k 3.4*k m further from the wall)
        t = np.array([self.vx * tau, 0.3 * np.sin(0.7 * tau + ph), self.vz * tau]) + 7.3 * stream * self.e1
        R = _rot_yp(0.05 * np.sin(0.5 * tau + ph), 0.03 * np.sin(0.8 * tau + ph))
        return R, t

    def ins_pose(self, k, fps=20.0, stream=0, seed=0):
        """INS prior = truth + N(0, 0.02 m / 0.1 deg), deterministic per (stream, frame)."""
        R, t = self.pose(k, fps, stream)
        rng = np.random.RandomState((seed * 1000003 + stream * 7919 + k) & 0x7fffffff)
        dt = rng.normal(0, 0.02, 3)
        dy, dp = rng.normal(0, np.deg2rad(0.1), 2)
        return R @ _rot_yp(dy, dp), t + dt

    def render(self, k, fps=20.0, stream=0):
        R, t = self.pose(k, fps, stream)
        pose12 = np.concatenate([R.ravel(), t]).astype(np.float64)
        out = np.zeros((self.h, self.w), np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.icgs_render(p(self.tex), self.tex_size, C.c_double(self.texels_per_m), p(self.rays), self.w, self.h, p(pose12),
                             p(self.plane), p(self.e1), p(self.e2), p(out), self.w, self.threads)
        return out


def pingpong(k, n):
    """frame index sequence 0,1,..,n-1,n-2,..,1,0,1,.. (continuous motion for arbitrarily long runs)"""
    period = 2 * (n - 1)
    m = k % period
    return m if m < n else period - m


class StreamBatch:
    """ctypes driver of icgh_batch (host/capi.cc): N independent streams tracked in lock-step on one device."""

    def __init__(self, lib_path, n_streams, width, height, cam10, max_features=300, window=10, min_parallax=20.0,
                 max_interval=0.5, check_hist=False, reproj_std=1.5, device=0, host_threads=1, groups=1, engine=None):
        """engine: None (ICG_TRACK_ENGINE or the default, the track table), "table" or "object" (the reference-shaped object graph
        throughout; on the table engine the entry points that work on the tracker's icg::Map — culling, window refinement, landmark
        tables — get an object view of the table and write their results back)."""
        if not os.path.exists(lib_path):
            raise RuntimeError(f"{lib_path} not found (build first; there is no fallback)")
        self.lib = C.CDLL(lib_path)
        old_engine = os.environ.get("ICG_TRACK_ENGINE")
        if engine is not None:
            os.environ["ICG_TRACK_ENGINE"] = engine
        self.lib.icgh_batch_create.restype = C.c_void_p
        self.lib.icgh_batch_ctx.restype = C.c_void_p
        self.lib.icgh_batch_ctx.argtypes = [C.c_void_p, C.c_int]
        self.lib.icgh_batch_destroy.argtypes = [C.c_void_p]
        self.n, self.w, self.h = n_streams, width, height
        err = C.create_string_buffer(512)
        cam = np.asarray(cam10, np.float64)
        self.h_ = self.lib.icgh_batch_create(device, n_streams, cam.ctypes.data_as(C.c_void_p), width, height, max_features,
                                             C.c_double(min_parallax), C.c_double(max_interval), 1 if check_hist else 0,
                                             C.c_double(reproj_std), window, host_threads, groups, err, 512)
        if engine is not None:
            if old_engine is None:
                del os.environ["ICG_TRACK_ENGINE"]
            else:
                os.environ["ICG_TRACK_ENGINE"] = old_engine
        if not self.h_:
            raise RuntimeError("icgh_batch_create failed: " + err.value.decode())
        self._err = err

    def engine(self):
        return ["table", "object", "core", "device"][self.lib.icgh_batch_engine(C.c_void_p(self.h_))]

    def dump(self, stream, kind=0):
        """canonical text of a stream's tracker + map state: kind 0 the engine's state, 1 the map part (table engine), 2 the map part
        computed from the materialized object graph (table engine's B2 view)"""
        self.lib.icgh_batch_dump.restype = C.c_long
        self.lib.icgh_batch_dump.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_long]
        n = self.lib.icgh_batch_dump(C.c_void_p(self.h_), stream, kind, None, C.c_long(0))
        if n < 0:
            raise RuntimeError(f"icgh_batch_dump failed: {n}")
        buf = C.create_string_buffer(n + 1)
        self.lib.icgh_batch_dump(C.c_void_p(self.h_), stream, kind, buf, C.c_long(n + 1))
        return buf.value.decode()

    def close(self):
        if getattr(self, "h_", None):
            self.lib.icgh_batch_destroy(C.c_void_p(self.h_))
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_groups(self):
        return self.lib.icgh_batch_groups(C.c_void_p(self.h_))

    def ctx_handle(self, group=0):
        return self.lib.icgh_batch_ctx(C.c_void_p(self.h_), group)

    def step(self, image_ptrs, stride, stamps, poses12, on_device=False, channels=1):
        """image_ptrs: list of int addresses (host or device) or None per stream."""
        ptrs = (C.c_void_p * self.n)(*[(p if p else None) for p in image_ptrs])
        stamps = np.ascontiguousarray(stamps, np.float64)
        poses12 = np.ascontiguousarray(poses12, np.float64)
        states = np.zeros(self.n, np.int32)
        rc = self.lib.icgh_batch_step(C.c_void_p(self.h_), ptrs, stride, channels, 1 if on_device else 0,
                                      stamps.ctypes.data_as(C.c_void_p), poses12.ctypes.data_as(C.c_void_p),
                                      states.ctypes.data_as(C.c_void_p), self._err, 512)
        if rc != 0:
            raise RuntimeError("icgh_batch_step failed: " + self._err.value.decode())
        return states

    def run(self, image_ptrs, stride, stamps, poses12, on_device=False, channels=1):
        """K steps in one call. image_ptrs: K x n nested list of addresses; stamps (K,n); poses12 (K,n,12)."""
        K = len(image_ptrs)
        flat = [(p if p else None) for row in image_ptrs for p in row]
        ptrs = (C.c_void_p * (K * self.n))(*flat)
        stamps = np.ascontiguousarray(stamps, np.float64).reshape(K, self.n)
        poses12 = np.ascontiguousarray(poses12, np.float64).reshape(K, self.n, 12)
        states = np.zeros((K, self.n), np.int32)
        rc = self.lib.icgh_batch_run(C.c_void_p(self.h_), K, ptrs, stride, channels, 1 if on_device else 0,
                                     stamps.ctypes.data_as(C.c_void_p), poses12.ctypes.data_as(C.c_void_p),
                                     states.ctypes.data_as(C.c_void_p), self._err, 512)
        if rc != 0:
            raise RuntimeError("icgh_batch_run failed: " + self._err.value.decode())
        return states

    def stats(self, stream):
        out = np.zeros(8, np.uint64)
        self.lib.icgh_batch_stats(C.c_void_p(self.h_), stream, out.ctypes.data_as(C.c_void_p))
        keys = ["frames", "keyframes", "tracked_sum", "digest", "mappoints_created", "window_keyframes", "landmarks", "last_state"]
        return dict(zip(keys, [int(v) for v in out]))

    def stats_all(self):
        """stats() of every stream in one call"""
        out = np.zeros((self.n, 8), np.uint64)
        if not hasattr(self.lib, "icgh_batch_stats_all") or self.lib.icgh_batch_stats_all(C.c_void_p(self.h_), out.ctypes.data_as(C.c_void_p)) != 0:
            return [self.stats(s) for s in range(self.n)]
        keys = ["frames", "keyframes", "tracked_sum", "digest", "mappoints_created", "window_keyframes", "landmarks", "last_state"]
        return [dict(zip(keys, [int(v) for v in row])) for row in out]

    def candidates(self, stream, cap=4096):
        """un-triangulated candidate points of the tracker (current and reference pixel), list order"""
        cur, ref = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        n = self.lib.icgh_batch_candidates(C.c_void_p(self.h_), stream, cap, cur.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise RuntimeError(f"icgh_batch_candidates failed: {n}")
        return cur[:n].copy(), ref[:n].copy()

    def counters(self, reset=True):
        out = np.zeros(8, np.uint64)
        self.lib.icgh_batch_counters(C.c_void_p(self.h_), out.ctypes.data_as(C.c_void_p), 1 if reset else 0)
        keys = ["lk_points", "lk_calls", "detect_jobs", "detect_calls", "ransac_sets", "ransac_calls", "frames", "tri_points"]
        return dict(zip(keys, [int(v) for v in out]))

    def timing(self, reset=True):
        out = np.zeros(5, np.float64)
        self.lib.icgh_batch_timing(C.c_void_p(self.h_), out.ctypes.data_as(C.c_void_p), 1 if reset else 0)
        return dict(zip(["host_logic", "gather", "device_execute", "scatter", "finalize"], [float(v) for v in out]))

    def timing_groups(self):
        out = []
        for g in range(self.n_groups()):
            t = np.zeros(5, np.float64)
            self.lib.icgh_batch_timing_group(C.c_void_p(self.h_), g, t.ctypes.data_as(C.c_void_p))
            out.append(t.copy())
        return np.stack(out)

    def step_log(self, reset=True, cap=1 << 14):
        """per group: array (steps, 3) = [steady-clock end time, host-logic s, device-execute s] of every step since the last reset"""
        out = []
        for g in range(self.n_groups()):
            buf = np.zeros((cap, 3), np.float64)
            n = self.lib.icgh_batch_step_log(C.c_void_p(self.h_), g, buf.ctypes.data_as(C.c_void_p), cap, 1 if reset else 0)
            out.append(buf[:max(0, n)].copy())
        return out

    def now(self):
        self.lib.icgh_now_s.restype = C.c_double
        return float(self.lib.icgh_now_s())

    def features(self, stream, max_n=2048):
        ids = np.zeros(max_n, np.uint64)
        px = np.zeros((max_n, 2), np.float32)
        n = self.lib.icgh_batch_features(C.c_void_p(self.h_), stream, max_n, ids.ctypes.data_as(C.c_void_p),
                                         px.ctypes.data_as(C.c_void_p))
        return ids[:n], px[:n]


def pose12(R, t):
    return np.concatenate([np.asarray(R, np.float64).ravel(), np.asarray(t, np.float64)])
