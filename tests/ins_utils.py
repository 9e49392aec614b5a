"""Test data and drivers for SURVEY.md §8 row f4 (INS mechanization, pose prior from the INS window, IMU series extraction,
redo-mechanization).  Three implementations are driven through the same flat-array layout (imu rows of 8: time, dt, dtheta3,
dvel3; state rows of 23: time, p3, q4 xyzw, v3, bg3, ba3, sg3, sa3; cfg8: gravity3, iewn3, iswithearth, iswithscale):

    RefMisc    the REFERENCE's own misc.cc (oracle/_ref/libref_misc.so) — golden generator, build container only
    OrcMisc    the CPU restatement (oracle/liboracle.so: orc_ins_*)
    HostMisc   the product's host layer (icgh_ins_*: icg::MISC on the C ABI) — oracle-backed on CPU, HIP-backed on the GPU
"""
import ctypes as C
import os
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_misc.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "ins_ref_golden.npz")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


# ---- data ---------------------------------------------------------------------------------------------------------------
def make_imu(n=161, rate=200.0, seed=0, t0=2000.0, jitter=False):
    """smooth vehicle-like motion + sensor noise; optional timing jitter so that dt varies from sample to sample"""
    rng = np.random.RandomState(seed)
    imu = np.zeros((n, 8))
    t = t0
    for k in range(n):
        dt = 1.0 / rate * (1.0 + (rng.uniform(-0.05, 0.05) if jitter else 0.0))
        t += dt
        w = np.array([0.03 * np.sin(0.9 * k / rate), -0.05 * np.cos(0.5 * k / rate), 0.25 + 0.1 * np.sin(0.3 * k / rate)])
        a = np.array([0.6 * np.sin(1.3 * k / rate), 0.4 * np.cos(0.7 * k / rate), -9.79 + 0.2 * np.sin(2.0 * k / rate)])
        imu[k] = [t, dt, *(w * dt + rng.normal(0, 2e-5, 3)), *(a * dt + rng.normal(0, 2e-4, 3))]
    return imu


def make_state(time, seed=0, scale=False):
    rng = np.random.RandomState(100 + seed)
    rv = rng.normal(0, 0.3, 3)
    a = np.linalg.norm(rv)
    q = np.array([*(np.sin(a / 2) * rv / a), np.cos(a / 2)])
    s = np.zeros(23)
    s[0] = time
    s[1:4] = rng.normal(0, 20, 3)
    s[4:8] = q
    s[8:11] = [6.0, 1.0, -0.2] + rng.normal(0, 0.5, 3)
    s[11:14] = rng.normal(0, 2e-4, 3)
    s[14:17] = rng.normal(0, 2e-3, 3)
    if scale:
        s[17:20] = rng.normal(0, 8e-4, 3)
        s[20:23] = rng.normal(0, 8e-4, 3)
    return s


def make_cfg(earth=False, scale=False):
    iewn = [7.292115e-5 * np.cos(0.53), 0.0, -7.292115e-5 * np.sin(0.53)] if earth else [0.0, 0.0, 0.0]
    return np.array([0.0, 0.0, 9.7936, *iewn, 1.0 if earth else 0.0, 1.0 if scale else 0.0])


POSE_B_C = None


def pose_b_c():
    """body -> camera extrinsic (R row-major 9, t 3): camera looking forward, small mounting misalignment"""
    rv = np.array([1.2, -1.2, 1.19])
    a = np.linalg.norm(rv)
    k = rv / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    return np.concatenate([R.reshape(-1), [0.12, -0.05, 0.31]])


MECH_CASES = [  # name, earth, scale, jitter, n, seed
    ("normal", False, False, False, 161, 0),
    ("normal_jitter", False, False, True, 97, 1),
    ("earth", True, False, False, 121, 2),
    ("earth_scale_jitter", True, True, True, 141, 3),
    ("normal_scale", False, True, False, 64, 4),
    ("two_samples", True, False, False, 2, 5),
    ("one_sample", False, False, False, 1, 6),
]


def mech_case(name):
    for nm, earth, scale, jitter, n, seed in MECH_CASES:
        if nm == name:
            imu = make_imu(n, seed=seed, jitter=jitter)
            return dict(cfg=make_cfg(earth, scale), imu=imu, s0=make_state(imu[0, 0], seed, scale))
    raise KeyError(name)


def query_times(imu):
    """times exercising every branch of the bracket search / near-node tests on a window with these IMU times"""
    t = imu[:, 0]
    n = len(t)
    q = [t[0] - 0.01, t[0], t[0] + 1e-5, 0.5 * (t[0] + t[1]), t[1] - 1e-5, t[1], t[n // 2] + 0.3 * (t[n // 2 + 1] - t[n // 2]), t[n // 3],
         t[n // 3] + 5e-5, t[n // 3] - 5e-5, t[-2] + 0.9 * (t[-1] - t[-2]), t[-1] - 1e-5, t[-1], t[-1] + 0.01]
    return np.array(q)


def _read_rows(d):
    """the single line each of nav.txt, err.txt, traj.txt holds after 10 calls of writeNavResult (trailing newline dropped)"""
    rows = []
    for name in ("nav.txt", "err.txt", "traj.txt"):
        with open(os.path.join(d, name)) as f:
            lines = f.read().split("\n")
        assert len(lines) == 2 and lines[1] == "", (name, lines)
        rows.append(lines[0])
    return np.array(rows, dtype=str)


ORIGIN = np.array([np.deg2rad(30.5278), np.deg2rad(114.3557), 21.3])


def nav_cases():
    """(cfg8, state23, sodo): a scale-factor state far from the origin, a plain state, a state pitched up 89.6 deg (the
    dcm(2,0) <= -0.999 branch of Rotation::matrix2euler)"""
    a = mech_case("earth_scale_jitter")
    s1 = a["s0"].copy()
    s1[1:4] += [820.5, -431.2, 12.7]
    b = mech_case("normal")
    s2 = b["s0"].copy()
    s3 = s2.copy()
    ang = np.deg2rad(89.6)
    s3[4:8] = [0.0, np.sin(ang / 2), 0.0, np.cos(ang / 2)]
    return [(a["cfg"], s1, 0.0123), (b["cfg"], s2, 0.0), (b["cfg"], s3, -0.004)]


# ---- drivers ------------------------------------------------------------------------------------------------------------
class _Flat:
    """common call layout; subclasses bind the symbols"""

    def mechanize(self, cfg8, imu, s0):
        raise NotImplementedError


class RefMisc(_Flat):
    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_ins_window_index.restype = C.c_size_t

    def mechanize(self, cfg8, imu, s0):
        imu, st = _f64(imu), _f64(s0).copy()
        traj = np.zeros((max(len(imu) - 1, 0), 23))
        self.lib.ref_ins_mechanize(_p(_f64(cfg8)), len(imu), _p(imu), _p(st), _p(traj) if len(traj) else None)
        return st, traj

    def window_index(self, imu, time):
        return int(self.lib.ref_ins_window_index(len(imu), _p(_f64(imu)), C.c_double(time)))

    def camera_pose(self, imu, states, pbc, time):
        out = np.zeros(12)
        ok = self.lib.ref_ins_camera_pose(len(imu), _p(_f64(imu)), _p(_f64(states)), _p(_f64(pbc)), C.c_double(time), _p(out))
        return out, int(ok)

    def imu_series(self, imu, start, end):
        out = np.zeros((len(imu) + 4, 8))
        n = self.lib.ref_imu_series(len(imu), _p(_f64(imu)), C.c_double(start), C.c_double(end), len(out), _p(out))
        return None if n < 0 else out[:n].copy()

    def nav_rows(self, cfg8, origin, state23, sodo):
        d = tempfile.mkdtemp(prefix="refnav_")
        assert self.lib.ref_write_nav_result(_p(_f64(cfg8)), _p(_f64(origin)), _p(_f64(state23)), C.c_double(sodo), d.encode(), 10) == 0
        return _read_rows(d)

    def redo(self, cfg8, updated, reserved, imu, states):
        imu, states = _f64(imu).copy(), _f64(states).copy()
        n = self.lib.ref_redo_ins(_p(_f64(cfg8)), _p(_f64(updated)), int(reserved), len(imu), _p(imu), _p(states))
        return imu[:n], states[:n]


class OrcMisc(_Flat):
    def __init__(self, lib):
        self.lib = lib
        self.lib.orc_ins_window_index.restype = C.c_int64

    def mechanize(self, cfg8, imu, s0):
        imu, st = _f64(imu), _f64(s0).copy()
        traj = np.zeros((max(len(imu) - 1, 0), 23))
        self.lib.orc_ins_mechanize(_p(_f64(cfg8)), len(imu), _p(imu), _p(st), _p(traj) if len(traj) else None)
        return st, traj

    def window_index(self, imu, time):
        return int(self.lib.orc_ins_window_index(len(imu), _p(_f64(imu)), C.c_double(time)))

    def camera_pose(self, imu, states, pbc, time):
        out = np.zeros(12)
        ok = self.lib.orc_ins_camera_pose(len(imu), _p(_f64(imu)), _p(_f64(states)), _p(_f64(pbc)), C.c_double(time), _p(out))
        return out, int(ok)

    def imu_series(self, imu, start, end):
        out = np.zeros((len(imu) + 4, 8))
        n = self.lib.orc_imu_series(len(imu), _p(_f64(imu)), C.c_double(start), C.c_double(end), len(out), _p(out))
        return None if n < 0 else out[:n].copy()

    def nav_rows(self, cfg8, origin, state23, sodo):
        nav, errrow, traj = np.zeros(11), np.zeros(14), np.zeros(8)
        n = self.lib.orc_nav_result_rows(_p(_f64(origin)), int(cfg8[7] != 0), _p(_f64(state23)), C.c_double(sodo), _p(nav), _p(errrow), _p(traj))
        fmt = lambda row: "".join("%-15.9f " % v for v in row)
        return np.array([fmt(nav), fmt(errrow[:n]), fmt(traj)], dtype=str)

    def redo(self, cfg8, updated, reserved, imu, states):
        imu, states = _f64(imu).copy(), _f64(states).copy()
        n = self.lib.orc_redo_ins(_p(_f64(cfg8)), _p(_f64(updated)), int(reserved), len(imu), _p(imu), _p(states))
        return imu[:n], states[:n]


class HostMisc(_Flat):
    """icg::MISC of the product's host layer; the *_batch methods take lists (one entry per stream) and issue ONE device call"""

    def __init__(self, lib):
        self.lib = lib
        self.lib.icgh_ins_window_index.restype = C.c_long
        self.err = C.create_string_buffer(512)

    def _ck(self, rc):
        assert rc == 0, (rc, self.err.value)

    def mechanize_batch(self, cfg8, imus, s0s):
        off = np.concatenate([[0], np.cumsum([len(i) for i in imus])]).astype(np.int32)
        imu = _f64(np.concatenate(imus)) if off[-1] else np.zeros((1, 8))
        st = _f64(np.stack(s0s)).copy()
        traj = np.zeros((max(int(off[-1]), 1), 23))
        self._ck(self.lib.icgh_ins_mechanize(len(imus), _p(off), _p(imu), _p(_f64(cfg8)), _p(st), _p(traj), self.err, 512))
        return st, [traj[off[s] + 1:off[s + 1]].copy() for s in range(len(imus))]

    def mechanize(self, cfg8, imu, s0):
        st, tr = self.mechanize_batch(cfg8, [imu], [s0])
        return st[0], tr[0]

    def window_index(self, imu, time):
        return int(self.lib.icgh_ins_window_index(len(imu), _p(_f64(imu)), C.c_double(time)))

    def camera_pose_batch(self, imus, states, pbc, times):
        off = np.concatenate([[0], np.cumsum([len(i) for i in imus])]).astype(np.int32)
        out, found = np.zeros((len(imus), 12)), np.zeros(len(imus), np.uint8)
        self._ck(self.lib.icgh_ins_camera_pose(len(imus), _p(off), _p(_f64(np.concatenate(imus))), _p(_f64(np.concatenate(states))),
                                               _p(_f64(pbc)), _p(_f64(times)), _p(out), _p(found), self.err, 512))
        return out, found

    def camera_pose(self, imu, states, pbc, time):
        o, f = self.camera_pose_batch([imu], [states], pbc, [time])
        return o[0], int(f[0])

    def imu_series(self, imu, start, end):
        out = np.zeros((len(imu) + 4, 8))
        n = self.lib.icgh_ins_imu_series(len(imu), _p(_f64(imu)), C.c_double(start), C.c_double(end), len(out), _p(out))
        return None if n < 0 else out[:n].copy()

    def nav_rows(self, cfg8, origin, state23, sodo):
        d = tempfile.mkdtemp(prefix="icgnav_")
        assert self.lib.icgh_ins_write_nav_result(_p(_f64(cfg8)), _p(_f64(origin)), _p(_f64(state23)), C.c_double(sodo), d.encode(), 10) == 0
        return _read_rows(d)

    def redo_batch(self, cfg8, updated, reserved, imus, states):
        off = np.concatenate([[0], np.cumsum([len(i) for i in imus])]).astype(np.int32)
        imu, st = _f64(np.concatenate(imus)).copy(), _f64(np.concatenate(states)).copy()
        nl = np.zeros(len(imus), np.int32)
        self._ck(self.lib.icgh_ins_redo(len(imus), _p(_f64(cfg8)), _p(_f64(np.stack(updated))), int(reserved), _p(off), _p(imu), _p(st),
                                        _p(nl), self.err, 512))
        return [(imu[off[s]:off[s] + nl[s]].copy(), st[off[s]:off[s] + nl[s]].copy()) for s in range(len(imus))]

    def redo(self, cfg8, updated, reserved, imu, states):
        return self.redo_batch(cfg8, [updated], reserved, [imu], [states])[0]


# ---- the scenario every implementation is run through -------------------------------------------------------------------------
def window_for(impl, name):
    """(imu, states) window: the mechanization trajectory of case `name` with the start state as entry 0"""
    c = mech_case(name)
    _, traj = impl.mechanize(c["cfg"], c["imu"], c["s0"])
    return c, np.concatenate([c["s0"][None, :], traj])


def redo_updates(c, states):
    """updated states (as the optimizer would hand them back) at times that hit every isNeedInterpolation outcome"""
    t = c["imu"][:, 0]
    n = len(t)
    outs = []
    for j, (k, frac) in enumerate([(n // 2, 0.4), (n // 2, 2e-5 / (t[n // 2 + 1] - t[n // 2])), (n // 2, 1 - 2e-5 / (t[n // 2 + 1] - t[n // 2])),
                                   (n // 4, 0.0), (5, 0.7)]):
        u = states[k].copy()
        u[0] = t[k] + frac * (t[k + 1] - t[k])
        rng = np.random.RandomState(50 + j)
        u[1:4] += rng.normal(0, 0.05, 3)
        u[8:11] += rng.normal(0, 0.02, 3)
        u[11:14] += rng.normal(0, 1e-5, 3)
        outs.append(u)
    return outs


def run_all(impl):
    """-> dict of arrays: everything the golden file pins"""
    out = {}
    pbc = pose_b_c()
    for name, *_ in MECH_CASES:
        c = mech_case(name)
        st, traj = impl.mechanize(c["cfg"], c["imu"], c["s0"])
        out[f"mech_{name}_final"] = st
        out[f"mech_{name}_traj"] = traj
    for name in ("normal_jitter", "earth"):
        c, states = window_for(impl, name)
        q = query_times(c["imu"])
        out[f"idx_{name}"] = np.array([impl.window_index(c["imu"], t) for t in q], np.int64)
        poses, found = [], []
        for t in q:
            p, f = impl.camera_pose(c["imu"], states, pbc, t)
            poses.append(p)
            found.append(f)
        out[f"pose_{name}"] = np.stack(poses)
        out[f"found_{name}"] = np.array(found, np.int32)
        # IMU series between pairs of query times that are both inside the window
        inside = [t for t in q if impl.window_index(c["imu"], t) > 0]
        k = 0
        for a in inside[:6]:
            for b in inside[-4:]:
                if b > a + 0.02:
                    s = impl.imu_series(c["imu"], a, b)
                    assert s is not None
                    out[f"series_{name}_{k}"] = s
                    k += 1
        out[f"series_{name}_count"] = np.array(k)
        for j, u in enumerate(redo_updates(c, states)):
            for reserved in (10, 10 ** 6):
                im, stt = impl.redo(c["cfg"], u, reserved, c["imu"], states)
                out[f"redo_{name}_{j}_{reserved}_imu"] = im
                out[f"redo_{name}_{j}_{reserved}_states"] = stt
    for k, (cfg, st, sodo) in enumerate(nav_cases()):
        out[f"navrows_{k}"] = impl.nav_rows(cfg, ORIGIN, st, sodo)
    return out


def compare(got, exp, tol_state=1e-12, tol_pose=1e-12, exact_series=True):
    """index/series/count decisions exact; states and poses relative to the magnitude of the golden array"""
    assert set(got.keys()) == set(exp.keys()), set(got.keys()) ^ set(exp.keys())
    for k in sorted(exp.keys()):
        g, e = np.asarray(got[k]), np.asarray(exp[k])
        assert g.shape == e.shape, (k, g.shape, e.shape)
        if k.startswith("navrows_"):  # the text of the three result lines (9 decimals)
            assert list(g) == [str(x) for x in e], (k, list(g), list(e))
        elif k.startswith(("idx_", "found_")) or k.endswith("_count"):
            assert np.array_equal(g, e), k
        elif k.startswith("series_") or k.endswith("_imu"):
            if exact_series:
                assert np.array_equal(g, e), k
            else:
                assert np.abs(g - e).max() <= 1e-15 * max(1.0, np.abs(e).max()), k
        else:
            tol = tol_pose if k.startswith("pose_") else tol_state
            if e.size:
                # per column scale: times ~2e3, positions ~1e2, biases ~1e-4 must each hold to the tolerance
                scale = np.maximum(np.abs(e).reshape(-1, e.shape[-1]).max(axis=0), 1e-3)
                assert (np.abs(g - e).reshape(-1, e.shape[-1]) / scale).max() < tol, (k, (np.abs(g - e).reshape(-1, e.shape[-1]) / scale).max())
