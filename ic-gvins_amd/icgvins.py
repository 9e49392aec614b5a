"""ctypes binding of libicgvins_hip.so (the C ABI in include/icgvins_hip.h).

This is plumbing for tests/bench: the product is the shared library.  There is NO CPU fallback here — if the
HIP library is missing or no GPU is visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicgvins_hip.so")

EXPORTS = [
    "icg_ctx_create", "icg_ctx_destroy", "icg_last_error", "icg_ctx_sync", "icg_ctx_set_wait_mode", "icg_ctx_stream", "icg_set_camera",
    "icg_version", "icg_pyramid_levels", "icg_prof_enable", "icg_prof_get", "icg_prof_names", "icg_dev_alloc",
    "icg_dev_free", "icg_dev_upload", "icg_dev_download", "icg_frames_preprocess", "icg_frame_download",
    "icg_lk_track", "icg_lk_track_fb", "icg_undistort_points", "icg_distort_points", "icg_predict_mappoints",
    "icg_predict_rotation", "icg_fm_ransac", "icg_fm_ransac_device", "icg_detect", "icg_triangulate", "icg_reproj_eval_batch",
    "icg_reproj_set_factors", "icg_reproj_stage_factors", "icg_reproj_commit_factors", "icg_reproj_eval_resident", "icg_reproj_accumulate_normal", "icg_preint_batch",
    "icg_ins_mechanize_batch", "icg_ins_camera_pose_batch", "icg_reproj_schur", "icg_reproj_backsub", "icg_reproj_cost", "icg_reproj_landmark_diag",
    "icg_reproj_error_batch", "icg_reproj_set_windows", "icg_reproj_eval_windows", "icg_reproj_schur_windows",
    "icg_reproj_schur_windows_view", "icg_reproj_reserve_windows", "icg_reproj_eval_resident_view", "icg_reproj_backsub_windows", "icg_reproj_schur_windows_resident", "icg_reproj_set_host_part_windows", "icg_reproj_solve_backsub_windows", "icg_reproj_cost_windows", "icg_reproj_fetch_residuals", "icg_reproj_chi2_cull",
]


class IcgError(RuntimeError):
    pass


class CtxConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("width", C.c_int), ("height", C.c_int), ("n_slots", C.c_int),
                ("max_batch", C.c_int), ("max_points", C.c_int), ("max_factors", C.c_int)]


class Camera(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("fx", "fy", "cx", "cy", "skew", "k1", "k2", "p1", "p2", "k3")]


class DetectGrid(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("block_cols", "block_rows", "block_w", "block_h", "min_dist", "max_per_block")]


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise IcgError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                       "There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.icg_last_error.restype = C.c_char_p
    lib.icg_last_error.argtypes = [C.c_void_p]
    lib.icg_version.restype = C.c_char_p
    lib.icg_ctx_stream.restype = C.c_void_p
    lib.icg_ctx_stream.argtypes = [C.c_void_p]
    lib.icg_ctx_destroy.argtypes = [C.c_void_p]
    lib.icg_ctx_destroy.restype = None
    return lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One icg_ctx: a HIP stream + resident frame slots on one MI355X."""

    def __init__(self, width, height, n_slots=4, max_batch=2, max_points=4096, max_factors=0, device=0, lib=None):
        self.lib = lib or load_library()
        cfg = CtxConfig(device, width, height, n_slots, max_batch, max_points, max_factors)
        h = C.c_void_p()
        rc = self.lib.icg_ctx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise IcgError(f"icg_ctx_create failed rc={rc}: {self.lib.icg_last_error(None).decode()}")
        self.h = h
        self.width, self.height = width, height
        self.n_slots, self.max_batch = n_slots, max_batch

    def close(self):
        if getattr(self, "h", None):
            self.lib.icg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise IcgError(f"{what} failed rc={rc}: {self.lib.icg_last_error(self.h).decode()}")

    # ---- misc
    def sync(self):
        self._ck(self.lib.icg_ctx_sync(self.h), "icg_ctx_sync")

    def levels(self):
        return self.lib.icg_pyramid_levels(self.h)

    def set_camera(self, cam10):
        cam = Camera(*[float(v) for v in cam10])
        self._ck(self.lib.icg_set_camera(self.h, C.byref(cam)), "icg_set_camera")

    def prof_enable(self, on=True):
        self._ck(self.lib.icg_prof_enable(self.h, 1 if on else 0), "icg_prof_enable")

    def prof(self):
        buf = C.create_string_buffer(4096)
        self._ck(self.lib.icg_prof_names(self.h, buf, 4096), "icg_prof_names")
        out = {}
        for name in buf.value.decode().split("\n"):
            if not name:
                continue
            n, ms = C.c_int(), C.c_double()
            self.lib.icg_prof_get(self.h, name.encode(), C.byref(n), C.byref(ms))
            out[name] = (n.value, ms.value)
        return out

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._ck(self.lib.icg_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(p)), "icg_dev_alloc")
        return p.value

    def dev_free(self, ptr):
        self._ck(self.lib.icg_dev_free(self.h, C.c_void_p(ptr)), "icg_dev_free")

    def dev_upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(self.lib.icg_dev_upload(self.h, C.c_void_p(ptr), _p(arr), C.c_size_t(arr.nbytes)), "icg_dev_upload")

    # ---- F1
    def preprocess(self, slots, images, want_hist=False):
        """images: list of HxW (or HxWx3) uint8 arrays (host)."""
        n = len(slots)
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        ch = 3 if imgs[0].ndim == 3 else 1
        stride = imgs[0].strides[0]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        hist = np.zeros(n, np.float64) if want_hist else None
        self._ck(self.lib.icg_frames_preprocess(self.h, n, _p(_i32(slots)), ptrs, stride, ch, 0, _p(hist)),
                 "icg_frames_preprocess")
        return hist

    def preprocess_device(self, slots, dev_ptrs, stride, channels=1):
        n = len(slots)
        ptrs = (C.c_void_p * n)(*dev_ptrs)
        self._ck(self.lib.icg_frames_preprocess(self.h, n, _p(_i32(slots)), ptrs, stride, channels, 1, None),
                 "icg_frames_preprocess")

    def download(self, slot, level=0):
        w, h = self.width, self.height
        for _ in range(level):
            w, h = (w + 1) // 2, (h + 1) // 2
        out = np.zeros((h, w), np.uint8)
        self._ck(self.lib.icg_frame_download(self.h, slot, level, _p(out), w), "icg_frame_download")
        return out

    # ---- F2/F3
    def lk_track(self, prev_slot, next_slot, prev_pts, guess, want_err=True):
        prev_pts = _f32(prev_pts).reshape(-1, 2)
        n = prev_pts.shape[0]
        nxt = _f32(guess).reshape(-1, 2).copy()
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32) if want_err else None
        ps = _i32(np.broadcast_to(prev_slot, (n,)))
        ns = _i32(np.broadcast_to(next_slot, (n,)))
        self._ck(self.lib.icg_lk_track(self.h, n, _p(ps), _p(ns), _p(prev_pts), _p(nxt), _p(st), _p(err)), "icg_lk_track")
        return nxt, st, err

    def lk_track_fb(self, prev_slot, next_slot, prev_pts, guess, want_undist=False, want_keep=False):
        prev_pts = _f32(prev_pts).reshape(-1, 2)
        guess = _f32(guess).reshape(-1, 2)
        n = prev_pts.shape[0]
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        und = np.zeros((n, 2), np.float32) if want_undist else None
        keep = np.zeros(max(n, 1), np.int32) if want_keep else None
        nkeep = np.zeros(1, np.int32) if want_keep else None
        ps = _i32(np.broadcast_to(prev_slot, (n,)))
        ns = _i32(np.broadcast_to(next_slot, (n,)))
        self._ck(self.lib.icg_lk_track_fb(self.h, n, _p(ps), _p(ns), _p(prev_pts), _p(guess), _p(out), _p(st), _p(und),
                                           _p(keep), _p(nkeep)), "icg_lk_track_fb")
        res = [out, st]
        if want_undist:
            res.append(und)
        if want_keep:
            res.append(keep[:int(nkeep[0])])
        return tuple(res)

    # ---- F4/F5
    def undistort(self, pts):
        p = _f32(pts).reshape(-1, 2).copy()
        self._ck(self.lib.icg_undistort_points(self.h, p.shape[0], _p(p)), "icg_undistort_points")
        return p

    def distort(self, pts):
        p = _f32(pts).reshape(-1, 2).copy()
        self._ck(self.lib.icg_distort_points(self.h, p.shape[0], _p(p)), "icg_distort_points")
        return p

    def predict_mappoints(self, pw, pose_idx, poses12):
        pw = _f64(pw).reshape(-1, 3)
        poses12 = _f64(poses12).reshape(-1, 12)
        n = pw.shape[0]
        out = np.zeros((n, 2), np.float32)
        self._ck(self.lib.icg_predict_mappoints(self.h, n, _p(pw), _p(_i32(np.broadcast_to(pose_idx, (n,)))),
                                                 poses12.shape[0], _p(poses12), _p(out)), "icg_predict_mappoints")
        return out

    def predict_rotation(self, pts, rot_idx, rots9):
        pts = _f32(pts).reshape(-1, 2)
        rots9 = _f64(rots9).reshape(-1, 9)
        n = pts.shape[0]
        out = np.zeros((n, 2), np.float32)
        self._ck(self.lib.icg_predict_rotation(self.h, n, _p(pts), _p(_i32(np.broadcast_to(rot_idx, (n,)))),
                                                rots9.shape[0], _p(rots9), _p(out)), "icg_predict_rotation")
        return out

    # ---- F6
    def fm_ransac(self, offsets, pts1, pts2, thresh=1.5, conf=0.99):
        offsets = _i32(offsets)
        pts1 = _f32(pts1).reshape(-1, 2)
        pts2 = _f32(pts2).reshape(-1, 2)
        mask = np.ones(pts1.shape[0], np.uint8)
        self._ck(self.lib.icg_fm_ransac(self.h, len(offsets) - 1, _p(offsets), _p(pts1), _p(pts2), C.c_double(thresh),
                                         C.c_double(conf), _p(mask)), "icg_fm_ransac")
        return mask

    def fm_ransac_device(self, offsets, pts1, pts2, thresh=1.5, conf=0.99):
        """icg_fm_ransac_device: the whole RANSAC run of every set in one launch (the kernel the device-resident tracker uses)"""
        offsets = _i32(offsets)
        pts1 = _f32(pts1).reshape(-1, 2)
        pts2 = _f32(pts2).reshape(-1, 2)
        mask = np.ones(pts1.shape[0], np.uint8)
        self._ck(self.lib.icg_fm_ransac_device(self.h, len(offsets) - 1, _p(offsets), _p(pts1), _p(pts2), C.c_double(thresh),
                                                C.c_double(conf), _p(mask)), "icg_fm_ransac_device")
        return mask

    # ---- F7
    def detect(self, slots, grid, mask_off, mask_pts, quota, max_per_job):
        slots = _i32(slots)
        n = len(slots)
        g = DetectGrid(*[int(v) for v in grid])
        mask_off = _i32(mask_off)
        mask_pts = _f32(mask_pts).reshape(-1, 2)
        quota = _i32(quota)
        out = np.zeros((n, max_per_job, 2), np.float32)
        cnt = np.zeros(n, np.int32)
        blk = np.zeros((n, max_per_job), np.int32)
        self._ck(self.lib.icg_detect(self.h, n, _p(slots), C.byref(g), _p(mask_off), _p(mask_pts), _p(quota), max_per_job,
                                      _p(out), _p(cnt), _p(blk)), "icg_detect")
        return out, cnt, blk

    # ---- F8
    def triangulate(self, T0_idx, T1_idx, Tcw12, pc0, pc1):
        Tcw12 = _f64(Tcw12).reshape(-1, 12)
        pc0 = _f64(pc0).reshape(-1, 3)
        pc1 = _f64(pc1).reshape(-1, 3)
        n = pc0.shape[0]
        pw = np.zeros((n, 3), np.float64)
        self._ck(self.lib.icg_triangulate(self.h, n, _p(_i32(np.broadcast_to(T0_idx, (n,)))),
                                           _p(_i32(np.broadcast_to(T1_idx, (n,)))), Tcw12.shape[0], _p(Tcw12), _p(pc0),
                                           _p(pc1), _p(pw)), "icg_triangulate")
        return pw

    # ---- R1/R2
    def reproj_eval(self, obs_soa, idx_i, idx_j, idx_lm, poses, ext, invdepth, td, want_jac=True, huber=0.0):
        obs_soa = _f64(obs_soa)
        n = obs_soa.shape[1]
        poses = _f64(poses).reshape(-1, 7)
        invdepth = _f64(invdepth).reshape(-1)
        r = np.zeros((n, 2))
        J = np.zeros((n, 46)) if want_jac else None
        self._ck(self.lib.icg_reproj_eval_batch(self.h, n, _p(obs_soa), _p(_i32(idx_i)), _p(_i32(idx_j)), _p(_i32(idx_lm)),
                                                 poses.shape[0], _p(poses), _p(_f64(ext)), invdepth.shape[0], _p(invdepth),
                                                 C.c_double(td), 1 if want_jac else 0, C.c_double(huber), _p(r), _p(J)),
                 "icg_reproj_eval_batch")
        return r, J

    def reproj_set_factors(self, obs_soa, idx_i, idx_j, idx_lm):
        obs_soa = _f64(obs_soa)
        n = obs_soa.shape[1]
        self._ck(self.lib.icg_reproj_set_factors(self.h, n, _p(obs_soa), _p(_i32(idx_i)), _p(_i32(idx_j)), _p(_i32(idx_lm))),
                 "icg_reproj_set_factors")
        self._nfac = n

    def reproj_set_factors_staged(self, obs_soa, idx_i, idx_j, idx_lm):
        """the same upload through icg_reproj_stage_factors / icg_reproj_commit_factors: the factor set is written into the context's pinned
        staging block in place (what the host layer does from its pool threads)"""
        obs_soa = _f64(obs_soa)
        n = obs_soa.shape[1]
        po, pi = C.c_void_p(), C.c_void_p()
        self._ck(self.lib.icg_reproj_stage_factors(self.h, n, C.byref(po), C.byref(pi)), "icg_reproj_stage_factors")
        if n:
            np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_double)), shape=(15, n))[:] = obs_soa
            idx = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_int32)), shape=(3, n))
            idx[0], idx[1], idx[2] = _i32(idx_i), _i32(idx_j), _i32(idx_lm)
        self._ck(self.lib.icg_reproj_commit_factors(self.h), "icg_reproj_commit_factors")
        self._nfac = n

    def reproj_eval_resident(self, poses, ext, invdepth, td, want_jac=True, huber=0.0, fetch=True):
        poses = _f64(poses).reshape(-1, 7)
        invdepth = _f64(invdepth).reshape(-1)
        n = self._nfac
        r = np.zeros((n, 2)) if fetch else None
        J = np.zeros((n, 46)) if (fetch and want_jac) else None
        self._ck(self.lib.icg_reproj_eval_resident(self.h, poses.shape[0], _p(poses), _p(_f64(ext)), invdepth.shape[0],
                                                    _p(invdepth), C.c_double(td), 1 if want_jac else 0, C.c_double(huber),
                                                    _p(r), _p(J)), "icg_reproj_eval_resident")
        return r, J

    def reproj_accumulate_normal(self, local_size, col_pose, col_ext, col_lm, col_td):
        H = np.zeros((local_size, local_size))
        b = np.zeros(local_size)
        self._ck(self.lib.icg_reproj_accumulate_normal(self.h, local_size, _p(_i32(col_pose)), int(col_ext), _p(_i32(col_lm)),
                                                        int(col_td), _p(H), _p(b)), "icg_reproj_accumulate_normal")
        return H, b

    # ---- f3
    def reproj_error_batch(self, pose_idx, lm_idx, poses12, pw, pix, max_error, min_depth=1.0, max_depth=200.0):
        poses12 = _f64(poses12).reshape(-1, 12)
        pw = _f64(pw).reshape(-1, 3)
        pix = _f32(pix).reshape(-1, 2)
        n = pix.shape[0]
        err, good = np.zeros(n), np.zeros(n, np.uint8)
        self._ck(self.lib.icg_reproj_error_batch(self.h, n, _p(_i32(pose_idx)), _p(_i32(lm_idx)), poses12.shape[0], _p(poses12), pw.shape[0],
                                                  _p(pw), _p(pix), C.c_double(max_error), C.c_double(min_depth), C.c_double(max_depth), _p(err),
                                                  _p(good)), "icg_reproj_error_batch")
        return err, good

    # ---- f1
    def reproj_schur(self, P, col_pose, col_ext, col_td, active=None, reassemble=True, damp=0.0, min_diag=1e-6, max_diag=1e32):
        S, s, dg, cost = np.zeros((P, P)), np.zeros(P), np.zeros(P), np.zeros(1)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        self._ck(self.lib.icg_reproj_schur(self.h, int(P), _p(_i32(col_pose)), int(col_ext), int(col_td), _p(act), 1 if reassemble else 0,
                                            C.c_double(damp), C.c_double(min_diag), C.c_double(max_diag), _p(S), _p(s), _p(dg), _p(cost)),
                 "icg_reproj_schur")
        return S, s, dg, float(cost[0])

    def reproj_backsub(self, P, delta_c, n_lm):
        out, terms = np.zeros(n_lm), np.zeros(2)
        self._ck(self.lib.icg_reproj_backsub(self.h, int(P), _p(_f64(delta_c)), _p(out), _p(terms)), "icg_reproj_backsub")
        return out, terms

    def reproj_cost(self, active=None):
        cost = np.zeros(1)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        self._ck(self.lib.icg_reproj_cost(self.h, _p(act), _p(cost)), "icg_reproj_cost")
        return float(cost[0])

    # ---- f1, many windows per launch
    def reproj_set_windows(self, fac_off, lm_off):
        fac_off, lm_off = _i32(fac_off), _i32(lm_off)
        self._nwin = len(fac_off) - 1
        self._ck(self.lib.icg_reproj_set_windows(self.h, self._nwin, _p(fac_off), _p(lm_off)), "icg_reproj_set_windows")

    def reproj_eval_windows(self, poses, ext, invdepth, td, want_jac=True, huber=0.0):
        poses, ext, invdepth, td = _f64(poses).reshape(-1, 7), _f64(ext).reshape(-1, 7), _f64(invdepth).reshape(-1), _f64(td).reshape(-1)
        self._ck(self.lib.icg_reproj_eval_windows(self.h, poses.shape[0], _p(poses), _p(ext), invdepth.shape[0], _p(invdepth), _p(td),
                                                   1 if want_jac else 0, C.c_double(huber)), "icg_reproj_eval_windows")

    def reproj_schur_windows(self, P, col_pose, col_ext, col_td, active=None, reassemble=None, damp=None, min_diag=1e-6, max_diag=1e32):
        W = self._nwin
        S, s, dg, cost = np.zeros((W, P, P)), np.zeros((W, P)), np.zeros((W, P)), np.zeros(W)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        re = np.ones(W, np.uint8) if reassemble is None else np.ascontiguousarray(reassemble, np.uint8)
        dm = np.zeros(W) if damp is None else _f64(damp)
        self._ck(self.lib.icg_reproj_schur_windows(self.h, int(P), _p(_i32(col_pose)), _p(_i32(col_ext)), _p(_i32(col_td)), _p(act), _p(re), _p(dm),
                                                    C.c_double(min_diag), C.c_double(max_diag), _p(S), _p(s), _p(dg), _p(cost)),
                 "icg_reproj_schur_windows")
        return S, s, dg, cost

    def reproj_reserve_windows(self, P):
        self._ck(self.lib.icg_reproj_reserve_windows(self.h, int(P)), "icg_reproj_reserve_windows")

    def reproj_schur_windows_view(self, P, col_pose, col_ext, col_td, active=None, reassemble=None, damp=None, min_diag=1e-6, max_diag=1e32):
        """icg_reproj_schur_windows_view: the reduced systems are read where the reduction kernel left them (a copy is returned; only the
        16 x 16 tiles on and below the diagonal are defined)"""
        W = self._nwin
        s, dg, cost = np.zeros((W, P)), np.zeros((W, P)), np.zeros(W)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        re = np.ones(W, np.uint8) if reassemble is None else np.ascontiguousarray(reassemble, np.uint8)
        dm = np.zeros(W) if damp is None else _f64(damp)
        view = C.POINTER(C.c_double)()
        self._ck(self.lib.icg_reproj_schur_windows_view(self.h, int(P), _p(_i32(col_pose)), _p(_i32(col_ext)), _p(_i32(col_td)), _p(act), _p(re), _p(dm),
                                                         C.c_double(min_diag), C.c_double(max_diag), C.byref(view), _p(s), _p(dg), _p(cost)),
                 "icg_reproj_schur_windows_view")
        S = np.ctypeslib.as_array(view, shape=(W, P, P)).copy()
        return S, s, dg, cost

    def reproj_backsub_windows(self, P, delta_c, n_lm):
        out, terms = np.zeros(n_lm), np.zeros((self._nwin, 2))
        self._ck(self.lib.icg_reproj_backsub_windows(self.h, int(P), _p(_f64(delta_c)), _p(out), _p(terms)), "icg_reproj_backsub_windows")
        return out, terms

    def reproj_cost_windows(self, active=None):
        cost = np.zeros(self._nwin)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        self._ck(self.lib.icg_reproj_cost_windows(self.h, _p(act), _p(cost)), "icg_reproj_cost_windows")
        return cost

    # ---- P1
    def preint_batch(self, variant, offsets, imu, state0, params):
        offsets = _i32(offsets)
        n = len(offsets) - 1
        imu = _f64(imu).reshape(-1, 8)
        state0 = _f64(state0).reshape(n, 16)
        cur = np.zeros((n, 16))
        delta = np.zeros((n, 16))
        jac = np.zeros((n, 15, 15))
        cov = np.zeros((n, 15, 15))
        dt = np.zeros(n)
        pn = np.zeros((imu.shape[0], 4))
        self._ck(self.lib.icg_preint_batch(self.h, int(variant), n, _p(offsets), _p(imu), _p(state0), _p(_f64(params)),
                                            _p(cur), _p(delta), _p(jac), _p(cov), _p(dt), _p(pn)), "icg_preint_batch")
        return cur, delta, jac, cov, dt, pn

    # ---- f4
    def ins_mechanize_batch(self, offsets, imu, cfg8, states23, want_traj=True):
        offsets = _i32(offsets)
        n = len(offsets) - 1
        imu = _f64(imu).reshape(-1, 8)
        st = _f64(states23).reshape(n, 23).copy()
        traj = np.zeros((imu.shape[0], 23)) if want_traj else None
        self._ck(self.lib.icg_ins_mechanize_batch(self.h, n, _p(offsets), _p(imu), _p(_f64(cfg8)), _p(st), _p(traj)),
                 "icg_ins_mechanize_batch")
        return st, traj

    def ins_camera_pose_batch(self, brackets16, interp, pose_b_c12, times):
        b = _f64(brackets16).reshape(-1, 16)
        n = b.shape[0]
        out = np.zeros((n, 12))
        self._ck(self.lib.icg_ins_camera_pose_batch(self.h, n, _p(b), _p(_i32(interp)), _p(_f64(pose_b_c12)), _p(_f64(times)), _p(out)),
                 "icg_ins_camera_pose_batch")
        return out
