"""ctypes wrappers around oracle/liboracle.so (TEST INFRASTRUCTURE ONLY: tests/, smoke(), bench cpu_baseline)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_histogram_mean.restype = C.c_double

    # ---- factors
    def reproj_eval(self, obs_soa, idx_i, idx_j, idx_lm, poses, ext, invdepth, td, want_jac=True, huber=0.0):
        obs_soa = _f64(obs_soa)
        n = obs_soa.shape[1]
        r = np.zeros((n, 2))
        J = np.zeros((n, 46))
        self.lib.orc_reproj_eval_batch(n, _p(obs_soa), _p(_i32(idx_i)), _p(_i32(idx_j)), _p(_i32(idx_lm)), _p(_f64(poses)),
                                       _p(_f64(ext)), _p(_f64(invdepth)), C.c_double(td), 1 if want_jac else 0, _p(r), _p(J))
        if huber > 0:
            self.lib.orc_huber_correct_2x46(n, C.c_double(huber), _p(r), _p(J) if want_jac else None)
        return r, (J if want_jac else None)

    def reproj_eval_one(self, obs15, pose_i, pose_j, ext, invdepth, td, want_jac=True):
        r = np.zeros(2)
        J = np.zeros(46)
        self.lib.orc_reproj_eval_one(_p(_f64(obs15)), _p(_f64(pose_i)), _p(_f64(pose_j)), _p(_f64(ext)), C.c_double(invdepth),
                                     C.c_double(td), 1 if want_jac else 0, _p(r), _p(J))
        return r, J

    # ---- image
    def bgr2gray(self, bgr):
        bgr = _u8(bgr)
        h, w, _ = bgr.shape
        out = np.zeros((h, w), np.uint8)
        self.lib.orc_bgr2gray(_p(bgr), w, h, w * 3, _p(out), w)
        return out

    def histogram_mean(self, img):
        img = _u8(img)
        h, w = img.shape
        return self.lib.orc_histogram_mean(_p(img), w, h, w)

    def clahe(self, img, clip=3.0, tiles=21, want_lut=False):
        img = _u8(img)
        h, w = img.shape
        out = np.zeros_like(img)
        lut = np.zeros((tiles * tiles, 256), np.uint8)
        self.lib.orc_clahe(_p(img), w, h, w, C.c_double(clip), tiles, _p(out), w, _p(lut))
        return (out, lut) if want_lut else out

    def pyrdown(self, img):
        img = _u8(img)
        h, w = img.shape
        out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.orc_pyrdown(_p(img), w, h, w, _p(out), out.shape[1])
        return out

    def scharr(self, img):
        img = _u8(img)
        h, w = img.shape
        out = np.zeros((h, w, 2), np.int16)
        self.lib.orc_scharr(_p(img), w, h, w, _p(out))
        return out

    # ---- LK
    def lk_track(self, prev, nxt, prev_pts, guess):
        prev, nxt = _u8(prev), _u8(nxt)
        h, w = prev.shape
        prev_pts = _f32(prev_pts).reshape(-1, 2)
        n = prev_pts.shape[0]
        out = _f32(guess).reshape(-1, 2).copy()
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self.lib.orc_lk_track(_p(prev), _p(nxt), w, h, w, n, _p(prev_pts), _p(out), _p(st), _p(err))
        return out, st, err

    def lk_track_fb(self, prev, nxt, prev_pts, guess):
        prev, nxt = _u8(prev), _u8(nxt)
        h, w = prev.shape
        prev_pts = _f32(prev_pts).reshape(-1, 2)
        guess = _f32(guess).reshape(-1, 2)
        n = prev_pts.shape[0]
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self.lib.orc_lk_track_fb(_p(prev), _p(nxt), w, h, w, n, _p(prev_pts), _p(guess), _p(out), _p(st))
        return out, st

    # ---- camera
    def undistort(self, cam, pts):
        p = _f32(pts).reshape(-1, 2).copy()
        self.lib.orc_undistort_points(_p(_f64(cam)), p.shape[0], _p(p))
        return p

    def distort(self, cam, pts):
        p = _f32(pts).reshape(-1, 2).copy()
        self.lib.orc_distort_points(_p(_f64(cam)), p.shape[0], _p(p))
        return p

    def world2pixel(self, cam, pose12, pw):
        pw = _f64(pw).reshape(-1, 3)
        out = np.zeros((pw.shape[0], 2), np.float32)
        self.lib.orc_world2pixel(_p(_f64(cam)), _p(_f64(pose12)), pw.shape[0], _p(pw), _p(out))
        return out

    def pixel2cam(self, cam, pts):
        pts = _f32(pts).reshape(-1, 2)
        out = np.zeros((pts.shape[0], 3))
        self.lib.orc_pixel2cam(_p(_f64(cam)), pts.shape[0], _p(pts), _p(out))
        return out

    def predict_rotation(self, cam, R_cur, R_pre, pts):
        pts = _f32(pts).reshape(-1, 2)
        out = np.zeros_like(pts)
        self.lib.orc_predict_rotation(_p(_f64(cam)), _p(_f64(R_cur)), _p(_f64(R_pre)), pts.shape[0], _p(pts), _p(out))
        return out


    # ---- detection
    def draw_circle(self, mask, cx, cy, r, value=0):
        h, w = mask.shape
        self.lib.orc_draw_filled_circle(_p(mask), w, h, mask.strides[0], int(cx), int(cy), int(r), int(value))

    def circle_halfwidths(self, r):
        hw = np.zeros(r + 1, np.int32)
        self.lib.orc_circle_halfwidths(int(r), _p(hw))
        return hw

    def min_eigen_map(self, img, roi):
        img = _u8(img)
        h, w = img.shape
        rx, ry, rw, rh = roi
        eig = np.zeros((rh, rw), np.float32)
        self.lib.orc_min_eigen_map(_p(img), w, h, w, rx, ry, rw, rh, _p(eig))
        return eig

    def good_features(self, img, mask, roi, max_corners, quality, min_dist):
        img = _u8(img)
        h, w = img.shape
        rx, ry, rw, rh = roi
        out = np.zeros((max_corners, 2), np.float32)
        m = None if mask is None else _u8(mask)
        n = self.lib.orc_good_features(_p(img), w, h, w, _p(m), w, rx, ry, rw, rh, max_corners, C.c_double(quality),
                                       C.c_double(min_dist), _p(out))
        return out[:n]

    def corner_subpix(self, img, roi, corners):
        img = _u8(img)
        h, w = img.shape
        rx, ry, rw, rh = roi
        c = _f32(corners).reshape(-1, 2).copy()
        self.lib.orc_corner_subpix(_p(img), w, h, w, rx, ry, rw, rh, c.shape[0], _p(c))
        return c

    def subpix_mask(self):
        m = np.zeros(121, np.float32)
        self.lib.orc_subpix_mask(_p(m))
        return m

    def detect(self, img, grid6, mask_pts, quota, max_out):
        img = _u8(img)
        h, w = img.shape
        mask_pts = _f32(mask_pts).reshape(-1, 2)
        out = np.zeros((max_out, 2), np.float32)
        blk = np.zeros(max_out, np.int32)
        n = self.lib.orc_detect(_p(img), w, h, w, _p(_i32(grid6)), mask_pts.shape[0], _p(mask_pts), _p(_i32(quota)), max_out,
                                _p(out), _p(blk))
        return out[:n], blk[:n]

    # ---- RANSAC / triangulation
    def seven_point(self, m1, m2):
        F = np.zeros((3, 9))
        n = self.lib.orc_seven_point(_p(_f64(m1)), _p(_f64(m2)), _p(F))
        return F[:n].reshape(n, 3, 3)

    def fm_score(self, F, pts1, pts2, thresh):
        pts1, pts2 = _f32(pts1).reshape(-1, 2), _f32(pts2).reshape(-1, 2)
        mask = np.zeros(pts1.shape[0], np.uint8)
        n = self.lib.orc_fm_score(_p(_f64(F)), pts1.shape[0], _p(pts1), _p(pts2), C.c_double(thresh), _p(mask))
        return n, mask

    def ransac_subsets(self, n_points, n_hyp, pts1=None, pts2=None):
        """hypothesis index stream of getSubset; with pts1/pts2 the FMEstimatorCallback::checkSubset rejection is applied"""
        idx = np.zeros((n_hyp, 7), np.int32)
        a = None if pts1 is None else _f32(pts1)
        b = None if pts2 is None else _f32(pts2)
        n = self.lib.orc_ransac_subsets(n_points, _p(a), _p(b), n_hyp, _p(idx))
        return idx[:n]

    def solve_cubic(self, coeffs4):
        r = np.zeros(3)
        n = self.lib.orc_solve_cubic(_p(_f64(coeffs4)), _p(r))
        return r[:n]

    def have_collinear_points(self, pts):
        pts = _f32(pts)
        return bool(self.lib.orc_have_collinear_points(_p(pts), pts.shape[0]))

    def fm_ransac(self, pts1, pts2, thresh=1.5, conf=0.99):
        pts1, pts2 = _f32(pts1).reshape(-1, 2), _f32(pts2).reshape(-1, 2)
        n = pts1.shape[0]
        mask = np.zeros(n, np.uint8)
        F = np.zeros(9)
        it = C.c_int(0)
        ok = self.lib.orc_find_fundamental_ransac(n, _p(pts1), _p(pts2), C.c_double(thresh), C.c_double(conf), _p(mask), _p(F),
                                                  C.byref(it))
        return ok, mask, F.reshape(3, 3), it.value

    def triangulate(self, T0, T1, pc0, pc1):
        pw = np.zeros(3)
        self.lib.orc_triangulate_point(_p(_f64(T0)), _p(_f64(T1)), _p(_f64(pc0)), _p(_f64(pc1)), _p(pw))
        return pw


    # ---- preintegration
    def preint_integrate(self, variant, imu, state0, params):
        imu = _f64(imu).reshape(-1, 8)
        n = imu.shape[0]
        cur, delta = np.zeros(16), np.zeros(16)
        jac, cov = np.zeros((15, 15)), np.zeros((15, 15))
        dt = C.c_double(0)
        pn = np.zeros((max(n - 1, 1), 4))
        self.lib.orc_preint_integrate(int(variant), n, _p(imu), _p(_f64(state0)), _p(_f64(params)), _p(cur), _p(delta), _p(jac),
                                      _p(cov), C.byref(dt), _p(pn))
        return dict(cur=cur, delta=delta, jac=jac, cov=cov, dt=dt.value, pn=pn[:n - 1])

    def preint_evaluate(self, variant, pre, gravity3, iewn3, pose0, mix0, pose1, mix1, want_jac=True):
        r = np.zeros(15)
        J = np.zeros(15 * 32) if want_jac else None
        pn = _f64(pre["pn"])
        self.lib.orc_preint_evaluate(int(variant), _p(_f64(pre["delta"])), _p(_f64(pre["jac"])), _p(_f64(pre["cov"])),
                                     C.c_double(pre["dt"]), _p(_f64(gravity3)), _p(_f64(iewn3)), pn.shape[0], _p(pn), None,
                                     _p(_f64(pose0)), _p(_f64(mix0)), _p(_f64(pose1)), _p(_f64(mix1)), _p(r), _p(J))
        if not want_jac:
            return r, None
        return r, (J[:105].reshape(15, 7), J[105:240].reshape(15, 9), J[240:345].reshape(15, 7), J[345:].reshape(15, 9))


    # ---- marginalization
    def sym_eigen(self, A):
        A = _f64(A)
        n = A.shape[0]
        ev, V = np.zeros(n), np.zeros((n, n))
        self.lib.orc_sym_eigen(n, _p(A), _p(ev), _p(V))
        return ev, V

    def reproj_accumulate_normal(self, r, J, idx_i, idx_j, idx_lm, col_pose, col_ext, col_lm, col_td, local_size):
        H, b = np.zeros((local_size, local_size)), np.zeros(local_size)
        r, J = _f64(r), _f64(J)
        self.lib.orc_reproj_accumulate_normal(r.shape[0], _p(r), _p(J), _p(_i32(idx_i)), _p(_i32(idx_j)), _p(_i32(idx_lm)),
                                              _p(_i32(col_pose)), int(col_ext), _p(_i32(col_lm)), int(col_td), local_size, _p(H), _p(b))
        return H, b

    def marginalize(self, H0, b0, m, eps=1e-8):
        H0, b0 = _f64(H0), _f64(b0)
        n = H0.shape[0]
        r = n - m
        J0, e0, Hp, bp = np.zeros((r, r)), np.zeros(r), np.zeros((r, r)), np.zeros(r)
        self.lib.orc_marginalize(n, m, _p(H0), _p(b0), C.c_double(eps), _p(J0), _p(e0), _p(Hp), _p(bp))
        return J0, e0, Hp, bp

    def marg_factor_eval(self, block_size, block_index, x0, x, J0, e0, want_jac=True):
        r = J0.shape[0]
        res = np.zeros(r)
        jac = np.zeros(r * int(np.sum(block_size))) if want_jac else None
        self.lib.orc_marg_factor_eval(r, len(block_size), _p(_i32(block_size)), _p(_i32(block_index)), _p(_f64(x0)), _p(_f64(x)),
                                      _p(_f64(J0)), _p(_f64(e0)), _p(res), _p(jac))
        return res, jac


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    if not os.path.exists(ORACLE_SO):
        build()
    return Oracle(C.CDLL(ORACLE_SO))
