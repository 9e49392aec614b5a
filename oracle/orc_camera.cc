// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the reference camera model
//   Camera::undistortPoints   ic_gvins/ic_gvins/tracking/camera.cc:72-74  -> cv::undistortPoints(pts,pts,K,D,Mat(),K)
//                             (OpenCV calib3d/src/undistort.dispatch.cpp cvUndistortPointsInternal, 5 fixed iterations,
//                              SURVEY.md Appendix B.6; input side ignores skew, output side applies full P)
//   Camera::distortPoints     camera.cc:76-89      Camera::distortCameraPoint  camera.cc:104-117
//   Camera::pixel2cam         camera.cc:123-127    Camera::cam2pixel           camera.cc:129-131
//   Camera::world2cam         camera.cc:145-147    Camera::world2pixel         camera.cc:141-143
//   Camera::reprojectionError camera.cc:153-157
// cam = {fx, fy, cx, cy, skew, k1, k2, p1, p2, k3}.  pose = {R row-major 9, t 3}.
// PARITY UNPINNED for undistortPoints (OpenCV absent); the closed-form functions are fully specified in-tree.
#include "oracle.h"
#include <cmath>

namespace {
struct Cam {
    double fx, fy, cx, cy, skew, k1, k2, p1, p2, k3;
};
static inline Cam load(const double *c) { return Cam{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9]}; }

static inline void pixel2cam(const Cam &c, float px, float py, double &x, double &y) {
    y = (py - c.cy) / c.fy;
    x = (px - c.cx - c.skew * y) / c.fx;
}
static inline void cam2pixel(const Cam &c, double X, double Y, double Z, float &px, float &py) {
    px = (float) ((c.fx * X + c.skew * Y) / Z + c.cx);
    py = (float) (c.fy * Y / Z + c.cy);
}
} // namespace

extern "C" {

void orc_undistort_points(const double *cam, int n, float *pts) {
    Cam c      = load(cam);
    double ifx = 1. / c.fx, ify = 1. / c.fy;
    for (int i = 0; i < n; i++) {
        double x = pts[2 * i], y = pts[2 * i + 1];
        double u = x, v = y;
        x         = (x - c.cx) * ifx;
        y         = (y - c.cy) * ify;
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2     = x * x + y * y;
            double icdist = 1. / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
            if (icdist < 0) {
                x = (u - c.cx) * ifx;
                y = (v - c.cy) * ify;
                break;
            }
            double deltaX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
            double deltaY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
            x             = (x0 - deltaX) * icdist;
            y             = (y0 - deltaY) * icdist;
        }
        double xx      = c.fx * x + c.skew * y + c.cx;
        double yy      = c.fy * y + c.cy;
        pts[2 * i]     = (float) xx;
        pts[2 * i + 1] = (float) yy;
    }
}

void orc_distort_points(const double *cam, int n, float *pts) {
    Cam c = load(cam);
    for (int i = 0; i < n; i++) {
        double x, y;
        pixel2cam(c, pts[2 * i], pts[2 * i + 1], x, y);
        double r2 = x * x + y * y;
        double rr = (1 + c.k1 * r2 + c.k2 * r2 * r2 + c.k3 * r2 * r2 * r2);
        double xd = x * rr + 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
        double yd = y * rr + c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
        cam2pixel(c, xd, yd, 1.0, pts[2 * i], pts[2 * i + 1]);
    }
}

// pc: n x 3 doubles (camera-frame points) -> distorted pixels
void orc_distort_camera_points(const double *cam, int n, const double *pc, float *pts) {
    Cam c = load(cam);
    for (int i = 0; i < n; i++) {
        double x  = pc[3 * i] / pc[3 * i + 2];
        double y  = pc[3 * i + 1] / pc[3 * i + 2];
        double r2 = x * x + y * y;
        double rr = (1 + c.k1 * r2 + c.k2 * r2 * r2 + c.k3 * r2 * r2 * r2);
        double xd = (double) (float) (x * rr + 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x));
        double yd = (double) (float) (y * rr + c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y);
        cam2pixel(c, xd, yd, 1.0, pts[2 * i], pts[2 * i + 1]);
    }
}

void orc_pixel2cam(const double *cam, int n, const float *pts, double *pc) {
    Cam c = load(cam);
    for (int i = 0; i < n; i++) {
        pixel2cam(c, pts[2 * i], pts[2 * i + 1], pc[3 * i], pc[3 * i + 1]);
        pc[3 * i + 2] = 1.0;
    }
}

// world2cam: R^T (pw - t)
void orc_world2cam(const double *pose, int n, const double *pw, double *pc) {
    const double *R = pose, *t = pose + 9;
    for (int i = 0; i < n; i++) {
        double d0 = pw[3 * i] - t[0], d1 = pw[3 * i + 1] - t[1], d2 = pw[3 * i + 2] - t[2];
        for (int j = 0; j < 3; j++) pc[3 * i + j] = R[0 * 3 + j] * d0 + R[1 * 3 + j] * d1 + R[2 * 3 + j] * d2;
    }
}

void orc_world2pixel(const double *cam, const double *pose, int n, const double *pw, float *pts) {
    Cam c = load(cam);
    for (int i = 0; i < n; i++) {
        double pc[3];
        orc_world2cam(pose, 1, pw + 3 * i, pc);
        cam2pixel(c, pc[0], pc[1], pc[2], pts[2 * i], pts[2 * i + 1]);
    }
}

// INS rotation-only prediction for reference features, tracking.cc:465-479:
//   r_cur_pre = R_cur^T R_pre ; p = distortCameraPoint(r_cur_pre * pixel2cam(undistort(p)))
void orc_predict_rotation(const double *cam, const double *R_cur, const double *R_pre, int n, const float *pts_in,
                          float *pts_out) {
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r[i * 3 + j] = R_cur[0 * 3 + i] * R_pre[0 * 3 + j] + R_cur[1 * 3 + i] * R_pre[1 * 3 + j] +
                           R_cur[2 * 3 + i] * R_pre[2 * 3 + j];
    Cam c = load(cam);
    for (int i = 0; i < n; i++) {
        float u[2] = {pts_in[2 * i], pts_in[2 * i + 1]};
        orc_undistort_points(cam, 1, u);
        double x, y;
        pixel2cam(c, u[0], u[1], x, y);
        double pc[3];
        for (int k = 0; k < 3; k++) pc[k] = r[k * 3 + 0] * x + r[k * 3 + 1] * y + r[k * 3 + 2] * 1.0;
        orc_distort_camera_points(cam, 1, pc, pts_out + 2 * i);
    }
}

// Per-observation arithmetic of GVINS::gvinsOutlierCulling (ic_gvins.cc:1068-1078) / parametersStatistic (:985), SURVEY.md §8 f3:
//   err[i]  = |Camera::reprojectionError(pose, pw, pp)|  (camera.cc:153-157: world2pixel in float, float differences, double norm)
//   good[i] = Tracking::isGoodToTrack(pp, pose, pw, scale, depth_scale)  (tracking.cc:813-829, isGoodDepth :247-249):
//             min_depth < z < max_depth  &&  !(err > max_error)      (max_error = reprojection_error_std * scale)
// PINNED against the reference's own Camera / Tracking code (oracle/_ref/libref_tracking.so, tests/golden/cull_ref_golden.npz).
void orc_reproj_error_batch(const double *cam, int n, const int32_t *pose_idx, const int32_t *lm_idx, const double *poses12, const double *pw,
                            const float *pix, double max_error, double min_depth, double max_depth, double *err_out, uint8_t *good_out) {
    Cam c = load(cam);
    for (int i = 0; i < n; i++) {
        double pc[3];
        orc_world2cam(poses12 + 12 * (size_t) pose_idx[i], 1, pw + 3 * (size_t) lm_idx[i], pc);
        float px, py;
        cam2pixel(c, pc[0], pc[1], pc[2], px, py);
        const double ex = (double) (px - pix[2 * i]), ey = (double) (py - pix[2 * i + 1]);
        const double e  = std::sqrt(ex * ex + ey * ey);
        if (err_out) err_out[i] = e;
        if (good_out) good_out[i] = ((pc[2] > min_depth) && (pc[2] < max_depth) && !(e > max_error)) ? 1 : 0;
    }
}

} // extern "C"
