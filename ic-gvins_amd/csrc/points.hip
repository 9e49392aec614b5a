// F4/F5: camera point maps and INS-aided flow predictions, one lane per point (FP64 math, FP32 storage).
// Reference: tracking/camera.cc:72-89,104-131 ; tracking/tracking.cc:367-378 (map-point prediction) and
// :465-479 (rotation-only prediction for un-triangulated reference features).
#include "dev_camera.h"
#include "icg_internal.h"

using namespace icgd;

__global__ void k_undistort(icg_camera cam, int n, float2 *pts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pts[i] = cam_undistort(cam, pts[i]);
}
__global__ void k_distort(icg_camera cam, int n, float2 *pts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pts[i] = cam_distort_pixel(cam, pts[i]);
}
// world2pixel(pw, pose) then distortPoints
__global__ void k_predict_mappoints(icg_camera cam, int n, const double *pw, const int32_t *pose_idx,
                                    const double *poses12, float2 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *R = poses12 + 12 * (size_t) pose_idx[i];
    const double *t = R + 9;
    double d0 = pw[3 * i] - t[0], d1 = pw[3 * i + 1] - t[1], d2 = pw[3 * i + 2] - t[2];
    double pc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) pc[j] = R[0 * 3 + j] * d0 + R[1 * 3 + j] * d1 + R[2 * 3 + j] * d2;
    float2 px = cam_cam2pixel(cam, pc[0], pc[1], pc[2]);
    out[i]    = cam_distort_pixel(cam, px);
}
__global__ void k_predict_rotation(icg_camera cam, int n, const float2 *in, const int32_t *rot_idx, const double *rots9,
                                   float2 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = rots9 + 9 * (size_t) rot_idx[i];
    float2 u        = cam_undistort(cam, in[i]);
    double x, y;
    cam_pixel2cam(cam, u.x, u.y, x, y);
    double pc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) pc[k] = r[k * 3 + 0] * x + r[k * 3 + 1] * y + r[k * 3 + 2] * 1.0;
    out[i] = cam_distort_campoint(cam, pc[0], pc[1], pc[2]);
}

// f3: per-observation arithmetic of GVINS::gvinsOutlierCulling (ic_gvins.cc:1068-1078) and parametersStatistic
// (:985): |Camera::reprojectionError| (camera.cc:153-157) and Tracking::isGoodToTrack (tracking.cc:813-829), one lane per
// observation.  The reference walks the weak_ptr graph under locks and calls both per observation; here the host flattens the
// window's observations once and the decisions run on the returned arrays.
__global__ void k_reproj_error(icg_camera cam, int n, const int32_t *pose_idx, const int32_t *lm_idx, const double *poses12, const double *pw,
                               const float2 *pix, double max_error, double min_depth, double max_depth, double *err, uint8_t *good) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *R = poses12 + 12 * (size_t) pose_idx[i];
    const double *t = R + 9;
    const double *p = pw + 3 * (size_t) lm_idx[i];
    double d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
    double pc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) pc[j] = R[0 * 3 + j] * d0 + R[1 * 3 + j] * d1 + R[2 * 3 + j] * d2;
    float2 px = cam_cam2pixel(cam, pc[0], pc[1], pc[2]);
    float2 pp = pix[i];
    const double ex = (double) (px.x - pp.x), ey = (double) (px.y - pp.y);
    const double e  = sqrt(ex * ex + ey * ey);
    err[i]  = e;
    good[i] = ((pc[2] > min_depth) && (pc[2] < max_depth) && !(e > max_error)) ? 1 : 0;
}

static int check_pts(icg_ctx *ctx, int n) {
    if (!ctx) return ICG_ERR_INVALID;
    if (!ctx->has_cam) return icg_fail(ctx, ICG_ERR_INVALID, "camera not set (icg_set_camera)");
    if (n < 0) return ICG_ERR_INVALID;
    if (n > ctx->cfg.max_points) return icg_fail(ctx, ICG_ERR_CAPACITY, "%d points > max_points %d", n, ctx->cfg.max_points);
    return 0;
}

extern "C" int icg_undistort_points(icg_ctx *ctx, int n, float *pts) {
    int rc = check_pts(ctx, n);
    if (rc || n == 0) return rc;
    if (!pts) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    if ((rc = c.reserve(sizeof(float) * 2 * (size_t) n * 2))) return rc;
    float2 *d_in = (float2 *) c.in(pts, 2 * (size_t) n);
    if ((rc = c.seal())) return rc;
    float2 *d_out = (float2 *) c.out(pts, 2 * (size_t) n);
    ICG_LAUNCH_GUARD(c); // before the device copy: on overflow d_out aliases the staged inputs
    ICG_HIP(ctx, hipMemcpyAsync(d_out, d_in, sizeof(float2) * n, hipMemcpyDeviceToDevice, ctx->stream));
    {
        icg_prof_scope ps(ctx, "undistort_points");
        hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_out);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_distort_points(icg_ctx *ctx, int n, float *pts) {
    int rc = check_pts(ctx, n);
    if (rc || n == 0) return rc;
    if (!pts) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    if ((rc = c.reserve(sizeof(float) * 2 * (size_t) n * 2))) return rc;
    float2 *d_in = (float2 *) c.in(pts, 2 * (size_t) n);
    if ((rc = c.seal())) return rc;
    float2 *d_out = (float2 *) c.out(pts, 2 * (size_t) n);
    ICG_LAUNCH_GUARD(c); // before the device copy: on overflow d_out aliases the staged inputs
    ICG_HIP(ctx, hipMemcpyAsync(d_out, d_in, sizeof(float2) * n, hipMemcpyDeviceToDevice, ctx->stream));
    {
        icg_prof_scope ps(ctx, "distort_points");
        hipLaunchKernelGGL(k_distort, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_out);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_predict_mappoints(icg_ctx *ctx, int n, const double *pw, const int32_t *pose_idx, int n_poses,
                                     const double *poses12, float *pts_out) {
    int rc = check_pts(ctx, n);
    if (rc || n == 0) return rc;
    if (!pw || !pose_idx || !poses12 || !pts_out || n_poses <= 0) return ICG_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (pose_idx[i] < 0 || pose_idx[i] >= n_poses) return icg_fail(ctx, ICG_ERR_INVALID, "pose_idx[%d] out of range", i);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    if ((rc = c.reserve((size_t) n * (24 + 4 + 8) + (size_t) n_poses * 96))) return rc;
    const double *d_pw  = c.in(pw, 3 * (size_t) n);
    const int32_t *d_pi = c.in(pose_idx, (size_t) n);
    const double *d_po  = c.in(poses12, 12 * (size_t) n_poses);
    if ((rc = c.seal())) return rc;
    float2 *d_out = (float2 *) c.out(pts_out, 2 * (size_t) n);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "predict_mappoints");
        hipLaunchKernelGGL(k_predict_mappoints, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_pw, d_pi,
                           d_po, d_out);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_predict_rotation(icg_ctx *ctx, int n, const float *pts_in, const int32_t *rot_idx, int n_rots,
                                    const double *rots9, float *pts_out) {
    int rc = check_pts(ctx, n);
    if (rc || n == 0) return rc;
    if (!pts_in || !rot_idx || !rots9 || !pts_out || n_rots <= 0) return ICG_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (rot_idx[i] < 0 || rot_idx[i] >= n_rots) return icg_fail(ctx, ICG_ERR_INVALID, "rot_idx[%d] out of range", i);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    if ((rc = c.reserve((size_t) n * (8 + 4 + 8) + (size_t) n_rots * 72))) return rc;
    const float2 *d_in  = (const float2 *) c.in(pts_in, 2 * (size_t) n);
    const int32_t *d_ri = c.in(rot_idx, (size_t) n);
    const double *d_r   = c.in(rots9, 9 * (size_t) n_rots);
    if ((rc = c.seal())) return rc;
    float2 *d_out = (float2 *) c.out(pts_out, 2 * (size_t) n);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "predict_rotation");
        hipLaunchKernelGGL(k_predict_rotation, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_in, d_ri, d_r,
                           d_out);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_reproj_error_batch(icg_ctx *ctx, int n, const int32_t *pose_idx, const int32_t *lm_idx, int n_poses, const double *poses12,
                                      int n_lm, const double *pw, const float *pix, double max_error, double min_depth, double max_depth,
                                      double *err_out, uint8_t *good_out) {
    if (!ctx) return ICG_ERR_INVALID;
    if (!ctx->has_cam) return icg_fail(ctx, ICG_ERR_INVALID, "camera not set (icg_set_camera)");
    if (n < 0) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (!pose_idx || !lm_idx || !poses12 || !pw || !pix || n_poses <= 0 || n_lm <= 0) return ICG_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (pose_idx[i] < 0 || pose_idx[i] >= n_poses || lm_idx[i] < 0 || lm_idx[i] >= n_lm)
            return icg_fail(ctx, ICG_ERR_INVALID, "observation %d: pose/landmark index out of range", i);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    int rc = c.reserve((size_t) n * (4 + 4 + 8 + 8 + 1) + sizeof(double) * (12 * (size_t) n_poses + 3 * (size_t) n_lm) + 4096);
    if (rc) return rc;
    const double *d_po = c.in(poses12, 12 * (size_t) n_poses);
    const double *d_pw = c.in(pw, 3 * (size_t) n_lm);
    const int32_t *d_pi = c.in_zc(pose_idx, (size_t) n);
    const int32_t *d_li = c.in_zc(lm_idx, (size_t) n);
    const float2 *d_px = (const float2 *) c.in_zc(pix, 2 * (size_t) n);
    if ((rc = c.seal())) return rc;
    double *d_e  = c.out_zc(err_out, (size_t) n);
    uint8_t *d_g = c.out_zc(good_out, (size_t) n);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_error");
        hipLaunchKernelGGL(k_reproj_error, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->cam, n, d_pi, d_li, d_po, d_pw, d_px, max_error,
                           min_depth, max_depth, d_e, d_g);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}
