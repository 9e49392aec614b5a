// Device-side camera model (FP64 math, FP32 point storage) shared by the point kernels and the fused LK epilogue.
// Reference: tracking/camera.cc:72-157; cv::undistortPoints semantics per SURVEY.md Appendix B.6
// (5 fixed-point iterations, input side ignores skew, output side applies the full K).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/icgvins_hip.h"

namespace icgd {

__device__ __forceinline__ void cam_pixel2cam(const icg_camera &c, float px, float py, double &x, double &y) {
    y = (py - c.cy) / c.fy;
    x = (px - c.cx - c.skew * y) / c.fx;
}

__device__ __forceinline__ float2 cam_cam2pixel(const icg_camera &c, double X, double Y, double Z) {
    float2 r;
    r.x = (float) ((c.fx * X + c.skew * Y) / Z + c.cx);
    r.y = (float) (c.fy * Y / Z + c.cy);
    return r;
}

__device__ __forceinline__ float2 cam_undistort(const icg_camera &c, float2 p) {
    double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = p.x, y = p.y;
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
    float2 r;
    r.x = (float) (c.fx * x + c.skew * y + c.cx);
    r.y = (float) (c.fy * y + c.cy);
    return r;
}

// radtan forward model on normalised coordinates
__device__ __forceinline__ void cam_distort_xy(const icg_camera &c, double x, double y, double &xd, double &yd) {
    double r2 = x * x + y * y;
    double rr = (1 + c.k1 * r2 + c.k2 * r2 * r2 + c.k3 * r2 * r2 * r2);
    xd        = x * rr + 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
    yd        = y * rr + c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
}

__device__ __forceinline__ float2 cam_distort_pixel(const icg_camera &c, float2 p) { // camera.cc:76-89
    double x, y, xd, yd;
    cam_pixel2cam(c, p.x, p.y, x, y);
    cam_distort_xy(c, x, y, xd, yd);
    return cam_cam2pixel(c, xd, yd, 1.0);
}

__device__ __forceinline__ float2 cam_distort_campoint(const icg_camera &c, double X, double Y, double Z) { // :104-117
    double x = X / Z, y = Y / Z, xd, yd;
    cam_distort_xy(c, x, y, xd, yd);
    return cam_cam2pixel(c, (double) (float) xd, (double) (float) yd, 1.0);
}

} // namespace icgd
