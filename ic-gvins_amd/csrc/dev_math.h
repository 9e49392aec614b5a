// Device-side FP64 small-vector helpers (gfx950).  Conventions follow the reference's use of Eigen:
// quaternion parameter storage [x,y,z,w]; rotation of a vector by a quaternion is the cross-product form;
// rotation matrices are built from raw (possibly slightly non-unit) coefficients
// (reference: common/rotation.h:32-120, factors/reprojection_factor.h:57-92).
#pragma once
#include <hip/hip_runtime.h>

namespace icgd {

struct d3 {
    double x, y, z;
};
struct dq {
    double x, y, z, w;
};
struct m33 {
    double a[9];
}; // row-major

__device__ __forceinline__ d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 add(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 sub(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 scl(double s, d3 a) { return d3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ d3 dvd(d3 a, double s) { return d3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ d3 crs(d3 a, d3 b) {
    return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ dq q_from_xyzw(const double *p) { return dq{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ dq q_inv(dq q) {
    double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (n2 > 0) return dq{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
    return dq{0, 0, 0, 0};
}
__device__ __forceinline__ dq q_mul(dq a, dq b) {
    return dq{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ d3 q_rot(dq q, d3 v) {
    d3 qv = mk3(q.x, q.y, q.z);
    d3 uv = crs(qv, v);
    uv    = add(uv, uv);
    return add(add(v, scl(q.w, uv)), crs(qv, uv));
}
__device__ __forceinline__ m33 q_mat(dq q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    m33 r;
    r.a[0] = 1 - (tyy + tzz);
    r.a[1] = txy - twz;
    r.a[2] = txz + twy;
    r.a[3] = txy + twz;
    r.a[4] = 1 - (txx + tzz);
    r.a[5] = tyz - twx;
    r.a[6] = txz - twy;
    r.a[7] = tyz + twx;
    r.a[8] = 1 - (txx + tyy);
    return r;
}
__device__ __forceinline__ m33 m_T(const m33 &m) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i * 3 + j] = m.a[j * 3 + i];
    return r;
}
__device__ __forceinline__ m33 m_mul(const m33 &p, const m33 &q) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            r.a[i * 3 + j] = p.a[i * 3 + 0] * q.a[0 * 3 + j] + p.a[i * 3 + 1] * q.a[1 * 3 + j] + p.a[i * 3 + 2] * q.a[2 * 3 + j];
    return r;
}
__device__ __forceinline__ m33 m_neg(const m33 &p) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.a[i] = -p.a[i];
    return r;
}
__device__ __forceinline__ m33 m_add(const m33 &p, const m33 &q) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.a[i] = p.a[i] + q.a[i];
    return r;
}
__device__ __forceinline__ m33 m_sub(const m33 &p, const m33 &q) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.a[i] = p.a[i] - q.a[i];
    return r;
}
__device__ __forceinline__ m33 m_eye() {
    m33 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.a[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return r;
}
__device__ __forceinline__ d3 m_vec(const m33 &m, d3 v) {
    return d3{m.a[0] * v.x + m.a[1] * v.y + m.a[2] * v.z, m.a[3] * v.x + m.a[4] * v.y + m.a[5] * v.z,
              m.a[6] * v.x + m.a[7] * v.y + m.a[8] * v.z};
}
__device__ __forceinline__ m33 m_skew(d3 v) {
    m33 r;
    r.a[0] = 0;
    r.a[1] = -v.z;
    r.a[2] = v.y;
    r.a[3] = v.z;
    r.a[4] = 0;
    r.a[5] = -v.x;
    r.a[6] = -v.y;
    r.a[7] = v.x;
    r.a[8] = 0;
    return r;
}
__device__ __forceinline__ dq q_normalized(dq q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return dq{q.x / n, q.y / n, q.z / n, q.w / n};
}
// Rotation::rotvec2quaternion (common/rotation.h:72-76)
__device__ __forceinline__ dq rotvec2quat(d3 rv) {
    double angle = sqrt(rv.x * rv.x + rv.y * rv.y + rv.z * rv.z);
    d3 axis      = rv;
    if (angle > 0) axis = dvd(rv, angle);
    double s = sin(0.5 * angle), c = cos(0.5 * angle);
    return dq{s * axis.x, s * axis.y, s * axis.z, c};
}
// Rotation::quaternion2vector (common/rotation.h:78-81): Eigen AngleAxis(q), angle * axis
__device__ __forceinline__ d3 quat2rotvec(dq q) {
    d3 vec   = mk3(q.x, q.y, q.z);
    double n = sqrt(vec.x * vec.x + vec.y * vec.y + vec.z * vec.z);
    if (n != 0.0) {
        double angle = 2.0 * atan2(n, fabs(q.w));
        if (q.w < 0) n = -n;
        return scl(angle, dvd(vec, n));
    }
    return mk3(0.0, 0.0, 0.0);
}
__device__ __forceinline__ m33 m_scale(const m33 &m, double s) {
    m33 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.a[i] = m.a[i] * s;
    return r;
}

} // namespace icgd
