// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Small fixed-size linear algebra restating the Eigen conventions the reference relies on
// (reference: ic_gvins/ic_gvins/common/rotation.h:32-120, Eigen::Quaternion semantics):
//   * quaternion storage order x,y,z,w (Eigen coeffs()), ctor order (w,x,y,z)
//   * q*v  = v + w*(2 q.vec x v) + q.vec x (2 q.vec x v)            (Eigen _transformVector)
//   * q.inverse() = conj / squaredNorm
//   * toRotationMatrix() uses raw (non-normalised) coefficients
// Compile with -ffp-contract=off so no FMA is formed.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V3 {
    double x, y, z;
};
struct Q4 {
    double x, y, z, w;
};
struct M3 {
    double m[3][3];
};

static inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

static inline Q4 quat_wxyz(double w, double x, double y, double z) { return Q4{x, y, z, w}; }
static inline Q4 quat_identity() { return Q4{0, 0, 0, 1}; }
static inline double qnorm2(Q4 q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
static inline Q4 qconj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
static inline Q4 qinv(Q4 q) {
    double n2 = qnorm2(q);
    if (n2 > 0) return Q4{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
    return Q4{0, 0, 0, 0};
}
static inline Q4 qnormalized(Q4 q) {
    double n = std::sqrt(qnorm2(q));
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
}
// Hamilton product a*b (Eigen operator*)
static inline Q4 qmul(Q4 a, Q4 b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
static inline V3 qrot(Q4 q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv    = uv + uv;
    return v + q.w * uv + cross(qv, uv);
}
static inline M3 qmat(Q4 q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 r;
    r.m[0][0] = 1 - (tyy + tzz);
    r.m[0][1] = txy - twz;
    r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;
    r.m[1][1] = 1 - (txx + tzz);
    r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;
    r.m[2][1] = tyz + twx;
    r.m[2][2] = 1 - (txx + tyy);
    return r;
}
// Rotation::rotvec2quaternion (rotation.h:72-76): AngleAxis(|v|, v/|v|); zero vector -> identity
static inline Q4 rotvec2quat(V3 rv) {
    double angle = norm(rv);
    V3 axis      = rv;
    if (angle > 0) axis = rv / angle;
    double s = std::sin(0.5 * angle), c = std::cos(0.5 * angle);
    return Q4{s * axis.x, s * axis.y, s * axis.z, c};
}

static inline M3 m3_identity() {
    M3 r;
    memset(&r, 0, sizeof r);
    r.m[0][0] = r.m[1][1] = r.m[2][2] = 1;
    return r;
}
static inline M3 m3_T(const M3 &a) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
static inline M3 m3_mul(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
static inline M3 m3_neg(const M3 &a) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = -a.m[i][j];
    return r;
}
static inline M3 m3_add(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
static inline M3 m3_sub(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j];
    return r;
}
static inline M3 m3_scale(const M3 &a, double s) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] * s;
    return r;
}
static inline V3 m3_vec(const M3 &a, V3 v) {
    return V3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
// Rotation::skewSymmetric (rotation.h:97-101)
static inline M3 skew(V3 v) {
    M3 r;
    r.m[0][0] = 0;
    r.m[0][1] = -v.z;
    r.m[0][2] = v.y;
    r.m[1][0] = v.z;
    r.m[1][1] = 0;
    r.m[1][2] = -v.x;
    r.m[2][0] = -v.y;
    r.m[2][1] = v.x;
    r.m[2][2] = 0;
    return r;
}

} // namespace orc
