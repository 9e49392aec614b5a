// Earth model and attitude helpers of the navigation layer — reference common/earth.h:36-260 (WGS-84 constants, gravity,
// blh <-> ecef, local (n-frame at an origin) <-> global, Earth rotation in the n-frame), common/rotation.h:44-100 (Euler angles),
// common/gpstime.h:30-48.  Host-only scalar math used a few times per GNSS fix / window solve by the GVINS orchestrator
// (gvins_hip.cc) and by MISC::writeNavResult; pinned against the reference's own headers in tests/golden/nav_ref_golden.npz.
#pragma once
#include <cmath>

#include "factors.h"
#include "types.h"

namespace icg {

const double D2R = (M_PI / 180.0); // common/angle.h:31-32
const double R2D = (180.0 / M_PI);

namespace Earth {
const double WGS84_WIE = 7.2921151467E-5;
const double WGS84_RA  = 6378137.0000000000;
const double WGS84_E1  = 0.0066943799901413156;

inline double gravity(const Vector3d &blh) { // earth.h:47-55
    double sin2 = std::sin(blh[0]);
    sin2 *= sin2;
    return 9.7803267715 * (1 + 0.0052790414 * sin2 + 0.0000232718 * sin2 * sin2) + blh[2] * (0.0000000043977311 * sin2 - 0.0000030876910891) +
           0.0000000000007211 * blh[2] * blh[2];
}
inline double RN(double lat) { // earth.h:71-74
    double sinlat = std::sin(lat);
    return WGS84_RA / std::sqrt(1.0 - WGS84_E1 * sinlat * sinlat);
}
inline Matrix3d cne(const Vector3d &blh) { // earth.h:77-101
    double sinlat = std::sin(blh[0]), sinlon = std::sin(blh[1]), coslat = std::cos(blh[0]), coslon = std::cos(blh[1]);
    Matrix3d d;
    d(0, 0) = -sinlat * coslon, d(0, 1) = -sinlon, d(0, 2) = -coslat * coslon;
    d(1, 0) = -sinlat * sinlon, d(1, 1) = coslon, d(1, 2) = -coslat * sinlon;
    d(2, 0) = coslat, d(2, 1) = 0, d(2, 2) = -sinlat;
    return d;
}
inline Vector3d blh2ecef(const Vector3d &blh) { // earth.h:128-141
    double coslat = std::cos(blh[0]), sinlat = std::sin(blh[0]), coslon = std::cos(blh[1]), sinlon = std::sin(blh[1]);
    double rn = RN(blh[0]), rnh = rn + blh[2];
    return {rnh * coslat * coslon, rnh * coslat * sinlon, (rnh - rn * WGS84_E1) * sinlat};
}
inline Vector3d ecef2blh(const Vector3d &ecef) { // earth.h:144-161
    double p = std::sqrt(ecef[0] * ecef[0] + ecef[1] * ecef[1]);
    double rn, lat, lon, h = 0, h2;
    lat = std::atan(ecef[2] / (p * (1.0 - WGS84_E1)));
    lon = 2.0 * std::atan2(ecef[1], ecef[0] + p);
    do {
        h2  = h;
        rn  = RN(lat);
        h   = p / std::cos(lat) - rn;
        lat = std::atan(ecef[2] / (p * (1.0 - WGS84_E1 * rn / (rn + h))));
    } while (std::fabs(h - h2) > 1.0e-4);
    return {lat, lon, h};
}
inline Vector3d local2global(const Vector3d &origin, const Vector3d &local) { // earth.h:188-197
    Vector3d ecef1 = blh2ecef(origin) + cne(origin) * local;
    return ecef2blh(ecef1);
}
inline Vector3d global2local(const Vector3d &origin, const Vector3d &global) { // earth.h:199-206
    Vector3d ecef0 = blh2ecef(origin);
    Matrix3d cn0e  = cne(origin);
    Vector3d ecef1 = blh2ecef(global);
    return cn0e.transpose() * (ecef1 - ecef0);
}
inline Vector3d iewn(double lat) { return {WGS84_WIE * std::cos(lat), 0, -WGS84_WIE * std::sin(lat)}; } // earth.h:244-246
inline Vector3d iewn(const Vector3d &origin, const Vector3d &local) { return iewn(local2global(origin, local)[0]); } // :248-252
} // namespace Earth

namespace Rotation {
// Rotation::euler2quaternion (rotation.h:87-91): Rz(yaw) * Ry(pitch) * Rx(roll), euler = (roll, pitch, yaw)
inline Quaterniond euler2quaternion(const Vector3d &euler) {
    double cr = std::cos(0.5 * euler[0]), sr = std::sin(0.5 * euler[0]);
    double cp = std::cos(0.5 * euler[1]), sp = std::sin(0.5 * euler[1]);
    double cy = std::cos(0.5 * euler[2]), sy = std::sin(0.5 * euler[2]);
    // qz * qy, then * qx (Hamilton products of axis quaternions, as Eigen composes the AngleAxis factors)
    double zw = cy, zz = sy;                                        // qz = (0, 0, sy ; cy)
    double aw = zw * cp, ax = -zz * sp, ay = zw * sp, az = zz * cp; // qz * qy, qy = (0, sp, 0 ; cp)
    Quaterniond q;
    q.w = aw * cr - ax * sr;
    q.x = aw * sr + ax * cr;
    q.y = ay * cr + az * sr;
    q.z = az * cr - ay * sr;
    return q;
}
inline Matrix3d quaternion2matrix(const Quaterniond &q) { // Eigen toRotationMatrix from raw coefficients
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y,
           tzz = tz * q.z;
    Matrix3d r;
    r(0, 0) = 1 - (tyy + tzz), r(0, 1) = txy - twz, r(0, 2) = txz + twy;
    r(1, 0) = txy + twz, r(1, 1) = 1 - (txx + tzz), r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy, r(2, 1) = tyz + twx, r(2, 2) = 1 - (txx + tyy);
    return r;
}
inline Quaterniond matrix2quaternion(const Matrix3d &m) { // Eigen Quaterniond(Matrix3d)
    Quaterniond q;
    double *c[4] = {&q.x, &q.y, &q.z, &q.w};
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t   = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t   = 0.5 / t;
        q.x = (m(2, 1) - m(1, 2)) * t;
        q.y = (m(0, 2) - m(2, 0)) * t;
        q.z = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t     = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        *c[i] = 0.5 * t;
        t     = 0.5 / t;
        q.w   = (m(k, j) - m(j, k)) * t;
        *c[j] = (m(j, i) + m(i, j)) * t;
        *c[k] = (m(k, i) + m(i, k)) * t;
    }
    return q;
}
inline Vector3d matrix2euler(const Matrix3d &dcm) { // rotation.h:46-66
    Vector3d e;
    e[1] = std::atan(-dcm(2, 0) / std::sqrt(dcm(2, 1) * dcm(2, 1) + dcm(2, 2) * dcm(2, 2)));
    if (dcm(2, 0) <= -0.999) {
        e[0] = std::atan2(dcm(2, 1), dcm(2, 2));
        e[2] = std::atan2((dcm(1, 2) - dcm(0, 1)), (dcm(0, 2) + dcm(1, 1)));
    } else if (dcm(2, 0) >= 0.999) {
        e[0] = std::atan2(dcm(2, 1), dcm(2, 2));
        e[2] = M_PI + std::atan2((dcm(1, 2) + dcm(0, 1)), (dcm(0, 2) - dcm(1, 1)));
    } else {
        e[0] = std::atan2(dcm(2, 1), dcm(2, 2));
        e[2] = std::atan2(dcm(1, 0), dcm(0, 0));
    }
    if (e[2] < 0) e[2] = M_PI * 2 + e[2];
    return e;
}
inline Quaterniond normalized(const Quaterniond &q) {
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Quaterniond{q.x / n, q.y / n, q.z / n, q.w / n};
}
inline Vector3d rotate(const Quaterniond &q, const Vector3d &v) { return quaternion2matrix(q) * v; }
} // namespace Rotation

namespace GpsTime { // common/gpstime.h:30-48
const int GPS_LEAP_SECOND = 18;
inline void unix2gps(double unixs, int &week, double &sow) {
    double seconds = unixs + GPS_LEAP_SECOND - 315964800;
    week           = (int) std::floor(seconds / 604800);
    sow            = seconds - week * 604800.0;
}
} // namespace GpsTime

} // namespace icg
