// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own small navigation factors and Earth / attitude helpers
// (/root/reference/ic_gvins/ic_gvins/factors/gnss_factor.h, preintegration/imu_{error,pose_prior,mix_prior}_factor.h, common/earth.h,
// common/rotation.h, common/gpstime.h, MISC::detectZeroVelocity of misc.cc — compiled unmodified from where they lie) behind C entry
// points with the argument layout of icgh_nav_factor / icgh_nav_helper / icgh_detect_zero_velocity (ic-gvins_amd/host/capi_replay.cc).
// Linear algebra comes from the Eigen-interface shim in shim/ (NOT real Eigen — stated in DESIGN.md).  SURVEY.md §8 row f2.
#include "misc.h"

#include "fileio/filesaver.cc" // the reference sources themselves (single translation unit)
#include "misc.cc"

#include "common/earth.h"
#include "common/gpstime.h"
#include "common/rotation.h"
#include "factors/gnss_factor.h"
#include "preintegration/imu_error_factor.h"
#include "preintegration/imu_mix_prior_factor.h"
#include "preintegration/imu_pose_prior_factor.h"

extern "C" {

int ref_nav_factor(int kind, const double *aux, const double *x, double *residuals, double *jacobian) {
    const double *params[1] = {x};
    double *J[1]            = {jacobian};
    if (kind == 0) {
        GNSS g;
        g.time = 0;
        g.blh  = Vector3d(aux[0], aux[1], aux[2]);
        g.std  = Vector3d(aux[3], aux[4], aux[5]);
        g.isyawvalid = false;
        g.yaw        = 0;
        GnssFactor f(g, Vector3d(aux[6], aux[7], aux[8]));
        return f.Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
    } else if (kind == 1) {
        ImuErrorFactor f(Preintegration::PREINTEGRATION_NORMAL);
        return f.Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
    } else if (kind == 2) {
        double pose[7], std6[6];
        memcpy(pose, aux, sizeof pose);
        memcpy(std6, aux + 7, sizeof std6);
        ImuPosePriorFactor f(pose, std6);
        return f.Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
    } else if (kind == 3) {
        double mix[18] = {0}, mix_std[18];
        for (int k = 0; k < 18; k++) mix_std[k] = 1.0;
        memcpy(mix, aux, sizeof(double) * 9);
        memcpy(mix_std, aux + 9, sizeof(double) * 9);
        ImuMixPriorFactor f(Preintegration::PREINTEGRATION_NORMAL, mix, mix_std);
        return f.Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
    }
    return -1;
}

int ref_nav_helper(int what, const double *a, const double *b, double *out) {
    Vector3d A(a[0], a[1], a[2]), B = b ? Vector3d(b[0], b[1], b[2]) : Vector3d(0, 0, 0);
    Vector3d r(0, 0, 0);
    switch (what) {
    case 0: out[0] = Earth::gravity(A); return 0;
    case 1: r = Earth::global2local(A, B); break;
    case 2: r = Earth::local2global(A, B); break;
    case 3: r = Earth::iewn(A, B); break;
    case 4: {
        Quaterniond q = Rotation::euler2quaternion(A);
        out[0] = q.x(), out[1] = q.y(), out[2] = q.z(), out[3] = q.w();
        return 0;
    }
    case 5: r = Rotation::matrix2euler(Rotation::quaternion2matrix(Quaterniond(a[3], a[0], a[1], a[2]))); break;
    case 6: {
        int week;
        double sow;
        GpsTime::unix2gps(a[0], week, sow);
        out[0] = week, out[1] = sow;
        return 0;
    }
    default: return -1;
    }
    out[0] = r[0], out[1] = r[1], out[2] = r[2];
    return 0;
}

int ref_detect_zero_velocity(int n, const double *rows6, double imudatarate, double *average6) {
    std::vector<IMU> buf((size_t) n);
    for (int k = 0; k < n; k++) {
        buf[(size_t) k].time = 0, buf[(size_t) k].dt = 0, buf[(size_t) k].odovel = 0;
        buf[(size_t) k].dtheta = Vector3d(rows6[6 * k], rows6[6 * k + 1], rows6[6 * k + 2]);
        buf[(size_t) k].dvel   = Vector3d(rows6[6 * k + 3], rows6[6 * k + 4], rows6[6 * k + 5]);
    }
    std::vector<double> avg;
    bool z = MISC::detectZeroVelocity(buf, imudatarate, avg);
    for (int k = 0; k < 6; k++) average6[k] = avg[(size_t) k];
    return z ? 1 : 0;
}
}
