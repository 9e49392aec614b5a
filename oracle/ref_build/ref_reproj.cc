// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own ReprojectionFactor::Evaluate
// (/root/reference/ic_gvins/ic_gvins/factors/reprojection_factor.h, compiled unmodified from where it lies) behind a C
// entry point.  Linear algebra comes from the minimal Eigen-interface shim in shim/ (NOT real Eigen — stated in DESIGN.md).
#include "factors/reprojection_factor.h"

extern "C" {
// obs15 = pts0[3], pts1[3], vel0[3], vel1[3], td0, td1, std ; J46 = 3 x [2x7] + 2 + 2
int ref_reproj_eval_one(const double *obs15, const double *pose_i, const double *pose_j, const double *ext, double invdepth,
                        double td, double *r2, double *J46) {
    ReprojectionFactor f(Vector3d(obs15[0], obs15[1], obs15[2]), Vector3d(obs15[3], obs15[4], obs15[5]),
                         Vector3d(obs15[6], obs15[7], obs15[8]), Vector3d(obs15[9], obs15[10], obs15[11]), obs15[12], obs15[13],
                         obs15[14]);
    const double *params[5] = {pose_i, pose_j, ext, &invdepth, &td};
    double *jac[5]          = {J46, J46 + 14, J46 + 28, J46 + 42, J46 + 44};
    return f.Evaluate(params, r2, J46 ? jac : nullptr) ? 0 : 1;
}
// Rotation helpers of common/rotation.h, for cross-checking the oracle's quaternion conventions
void ref_rotvec2quaternion(const double *rv, double *q_xyzw) {
    Quaterniond q = Rotation::rotvec2quaternion(Vector3d(rv[0], rv[1], rv[2]));
    q_xyzw[0] = q.x(), q_xyzw[1] = q.y(), q_xyzw[2] = q.z(), q_xyzw[3] = q.w();
}
}
