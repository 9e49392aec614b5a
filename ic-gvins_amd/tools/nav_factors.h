// The small host-evaluated factors of the GVINS window (tens of blocks per solve; WindowSolver adds them to the reduced system):
//   GnssFactor          reference factors/gnss_factor.h:33-80            SizedCostFunction<3, 7>
//   ImuErrorFactor      reference preintegration/imu_error_factor.h:30-95   6 residuals on the 9-vector mix block (Normal / Earth)
//   ImuPosePriorFactor  reference preintegration/imu_pose_prior_factor.h:30-72
//   ImuMixPriorFactor   reference preintegration/imu_mix_prior_factor.h:30-80
// Same constructors, block sizes, residual order and Jacobian layout (row-major, global size) as the reference; pinned against the
// reference's own headers (oracle/_ref/libref_nav.so, tests/golden/nav_ref_golden.npz).
#pragma once
#include <cstring>

#include "earth.h"
#include "factors.h"

namespace icg {

struct GNSS { // common/types.h:35-43
    double time{0};
    Vector3d blh, std;
    bool isyawvalid{false};
    double yaw{0};
};

class GnssFactor : public ceres::SizedCostFunction<3, 7> {
public:
    GnssFactor(GNSS gnss, Vector3d lever) : gnss_(gnss), lever_(lever) {}
    void updateGnssState(const GNSS &gnss) { gnss_ = gnss; }
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        const double *x = parameters[0];
        Matrix3d R      = Rotation::quaternion2matrix(Quaterniond{x[3], x[4], x[5], x[6]});
        Vector3d rl     = R * lever_;
        for (int k = 0; k < 3; k++) residuals[k] = (1.0 / gnss_.std[k]) * (x[k] + rl[k] - gnss_.blh[k]);
        if (jacobians && jacobians[0]) {
            double *J = jacobians[0];
            memset(J, 0, sizeof(double) * 21);
            const double S[3][3] = {{0, -lever_[2], lever_[1]}, {lever_[2], 0, -lever_[0]}, {-lever_[1], lever_[0], 0}};
            for (int i = 0; i < 3; i++) {
                const double w = 1.0 / gnss_.std[i];
                J[i * 7 + i]   = w * 1.0;
                for (int j = 0; j < 3; j++) J[i * 7 + 3 + j] = w * -(R(i, 0) * S[0][j] + R(i, 1) * S[1][j] + R(i, 2) * S[2][j]);
            }
        }
        return true;
    }

private:
    GNSS gnss_;
    Vector3d lever_;
};

class ImuErrorFactor : public ceres::SizedCostFunction<6, 9> {
public:
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        for (int k = 0; k < 3; k++) {
            residuals[k + 0] = parameters[0][k + 3] / IMU_GRY_BIAS_STD;
            residuals[k + 3] = parameters[0][k + 6] / IMU_ACC_BIAS_STD;
        }
        if (jacobians && jacobians[0]) {
            memset(jacobians[0], 0, sizeof(double) * 54);
            for (int k = 0; k < 3; k++) {
                jacobians[0][(k + 0) * 9 + k + 3] = 1.0 / IMU_GRY_BIAS_STD;
                jacobians[0][(k + 3) * 9 + k + 6] = 1.0 / IMU_ACC_BIAS_STD;
            }
        }
        return true;
    }

private:
    static constexpr double IMU_GRY_BIAS_STD = 7200 / 3600.0 * M_PI / 180.0; // 7200 deg / hr
    static constexpr double IMU_ACC_BIAS_STD = 2.0e4 * 1.0e-5;               // 20000 mGal
};

class ImuPosePriorFactor : public ceres::SizedCostFunction<6, 7> {
public:
    ImuPosePriorFactor(const double *pose, const double *std) {
        memcpy(pose_, pose, sizeof(double) * 7);
        for (int k = 0; k < 6; k++) w_[k] = 1.0 / std[k];
    }
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        const double *x = parameters[0];
        for (int k = 0; k < 3; k++) residuals[k] = (x[k] - pose_[k]);
        // d = q^-1 * q_p (Eigen inverse(): conjugate / squared norm)
        const double n2 = x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6];
        const double ax = -x[3] / n2, ay = -x[4] / n2, az = -x[5] / n2, aw = x[6] / n2;
        const double bx = pose_[3], by = pose_[4], bz = pose_[5], bw = pose_[6];
        const double dx = aw * bx + ax * bw + ay * bz - az * by;
        const double dy = aw * by + ay * bw + az * bx - ax * bz;
        const double dz = aw * bz + az * bw + ax * by - ay * bx;
        const double dw = aw * bw - ax * bx - ay * by - az * bz;
        residuals[3] = 2 * dx, residuals[4] = 2 * dy, residuals[5] = 2 * dz;
        for (int k = 0; k < 6; k++) residuals[k] = w_[k] * residuals[k];
        if (jacobians && jacobians[0]) {
            double *J = jacobians[0];
            memset(J, 0, sizeof(double) * 42);
            for (int k = 0; k < 3; k++) J[k * 7 + k] = w_[k] * 1.0;
            // -quaternionright(d).bottomRightCorner<3,3>() = -(dw I - skew(d.vec))
            const double B[3][3] = {{dw, dz, -dy}, {-dz, dw, dx}, {dy, -dx, dw}};
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) J[(3 + i) * 7 + 3 + j] = w_[3 + i] * -B[i][j];
        }
        return true;
    }

private:
    double pose_[7], w_[6];
};

class ImuMixPriorFactor : public ceres::SizedCostFunction<9, 9> {
public:
    ImuMixPriorFactor(const double *mix, const double *mix_std) {
        memcpy(mix_, mix, sizeof(double) * 9);
        memcpy(mix_std_, mix_std, sizeof(double) * 9);
    }
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        for (int k = 0; k < 9; k++) residuals[k] = (parameters[0][k] - mix_[k]) / mix_std_[k];
        if (jacobians && jacobians[0]) {
            memset(jacobians[0], 0, sizeof(double) * 81);
            for (int k = 0; k < 9; k++) jacobians[0][k * 9 + k] = 1.0 / mix_std_[k];
        }
        return true;
    }

private:
    double mix_[9], mix_std_[9];
};

} // namespace icg
