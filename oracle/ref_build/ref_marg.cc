// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own marginalization pipeline (ResidualBlockInfo,
// MarginalizationInfo, MarginalizationFactor and ReprojectionFactor: /root/reference/ic_gvins/ic_gvins/factors/*.h, compiled
// unmodified from where they lie) behind the SAME C entry point the product's host layer exposes for its tests
// (icgh_backend_marginalize, ic-gvins_amd/host/capi.cc), so tests/backend_utils.py can drive either.
// Linear algebra comes from the Eigen-interface shim in shim/ (NOT real Eigen — stated in DESIGN.md).
// The reference keeps H0/Hp/bp private; Hp and bp are therefore returned as J0^T J0 and -J0^T e0 (identical up to the
// eigenvalues <= 1e-8 the reference truncates in linearization()).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>
using std::vector;

#include "factors/marginalization_factor.h"
#include "factors/marginalization_info.h"
#include "factors/reprojection_factor.h"
#include "factors/residual_block_info.h"

namespace {
// the generic host factor of capi.cc's scenario: residual = w * [p - p0 ; 2 vec(q0^-1 q)] on one pose block
class PosePriorFactor : public ceres::SizedCostFunction<6, 7> {
public:
    PosePriorFactor(const double *pose0, double weight) : w_(weight) { memcpy(x0_, pose0, sizeof x0_); }
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        const double *x = parameters[0];
        const double n2 = x0_[3] * x0_[3] + x0_[4] * x0_[4] + x0_[5] * x0_[5] + x0_[6] * x0_[6];
        const double ax = -x0_[3] / n2, ay = -x0_[4] / n2, az = -x0_[5] / n2, aw = x0_[6] / n2;
        const double bx = x[3], by = x[4], bz = x[5], bw = x[6];
        const double dq[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                              aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz};
        for (int k = 0; k < 3; k++) {
            residuals[k]     = w_ * (x[k] - x0_[k]);
            residuals[3 + k] = w_ * 2.0 * dq[k];
        }
        if (jacobians && jacobians[0]) {
            memset(jacobians[0], 0, sizeof(double) * 42);
            for (int k = 0; k < 3; k++) {
                jacobians[0][k * 7 + k]           = w_;
                jacobians[0][(3 + k) * 7 + 3 + k] = w_ * dq[3]; // same scaffolding factor as ic-gvins_amd/host/capi.cc
            }
        }
        return true;
    }

private:
    double x0_[7];
    double w_;
};
} // namespace

extern "C" int icgh_backend_marginalize(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j,
                                        const int32_t *idx_lm, int n_poses, const double *poses, const double *ext, int n_lm,
                                        const double *invdepth, double td, double huber_delta, double prior_weight, int, int,
                                        int32_t *sizes, int64_t *rem_ids, int32_t *rem_index, int32_t *rem_size, int32_t *n_rem,
                                        double *Hp, double *bp, double *J0, double *e0, const double *x_eval, double *marg_res,
                                        char *err, int errlen) {
    vector<double> P(poses, poses + 7 * (size_t) n_poses), E(ext, ext + 7), D(invdepth, invdepth + n_lm);
    double TD = td;
    std::unordered_map<long, long> ids;
    std::unordered_map<long, double *> address;
    auto reg = [&](double *p, long id) {
        ids[reinterpret_cast<long>(p)] = id;
        address[id]                     = p;
    };
    for (int k = 0; k < n_poses; k++) reg(&P[7 * (size_t) k], k);
    for (int l = 0; l < n_lm; l++) reg(&D[(size_t) l], 100000 + l);
    reg(E.data(), 900000);
    reg(&TD, 900001);

    auto info = std::make_shared<MarginalizationInfo>();
    info->updateParamtersIds(ids);
    std::shared_ptr<ceres::LossFunction> loss;
    if (huber_delta > 0) loss = std::make_shared<ceres::HuberLoss>(huber_delta);
    for (int k = 0; k < n; k++) {
        auto o = [&](int c) { return obs_soa[(size_t) c * n + k]; };
        auto f = std::make_shared<ReprojectionFactor>(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                      Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14));
        double *pi = &P[7 * (size_t) idx_i[k]], *pj = &P[7 * (size_t) idx_j[k]], *lm = &D[(size_t) idx_lm[k]];
        info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(f, loss, vector<double *>{pi, pj, E.data(), lm, &TD}, vector<int>{0, 3}));
    }
    vector<double> pose0_prior(P.begin(), P.begin() + 7);
    pose0_prior[0] += 0.01;
    info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(std::make_shared<PosePriorFactor>(pose0_prior.data(), prior_weight),
                                                                   nullptr, vector<double *>{&P[0]}, vector<int>{0}));
    if (!info->marginalization()) {
        if (err && errlen > 0) snprintf(err, (size_t) errlen, "reference marginalization() returned false");
        return -2;
    }
    auto blocks = info->getParamterBlocks(address);
    sizes[0]    = info->marginalizedSize();
    sizes[1]    = info->remainedSize();
    *n_rem      = (int32_t) blocks.size();
    for (size_t b = 0; b < blocks.size(); b++) {
        rem_ids[b]   = ids[reinterpret_cast<long>(blocks[b])];
        rem_index[b] = info->remainedBlockIndex()[b];
        rem_size[b]  = info->remainedBlockSize()[b];
    }
    const int r = info->remainedSize();
    const Eigen::MatrixXd &Jl = info->linearizedJacobians();
    const Eigen::VectorXd &el = info->linearizedResiduals();
    for (int i = 0; i < r; i++) {
        e0[i] = el(i);
        for (int j = 0; j < r; j++) J0[(size_t) i * r + j] = Jl.get(i, j); // row-major like the product's accessor
    }
    Eigen::MatrixXd H = Jl.transpose() * Jl;
    Eigen::MatrixXd g = Jl.transpose() * el;
    for (int i = 0; i < r; i++) {
        bp[i] = -g.get(i, 0);
        for (int j = 0; j < r; j++) Hp[(size_t) i * r + j] = H.get(i, j);
    }
    if (x_eval && marg_res) {
        MarginalizationFactor factor(info);
        vector<const double *> params;
        size_t off = 0;
        for (size_t b = 0; b < blocks.size(); b++) {
            params.push_back(x_eval + off);
            off += (size_t) rem_size[b];
        }
        if (!factor.Evaluate(params.data(), marg_res, nullptr)) return -3;
    }
    return 0;
}
