// ORACLE / TEST INFRASTRUCTURE ONLY — the slice of the Ceres interface the reference factor headers are written against
// (ceres::CostFunction / SizedCostFunction), so reference headers compile unmodified into oracle/_ref/.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric> // real Ceres / Eigen headers pull these in; ic_gvins.{h,cc} rely on it
#include <queue>
#include <vector>

#include <Eigen/Geometry> // real Ceres headers pull Eigen in; residual_block_info.h relies on that

namespace ceres {
class CostFunction {
public:
    virtual ~CostFunction() = default;
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    const std::vector<int32_t> &parameter_block_sizes() const { return sizes_; }
    int num_residuals() const { return nres_; }
protected:
    std::vector<int32_t> *mutable_parameter_block_sizes() { return &sizes_; }
    void set_num_residuals(int n) { nres_ = n; }
private:
    std::vector<int32_t> sizes_;
    int nres_ = 0;
};
template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction {
public:
    SizedCostFunction() {
        set_num_residuals(kNumResiduals);
        *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
    }
};
// robust kernels (ceres/loss_function.h): rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s)
class LossFunction {
public:
    virtual ~LossFunction() = default;
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunction {
public:
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    void Evaluate(double s, double rho[3]) const override {
        if (s > b_) {
            const double r = std::sqrt(s);
            rho[0]         = 2.0 * a_ * r - b_;
            rho[1]         = std::max(std::numeric_limits<double>::min(), a_ / r);
            rho[2]         = -rho[1] / (2.0 * s);
        } else {
            rho[0] = s;
            rho[1] = 1.0;
            rho[2] = 0.0;
        }
    }
    double delta() const { return a_; }
private:
    const double a_, b_;
};
// ceres/evaluation_callback.h (Ceres >= 2.0): called once per evaluation point before the residual blocks are evaluated
class EvaluationCallback {
public:
    virtual ~EvaluationCallback() = default;
    virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};
} // namespace ceres
#include "problem_shim.h"
