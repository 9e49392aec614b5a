// The interface of Ceres Solver that the reference's factors are written against (ceres::CostFunction,
// SizedCostFunction, LossFunction / HuberLoss, EvaluationCallback; Ceres 2.0/2.1 — README.md:46).
// When the real library is installed its headers are used and the factors below derive from the real base classes, so
// they can be handed to ceres::Problem::AddResidualBlock unchanged.  Ceres is not available in this environment, so a
// minimal declaration of the same virtual interface (same names, same signatures) stands in for it.
#pragma once

#if __has_include(<ceres/ceres.h>)
#include <ceres/ceres.h>
#else
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace ceres {

class CostFunction {
public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() = default;
    // parameters[i]: block i; residuals: num_residuals(); jacobians may be null, jacobians[i] may be null;
    // jacobians[i] is row-major num_residuals x parameter_block_sizes()[i].  false = evaluation failed.
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    const std::vector<int32_t> &parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }

protected:
    std::vector<int32_t> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }

private:
    std::vector<int32_t> parameter_block_sizes_;
    int num_residuals_;
};

template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction {
public:
    SizedCostFunction() {
        set_num_residuals(kNumResiduals);
        *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
    }
};

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

class EvaluationCallback {
public:
    virtual ~EvaluationCallback() = default;
    virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

} // namespace ceres
#endif
