// ORACLE / TEST INFRASTRUCTURE ONLY — the slice of the Ceres interface the reference factor headers are written against
// (ceres::CostFunction / SizedCostFunction), so reference headers compile unmodified into oracle/_ref/.
#pragma once
#include <cstdint>
#include <vector>
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
} // namespace ceres
