// Window optimizer of the MI355X host layer — SURVEY.md §8 row f1.
//
// The reference solves its sliding window with Ceres (GVINS::gvinsOptimization, ic_gvins.cc:1130-1239: LEVENBERG_MARQUARDT +
// DENSE_SCHUR, two solves with a chi-square culling pass in between, :1269-1297).  Ceres is not available here, and the design
// point of the device back-end is to keep the visual residuals/Jacobians resident instead of copying 48 doubles per factor per
// evaluation through the CostFunction callback.  WindowSolver is a small Levenberg-Marquardt driver with the same problem-building
// surface (parameter blocks with the pose manifold, residual blocks with loss functions, constant blocks, removal of residual
// blocks, per-block evaluation) in which
//   * the reprojection factors live in a ReprojectionBatch: one batched evaluation per linearization point, normal equations and
//     the elimination of the 1x1 inverse-depth blocks on the device (icg_reproj_schur), back-substitution on the device
//     (icg_reproj_backsub) — per iteration only the reduced camera system (P x P, P ~ 160) crosses PCIe;
//   * every other factor (preintegration, marginalization prior, GNSS, pose priors: tens of blocks per solve) is an ordinary
//     ceres::CostFunction evaluated on the host and added to the reduced system.
// Step control follows the published Ceres trust-region loop (LM diagonal clamp(diag(J^T J), 1e-6, 1e32)/radius, radius update
// radius / max(1/3, 1 - (2 rho - 1)^3) on success, halving with a doubling factor on failure, function / gradient / parameter
// tolerances 1e-6 / 1e-10 / 1e-8) without Jacobi column scaling; iterates therefore agree with Ceres only to the extent LM
// implementations do — "parity unpinned" for this row, stated in DESIGN.md.  No CPU fallback for the visual part: a device error
// fails the solve.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "factors.h"
#include "host_pool.h"
#include "solver_detail.h"

namespace icg {

class WindowSolver {
public:
    struct Options {
        int max_num_iterations{50};
        double initial_trust_region_radius{1e4}, max_trust_region_radius{1e16}, min_trust_region_radius{1e-32};
        double min_relative_decrease{1e-3}, min_lm_diagonal{1e-6}, max_lm_diagonal{1e32};
        double function_tolerance{1e-6}, gradient_tolerance{1e-10}, parameter_tolerance{1e-8};
    };
    struct Summary {
        double initial_cost{0}, final_cost{0};
        int num_successful_steps{0}, num_unsuccessful_steps{0};
        std::string termination;
        std::string BriefReport() const;
    };
    typedef int ResidualBlockId;

    // visual: the reprojection factors (may be null); huber_delta: their loss (ceres::HuberLoss(delta), 0 = none)
    explicit WindowSolver(ReprojectionBatch *visual = nullptr, double huber_delta = 0.0);

    // problem.AddParameterBlock(values, size[, new PoseParameterization()]) — pose blocks are [p3, q4 xyzw], tangent size 6
    void addParameterBlock(double *values, int size, bool pose_manifold = false);
    void setParameterBlockConstant(double *values);
    // problem.AddResidualBlock(cost, loss, blocks) for the host-evaluated factors
    ResidualBlockId addResidualBlock(std::shared_ptr<ceres::CostFunction> cost, std::shared_ptr<ceres::LossFunction> loss,
                                     const std::vector<double *> &blocks);
    void removeResidualBlock(ResidualBlockId id);
    // problem.EvaluateResidualBlock(id, apply_loss_function, &cost, nullptr, nullptr)
    bool evaluateResidualBlock(ResidualBlockId id, bool apply_loss_function, double *cost) const;

    bool solve(const Options &options, Summary *summary);
    // Process-wide: evaluate the host factors of a linearization / trial point on ONE helper thread while the calling thread drives the
    // device calls for the visual factors (the two halves are independent; results are identical).  For a single camera stream, where the
    // window solve is a latency chain; leave it off when every core already runs an estimator of its own.
    static void setHostFactorOverlap(bool on);

    // removeReprojectionFactorsByChi2 (ic_gvins.cc:1269-1297): raw cost of every active visual factor at the current state,
    // factors with 2 cost > chi2 are deactivated for the following solves; returns how many
    int removeReprojectionFactorsByChi2(double chi2);
    int numActiveReprojectionFactors() const;
    const std::vector<uint8_t> &activeReprojectionFactors() const { return active_; }
    const std::string &error() const { return error_; }

private:
    typedef solver_detail::Block Block;
    typedef solver_detail::Residual Residual;
    bool layout();
    bool linearize(double damp, bool reassemble, const Options &o, std::vector<double> &S, std::vector<double> &s, std::vector<double> &diag,
                   double *cost);
    bool evaluateCost(double *cost);
    void applyStep(const std::vector<double> &delta_c, const std::vector<double> &delta_l);
    void backup();
    void restore();

    ReprojectionBatch *visual_;
    double huber_;
    std::vector<Block> blocks_;
    std::unordered_map<const double *, int> block_of_;
    std::vector<Residual> residuals_;
    std::vector<uint8_t> active_;
    std::vector<int32_t> col_pose_;
    int col_ext_{-1}, col_td_{-1}, P_{0};
    std::vector<std::vector<double>> saved_;
    std::vector<double> host_S_, host_s_, host_diag_; // host factors' part of the current linearization (kept for re-damping)
    std::string error_;
};

} // namespace icg
