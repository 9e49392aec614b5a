// Many sliding windows optimized together — the multi-stream form of WindowSolver (solver_hip.h), SURVEY.md §8 row f1.
//
// One WindowSolver per camera stream is bounded by the runtime's rate of small launches and copies (~100 per window and solve).
// WindowSolverBatch advances the LM iterations of W independent windows in lock-step on ONE device context: per iteration one
// evaluation launch for the reprojection factors of all windows, one assembly + elimination launch sequence, one back-substitution
// launch; each window keeps its own trust-region radius, accepts or rejects its own step and stops by its own tolerances, exactly as
// a WindowSolver of its own would (tests: identical step sequences and optima).  Host-evaluated factors (preintegration,
// marginalization prior, priors) and the P x P reduced solves stay per window on the host.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_pool.h"
#include "solver_hip.h"

namespace icg {

class WindowSolverBatch {
public:
    typedef WindowSolver::Options Options;
    typedef WindowSolver::Summary Summary;
    // Widest reduced system of the device path: the assembly (csrc/reproj.hip k_asm_*) keeps no per-window tile in LDS any more — rounds 2-5
    // held the camera block there (82, then 138 columns) — and the reduction kernel stages 32 landmark rows of 4 ceil(P/4) doubles (64.5 KB at
    // 512).  A window that can exceed it is the caller's to solve on the host.
    static constexpr int kMaxCameraColumns = 512;

    // host_threads: the per-window host phases (host factors, reduced solves, cost bookkeeping) are spread over this many threads
    explicit WindowSolverBatch(int device = 0, double huber_delta = 1.0, int host_threads = 0 /* 0 = hardware concurrency, at most 16 */);
    ~WindowSolverBatch();
    WindowSolverBatch(const WindowSolverBatch &) = delete;
    WindowSolverBatch &operator=(const WindowSolverBatch &) = delete;

    int addWindow();
    int numWindows() const { return (int) windows_.size(); }
    // drops every window but keeps the device context (a batch object that is re-used for one set of windows after another)
    void clear();
    void addParameterBlock(int w, double *values, int size, bool pose_manifold = false);
    void setParameterBlockConstant(int w, double *values);
    int addResidualBlock(int w, std::shared_ptr<ceres::CostFunction> cost, std::shared_ptr<ceres::LossFunction> loss, const std::vector<double *> &blocks);
    void removeResidualBlock(int w, int id);
    // problem.EvaluateResidualBlock(id, apply_loss_function, &cost, nullptr, nullptr) for a host factor of window w
    bool evaluateResidualBlock(int w, int id, bool apply_loss_function, double *cost) const;
    int numReprojectionFactors(int w) const { return (int) windows_.at((size_t) w).visual.size(); }
    // a reprojection factor of window w with the five blocks it would get in AddResidualBlock (only its observation constants are
    // read from `factor`); all factors of a window share its extrinsic and td blocks
    void addReprojectionFactor(int w, const ReprojectionFactor *factor, double *pose_i, double *pose_j, double *extrinsic, double *invdepth, double *td);

    // uploads the factor set and the window partition (done by the first solve() otherwise): problem setup, not part of a solve
    bool prepare();
    bool solve(const Options &options, std::vector<Summary> *summaries);
    // removeReprojectionFactorsByChi2 (ic_gvins.cc:1269-1297) for every window; returns the number removed per window
    std::vector<int> removeReprojectionFactorsByChi2(double chi2);
    const std::string &error() const { return error_; }

private:
    typedef solver_detail::Block Block;
    typedef solver_detail::Residual Residual;
    struct VisualFactor {
        double obs[15];
        double *pose_i, *pose_j, *invdepth;
    };
    struct Window {
        std::vector<Block> blocks;
        std::unordered_map<const double *, int> block_of;
        std::vector<Residual> residuals;
        std::vector<VisualFactor> visual;
        double *ext{nullptr}, *td{nullptr};
        std::vector<double *> poses, landmarks; // first-seen order of the visual factors
        std::unordered_map<const double *, int> pose_index, lm_index;
        int P{0};
        int fac_begin{0}, lm_begin{0}, pose_begin{0};
        std::vector<std::vector<double>> saved;
        std::vector<double> host_S, host_s, host_diag;
    };
    bool finalize();
    bool layout();
    void gather(std::vector<double> &poses, std::vector<double> &ext, std::vector<double> &inv, std::vector<double> &td);

    // fn(w) for every window on the persistent helper threads (created on first use: a batch of one or two windows never needs them)
    template <typename F> void forEachWindow(size_t n, F &&fn);

    icg_ctx *ctx_{nullptr};
    double huber_;
    int host_threads_;
    std::unique_ptr<HostPool> pool_;
    std::unique_ptr<SideThread> side_; // runs the device call of a phase beside the pool's host half
    std::vector<Window> windows_;
    std::vector<uint8_t> active_;
    std::vector<int32_t> col_pose_, col_ext_, col_td_;
    int P_{0}, n_factors_{0}, n_poses_{0}, n_lm_{0};
    bool finalized_{false};
    std::string error_;
};

} // namespace icg
