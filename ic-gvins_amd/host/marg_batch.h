// The marginalizations of many camera streams in one pass — the multi-stream form of MarginalizationInfo::marginalization()
// (reference factors/marginalization_info.h:73-101, one call per stream and keyframe from GVINS::gvinsMarginalization,
// ic_gvins.cc:1412-1693), SURVEY.md §8 rows M2 / M3.
//
// One MarginalizationInfo per stream costs three device round trips per marginalization (evaluation with the robust correction,
// assembly + landmark elimination, the h_ll read-back) for ~1 k factors: launch- and wait-bound, the device idles.
// MarginalizationBatch keeps every stream's MarginalizationInfo exactly as the caller built it and runs the windows in lock-step:
//
//   1  ONE evaluation launch for the reprojection factors of all windows (icg_reproj_eval_windows, Huber correction on the device)
//   2  per window, on the host threads: M1 bookkeeping (updateParameterBlocksIndex, marginalization_info.h:232-253), the host-evaluated
//      factors (prior, preintegration, GNSS: ResidualBlockInfo::Evaluate) and the layout of the compact camera system
//   3  ONE assembly + landmark-elimination launch sequence (icg_reproj_schur_windows, no damping) and ONE read-back of the landmark
//      diagonals (icg_reproj_landmark_diag_windows) for all windows
//   4  per window, on the host threads: conditioning guard, the reference's eigen / 1e-8-floor procedure on the few pose / mix columns
//      (M3, :170-192) and the linearization (:153-167)
//
// A window whose marginalized set does not have the structure of gvinsMarginalization, or that fails the conditioning guard, takes the
// reference's dense M2 + M3 on its own (a one-window context created on first use) — the same rule MarginalizationInfo applies alone.
// Results per window are those of MarginalizationInfo::marginalization() on a ReprojectionBatch of its own (tests: Hp, bp, J0, e0).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "factors.h"
#include "host_pool.h"

namespace icg {

class MarginalizationBatch {
public:
    explicit MarginalizationBatch(int device = 0, double huber_delta = 1.0, int host_threads = 0 /* 0 = hardware concurrency, at most 16 */);
    ~MarginalizationBatch();
    MarginalizationBatch(const MarginalizationBatch &)            = delete;
    MarginalizationBatch &operator=(const MarginalizationBatch &) = delete;

    // a stream's marginalization: `info` is filled by the caller as for marginalization() (updateParamtersIds, addResidualBlockInfo for
    // every factor incl. the reprojection ones); the batch becomes the info's device factor set
    int addWindow(const std::shared_ptr<MarginalizationInfo> &info);
    // the reprojection factors of window w with the five blocks of their ResidualBlockInfo (instead of ReprojectionBatch::add);
    // all factors of a window share its extrinsic and td blocks, a pose block belongs to one window
    void addReprojectionFactor(int w, const ReprojectionFactor *factor, double *pose_i, double *pose_j, double *extrinsic, double *invdepth, double *td);
    int numWindows() const { return (int) windows_.size(); }
    int numReprojectionFactors(int w) const { return (int) windows_.at((size_t) w)->obs.size() / 15; }
    // drops every window, keeps the device context (a batch object lives as long as the group of streams it serves)
    void clear();

    // marginalization() of every window.  ok[w] = what the window's own call would have returned (false: info->isValid() is false, as
    // there); the return value is false only when a device call all windows share failed (error()).  A window that failed by itself
    // (its dense fallback could not run) has ok[w] == 0 and its message in windowError(); the other windows are unaffected.
    bool marginalize(std::vector<char> *ok);

    // diagnostics of the last marginalize(): windows that took the landmark-eliminated device path / the dense path; wall time of the
    // four phases above [ms]
    int structuredWindows() const { return n_structured_; }
    int denseWindows() const { return n_dense_; }
    const double *lastPhaseMs() const { return phase_ms_; }
    const std::string &error() const { return error_; }
    const std::string &windowError() const { return window_error_; } // first per-window failure of the last marginalize()

private:
    // one window's view of the batch: what MarginalizationInfo asks of its device factors
    struct Slice : public DeviceFactorSet {
        MarginalizationBatch *owner{nullptr};
        std::shared_ptr<MarginalizationInfo> info;
        std::unordered_set<const ReprojectionFactor *> members;
        std::vector<double> obs; // 15 per factor, factor-major
        std::vector<int32_t> idx_i, idx_j, idx_lm; // window-local pose / landmark indices
        std::vector<double *> poses, landmarks;   // first-seen order
        std::unordered_map<const double *, int> pose_index, lm_index;
        double *ext{nullptr}, *td{nullptr};
        int fac_begin{0}, pose_begin{0}, lm_begin{0};
        bool evaluated{false};
        std::string err;
        bool owns(const ReprojectionFactor *factor) const override { return members.count(factor) != 0; }
        int size() const override { return (int) (obs.size() / 15); }
        const std::vector<double *> &landmarkBlocks() const override { return landmarks; }
        bool evaluateCorrected(double huber_delta) override;
        bool accumulateNormal(const std::unordered_map<const double *, int> &column_of, int local_size, double *H0, double *b0) override;
        bool accumulateLandmarkEliminated(const std::unordered_map<const double *, int> &, int, double *, double *, double *) override;
        const std::string &error() const override { return err; }
    };
    bool layout();
    bool denseNormalOfWindow(Slice &W, const std::unordered_map<const double *, int> &column_of, int local_size, double *H0, double *b0);
    template <typename F> void forEachWindow(size_t n, F &&fn);

    icg_ctx *ctx_{nullptr};
    icg_ctx *dense_ctx_{nullptr}; // one-window context of the dense path (created on first use)
    std::mutex dense_mutex_;
    int device_{0};
    double huber_{1.0};
    int host_threads_{1};
    std::unique_ptr<HostPool> pool_;
    std::vector<std::unique_ptr<Slice>> windows_;
    bool laid_out_{false};
    std::vector<std::shared_ptr<ResidualBlockInfo>> retired_; // factor records of marginalized windows, freed by clear() / the destructor
    int n_factors_{0}, n_poses_{0}, n_lm_{0};
    int n_structured_{0}, n_dense_{0};
    double phase_ms_[4]{0, 0, 0, 0};
    std::string error_, window_error_;
};

} // namespace icg
