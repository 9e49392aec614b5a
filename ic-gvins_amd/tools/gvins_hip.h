// GVINS — the estimator that calls the hot path from both sides ("replay harness", i.e. the caller the
// reference's ROS shell feeds): IMU / GNSS / image ingestion, GNSS-aided initialization, INS mechanization window, INS-aided
// tracking, sliding-window optimization (GNSS + IMU preintegration + reprojection + marginalization prior), chi-square culling,
// marginalization, result files.  Mirrors the reference's `GVINS` class (ic_gvins.h:44-270, ic_gvins.cc) with the same public
// surface: GVINS(configfile, outputpath, drawer), addNewImu / addNewGnss / addNewFrame, setFinished, isRunning, gvinsState.
//
// What is different, deliberately:
//   * The reference runs three free-running threads (fusion / tracking / optimization) that hand work over with try_lock and
//     condition variables, so its output depends on thread timing.  Here the same three loop bodies (runFusion :237-393,
//     runTracking :479-552, runOptimization :395-477) are executed as ONE deterministic event loop inside addNewImu(): after
//     every IMU epoch the tracker consumes every frame the INS has passed, then a signalled optimization runs to completion.
//     That is the schedule the reference follows when tracker and optimizer finish within one IMU period; results are reproducible.
//   * All floating-point work goes through the device paths of this library: tracking (icg::Tracking), INS mechanization of the
//     pending epochs as one series launch per event (MISC::insMechanizationBatch, lazily: the state machine only looks at IMU
//     *times*), pose prior (MISC::getCameraPoseFromInsWindowBatch), preintegration (Preintegration::integrateBatch: all dirty
//     intervals per launch), window solve with device-side landmark elimination (WindowSolver replaces Ceres LM + DENSE_SCHUR),
//     culling / statistics (WindowCulling), marginalization assembly (MarginalizationInfo::setReprojectionBatch).  No CPU fallback.
//   * Function-local statics of the reference (initial attitude / gyro bias of gvinsInitialization :606-608, iteration split of
//     gvinsOptimization :1131-1132) are members, so several estimators can live in one process (one per camera stream).
// Parity: the solver is unpinned (Ceres is an absent dependency); the small factors and Earth / attitude helpers are
// pinned against the reference's headers (tests/golden/nav_ref_golden.npz); behaviour is checked end to end on a synthetic
// GNSS + IMU + camera sequence with known truth (tests/gvins_checks.py).
#pragma once
#include <chrono>
#include <deque>
#include <functional>
#include <memory>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

#include "culling_hip.h"
#include "factors.h"
#include "misc_hip.h"
#include "nav_factors.h"
#include "solver_batch_hip.h"
#include "solver_hip.h"
#include "tracking.h"

namespace icg {

struct IntegrationStateData { // preintegration/integration_state.h:54-60
    double time{0};
    double pose[7]{0, 0, 0, 0, 0, 0, 1};
    double mix[18]{0};
};

// What the window problem of the estimator is built on: a WindowSolver of its own (one stream), or one window of a WindowSolverBatch
// shared with the estimators of other camera streams (GvinsLockstep, replay.h).  The surface is ceres::Problem's.
class WindowProblem {
public:
    virtual ~WindowProblem() = default;
    virtual void addParameterBlock(double *values, int size, bool pose_manifold = false)                                     = 0;
    virtual void setParameterBlockConstant(double *values)                                                                   = 0;
    virtual int addResidualBlock(std::shared_ptr<ceres::CostFunction> cost, std::shared_ptr<ceres::LossFunction> loss,
                                 const std::vector<double *> &blocks)                                                        = 0;
    virtual void removeResidualBlock(int id)                                                                                 = 0;
    virtual bool evaluateResidualBlock(int id, bool apply_loss_function, double *cost)                                       = 0;
    virtual void addReprojectionFactor(const ReprojectionFactor *factor, double *pose_i, double *pose_j, double *extrinsic, double *invdepth,
                                       double *td)                                                                           = 0;
};
class SingleWindowProblem : public WindowProblem { // the visual factors already sit in the ReprojectionBatch the solver was built on
public:
    explicit SingleWindowProblem(WindowSolver &solver) : s_(solver) {}
    void addParameterBlock(double *v, int size, bool pose) override { s_.addParameterBlock(v, size, pose); }
    void setParameterBlockConstant(double *v) override { s_.setParameterBlockConstant(v); }
    int addResidualBlock(std::shared_ptr<ceres::CostFunction> c, std::shared_ptr<ceres::LossFunction> l, const std::vector<double *> &b) override {
        return s_.addResidualBlock(std::move(c), std::move(l), b);
    }
    void removeResidualBlock(int id) override { s_.removeResidualBlock(id); }
    bool evaluateResidualBlock(int id, bool apply, double *cost) override { return s_.evaluateResidualBlock(id, apply, cost); }
    void addReprojectionFactor(const ReprojectionFactor *, double *, double *, double *, double *, double *) override {}

private:
    WindowSolver &s_;
};
class BatchWindowProblem : public WindowProblem {
public:
    BatchWindowProblem(WindowSolverBatch &batch, int window) : b_(batch), w_(window) {}
    int window() const { return w_; }
    void addParameterBlock(double *v, int size, bool pose) override { b_.addParameterBlock(w_, v, size, pose); }
    void setParameterBlockConstant(double *v) override { b_.setParameterBlockConstant(w_, v); }
    int addResidualBlock(std::shared_ptr<ceres::CostFunction> c, std::shared_ptr<ceres::LossFunction> l, const std::vector<double *> &b) override {
        return b_.addResidualBlock(w_, std::move(c), std::move(l), b);
    }
    void removeResidualBlock(int id) override { b_.removeResidualBlock(w_, id); }
    bool evaluateResidualBlock(int id, bool apply, double *cost) override { return b_.evaluateResidualBlock(w_, id, apply, cost); }
    void addReprojectionFactor(const ReprojectionFactor *f, double *pi, double *pj, double *ext, double *inv, double *td) override {
        b_.addReprojectionFactor(w_, f, pi, pj, ext, inv, td);
    }

private:
    WindowSolverBatch &b_;
    int w_;
};

class GVINS {
public:
    enum GVINSState { // ic_gvins.h:47-55
        GVINS_ERROR                 = -1,
        GVINS_INITIALIZING          = 0,
        GVINS_INITIALIZING_INS      = 1,
        GVINS_INITIALIZING_VIO      = 2,
        GVINS_TRACKING_INITIALIZING = 3,
        GVINS_TRACKING_NORMAL       = 4,
        GVINS_TRACKING_LOST         = 5,
    };
    typedef std::shared_ptr<GVINS> Ptr;

    GVINS() = delete;
    explicit GVINS(const std::string &configfile, const std::string &outputpath, Drawer::Ptr drawer = nullptr, int device = 0);
    ~GVINS();

    bool addNewImu(const IMU &imu);
    bool addNewGnss(const GNSS &gnss);
    bool addNewFrame(const Frame::Ptr &frame);
    void setFinished();
    // completion waits of this estimator's device contexts (tracker, INS / preintegration / culling, the two reprojection batches):
    // ICG_WAIT_SPIN (default: lowest latency for one estimator) or ICG_WAIT_POLL + sleep (many estimators on few host cores)
    void setWaitMode(int icg_wait_mode, int sleep_us);
    bool isRunning() const { return !isfinished_; }
    GVINSState gvinsState() const { return gvinsstate_; }

    // ---- lock-step interface: the window solve of several estimators shared through one WindowSolverBatch (replay.h) ----
    // With deferred solves on, an optimization that addNewImu signals in the tracking states is not run inside addNewImu; the caller asks
    // windowSolvePending() after the call and drives the phases: n = beginWindowSolve(); populateWindow(problem, n); [solve 1];
    // betweenWindowSolves(problem); [chi-square removal of reprojection factors]; [solve 2]; finishWindowSolve(...); afterWindowSolve().
    // solveWindowAlone() runs the same phases on a WindowSolver of this estimator (what addNewImu does without deferral).
    void setDeferredWindowSolves(bool on) { deferred_window_solves_ = on; }
    bool windowSolvePending() const { return window_solve_pending_; }
    int beginWindowSolve();
    // upper bound of the free camera columns of this estimator's window in a batched solve: 6 per keyframe in the map, extrinsic 6, td 1
    int windowCameraColumnsBound() const;
    void populateWindow(WindowProblem &problem, int n_visual);
    void betweenWindowSolves(WindowProblem &problem);
    void finishWindowSolve(const WindowSolver::Summary &first, const WindowSolver::Summary &second, double first_ms, double second_ms, int chi2_removed);
    void afterWindowSolve();
    // afterWindowSolve() in phases, for a caller that marginalizes the windows of several estimators together (MarginalizationBatch,
    // host/marg_batch.h): afterWindowSolveBegin(); while (marginalizationDue()) { beginMarginalization(job, sink); [job.info marginalized: by
    // the caller's batch when job.device_factors > 0, else job.info->marginalization()]; finishMarginalization(job); } afterWindowSolveEnd().
    struct MarginalizationJob {
        std::shared_ptr<MarginalizationInfo> info;
        std::vector<std::shared_ptr<ReprojectionFactor>> factors; // the reprojection factors handed to the sink (kept alive until finish)
        int device_factors{0};
        std::unordered_map<long, long> parameters_ids;
        size_t num_marg{0};
        double last_time{0};
        std::shared_ptr<Frame> frame;
    };
    // (factor, pose_ref, pose_obs, extrinsic, inverse depth, td): ReprojectionBatch::add / MarginalizationBatch::addReprojectionFactor
    typedef std::function<void(ReprojectionFactor *, double *, double *, double *, double *, double *)> MarginalizationSink;
    void afterWindowSolveBegin();
    bool marginalizationDue() const;
    void beginMarginalization(MarginalizationJob &job, const MarginalizationSink &sink);
    void finishMarginalization(MarginalizationJob &job, bool valid);
    void afterWindowSolveEnd();
    void solveWindowAlone(int prepared_n_visual = -1); // >= 0: beginWindowSolve() already ran for this window and returned this count
    int firstNumIterations() const { return first_num_iterations_; }
    int secondNumIterations() const { return second_num_iterations_; }

    // ---- read-only views for the replay harness and the tests ----
    struct Counters {
        long imu{0}, gnss{0}, frames_tracked{0}, keyframes{0}, optimizations{0}, marginalizations{0}, ins_launches{0}, lost{0};
        long reprojection_factors{0}, chi2_removed{0};
    };
    const Counters &counters() const { return counters_; }
    const std::string &error() const { return error_; }
    size_t numTimeNodes() const { return timelist_.size(); }
    const Map::Ptr &map() const { return map_; }
    // the id space of this estimator's frames / keyframes / landmarks: Frame::createFrame(stamp, image, gvins.ids())
    const std::shared_ptr<IdSpace> &ids() const { return ids_; }
    Pose extrinsic() const { return pose_b_c_; }
    double timeDelay() const { return td_b_c_; }
    // latest mechanized INS state (flushes pending epochs)
    bool latestState(IntegrationState &state);

    static IntegrationStateData stateToData(const IntegrationState &state);
    static IntegrationState stateFromData(const IntegrationStateData &data);
    // MISC::detectZeroVelocity (misc.cc:363-415)
    static bool detectZeroVelocity(const std::vector<IMU> &imu_buffer, double imudatarate, std::vector<double> &average);

private:
    // ---- the three loop bodies of the reference, run from addNewImu / addNewFrame ----
    void fusionStep(const IMU &imu);
    void processTracking();
    void runOptimizationOnce();

    void flushIns();
    void parametersStatistic();
    bool gvinsInitialization();
    bool gvinsInitializationOptimization();
    void addNewTimeNode(double time);
    void addNewGnssTimeNode();
    bool insertNewGnssTimeNode();
    void addNewKeyFrameTimeNode();
    bool removeUnusedTimeNode();
    void constructPrior(bool is_zero_velocity);

    void addStateParameters(WindowProblem &problem);
    void addReprojectionParameters();
    void registerReprojectionBlocks(WindowProblem &problem);
    void addImuFactors(WindowProblem &problem);
    std::vector<std::pair<WindowSolver::ResidualBlockId, GNSS *>> addGnssFactors(WindowProblem &problem, bool isusekernel);
    int addReprojectionFactors();
    void doReintegration();
    void updateParametersFromOptimizer();
    int getStateDataIndex(double time);
    bool gvinsOptimization(int prepared_n_visual = -1);
    bool gvinsMarginalization();
    bool gvinsOutlierCulling();
    bool gvinsRemoveAllSecondNewFrame();
    void gnssOutlierCullingByChi2(WindowProblem &problem, std::vector<std::pair<WindowSolver::ResidualBlockId, GNSS *>> &residual_block);

    std::shared_ptr<Preintegration> createPreintegration(const IMU &imu0, const IntegrationState &state);
    void integrate(const std::vector<Preintegration *> &list);
    void fail(const std::string &what);

private:
    const double NORMAL_GRAVITY                = 9.80;  // ic_gvins.h:120-141
    const size_t MAXIMUM_INS_NUMBER            = 1000;
    const double MINMUM_ALIGN_VELOCITY         = 0.5;
    const double MINMUM_SYNC_INTERVAL          = 0.025;
    const double MAXIMUM_PREINTEGRATION_LENGTH = 10.0;
    const double GYROSCOPE_BIAS_PRIOR_STD      = 7200 * D2R / 3600;
    const double ACCELEROMETER_BIAS_PRIOR_STD  = 20000 * 1.0e-5;

    std::deque<std::shared_ptr<Preintegration>> preintegrationlist_;
    std::deque<IntegrationStateData> statedatalist_;
    std::deque<GNSS> gnsslist_;
    std::deque<double> timelist_;
    std::unordered_map<ulong, double> invdepthlist_;
    double extrinsic_[8]{0};
    std::vector<double> unused_time_nodes_;

    std::shared_ptr<MarginalizationInfo> last_marginalization_info_;
    std::vector<double *> last_marginalization_parameter_blocks_;

    bool is_use_prior_{false};
    double mix_prior_[18], mix_prior_std_[18], pose_prior_[7], pose_prior_std_[6];

    Tracking::Ptr tracking_;
    DeviceContext::Ptr tracking_device_;
    int nav_counter_{0}; // MISC::writeNavResult's every-10th-call counter, per estimator
    std::shared_ptr<IdSpace> ids_;
    Map::Ptr map_;
    Camera::Ptr camera_;
    Drawer::Ptr drawer_;

    bool isoptimized_{false}, isfinished_{false}, isgnssready_{false}, isframeready_{false}, isgnssobs_{false}, isvisualobs_{false};
    bool optimization_signalled_{false};

    std::queue<Frame::Ptr> keyframes_;
    GNSS gnss_, last_gnss_, last_last_gnss_;
    std::queue<Frame::Ptr> frame_buffer_;
    IMU imu_pre_, imu_cur_;
    InsWindow ins_window_;
    size_t ins_pending_{0}; // trailing entries of ins_window_ whose state has not been mechanized yet

    std::shared_ptr<IntegrationParameters> integration_parameters_;
    Preintegration::Variant preintegration_options_{Preintegration::NORMAL};
    IntegrationConfiguration integration_config_;
    double imudatarate_{200}, imudatadt_{0.005};
    size_t reserved_ins_num_{2};
    Vector3d antlever_;
    int initlength_{1};
    Pose pose_b_c_;
    double td_b_c_{0};
    bool is_use_visualization_{false};
    bool optimize_estimate_extrinsic_{false}, optimize_estimate_td_{false};
    double optimize_reprojection_error_std_{0};
    int optimize_num_iterations_{20};
    size_t optimize_windows_size_{10};
    double reprojection_error_std_{1.5};
    int first_num_iterations_{5}, second_num_iterations_{15};

    // gvinsInitialization's function-local statics (ic_gvins.cc:606-608)
    Vector3d init_bg_, init_att_;
    bool is_has_zero_velocity_{false};

    int iterations_[2]{0, 0};
    double timecosts_[3]{0, 0, 0};
    int outliers_[2]{0, 0};

    FileSaver::Ptr navfilesaver_, imuerrfilesaver_, ptsfilesaver_, statfilesaver_, extfilesaver_, trajfilesaver_;
    GVINSState gvinsstate_{GVINS_ERROR};

    // device side
    icg_ctx *ctx_{nullptr}; // INS / preintegration / culling launches of this estimator
    std::unique_ptr<ReprojectionBatch> visual_batch_, marg_batch_;
    std::vector<std::unique_ptr<ReprojectionFactor>> visual_factors_;
    std::vector<double *> visual_invdepth_blocks_; // inverse depths with at least one factor, first-seen order
    struct VisualBlocks {
        double *pose_i, *pose_j, *invdepth;
    };
    std::vector<VisualBlocks> visual_blocks_; // the blocks of visual_factors_[k]
    std::vector<std::pair<WindowSolver::ResidualBlockId, GNSS *>> gnss_blocks_; // GNSS residual blocks of the window being solved
    bool deferred_window_solves_{false}, window_solve_pending_{false};
    std::chrono::steady_clock::time_point after_solve_t0_; // afterWindowSolveBegin() .. End(): column 13 of statistics.txt (timecosts_[2])
    Counters counters_;
    double phase_ms_[8]{0, 0, 0, 0, 0, 0, 0, 0}; // host wall time per phase (tracking, INS, build, solves, write-back, marginalization, statistics, nodes)
    std::string error_;
};

} // namespace icg
