// GVINS on the MI355X host layer: the reference's estimator (ic_gvins.cc) as one deterministic event loop over the device paths of
// this library.  See gvins_hip.h for what is kept and what is deliberately different.
#include "gvins_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "yaml_lite.h"

namespace icg {

namespace {
struct TimeCost { // common/timecost.h
    std::chrono::steady_clock::time_point t0{std::chrono::steady_clock::now()};
    void restart() { t0 = std::chrono::steady_clock::now(); }
    double costInMillisecond() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
bool isTheSameTimeNode(double time0, double time1, double interval) { return std::fabs(time0 - time1) < interval; } // misc.cc:119-121
size_t stateDataIndex(const std::deque<double> &timelist, double time, double interval) { // MISC::getStateDataIndex, misc.cc:123-149
    size_t index = 0, sta = 0, end = timelist.size();
    int counts = 0;
    while (true) {
        size_t mid      = (sta + end) / 2;
        double mid_time = timelist[mid];
        if (isTheSameTimeNode(mid_time, time, interval)) {
            index = mid;
            break;
        } else if (mid_time > time) {
            end = mid;
        } else if (mid_time < time) {
            sta = mid;
        }
        if (counts++ > 10) break;
    }
    return index;
}
Pose stateToCameraPose(const IntegrationState &state, const Pose &pose_b_c) { // misc.cc:102-108
    Matrix3d R = Rotation::quaternion2matrix(state.q);
    Pose pose;
    pose.t = state.p + R * pose_b_c.t;
    pose.R = R * pose_b_c.R;
    return pose;
}
bool debugOn() {
    static bool on = getenv("ICG_GVINS_DEBUG") != nullptr;
    return on;
}
struct PhaseTimer { // adds the wall time of a scope to one slot of the estimator's phase clock (printed under ICG_GVINS_DEBUG=1)
    double *acc;
    int k;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer(double *a, int k_) : acc(a), k(k_), t0(std::chrono::steady_clock::now()) {}
    ~PhaseTimer() { acc[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
enum { PH_TRACK = 0, PH_INS, PH_BUILD, PH_SOLVE, PH_FINISH, PH_MARG, PH_STAT, PH_NODES };
#define GLOG(...)                                                                                                                               \
    do {                                                                                                                                        \
        if (debugOn()) {                                                                                                                        \
            fprintf(stderr, "[gvins] " __VA_ARGS__);                                                                                            \
            fprintf(stderr, "\n");                                                                                                              \
        }                                                                                                                                       \
    } while (0)
} // namespace

IntegrationStateData GVINS::stateToData(const IntegrationState &state) { // preintegration_base.cc:104-113
    IntegrationStateData data;
    data.time = state.time;
    for (int k = 0; k < 3; k++) {
        data.pose[k]    = state.p[k];
        data.mix[k]     = state.v[k];
        data.mix[3 + k] = state.bg[k];
        data.mix[6 + k] = state.ba[k];
    }
    data.pose[3] = state.q.x, data.pose[4] = state.q.y, data.pose[5] = state.q.z, data.pose[6] = state.q.w;
    return data;
}

IntegrationState GVINS::stateFromData(const IntegrationStateData &data) { // preintegration_base.cc:115-125
    IntegrationState state;
    state.time = data.time;
    state.p    = Vector3d(data.pose[0], data.pose[1], data.pose[2]);
    state.q    = Rotation::normalized(Quaterniond{data.pose[3], data.pose[4], data.pose[5], data.pose[6]});
    state.v    = Vector3d(data.mix[0], data.mix[1], data.mix[2]);
    state.bg   = Vector3d(data.mix[3], data.mix[4], data.mix[5]);
    state.ba   = Vector3d(data.mix[6], data.mix[7], data.mix[8]);
    return state;
}

bool GVINS::detectZeroVelocity(const std::vector<IMU> &imu_buffer, double imudatarate, std::vector<double> &average) { // misc.cc:363-415
    const double ZERO_VELOCITY_GYR_THRESHOLD = 0.002, ZERO_VELOCITY_ACC_THRESHOLD = 0.1; // misc.h:75-76
    double size_invert = 1.0 / static_cast<double>(imu_buffer.size());
    double sum[6], std_[6];
    average.assign(6, 0.0);
    for (const auto &imu : imu_buffer)
        for (int k = 0; k < 3; k++) average[k] += imu.dtheta[k], average[3 + k] += imu.dvel[k];
    for (int k = 0; k < 6; k++) average[k] *= size_invert, sum[k] = 0;
    for (const auto &imu : imu_buffer)
        for (int k = 0; k < 3; k++) {
            sum[k] += (imu.dtheta[k] - average[k]) * (imu.dtheta[k] - average[k]);
            sum[3 + k] += (imu.dvel[k] - average[3 + k]) * (imu.dvel[k] - average[3 + k]);
        }
    for (int k = 0; k < 6; k++) std_[k] = std::sqrt(sum[k] * size_invert) * imudatarate;
    return (std_[0] < ZERO_VELOCITY_GYR_THRESHOLD) && (std_[1] < ZERO_VELOCITY_GYR_THRESHOLD) && (std_[2] < ZERO_VELOCITY_GYR_THRESHOLD) &&
           (std_[3] < ZERO_VELOCITY_ACC_THRESHOLD) && (std_[4] < ZERO_VELOCITY_ACC_THRESHOLD) && (std_[5] < ZERO_VELOCITY_ACC_THRESHOLD);
}

void GVINS::fail(const std::string &what) {
    error_      = what;
    gvinsstate_ = GVINS_ERROR;
    isfinished_ = true;
    throw std::runtime_error("GVINS: " + what);
}

GVINS::GVINS(const std::string &configfile, const std::string &outputpath, Drawer::Ptr drawer, int device) { // ic_gvins.cc:46-167
    gvinsstate_ = GVINS_ERROR;
    isfinished_ = true;
    YamlLite config;
    std::string err;
    if (!YamlLite::load(configfile, config, &err)) {
        error_ = "Failed to open configuration file: " + err;
        return;
    }
    navfilesaver_    = FileSaver::create(outputpath + "/gvins.nav", 11);
    ptsfilesaver_    = FileSaver::create(outputpath + "/mappoint.txt", 3);
    statfilesaver_   = FileSaver::create(outputpath + "/statistics.txt", 3);
    extfilesaver_    = FileSaver::create(outputpath + "/extrinsic.txt", 3);
    imuerrfilesaver_ = FileSaver::create(outputpath + "/IMU_ERR.bin", 7, FileSaver::BINARY);
    trajfilesaver_   = FileSaver::create(outputpath + "/trajectory.csv", 8);
    if (!navfilesaver_->isOpen() || !ptsfilesaver_->isOpen() || !statfilesaver_->isOpen() || !extfilesaver_->isOpen()) {
        error_ = "Failed to open data file";
        return;
    }
    { // a copy of the configuration in the output directory (:75-77)
        std::ofstream ofconfig(outputpath + "/gvins.yaml");
        ofconfig << config.text();
    }
    try {
        initlength_       = (int) config.integer("initlength");
        imudatarate_      = config.real("imudatarate");
        imudatadt_        = 1.0 / imudatarate_;
        reserved_ins_num_ = 2;
        std::vector<double> vecdata = config.reals("antlever");
        antlever_                   = Vector3d(vecdata.at(0), vecdata.at(1), vecdata.at(2));

        integration_parameters_               = std::make_shared<IntegrationParameters>();
        integration_parameters_->gyr_arw      = config.real("imumodel.arw") * D2R / 60.0;
        integration_parameters_->gyr_bias_std = config.real("imumodel.gbstd") * D2R / 3600.0;
        integration_parameters_->acc_vrw      = config.real("imumodel.vrw") / 60.0;
        integration_parameters_->acc_bias_std = config.real("imumodel.abstd") * 1.0e-5;
        integration_parameters_->corr_time    = config.real("imumodel.corrtime") * 3600;
        integration_parameters_->gravity      = NORMAL_GRAVITY;

        integration_config_.iswithearth = config.boolean("iswithearth");
        integration_config_.isuseodo    = false;
        integration_config_.iswithscale = false;
        integration_config_.gravity     = Vector3d(0, 0, integration_parameters_->gravity);
        integration_config_.origin      = Vector3d(0, 0, 0);
        preintegration_options_         = integration_config_.iswithearth ? Preintegration::EARTH : Preintegration::NORMAL;

        std::vector<double> intrinsic  = config.reals("cam0.intrinsic");
        std::vector<double> distortion = config.reals("cam0.distortion");
        std::vector<double> res        = config.reals("cam0.resolution");
        camera_ = Camera::createCamera(intrinsic, distortion, std::vector<int>{(int) res.at(0), (int) res.at(1)});

        vecdata = config.reals("cam0.q_b_c"); // Eigen::Quaterniond(const double*) reads x, y, z, w
        Quaterniond q_b_c{vecdata.at(0), vecdata.at(1), vecdata.at(2), vecdata.at(3)};
        vecdata     = config.reals("cam0.t_b_c");
        pose_b_c_.R = Rotation::quaternion2matrix(q_b_c);
        pose_b_c_.t = Vector3d(vecdata.at(0), vecdata.at(1), vecdata.at(2));
        td_b_c_     = config.real("cam0.td_b_c");

        reprojection_error_std_      = config.real("reprojection_error_std");
        optimize_estimate_extrinsic_ = config.boolean("optimize_estimate_extrinsic");
        optimize_estimate_td_        = config.boolean("optimize_estimate_td");
        optimize_num_iterations_     = (int) config.integer("optimize_num_iterations");
        optimize_windows_size_       = (size_t) config.integer("optimize_windows_size");
        optimize_reprojection_error_std_ = reprojection_error_std_ / camera_->focalLength();
        is_use_visualization_            = config.boolean("is_use_visualization");
        first_num_iterations_            = optimize_num_iterations_ / 4; // :1131-1132
        second_num_iterations_           = optimize_num_iterations_ - first_num_iterations_;

        map_      = std::make_shared<Map>(optimize_windows_size_);
        drawer_   = drawer ? std::move(drawer) : std::make_shared<Drawer>();
        // one id space per estimator: the reference's process-wide id factories (frame.cc:38-54, mappoint.cc:47) make the first frame's
        // map key (keyframe id 0 before Frame::setKeyFrame) coincide with the id it is given later only for the FIRST estimator of a process
        ids_ = std::make_shared<IdSpace>();
        TrackingConfig tracking_config;
        std::string terr;
        if (!TrackingConfig::fromYamlFile(configfile, tracking_config, &terr)) throw std::runtime_error(terr);
        auto tracking_device = std::make_shared<DeviceContext>(device, camera_->width(), camera_->height(), 1, tracking_config.track_max_features);
        tracking_device->setCamera(*camera_);
        tracking_ = std::make_shared<Tracking>(camera_, map_, drawer_, tracking_config, outputpath, tracking_device, ids_);
        tracking_device_ = tracking_device;

        icg_ctx_config cfg{};
        cfg.device = device, cfg.width = 64, cfg.height = 64, cfg.n_slots = 1, cfg.max_batch = 1, cfg.max_points = 64;
        if (icg_ctx_create(&cfg, &ctx_) != ICG_OK) throw std::runtime_error(icg_last_error(nullptr));
        icg_camera cam = camera_->abi();
        if (icg_set_camera(ctx_, &cam) != ICG_OK) throw std::runtime_error(icg_last_error(ctx_));
        visual_batch_.reset(new ReprojectionBatch(device));
        marg_batch_.reset(new ReprojectionBatch(device));
    } catch (const std::exception &e) {
        error_ = e.what();
        return;
    }
    gnss_.blh = last_gnss_.blh = Vector3d(0, 0, 0);
    gnss_.time = last_gnss_.time = last_last_gnss_.time = 0;
    imu_pre_.time = imu_cur_.time = 0;
    isfinished_ = false;
    gvinsstate_ = GVINS_INITIALIZING;
}

GVINS::~GVINS() {
    visual_batch_.reset(); // the batch un-registers itself from its factors: before they go
    marg_batch_.reset();
    visual_factors_.clear();
    tracking_.reset();
    if (ctx_) icg_ctx_destroy(ctx_);
}

void GVINS::setWaitMode(int icg_wait_mode, int sleep_us) {
    if (tracking_device_) (void) icg_ctx_set_wait_mode(tracking_device_->ctx(), icg_wait_mode, sleep_us);
    if (ctx_) (void) icg_ctx_set_wait_mode(ctx_, icg_wait_mode, sleep_us);
    if (visual_batch_) visual_batch_->setWaitMode(icg_wait_mode, sleep_us);
    if (marg_batch_) marg_batch_->setWaitMode(icg_wait_mode, sleep_us);
}

void GVINS::setFinished() { // ic_gvins.cc:554-582
    if (isfinished_) return;
    try {
        flushIns();
    } catch (const std::exception &) {
    }
    isfinished_ = true;
    if (debugOn()) {
        static const char *names[8] = {"tracking", "INS launches", "problem build", "window solves", "write-back + culling", "marginalization", "statistics", "time nodes"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "[gvins-phase] %-22s %9.3f ms\n", names[k], phase_ms_[k]);
    }
    for (auto &f : {navfilesaver_, imuerrfilesaver_, ptsfilesaver_, statfilesaver_, extfilesaver_, trajfilesaver_})
        if (f) f->flush();
}

bool GVINS::latestState(IntegrationState &state) {
    if (ins_window_.empty() || gvinsstate_ <= GVINS_INITIALIZING) return false;
    flushIns();
    state = ins_window_.back().second;
    return true;
}

// ---- ingestion ----------------------------------------------------------------------------------------------------------
bool GVINS::addNewImu(const IMU &imu) { // ic_gvins.cc:169-197 + the body of runFusion
    if (isfinished_) return false;
    counters_.imu++;
    if (imu.dt > (imudatadt_ * 1.5)) { // lost samples are filled with copies (:171-183)
        long cnts    = lround(imu.dt / imudatadt_) - 1;
        IMU imudata  = imu;
        imudata.time = imu.time - imu.dt;
        while (cnts-- > 0) {
            imudata.time += imudatadt_;
            imudata.dt = imudatadt_;
            fusionStep(imudata);
        }
    } else {
        fusionStep(imu);
    }
    return true;
}

bool GVINS::addNewGnss(const GNSS &gnss) { // ic_gvins.cc:199-220
    if (isfinished_) return false;
    counters_.gnss++;
    bool origin_zero = integration_config_.origin[0] == 0 && integration_config_.origin[1] == 0 && integration_config_.origin[2] == 0;
    if (origin_zero) {
        integration_config_.origin       = gnss.blh;
        integration_parameters_->gravity = Earth::gravity(gnss.blh);
    } else {
        last_last_gnss_ = last_gnss_;
        last_gnss_      = gnss_;
    }
    gnss_        = gnss;
    gnss_.blh    = Earth::global2local(integration_config_.origin, gnss_.blh);
    isgnssready_ = true;
    return true;
}

bool GVINS::addNewFrame(const Frame::Ptr &frame) { // ic_gvins.cc:222-235
    if (isfinished_) return false;
    if (gvinsstate_ > GVINS_INITIALIZING_INS) {
        frame_buffer_.push(frame);
        processTracking();
    }
    return true;
}

// ---- INS ---------------------------------------------------------------------------------------------------------------
// mechanize every pending epoch of the INS window with ONE series launch and write the navigation lines the reference writes
// after each epoch (runFusion :284-286, :387-389)
void GVINS::flushIns() {
    if (ins_pending_ == 0) return;
    PhaseTimer pt(phase_ms_, PH_INS);
    const size_t n = ins_window_.size(), first = n - ins_pending_;
    if (first == 0) fail("INS window has no mechanized state to start from");
    std::vector<IMU> series;
    series.reserve(ins_pending_ + 1);
    for (size_t k = first - 1; k < n; k++) series.push_back(ins_window_[k].first);
    IntegrationState state = ins_window_[first - 1].second;
    std::vector<std::vector<IntegrationState>> traj;
    std::string err;
    if (!MISC::insMechanizationBatch(ctx_, integration_config_, {&series}, {&state}, &traj, &err)) fail("INS mechanization: " + err);
    counters_.ins_launches++;
    for (size_t k = 0; k < ins_pending_; k++) {
        ins_window_[first + k].second = traj[0][k];
        MISC::writeNavResult(integration_config_, traj[0][k], navfilesaver_, imuerrfilesaver_, trajfilesaver_, &nav_counter_);
    }
    ins_pending_ = 0;
}

void GVINS::fusionStep(const IMU &imu) { // one pass of the IMU BUFFER loop of runFusion (ic_gvins.cc:249-391)
    imu_pre_ = imu_cur_;
    imu_cur_ = imu;
    bool output_now = false;
    if (gvinsstate_ > GVINS_INITIALIZING) {
        if (isoptimized_) { // the optimizer has finished: re-propagate the window from the newest optimized state (:272-280)
            isoptimized_ = false;
            flushIns(); // epochs before this one keep (and have written) the states they had
            ins_window_.emplace_back(imu_cur_, IntegrationState());
            IntegrationState state = stateFromData(statedatalist_.back());
            std::string err;
            if (!MISC::redoInsMechanizationBatch(ctx_, integration_config_, {state}, reserved_ins_num_, {&ins_window_}, &err))
                fail("INS re-mechanization: " + err);
            counters_.ins_launches++;
            output_now = true;
        } else {
            ins_window_.emplace_back(imu_cur_, IntegrationState());
            ins_window_.back().second.time = imu_cur_.time;
            ins_pending_++;
        }
    } else {
        ins_window_.emplace_back(imu_cur_, IntegrationState());
        if (ins_window_.size() > MAXIMUM_INS_NUMBER) ins_window_.pop_front(); // :289-293
    }

    bool skip_output = false;
    if (gvinsstate_ == GVINS_INITIALIZING) {
        if (isgnssready_) {
            if (gvinsInitialization()) {
                gvinsstate_  = GVINS_INITIALIZING_INS;
                isoptimized_ = true; // the INS window is re-propagated at the next epoch (:304-306)
            }
            isgnssready_ = false;
            skip_output  = true; // `continue` (:311)
        }
    } else if (gvinsstate_ == GVINS_INITIALIZING_INS) {
        if (isgnssready_) {
            if (gnss_.time < ins_window_.back().first.time) { // data alignment (:319)
                addNewGnssTimeNode();
                isgnssready_            = false;
                isgnssobs_              = true;
                optimization_signalled_ = true;
            }
        }
    } else if (gvinsstate_ == GVINS_INITIALIZING_VIO) {
        if (isframeready_ || isgnssready_) {
            if (isframeready_ && (keyframes_.front()->stamp() < ins_window_.back().first.time)) {
                addNewKeyFrameTimeNode();
                isframeready_ = false;
                gvinsstate_   = GVINS_TRACKING_INITIALIZING;
            }
            if (isgnssready_) {
                if (insertNewGnssTimeNode()) isgnssready_ = false;
            }
        }
    } else if (gvinsstate_ >= GVINS_TRACKING_INITIALIZING) {
        if (isframeready_ || isgnssready_) {
            if (isframeready_ && (keyframes_.front()->stamp() < ins_window_.back().first.time)) {
                addNewKeyFrameTimeNode();
                isframeready_ = false;
                isvisualobs_  = true;
            }
            if (isgnssready_) {
                if (insertNewGnssTimeNode()) {
                    isgnssready_ = false;
                    isgnssobs_   = true;
                }
            }
            if (isvisualobs_) optimization_signalled_ = true;
        }
    }
    if (output_now && !skip_output && gvinsstate_ > GVINS_INITIALIZING)
        MISC::writeNavResult(integration_config_, ins_window_.back().second, navfilesaver_, imuerrfilesaver_, trajfilesaver_, &nav_counter_);

    // the other two loops of the reference, run to completion before the next IMU epoch
    processTracking();
    if (optimization_signalled_) {
        optimization_signalled_ = false;
        runOptimizationOnce();
    }
}

// ---- tracking ------------------------------------------------------------------------------------------------------------
void GVINS::processTracking() { // body of runTracking (ic_gvins.cc:493-550)
    while (!frame_buffer_.empty()) {
        Frame::Ptr frame = frame_buffer_.front();
        const double td  = td_b_c_;
        if (ins_window_.empty() || (ins_window_.back().first.time <= (frame->stamp() + td))) return; // wait for the INS (:512)
        frame_buffer_.pop();
        flushIns();
        frame->setStamp(frame->stamp() + td);
        frame->setTimeDelay(td);
        std::vector<Pose> poses;
        std::vector<uint8_t> found;
        std::string err;
        if (!MISC::getCameraPoseFromInsWindowBatch(ctx_, {&ins_window_}, pose_b_c_, {frame->stamp()}, poses, found, &err)) fail("pose prior: " + err);
        frame->setPose(poses[0]);
        TrackState trackstate;
        {
            PhaseTimer pt(phase_ms_, PH_TRACK);
            trackstate = tracking_->track(frame);
        }
        counters_.frames_tracked++;
        if (trackstate == TRACK_LOST) counters_.lost++;
        GLOG("track %.3f -> state %d, features %zu, new keyframe %d", frame->stamp(), (int) trackstate, frame->numFeatures(), (int) tracking_->isNewKeyFrame());
        if (tracking_->isNewKeyFrame() || (trackstate == TRACK_FIRST_FRAME) || trackstate == TRACK_LOST) {
            keyframes_.push(frame);
            isframeready_ = true;
        }
    }
}

// ---- optimization loop body ------------------------------------------------------------------------------------------------
void GVINS::runOptimizationOnce() { // body of runOptimization (ic_gvins.cc:404-475)
    if (!(isgnssobs_ || isvisualobs_)) return;
    if (gvinsstate_ == GVINS_INITIALIZING_INS) {
        bool isinitialized = gvinsInitializationOptimization();
        if (preintegrationlist_.size() >= static_cast<size_t>(initlength_)) {
            gvinsstate_ = GVINS_INITIALIZING_VIO;
            GLOG("GINS initialization %s", isinitialized ? "is finished" : "is not convergence");
        }
        afterWindowSolve();
    } else if (gvinsstate_ >= GVINS_TRACKING_INITIALIZING) {
        if (map_->isMaximumKeframes()) gvinsstate_ = GVINS_TRACKING_NORMAL;
        if (deferred_window_solves_) { // the caller runs the phases (possibly together with other estimators' windows)
            window_solve_pending_ = true;
            return;
        }
        solveWindowAlone();
    } else {
        afterWindowSolve();
    }
}

void GVINS::solveWindowAlone(int prepared_n_visual) {
    gvinsOptimization(prepared_n_visual);
    afterWindowSolve();
}

// what follows the solve in runOptimization (:436-471): window maintenance, statistics, the flags the fusion loop looks at
void GVINS::afterWindowSolve() {
    afterWindowSolveBegin();
    while (marginalizationDue()) gvinsMarginalization();
    afterWindowSolveEnd();
}

void GVINS::afterWindowSolveBegin() {
    after_solve_t0_ = std::chrono::steady_clock::now();
    if (gvinsstate_ >= GVINS_TRACKING_INITIALIZING) gvinsRemoveAllSecondNewFrame();
}

bool GVINS::marginalizationDue() const { return gvinsstate_ >= GVINS_TRACKING_INITIALIZING && map_->isMaximumKeframes(); }

void GVINS::afterWindowSolveEnd() {
    if (gvinsstate_ >= GVINS_TRACKING_INITIALIZING) {
        timecosts_[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - after_solve_t0_).count();
        parametersStatistic();
    }
    window_solve_pending_ = false;
    isgnssobs_ = isvisualobs_ = false;
    isoptimized_              = true;
    counters_.optimizations++;
    if (debugOn()) { // every time node after the solve: time, p, q, v, bg, ba
        for (const auto &d : statedatalist_)
            fprintf(stderr, "[gvins-state] %d %.4f %.6f %.6f %.6f %.8f %.8f %.8f %.8f %.5f %.5f %.5f %.3e %.3e %.3e %.3e %.3e %.3e\n", (int) counters_.optimizations,
                    d.time, d.pose[0], d.pose[1], d.pose[2], d.pose[3], d.pose[4], d.pose[5], d.pose[6], d.mix[0], d.mix[1], d.mix[2], d.mix[3], d.mix[4],
                    d.mix[5], d.mix[6], d.mix[7], d.mix[8]);
    }
}

// ---- initialization --------------------------------------------------------------------------------------------------------
bool GVINS::gvinsInitialization() { // ic_gvins.cc:584-692
    if ((gnss_.time == 0) || (last_gnss_.time == 0)) return false;
    std::vector<IMU> imu_buff;
    for (const auto &ins : ins_window_) {
        const IMU &imu = ins.first;
        if ((imu.time > last_gnss_.time) && (imu.time < gnss_.time)) imu_buff.push_back(imu);
    }
    if (imu_buff.size() < 20) return false;

    std::vector<double> average;
    bool is_zero_velocity = detectZeroVelocity(imu_buff, imudatarate_, average);
    if (is_zero_velocity) {
        init_bg_ = Vector3d(average[0], average[1], average[2]) * imudatarate_;
        Vector3d fb = Vector3d(average[3], average[4], average[5]) * imudatarate_;
        init_att_[0] = -std::asin(fb[1] / integration_parameters_->gravity);
        init_att_[1] = std::asin(fb[0] / integration_parameters_->gravity);
        is_has_zero_velocity_ = true;
    }
    if (!is_zero_velocity) {
        if (last_gnss_.isyawvalid) {
            init_att_[2] = last_gnss_.yaw;
        } else {
            Vector3d vel = gnss_.blh - last_gnss_.blh;
            if (vel.norm() < MINMUM_ALIGN_VELOCITY) return false;
            if (!is_has_zero_velocity_) {
                init_att_[0] = 0;
                init_att_[1] = std::atan(-vel.z() / std::sqrt(vel.x() * vel.x() + vel.y() * vel.y()));
            }
            init_att_[2] = std::atan2(vel.y(), vel.x());
        }
    } else {
        return false;
    }

    IntegrationState state;
    state.time = last_gnss_.time;
    state.q    = Rotation::euler2quaternion(init_att_);
    state.p    = last_gnss_.blh - Rotation::rotate(state.q, antlever_);
    state.v    = Vector3d(0, 0, 0);
    state.bg   = init_bg_;
    state.ba   = Vector3d(0, 0, 0);
    statedatalist_.emplace_back(stateToData(state));
    gnsslist_.push_back(last_gnss_);
    timelist_.push_back(last_gnss_.time);
    constructPrior(is_has_zero_velocity_);

    integration_config_.gravity = Vector3d(0, 0, integration_parameters_->gravity);
    if (integration_config_.iswithearth) integration_config_.iewn = Earth::iewn(integration_config_.origin, state.p);

    // the first second of INS results (:680-683)
    state = stateFromData(statedatalist_.back());
    std::string err;
    if (!MISC::redoInsMechanizationBatch(ctx_, integration_config_, {state}, reserved_ins_num_, {&ins_window_}, &err)) fail("INS re-mechanization: " + err);
    counters_.ins_launches++;
    ins_pending_ = 0;
    GLOG("Initialization at %.3f, heading %.2f deg", gnss_.time, init_att_[2] * R2D);
    addNewGnssTimeNode();
    return true;
}

bool GVINS::gvinsInitializationOptimization() { // ic_gvins.cc:694-722 (Ceres SPARSE_NORMAL_CHOLESKY, 50 iterations)
    WindowSolver solver(nullptr, 0.0);
    SingleWindowProblem problem(solver);
    addStateParameters(problem);
    addGnssFactors(problem, true);
    addImuFactors(problem);
    WindowSolver::Options options;
    options.max_num_iterations = 50;
    WindowSolver::Summary summary;
    if (!solver.solve(options, &summary)) fail("GNSS/INS initialization solve: " + solver.error());
    GLOG("%s", summary.BriefReport().c_str());
    // ceres::CONVERGENCE: one of the three tolerances was met (only logged by the caller, ic_gvins.cc:420-424)
    return summary.termination == "function_tolerance" || summary.termination == "gradient_tolerance" || summary.termination == "parameter_tolerance";
}

// ---- time nodes ------------------------------------------------------------------------------------------------------------
std::shared_ptr<Preintegration> GVINS::createPreintegration(const IMU &imu0, const IntegrationState &state) {
    std::shared_ptr<IntegrationParameters> parameters = integration_parameters_;
    if (preintegration_options_ == Preintegration::EARTH) {
        // the reference derives the Earth rate from IntegrationParameters::station, which nothing ever assigns (zero-initialised):
        // Earth::iewn(station = 0, p) (preintegration_earth.cc:320, SURVEY.md hazard H9).  Reproduced, per interval.
        // resetState recomputes it from the interval's start position at every reintegration too (:319-321): Preintegration does the same
        parameters              = std::make_shared<IntegrationParameters>(*integration_parameters_);
        parameters->has_station = true;
        parameters->station     = Vector3d(0, 0, 0);
    }
    return std::make_shared<Preintegration>(parameters, imu0, state, preintegration_options_);
}

void GVINS::integrate(const std::vector<Preintegration *> &list) {
    std::string err;
    if (!Preintegration::integrateBatch(ctx_, list, &err)) fail("preintegration: " + err);
}

void GVINS::addNewKeyFrameTimeNode() { // ic_gvins.cc:724-752
    while (!keyframes_.empty()) {
        auto frame       = keyframes_.front();
        double frametime = frame->stamp();
        if (frametime > ins_window_.back().first.time) break;
        keyframes_.pop();
        map_->insertKeyFrame(frame);
        counters_.keyframes++;
        addNewTimeNode(frametime);
        GLOG("keyframe %lu at %.3f (%zu new mappoints), %zu time nodes", frame->keyFrameId(), frametime, frame->unupdatedMappoints().size(), timelist_.size());
    }
    removeUnusedTimeNode();
}

bool GVINS::removeUnusedTimeNode() { // ic_gvins.cc:754-789
    if (unused_time_nodes_.empty()) return false;
    std::vector<Preintegration *> dirty;
    for (double node : unused_time_nodes_) {
        int index = getStateDataIndex(node);
        if (index < 1) continue; // the reference tests < 0 and then indexes [index - 1]
        auto first_preintegration  = preintegrationlist_[(size_t) index - 1];
        auto second_preintegration = preintegrationlist_[(size_t) index];
        auto imu_buffer            = second_preintegration->imuBuffer();
        for (size_t k = 1; k < imu_buffer.size(); k++) first_preintegration->addNewImu(imu_buffer[k]);
        preintegrationlist_.erase(preintegrationlist_.begin() + index);
        timelist_.erase(timelist_.begin() + index);
        statedatalist_.erase(statedatalist_.begin() + index);
        dirty.push_back(first_preintegration.get());
    }
    unused_time_nodes_.clear();
    if (!dirty.empty()) integrate(dirty); // merged intervals: one launch
    return true;
}

bool GVINS::insertNewGnssTimeNode() { // ic_gvins.cc:791-888
    if (gnss_.time > timelist_.back()) return false;
    double sta = 0, end = 0;
    size_t index = 0;
    for (size_t k = timelist_.size() - 1; k > 1; k--) {
        if ((gnss_.time <= timelist_[k]) && (gnss_.time > timelist_[k - 1])) {
            sta   = timelist_[k - 1];
            end   = timelist_[k];
            index = k;
        }
    }
    if (sta == 0) return false;

    bool is_need_gnss = false;
    auto keyframeids  = map_->orderedKeyFrames();
    for (int k = (int) keyframeids.size() - 1; k >= 0; k--) {
        auto frame = map_->keyframes().find(keyframeids[(size_t) k])->second;
        if (isTheSameTimeNode(frame->stamp(), end, MISC::MINIMUM_TIME_INTERVAL)) {
            if (frame->keyFrameState() != KEYFRAME_REMOVE_SECOND_NEW) is_need_gnss = true;
        }
    }
    if (!is_need_gnss) return true;

    if (gnss_.time - sta < MINMUM_SYNC_INTERVAL) { // align to the previous node
        GNSS gnss = gnss_;
        gnss.time = sta;
        double dt = gnss_.time - sta;
        for (int k = 0; k < 3; k++) gnss.blh[k] -= statedatalist_[index - 1].mix[k] * dt;
        gnss.std = gnss.std * 1.2;
        gnsslist_.push_back(gnss);
    } else if (end - gnss_.time < MINMUM_SYNC_INTERVAL) { // align to the current node
        GNSS gnss = gnss_;
        gnss.time = end;
        double dt = end - gnss_.time;
        for (int k = 0; k < 3; k++) gnss.blh[k] += statedatalist_[index].mix[k] * dt;
        gnss.std = gnss.std * 1.2;
        gnsslist_.push_back(gnss);
    } else {
        if (preintegrationlist_[index - 1]->deltaTime() > MAXIMUM_PREINTEGRATION_LENGTH) return true;
        std::vector<double> timelist;
        for (size_t k = index; k < timelist_.size(); k++) timelist.push_back(timelist_[k]);
        size_t num_remove = timelist_.size() - index;
        for (size_t k = num_remove; k > 0; k--) {
            timelist_.pop_back();
            statedatalist_.pop_back();
            preintegrationlist_.pop_back();
        }
        addNewGnssTimeNode();
        for (size_t k = 0; k < timelist.size(); k++) addNewTimeNode(timelist[k]);
    }
    return true;
}

void GVINS::addNewGnssTimeNode() { // ic_gvins.cc:890-895
    addNewTimeNode(gnss_.time);
    gnsslist_.push_back(gnss_);
}

void GVINS::addNewTimeNode(double time) { // ic_gvins.cc:897-928
    PhaseTimer pt(phase_ms_, PH_NODES);
    std::vector<IMU> series;
    double start = timelist_.back();
    if (!MISC::getImuSeriesFromTo(ins_window_, start, time, series)) fail("no IMU samples between two time nodes");
    IntegrationState state = stateFromData(statedatalist_.back());
    preintegrationlist_.emplace_back(createPreintegration(series[0], state));
    for (size_t k = 1; k < series.size(); k++) preintegrationlist_.back()->addNewImu(series[k]);
    integrate({preintegrationlist_.back().get()});
    state      = preintegrationlist_.back()->currentState();
    state.time = time;
    GLOG("time node %.4f -> %.4f: %zu IMU samples, delta time %.4f, p (%.3f %.3f %.3f)", start, time, series.size(), preintegrationlist_.back()->deltaTime(), state.p[0],
         state.p[1], state.p[2]);
    statedatalist_.emplace_back(stateToData(state));
    timelist_.push_back(time);
}

int GVINS::getStateDataIndex(double time) { // ic_gvins.cc:1839-1848
    size_t index = stateDataIndex(timelist_, time, MISC::MINIMUM_TIME_INTERVAL);
    if (!isTheSameTimeNode(timelist_[index], time, MISC::MINIMUM_TIME_INTERVAL)) return -1;
    return static_cast<int>(index);
}

void GVINS::constructPrior(bool is_zero_velocity) { // ic_gvins.cc:1911-1936
    double pos_prior_std = 0.1, att_prior_std = 0.5 * D2R, vel_prior_std = 0.1;
    double bg_prior_std = integration_parameters_->gyr_bias_std * 3, ba_prior_std = ACCELEROMETER_BIAS_PRIOR_STD, sodo_prior_std = 0.005;
    if (!is_zero_velocity) bg_prior_std = GYROSCOPE_BIAS_PRIOR_STD;
    memcpy(pose_prior_, statedatalist_[0].pose, sizeof(double) * 7);
    memcpy(mix_prior_, statedatalist_[0].mix, sizeof(double) * 18);
    for (int k = 0; k < 18; k++) mix_prior_std_[k] = 1.0;
    for (size_t k = 0; k < 3; k++) {
        pose_prior_std_[k + 0] = pos_prior_std;
        pose_prior_std_[k + 3] = att_prior_std;
        mix_prior_std_[k + 0]  = vel_prior_std;
        mix_prior_std_[k + 3]  = bg_prior_std;
        mix_prior_std_[k + 6]  = ba_prior_std;
    }
    pose_prior_std_[5] = att_prior_std * 3;
    mix_prior_std_[9]  = sodo_prior_std;
    is_use_prior_      = true;
}

// ---- problem construction -----------------------------------------------------------------------------------------------------
void GVINS::addStateParameters(WindowProblem &problem) { // ic_gvins.cc:1850-1864
    for (auto &statedata : statedatalist_) {
        problem.addParameterBlock(statedata.pose, 7, true);
        problem.addParameterBlock(statedata.mix, 9);
    }
}

void GVINS::addImuFactors(WindowProblem &problem) { // ic_gvins.cc:1866-1889
    for (size_t k = 0; k < preintegrationlist_.size(); k++)
        problem.addResidualBlock(std::make_shared<PreintegrationFactor>(preintegrationlist_[k]), nullptr,
                                 {statedatalist_[k].pose, statedatalist_[k].mix, statedatalist_[k + 1].pose, statedatalist_[k + 1].mix});
    problem.addResidualBlock(std::make_shared<ImuErrorFactor>(), nullptr, {statedatalist_[preintegrationlist_.size()].mix});
    if (is_use_prior_) {
        problem.addResidualBlock(std::make_shared<ImuPosePriorFactor>(pose_prior_, pose_prior_std_), nullptr, {statedatalist_[0].pose});
        problem.addResidualBlock(std::make_shared<ImuMixPriorFactor>(mix_prior_, mix_prior_std_), nullptr, {statedatalist_[0].mix});
    }
}

std::vector<std::pair<WindowSolver::ResidualBlockId, GNSS *>> GVINS::addGnssFactors(WindowProblem &problem, bool isusekernel) { // :1891-1909
    std::vector<std::pair<WindowSolver::ResidualBlockId, GNSS *>> residual_block;
    std::shared_ptr<ceres::LossFunction> loss_function;
    if (isusekernel) loss_function = std::make_shared<HuberLossHip>(1.0);
    for (auto &data : gnsslist_) {
        int index = getStateDataIndex(data.time);
        if (index >= 0) {
            auto id = problem.addResidualBlock(std::make_shared<GnssFactor>(data, antlever_), loss_function, {statedatalist_[(size_t) index].pose});
            residual_block.push_back(std::make_pair(id, &data));
        }
    }
    return residual_block;
}

void GVINS::addReprojectionParameters() { // ic_gvins.cc:1697-1761 (the blocks themselves are registered by registerReprojectionBlocks)
    if (map_->landmarks().empty()) return;
    invdepthlist_.clear();
    for (const auto &landmark : map_->landmarks()) {
        const auto &mappoint = landmark.second;
        if (!mappoint || mappoint->isOutlier()) continue;
        if (invdepthlist_.find(mappoint->id()) == invdepthlist_.end()) {
            auto frame = mappoint->referenceFrame();
            if (!frame || !map_->isKeyFrameInMap(frame)) continue;
            double depth         = mappoint->depth();
            double inverse_depth = 1.0 / depth;
            if (std::isnan(inverse_depth)) {
                mappoint->setOutlier(true);
                continue;
            }
            invdepthlist_[mappoint->id()] = inverse_depth;
            mappoint->addOptimizedTimes();
        }
    }
    extrinsic_[0] = pose_b_c_.t[0], extrinsic_[1] = pose_b_c_.t[1], extrinsic_[2] = pose_b_c_.t[2];
    Quaterniond qic = Rotation::normalized(Rotation::matrix2quaternion(pose_b_c_.R));
    extrinsic_[3] = qic.x, extrinsic_[4] = qic.y, extrinsic_[5] = qic.z, extrinsic_[6] = qic.w;
    extrinsic_[7] = td_b_c_;
}

int GVINS::addReprojectionFactors() { // ic_gvins.cc:1763-1837 (the Huber kernel is the solver's huber_delta); collected, registered later
    visual_batch_->clear();
    visual_factors_.clear();
    visual_invdepth_blocks_.clear();
    visual_blocks_.clear();
    std::unordered_map<const double *, bool> seen;
    if (map_->keyframes().empty()) return 0;
    for (const auto &landmark : map_->landmarks()) {
        const auto &mappoint = landmark.second;
        if (!mappoint || mappoint->isOutlier()) continue;
        auto it = invdepthlist_.find(mappoint->id());
        if (it == invdepthlist_.end()) continue;
        auto ref_frame = mappoint->referenceFrame();
        if (!ref_frame || !map_->isKeyFrameInMap(ref_frame)) continue;
        Vector3d ref_frame_pc = camera_->pixel2cam(mappoint->referenceKeypoint());
        int ref_frame_index   = getStateDataIndex(ref_frame->stamp());
        if (ref_frame_index < 0) continue;
        double *invdepth = &it->second;
        if (*invdepth == 0) *invdepth = 1.0 / MapPoint::DEFAULT_DEPTH;
        auto ref_features = ref_frame->features();
        auto rf           = ref_features.find(mappoint->id());
        if (rf == ref_features.end()) continue; // the reference dereferences end() here
        auto ref_feature = rf->second;
        for (auto &observation : mappoint->observations()) {
            auto obs_feature = observation.lock();
            if (!obs_feature || obs_feature->isOutlier()) continue;
            auto obs_frame = obs_feature->getFrame();
            if (!obs_frame || !obs_frame->isKeyFrame() || !map_->isKeyFrameInMap(obs_frame) || (obs_frame == ref_frame)) continue;
            Vector3d obs_frame_pc = camera_->pixel2cam(obs_feature->keyPoint());
            int obs_frame_index   = getStateDataIndex(obs_frame->stamp());
            if ((obs_frame_index < 0) || (ref_frame_index == obs_frame_index)) continue;
            visual_factors_.emplace_back(new ReprojectionFactor(ref_frame_pc, obs_frame_pc, ref_feature->velocityInPixel(), obs_feature->velocityInPixel(),
                                                                ref_frame->timeDelay(), obs_frame->timeDelay(), optimize_reprojection_error_std_));
            if (!seen[invdepth]) {
                seen[invdepth] = true;
                visual_invdepth_blocks_.push_back(invdepth);
            }
            visual_blocks_.push_back({statedatalist_[(size_t) ref_frame_index].pose, statedatalist_[(size_t) obs_frame_index].pose, invdepth});
        }
    }
    return (int) visual_factors_.size();
}

// the parameter blocks of the visual factors: inverse depths that carry at least one factor (Ceres drops blocks without residuals
// from the reduced program), the extrinsic and the time delay, constant unless estimated in the normal tracking state (:1747-1760)
void GVINS::registerReprojectionBlocks(WindowProblem &problem) {
    for (double *p : visual_invdepth_blocks_) problem.addParameterBlock(p, 1);
    problem.addParameterBlock(extrinsic_, 7, true);
    problem.addParameterBlock(&extrinsic_[7], 1);
    if (!optimize_estimate_extrinsic_ || gvinsstate_ != GVINS_TRACKING_NORMAL) problem.setParameterBlockConstant(extrinsic_);
    if (!optimize_estimate_td_ || gvinsstate_ != GVINS_TRACKING_NORMAL) problem.setParameterBlockConstant(&extrinsic_[7]);
}

void GVINS::doReintegration() { // ic_gvins.cc:1680-1695: every interval that needs it in ONE launch
    std::vector<Preintegration *> dirty;
    for (size_t k = 0; k < preintegrationlist_.size(); k++) {
        IntegrationState state = stateFromData(statedatalist_[k]);
        Vector3d dbg           = preintegrationlist_[k]->deltaState().bg - state.bg;
        Vector3d dba           = preintegrationlist_[k]->deltaState().ba - state.ba;
        if ((dbg.norm() > 6 * integration_parameters_->gyr_bias_std) || (dba.norm() > 6 * integration_parameters_->acc_bias_std)) {
            preintegrationlist_[k]->reintegration(state);
            dirty.push_back(preintegrationlist_[k].get());
        }
    }
    if (!dirty.empty()) integrate(dirty);
}

// ---- the window solve ------------------------------------------------------------------------------------------------------------
// ---- the window solve in phases (ic_gvins.cc:1130-1239) --------------------------------------------------------------------------
int GVINS::windowCameraColumnsBound() const { return 6 * (int) map_->keyframes().size() + 7; }

int GVINS::beginWindowSolve() { // parameters and factors of the visual part (:1150-1154, 1173)
    PhaseTimer pt(phase_ms_, PH_BUILD);
    addReprojectionParameters();
    const int n_visual = addReprojectionFactors();
    counters_.reprojection_factors += n_visual;
    return n_visual;
}

void GVINS::populateWindow(WindowProblem &problem, int n_visual) { // :1148-1176
    PhaseTimer pt(phase_ms_, PH_BUILD);
    addStateParameters(problem);
    if (n_visual > 0) {
        registerReprojectionBlocks(problem);
        for (size_t k = 0; k < visual_factors_.size(); k++)
            problem.addReprojectionFactor(visual_factors_[k].get(), visual_blocks_[k].pose_i, visual_blocks_[k].pose_j, extrinsic_, visual_blocks_[k].invdepth,
                                          &extrinsic_[7]);
    }
    if (last_marginalization_info_ && last_marginalization_info_->isValid())
        problem.addResidualBlock(std::make_shared<MarginalizationFactor>(last_marginalization_info_), nullptr, last_marginalization_parameter_blocks_);
    gnss_blocks_ = addGnssFactors(problem, true);
    addImuFactors(problem);
    GLOG("Add %zu preintegration, %zu GNSS, %d reprojection", preintegrationlist_.size(), gnsslist_.size(), n_visual);
}

void GVINS::betweenWindowSolves(WindowProblem &problem) { // outlier detection for GNSS (:1192-1208; the visual factors are the solver's part)
    gnssOutlierCullingByChi2(problem, gnss_blocks_);
    for (auto &block : gnss_blocks_) problem.removeResidualBlock(block.first);
    addGnssFactors(problem, false);
}

void GVINS::finishWindowSolve(const WindowSolver::Summary &first, const WindowSolver::Summary &second, double first_ms, double second_ms, int chi2_removed) {
    PhaseTimer pt(phase_ms_, PH_FINISH);
    GLOG("%s", first.BriefReport().c_str());
    GLOG("%s", second.BriefReport().c_str());
    iterations_[0] = first.num_successful_steps, iterations_[1] = second.num_successful_steps;
    timecosts_[0] = first_ms, timecosts_[1] = second_ms;
    counters_.chi2_removed += chi2_removed;
    if (!map_->isMaximumKeframes()) doReintegration(); // :1223-1227
    updateParametersFromOptimizer();
    gvinsOutlierCulling();
    gnss_blocks_.clear();
}

bool GVINS::gvinsOptimization(int prepared_n_visual) { // the phases on a WindowSolver of this estimator
    TimeCost timecost;
    // a caller that already ran beginWindowSolve() for this window (lock-step replay) hands its factor count over: running it a
    // second time would count every landmark's addOptimizedTimes() twice and rebuild invdepthlist_
    const int n_visual = prepared_n_visual >= 0 ? prepared_n_visual : beginWindowSolve();
    if (n_visual > 0) { // the reprojection batch has to be complete before the solver is built on it (WindowSolver drives it on the device)
        for (size_t k = 0; k < visual_factors_.size(); k++)
            visual_batch_->add(visual_factors_[k].get(), visual_blocks_[k].pose_i, visual_blocks_[k].pose_j, extrinsic_, visual_blocks_[k].invdepth, &extrinsic_[7]);
        visual_batch_->finalize();
    }
    WindowSolver solver(n_visual > 0 ? visual_batch_.get() : nullptr, 1.0);
    SingleWindowProblem problem(solver);
    populateWindow(problem, n_visual);
    WindowSolver::Options options;
    WindowSolver::Summary first, second;
    timecost.restart();
    options.max_num_iterations = first_num_iterations_;
    {
        PhaseTimer pt(phase_ms_, PH_SOLVE);
        if (!solver.solve(options, &first)) fail("window solve: " + solver.error());
    }
    const double first_ms = timecost.costInMillisecond();
    betweenWindowSolves(problem);
    const int removed = n_visual > 0 ? solver.removeReprojectionFactorsByChi2(5.991) : 0;
    options.max_num_iterations = second_num_iterations_;
    timecost.restart();
    {
        PhaseTimer pt(phase_ms_, PH_SOLVE);
        if (!solver.solve(options, &second)) fail("window solve: " + solver.error());
    }
    finishWindowSolve(first, second, first_ms, timecost.costInMillisecond(), removed);
    return true;
}

void GVINS::gnssOutlierCullingByChi2(WindowProblem &problem, std::vector<std::pair<WindowSolver::ResidualBlockId, GNSS *>> &residual_block) { // :1241-1267
    const double chi2_threshold = 7.815;
    for (auto &block : residual_block) {
        double cost = 0;
        problem.evaluateResidualBlock(block.first, false, &cost);
        double chi2 = cost * 2;
        if (chi2 > chi2_threshold) block.second->std = block.second->std * std::sqrt(chi2 / chi2_threshold);
    }
}

void GVINS::updateParametersFromOptimizer() { // ic_gvins.cc:1299-1389
    if (map_->keyframes().empty()) return;
    if (!visual_factors_.empty()) {
        if (optimize_estimate_td_) td_b_c_ = extrinsic_[7];
        if (optimize_estimate_extrinsic_) {
            Pose ext;
            ext.t = Vector3d(extrinsic_[0], extrinsic_[1], extrinsic_[2]);
            ext.R = Rotation::quaternion2matrix(Rotation::normalized(Quaterniond{extrinsic_[3], extrinsic_[4], extrinsic_[5], extrinsic_[6]}));
            double dt      = (ext.t - pose_b_c_.t).norm();
            Quaterniond dq = Rotation::matrix2quaternion(ext.R * pose_b_c_.R.transpose());
            double dr      = std::sqrt(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z) * R2D;
            if (!((dt > 1.0) || (dr > 5.0))) pose_b_c_ = ext; // 1 m or 5 deg: rejected
            Vector3d euler = Rotation::matrix2euler(ext.R) * R2D;
            extfilesaver_->dump({timelist_.back(), ext.t[0], ext.t[1], ext.t[2], euler[0], euler[1], euler[2], td_b_c_});
            extfilesaver_->flush();
        }
    }
    for (auto &keyframe : map_->keyframes()) {
        auto &frame = keyframe.second;
        int index   = getStateDataIndex(frame->stamp());
        if (index < 0) continue;
        frame->setPose(stateToCameraPose(stateFromData(statedatalist_[(size_t) index]), pose_b_c_));
    }
    for (const auto &landmark : map_->landmarks()) {
        const auto &mappoint = landmark.second;
        if (!mappoint || mappoint->isOutlier()) continue;
        auto frame = mappoint->referenceFrame();
        if (!frame || !map_->isKeyFrameInMap(frame)) continue;
        auto it = invdepthlist_.find(mappoint->id());
        if (it == invdepthlist_.end()) continue;
        double depth = 1.0 / it->second;
        Vector3d pc0 = camera_->pixel2cam(mappoint->referenceKeypoint());
        Vector3d pc00(pc0.x() * depth, pc0.y() * depth, 1.0 * depth);
        mappoint->setPos(Camera::cam2world(pc00, frame->pose()));
        mappoint->updateDepth(depth);
    }
}

bool GVINS::gvinsOutlierCulling() { // ic_gvins.cc:1035-1128: one device launch for all observations of the window
    if (map_->keyframes().empty()) return false;
    std::vector<CullingResult> results;
    std::string err;
    if (!WindowCulling::gvinsOutlierCulling(ctx_, {WindowCulling::Stream{map_, &invdepthlist_}}, reprojection_error_std_, results, &err))
        fail("outlier culling: " + err);
    outliers_[0] = results[0].outlier_mappoints;
    outliers_[1] = results[0].outlier_features;
    GLOG("Culled %d mappoint with %d bad observed features %d, %d, %d", results[0].outlier_mappoints, results[0].outlier_features,
         results[0].by_reference_frame, results[0].by_observation_count, results[0].by_mean_error);
    return true;
}

void GVINS::parametersStatistic() { // ic_gvins.cc:930-1033
    PhaseTimer pt(phase_ms_, PH_STAT);
    std::vector<ReprojectionStatistics> stats;
    std::string err;
    if (map_->orderedKeyFrames().size() < 2) return;
    if (!WindowCulling::reprojectionStatistics(ctx_, {WindowCulling::Stream{map_, &invdepthlist_}}, stats, &err)) fail("statistics: " + err);
    std::vector<double> row = WindowCulling::statisticsRow(map_, stats[0], iterations_, timecosts_, outliers_);
    if (row.empty()) return;
    statfilesaver_->dump(row);
    statfilesaver_->flush();
}

bool GVINS::gvinsRemoveAllSecondNewFrame() { // ic_gvins.cc:1391-1410
    std::vector<ulong> keyframeids = map_->orderedKeyFrames();
    for (auto id : keyframeids) {
        auto frame = map_->keyframes().find(id)->second;
        if ((frame->keyFrameState() == KEYFRAME_REMOVE_SECOND_NEW) || ((frame->numFeatures() == 0) && (id != keyframeids.back()))) {
            unused_time_nodes_.push_back(frame->stamp());
            frame->resetKeyFrame();
            map_->removeKeyFrame(frame, false);
        }
    }
    return true;
}

// ---- marginalization -----------------------------------------------------------------------------------------------------------
bool GVINS::gvinsMarginalization() { // ic_gvins.cc:1412-1678
    MarginalizationJob job;
    marg_batch_->clear();
    beginMarginalization(job, [this](ReprojectionFactor *f, double *pi, double *pj, double *ext, double *invdepth, double *td) {
        marg_batch_->add(f, pi, pj, ext, invdepth, td);
    });
    bool valid;
    {
        PhaseTimer pt(phase_ms_, PH_MARG);
        if (marg_batch_->size() > 0) {
            marg_batch_->finalize();
            job.info->setReprojectionBatch(marg_batch_.get());
        }
        valid = job.info->marginalization();
        job.info->setReprojectionBatch(nullptr);
        marg_batch_->clear();
    }
    finishMarginalization(job, valid);
    return true;
}

// the factors of the marginalization (:1412-1610): the MarginalizationInfo with every host factor, the reprojection factors handed to `sink`
void GVINS::beginMarginalization(MarginalizationJob &job, const MarginalizationSink &sink) {
    PhaseTimer pt(phase_ms_, PH_MARG);
    std::vector<ulong> keyframeids = map_->orderedKeyFrames();
    auto latest_keyframe           = map_->latestKeyFrame();
    latest_keyframe->setKeyFrameState(KEYFRAME_NORMAL);

    auto frame = map_->keyframes().find(keyframeids[1])->second;
    int marg_i = getStateDataIndex(frame->stamp());
    if (marg_i < 1) fail("marginalization: the second keyframe has no time node");
    size_t num_marg  = (size_t) marg_i;
    double last_time = timelist_[num_marg];
    GLOG("Marginalize %zu states, last time %.3f", num_marg, last_time);

    auto marginalization_info = std::make_shared<MarginalizationInfo>();
    std::unordered_map<long, long> parameters_ids;
    long parameters_id = 0;
    auto key           = [](const double *p) { return reinterpret_cast<long>(p); };
    frame              = map_->keyframes().at(keyframeids[0]);
    auto features      = frame->features();
    {
        for (auto &block : last_marginalization_parameter_blocks_) parameters_ids[key(block)] = parameters_id++;
        parameters_ids[key(extrinsic_)]     = parameters_id++;
        parameters_ids[key(extrinsic_ + 7)] = parameters_id++;
        for (const auto &statedata : statedatalist_) {
            parameters_ids[key(statedata.pose)] = parameters_id++;
            parameters_ids[key(statedata.mix)]  = parameters_id++;
        }
        for (auto const &feature : features) {
            auto mappoint = feature.second->getMapPoint();
            if (feature.second->isOutlier() || !mappoint || mappoint->isOutlier()) continue;
            if (mappoint->referenceFrame() != frame) continue;
            double *invdepth               = &invdepthlist_[mappoint->id()];
            parameters_ids[key(invdepth)] = parameters_id++;
        }
        marginalization_info->updateParamtersIds(parameters_ids);
    }

    if (last_marginalization_info_ && last_marginalization_info_->isValid()) { // the previous prior
        std::vector<int> marginalized_index;
        for (size_t i = 0; i < num_marg; i++)
            for (size_t k = 0; k < last_marginalization_parameter_blocks_.size(); k++)
                if (last_marginalization_parameter_blocks_[k] == statedatalist_[i].pose || last_marginalization_parameter_blocks_[k] == statedatalist_[i].mix)
                    marginalized_index.push_back((int) k);
        auto factor = std::make_shared<MarginalizationFactor>(last_marginalization_info_);
        marginalization_info->addResidualBlockInfo(
            std::make_shared<ResidualBlockInfo>(factor, nullptr, last_marginalization_parameter_blocks_, marginalized_index));
    }
    for (auto &gnss : gnsslist_) // GNSS factors on the removed nodes
        for (size_t k = 0; k < num_marg; k++)
            if (isTheSameTimeNode(gnss.time, timelist_[k], MISC::MINIMUM_TIME_INTERVAL)) {
                marginalization_info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(
                    std::make_shared<GnssFactor>(gnss, antlever_), nullptr, std::vector<double *>{statedatalist_[k].pose}, std::vector<int>{0}));
                break;
            }
    for (size_t k = 0; k < num_marg; k++) { // preintegration factors
        std::vector<int> marg_index = (k == (num_marg - 1)) ? std::vector<int>{0, 1} : std::vector<int>{0, 1, 2, 3};
        marginalization_info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(
            std::make_shared<PreintegrationFactor>(preintegrationlist_[k]), nullptr,
            std::vector<double *>{statedatalist_[k].pose, statedatalist_[k].mix, statedatalist_[k + 1].pose, statedatalist_[k + 1].mix}, marg_index));
    }
    if (is_use_prior_) {
        marginalization_info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(
            std::make_shared<ImuPosePriorFactor>(pose_prior_, pose_prior_std_), nullptr, std::vector<double *>{statedatalist_[0].pose}, std::vector<int>{0}));
        marginalization_info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(
            std::make_shared<ImuMixPriorFactor>(mix_prior_, mix_prior_std_), nullptr, std::vector<double *>{statedatalist_[0].mix}, std::vector<int>{0}));
        is_use_prior_ = false;
    }

    // reprojection factors of the landmarks anchored in the oldest keyframe: evaluated and assembled on the device
    std::vector<std::shared_ptr<ReprojectionFactor>> &marg_factors = job.factors;
    // ic_gvins.cc:1556 constructs a HuberLoss here but :1600-1606 hands nullptr to every ResidualBlockInfo: the reference's
    // prior is built from UNCORRECTED reprojection residuals / Jacobians (the device batch therefore runs with delta 0)
    const std::shared_ptr<ceres::LossFunction> loss_function; // null, as the reference passes
    for (auto const &feature : features) {
        auto mappoint = feature.second->getMapPoint();
        if (feature.second->isOutlier() || !mappoint || mappoint->isOutlier()) continue;
        auto ref_frame = mappoint->referenceFrame();
        if (ref_frame != frame) continue;
        Vector3d ref_frame_pc = camera_->pixel2cam(mappoint->referenceKeypoint());
        int ref_frame_index   = getStateDataIndex(ref_frame->stamp());
        if (ref_frame_index < 0) continue;
        auto idit = invdepthlist_.find(mappoint->id());
        if (idit == invdepthlist_.end() || idit->second == 0) {
            GLOG("marginalization: mappoint %lu (type %d, depth %.3f, used %d, observed %d) has no inverse depth in the window", mappoint->id(), (int) mappoint->mapPointType(),
                 mappoint->depth(), mappoint->usedTimes(), mappoint->observedTimes());
            continue;
        }
        double *invdepth  = &idit->second;
        auto ref_features = ref_frame->features();
        auto rf           = ref_features.find(mappoint->id());
        if (rf == ref_features.end()) continue;
        auto ref_feature = rf->second;
        for (auto &observation : mappoint->observations()) {
            auto obs_feature = observation.lock();
            if (!obs_feature || obs_feature->isOutlier()) continue;
            auto obs_frame = obs_feature->getFrame();
            if (!obs_frame || !obs_frame->isKeyFrame() || !map_->isKeyFrameInMap(obs_frame) || (obs_frame == ref_frame)) continue;
            Vector3d obs_frame_pc = camera_->pixel2cam(obs_feature->keyPoint());
            int obs_frame_index   = getStateDataIndex(obs_frame->stamp());
            if ((obs_frame_index < 0) || (ref_frame_index == obs_frame_index)) continue;
            auto factor = std::make_shared<ReprojectionFactor>(ref_frame_pc, obs_frame_pc, ref_feature->velocityInPixel(), obs_feature->velocityInPixel(),
                                                               ref_frame->timeDelay(), obs_frame->timeDelay(), optimize_reprojection_error_std_);
            marg_factors.push_back(factor);
            double *pi = statedatalist_[(size_t) ref_frame_index].pose, *pj = statedatalist_[(size_t) obs_frame_index].pose;
            sink(factor.get(), pi, pj, extrinsic_, invdepth, &extrinsic_[7]);
            marginalization_info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(
                factor, loss_function, std::vector<double *>{pi, pj, extrinsic_, invdepth, &extrinsic_[7]}, std::vector<int>{0, 3}));
        }
    }
    job.device_factors = (int) marg_factors.size();
    job.info           = std::move(marginalization_info);
    job.parameters_ids = std::move(parameters_ids);
    job.num_marg       = num_marg;
    job.last_time      = last_time;
    job.frame          = frame;
}

// what follows MarginalizationInfo::marginalization() (:1612-1678): the prior's retained blocks, the window bookkeeping, the keyframe leaves the map
void GVINS::finishMarginalization(MarginalizationJob &job, bool valid) {
    PhaseTimer pt(phase_ms_, PH_MARG);
    if (!valid) GLOG("marginalization produced no valid prior");
    std::shared_ptr<MarginalizationInfo> marginalization_info = std::move(job.info);
    std::unordered_map<long, long> &parameters_ids            = job.parameters_ids;
    const size_t num_marg                                     = job.num_marg;
    const double last_time                                    = job.last_time;
    auto frame                                                = job.frame;
    auto features                                             = frame->features();
    auto key                                                  = [](const double *p) { return reinterpret_cast<long>(p); };
    job.factors.clear();
    counters_.marginalizations++;

    std::unordered_map<long, double *> address;
    for (size_t k = num_marg; k < statedatalist_.size(); k++) {
        address[parameters_ids[key(statedatalist_[k].pose)]] = statedatalist_[k].pose;
        address[parameters_ids[key(statedatalist_[k].mix)]]  = statedatalist_[k].mix;
    }
    address[parameters_ids[key(extrinsic_)]]     = extrinsic_;
    address[parameters_ids[key(extrinsic_ + 7)]] = &extrinsic_[7];
    last_marginalization_parameter_blocks_       = marginalization_info->getParamterBlocks(address);
    last_marginalization_info_                   = std::move(marginalization_info);

    size_t num_gnss = gnsslist_.size();
    for (size_t k = 0; k < gnsslist_.size(); k++)
        if (gnsslist_[k].time > last_time) {
            num_gnss = k;
            break;
        }
    for (size_t k = 0; k < num_gnss; k++) gnsslist_.pop_front();
    for (size_t k = 0; k < num_marg; k++) {
        timelist_.pop_front();
        statedatalist_.pop_front();
        preintegrationlist_.pop_front();
    }
    for (const auto &feature : features) { // the landmarks that leave with the keyframe (:1653-1671)
        auto mappoint = feature.second->getMapPoint();
        if (feature.second->isOutlier() || !mappoint || mappoint->isOutlier()) continue;
        Vector3d pw = mappoint->pos();
        ptsfilesaver_->dump({pw.x(), pw.y(), pw.z()});
    }
    map_->removeKeyFrame(frame, true);
}

} // namespace icg
