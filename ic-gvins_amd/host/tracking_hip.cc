// Staged implementation of the reference front-end state machine on the HIP C ABI.
// Reference control flow: ic_gvins/ic_gvins/tracking/tracking.cc:144-245 (track), :351-455 (trackMappoint),
// :457-574 (trackReferenceFrame), :576-688 (featuresDetection), :690-798 (triangulation), :263-307 (keyframe
// decision).  Every device stage replaces the OpenCV call(s) named in include/icgvins_hip.h; the host code between
// stages keeps the reference's containers and iteration order (std::unordered_map copies, push_back order, stable
// reduceVector) so indices and ids are reproduced.
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "tracking.h"
#include "hostprof.h"

namespace icg {

// ---- configuration ---------------------------------------------------------------------------------------------
static std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

bool TrackingConfig::fromYamlFile(const std::string &path, TrackingConfig &cfg, std::string *err) {
    std::ifstream f(path);
    if (!f) {
        if (err) *err = "cannot open " + path;
        return false;
    }
    std::string line;
    auto as_bool = [](const std::string &v) { return v == "true" || v == "True" || v == "1"; };
    while (std::getline(f, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
        if (val.empty()) continue;
        if (key == "track_check_histogram") cfg.track_check_histogram = as_bool(val);
        else if (key == "track_min_parallax") cfg.track_min_parallax = atof(val.c_str());
        else if (key == "track_max_features") cfg.track_max_features = atoi(val.c_str());
        else if (key == "track_max_interval") cfg.track_max_interval = atof(val.c_str());
        else if (key == "is_use_visualization") cfg.is_use_visualization = as_bool(val);
        else if (key == "reprojection_error_std") cfg.reprojection_error_std = atof(val.c_str());
    }
    return true;
}

void StageBatch::clear() {
    pre_slots.clear();
    pre_imgs.clear();
    pre_hist.clear();
    pre_want_hist = false;
    det_slots.clear();
    det_mask_off.assign(1, 0);
    det_quota.clear();
    det_mask_pts.clear();
    det_out.clear();
    det_count.clear();
    lk_prev_slot.clear();
    lk_next_slot.clear();
    lk_prev.clear();
    lk_guess.clear();
    lk_out.clear();
    lk_undist.clear();
    lk_status.clear();
    rs_off.assign(1, 0);
    rs_p1.clear();
    rs_p2.clear();
    rs_mask.clear();
    tri_T0.clear();
    tri_T1.clear();
    tri_Tcw.clear();
    tri_pc0.clear();
    tri_pc1.clear();
    tri_pw.clear();
}

// ---- device context ----------------------------------------------------------------------------------------------
static void abi_check(icg_ctx *ctx, int rc, const char *what) {
    if (rc != ICG_OK) throw std::runtime_error(std::string(what) + " failed: " + icg_last_error(ctx));
}

DeviceContext::DeviceContext(int device, int width, int height, int n_streams, int max_features) {
    icg_ctx_config cfg{};
    cfg.device      = device;
    cfg.width       = width;
    cfg.height      = height;
    cfg.n_slots     = 4 * n_streams;
    cfg.max_batch   = n_streams;
    cfg.max_points  = std::max(1024, 2 * (max_features + 64) * n_streams);
    cfg.max_factors = 0;
    int rc          = icg_ctx_create(&cfg, &ctx_);
    if (rc != ICG_OK) throw std::runtime_error(std::string("icg_ctx_create failed: ") + icg_last_error(nullptr));
    for (int s = cfg.n_slots - 1; s >= 0; s--) free_slots_.push_back(s);
}

DeviceContext::~DeviceContext() { icg_ctx_destroy(ctx_); }

void DeviceContext::setCamera(const Camera &cam) {
    icg_camera c = cam.abi();
    abi_check(ctx_, icg_set_camera(ctx_, &c), "icg_set_camera");
}

int DeviceContext::allocSlot() {
    std::unique_lock<std::mutex> lock(slot_mutex_);
    if (free_slots_.empty()) throw std::runtime_error("DeviceContext: out of frame slots");
    int s = free_slots_.back();
    free_slots_.pop_back();
    return s;
}
void DeviceContext::freeSlot(int s) {
    std::unique_lock<std::mutex> lock(slot_mutex_);
    free_slots_.push_back(s);
}

void DeviceContext::execute(StageBatch &b, const icg_detect_grid &grid, int max_per_job) {
    if (recording_) recorded_.push_back(b);
    if (!b.pre_slots.empty()) {
        hostprof::Scope hp(hostprof::DEV_PREPROCESS);
        int n = (int) b.pre_slots.size();
        b.pre_hist.assign((size_t) n, 0.0);
        abi_check(ctx_,
                  icg_frames_preprocess(ctx_, n, b.pre_slots.data(), b.pre_imgs.data(), b.pre_stride, b.pre_channels,
                                        b.pre_device ? 1 : 0, b.pre_want_hist ? b.pre_hist.data() : nullptr),
                  "icg_frames_preprocess");
    }
    if (!b.det_slots.empty()) {
        hostprof::Scope hp(hostprof::DEV_DETECT);
        int n = (int) b.det_slots.size();
        b.det_out.assign((size_t) n * max_per_job * 2, 0.f);
        b.det_count.assign((size_t) n, 0);
        abi_check(ctx_,
                  icg_detect(ctx_, n, b.det_slots.data(), &grid, b.det_mask_off.data(), b.det_mask_pts.data(),
                             b.det_quota.data(), max_per_job, b.det_out.data(), b.det_count.data(), nullptr),
                  "icg_detect");
    }
    if (!b.lk_prev_slot.empty()) {
        hostprof::Scope hp(hostprof::DEV_LK);
        int n = (int) b.lk_prev_slot.size();
        b.lk_out.assign((size_t) n * 2, 0.f);
        b.lk_undist.assign((size_t) n * 2, 0.f);
        b.lk_status.assign((size_t) n, 0);
        abi_check(ctx_,
                  icg_lk_track_fb(ctx_, n, b.lk_prev_slot.data(), b.lk_next_slot.data(), b.lk_prev.data(), b.lk_guess.data(), b.lk_out.data(),
                                  b.lk_status.data(), b.lk_undist.data(), nullptr, nullptr),
                  "icg_lk_track_fb");
    }
    if (b.rs_off.size() > 1) {
        hostprof::Scope hp(hostprof::DEV_RANSAC);
        int n = (int) b.rs_off.size() - 1;
        b.rs_mask.assign((size_t) b.rs_off.back(), 1);
        // the one-launch form (every set's whole run inside its workgroup — the kernel of the device-resident tracker) rather than one
        // launch + one wait per RANSAC chunk (icg_fm_ransac): identical masks (tests/test_gpu_geometry.py), one round trip
        abi_check(ctx_, icg_fm_ransac_device(ctx_, n, b.rs_off.data(), b.rs_p1.data(), b.rs_p2.data(), b.rs_thresh, b.rs_conf, b.rs_mask.data()),
                  "icg_fm_ransac_device");
    }
    if (!b.tri_T0.empty()) {
        hostprof::Scope hp(hostprof::DEV_TRIANGULATE);
        int n = (int) b.tri_T0.size();
        b.tri_pw.assign((size_t) n * 3, 0.0);
        abi_check(ctx_,
                  icg_triangulate(ctx_, n, b.tri_T0.data(), b.tri_T1.data(), (int) (b.tri_Tcw.size() / 12), b.tri_Tcw.data(),
                                  b.tri_pc0.data(), b.tri_pc1.data(), b.tri_pw.data()),
                  "icg_triangulate");
    }
}

// ---- construction (tracking.cc:32-86) ----------------------------------------------------------------------------
Tracking::Tracking(Camera::Ptr camera, Map::Ptr map, Drawer::Ptr drawer, const std::string &configfile,
                   const std::string &outputpath)
    : camera_(std::move(camera)), map_(std::move(map)), drawer_(std::move(drawer)), ids_(IdSpace::global()) {
    std::string err;
    if (!TrackingConfig::fromYamlFile(configfile, cfg_, &err)) throw std::runtime_error("Tracking: " + err);
    device_ = std::make_shared<DeviceContext>(0, camera_->width(), camera_->height(), 1, cfg_.track_max_features);
    device_->setCamera(*camera_);
    init(outputpath);
}

Tracking::Tracking(Camera::Ptr camera, Map::Ptr map, Drawer::Ptr drawer, const TrackingConfig &config,
                   const std::string &outputpath, DeviceContext::Ptr device, std::shared_ptr<IdSpace> ids)
    : camera_(std::move(camera)), map_(std::move(map)), drawer_(std::move(drawer)), device_(std::move(device)),
      ids_(std::move(ids)), cfg_(config) {
    init(outputpath);
}

void Tracking::init(const std::string &outputpath) {
    if (!drawer_) drawer_ = std::make_shared<Drawer>(); // the reference dereferences it unconditionally (:515,:559)
    if (cfg_.is_use_visualization) Frame::retainRawImages(true); // the drawer reads frame->rawImage()
    if (!outputpath.empty()) { // :44-50; the reference logs an error and leaves the tracker half-constructed, this one throws
        logfile_ = fopen((outputpath + "/tracking.txt").c_str(), "w");
        if (!logfile_) throw std::runtime_error("Tracking: failed to open " + outputpath + "/tracking.txt");
    }
    track_max_interval_ = cfg_.track_max_interval * 0.95; // :57
    block_cols_ = static_cast<int>(lround(camera_->width() / TRACK_BLOCK_SIZE));  // :66
    block_rows_ = static_cast<int>(lround(camera_->height() / TRACK_BLOCK_SIZE)); // :67
    block_cnts_ = block_cols_ * block_rows_;
    block_h_    = camera_->height() / block_rows_; // :71
    block_w_    = camera_->width() / block_cols_;  // :72
    track_max_block_features_ =
        static_cast<int>(lround(static_cast<double>(cfg_.track_max_features) / static_cast<double>(block_cnts_))); // :81
    track_min_pixel_distance_ = static_cast<int>(round(TRACK_BLOCK_SIZE / sqrt(track_max_block_features_ * 1.5))); // :85
    grid_.block_cols    = block_cols_;
    grid_.block_rows    = block_rows_;
    grid_.block_w       = block_w_;
    grid_.block_h       = block_h_;
    grid_.min_dist      = track_min_pixel_distance_;
    grid_.max_per_block = track_max_block_features_;
}

Tracking::~Tracking() {
    if (logfile_) fclose(logfile_);
    for (int s : owned_slots_) device_->freeSlot(s);
    if (pending_slot_ >= 0) device_->freeSlot(pending_slot_);
}

// ---- helpers ---------------------------------------------------------------------------------------------------
template <typename T> void Tracking::reduceVector(T &vec, const vector<uint8_t> &status) { // :831-839
    size_t index = 0;
    for (size_t k = 0; k < vec.size(); k++)
        if (status[k]) {
            if (index != k) vec[index] = std::move(vec[k]); // (shared_ptr lists: no reference-count round trip per kept element)
            index++;
        }
    vec.resize(index);
}

bool Tracking::isOnBorder(const Point2f &pts) { // :847-849
    return pts.x < 5.0 || pts.y < 5.0 || (pts.x > (camera_->width() - 5.0)) || (pts.y > (camera_->height() - 5.0));
}

bool Tracking::isGoodDepth(double depth, double scale) { // :247-249
    return ((depth > MapPoint::NEAREST_DEPTH) && (depth < MapPoint::FARTHEST_DEPTH * scale));
}

bool Tracking::isGoodToTrack(const Point2f &pp, const Pose &pose, const Vector3d &pw, double scale, double depth_scale) { // :813-829
    Vector3d pc = Camera::world2cam(pw, pose);
    if (!isGoodDepth(pc[2], depth_scale)) return false;
    if (camera_->reprojectionError(pose, pw, pp).norm() > cfg_.reprojection_error_std * scale) return false;
    return true;
}

Matrix4d Tracking::pose2Tcw(const Pose &pose) { // :851-859
    Matrix4d T   = Matrix4d::Zero();
    T(3, 3)      = 1;
    Matrix3d Rt  = pose.R.transpose();
    Vector3d t   = Rt * pose.t;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T(i, j) = Rt(i, j);
        T(i, 3) = -t[i];
    }
    return T;
}

double Tracking::keyPointParallax(const Point2f &pp0, const Point2f &pp1, const Pose &pose0, const Pose &pose1) { // :861-871
    Vector3d pc0  = camera_->pixel2cam(pp0);
    Vector3d pc1  = camera_->pixel2cam(pp1);
    Vector3d pc01 = (pose1.R.transpose() * pose0.R) * pc0;
    return Vector2d(pc01[0] - pc1[0], pc01[1] - pc1[1]).norm() * camera_->focalLength();
}

double Tracking::keyPointParallax(const Point2f &pp0, const Point2f &pp1, const Matrix3d &R10) {
    Vector3d pc0  = camera_->pixel2cam(pp0);
    Vector3d pc1  = camera_->pixel2cam(pp1);
    Vector3d pc01 = R10 * pc0;
    return Vector2d(pc01[0] - pc1[0], pc01[1] - pc1[1]).norm() * camera_->focalLength();
}

int Tracking::parallaxFromReferenceMapPoints(double &parallax) { // :873-905
    parallax   = 0;
    int counts = 0;
    const Matrix3d R10 = frame_cur_->pose().R.transpose() * frame_ref_->pose().R; // loop invariant of :890
    frame_ref_->forEachFeaturePipelined([&](ulong, const Feature::Ptr &feature) {
        auto mappoint = feature->getMapPoint();
        if (mappoint) {
            std::shared_ptr<Feature> feat; // !isOutlier() && observations().back().lock() (:880-887)
            if (!mappoint->lastObservationUnlessOutlier(feat)) return;
            if (feat && !feat->isOutlier()) {
                auto frame = feat->getFrame();
                if (frame && (frame == frame_cur_)) {
                    parallax += keyPointParallax(feature->keyPoint(), feat->keyPoint(), R10);
                    counts++;
                }
            }
        }
    });
    if (counts != 0) parallax /= counts;
    return counts;
}

// ICG_DEBUG_TRI=1: per-candidate triangulation / detection decisions on stderr (used to localise divergences from the
// reference tracker build, tests/ref_tracking_utils.py)
static bool debugTriangulation() {
    static const bool on = getenv("ICG_DEBUG_TRI") != nullptr;
    return on;
}

// ICG_HOST_CHECK=1: the carried undistorted twins must equal a fresh Camera::undistortPoints of their sources, bit for bit
void Tracking::checkCarriedUndistortion(const char *where) {
    static const bool on = getenv("ICG_HOST_CHECK") != nullptr;
    if (!on) return;
    auto same = [&](const vector<Point2f> &src, const vector<Point2f> &carried, const char *what) {
        if (src.size() != carried.size())
            throw std::runtime_error(std::string("carried undistortion size mismatch (") + what + ") at " + where);
        vector<Point2f> u = src;
        camera_->undistortPoints(u);
        for (size_t k = 0; k < u.size(); k++)
            if (memcmp(&u[k], &carried[k], sizeof(Point2f)) != 0)
                throw std::runtime_error(std::string("carried undistortion differs (") + what + ") at " + where);
    };
    same(pts2d_ref_, pts2d_ref_undis_, "ref");
    if (where[0] == 't' && where[2] == 'i') // "triangulation": pts2d_new_ is stale here, pts2d_cur_ is live
        same(pts2d_cur_, tr_cur_undis_, "cur");
    else
        same(pts2d_new_, pts2d_new_undis_, "new");
}

int Tracking::parallaxFromReferenceKeyPoints(const vector<Point2f> &ref, const vector<Point2f> &cur, double &parallax) { // :907-922
    parallax   = 0;
    int counts = 0;
    const Matrix3d R10 = frame_cur_->pose().R.transpose() * frame_ref_->pose().R;
    for (size_t k = 0; k < pts2d_ref_frame_.size(); k++) {
        if (pts2d_ref_frame_[k] == frame_ref_) {
            parallax += keyPointParallax(ref[k], cur[k], R10);
            counts++;
        }
    }
    if (counts != 0) parallax /= counts;
    return counts;
}

double Tracking::relativeTranslation() { return (frame_cur_->pose().t - frame_ref_->pose().t).norm(); } // :331-333

double Tracking::relativeRotation() { // :335-341 (Rotation::matrix2euler pitch component, common/rotation.h:47)
    Matrix3d R   = frame_cur_->pose().R.transpose() * frame_ref_->pose().R;
    double pitch = atan(-R(2, 0) / sqrt(R(2, 1) * R(2, 1) + R(2, 2) * R(2, 2)));
    return fabs(pitch * (180.0 / M_PI));
}

void Tracking::showTracking() { // :343-349
    if (!cfg_.is_use_visualization) return;
    drawer_->updateFrame(frame_cur_);
}

bool Tracking::doResetTracking() { // :317-329
    if (!frame_cur_->numFeatures()) {
        isinitializing_ = true;
        frame_ref_      = frame_cur_;
        pts2d_new_.clear();
        pts2d_ref_.clear();
        pts2d_ref_undis_.clear();
        pts2d_new_undis_.clear();
        pts2d_ref_frame_.clear();
        velocity_ref_.clear();
        return true;
    }
    return false;
}

void Tracking::writeLoggingMessage() { // :309-315 ; FileSaver text format "%-15.9lf " (fileio/filesaver.cc:51-66)
    logging_data_.push_back(static_cast<double>(frame_cur_->numFeatures()));
    logging_data_.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start_).count());
    if (logfile_) {
        for (double v : logging_data_) fprintf(logfile_, "%-15.9lf ", v);
        fprintf(logfile_, "\n");
        fflush(logfile_);
    }
}

keyFrameState Tracking::checkKeyFrameSate() { // :263-307
    keyFrameState keyframe_state = KEYFRAME_NONE;
    double dt                    = frame_cur_->stamp() - last_keyframe_->stamp();
    if (dt < TRACK_MIN_INTERVAl) return keyframe_state;
    double parallax = (parallax_map_ * parallax_map_counts_ + parallax_ref_ * parallax_ref_counts_) /
                      (parallax_map_counts_ + parallax_ref_counts_);
    if (parallax > cfg_.track_min_parallax) {
        keyframe_state = map_->isWindowFull() ? KEYFRAME_REMOVE_OLDEST : KEYFRAME_NORMAL;
    } else if (dt > track_max_interval_) {
        keyframe_state = KEYFRAME_REMOVE_SECOND_NEW;
    }
    if (keyframe_state != KEYFRAME_NONE) {
        last_keyframe_ = frame_cur_;
        for (auto &mappoint : tracked_mappoint_) mappoint->increaseUsedTimes();
        logging_data_.clear();
        logging_data_.push_back(frame_cur_->stamp());
        logging_data_.push_back(dt);
        logging_data_.push_back(parallax);
        logging_data_.push_back(relativeTranslation());
        logging_data_.push_back(relativeRotation());
    }
    return keyframe_state;
}

// ---- device slots: at most {ref, pre, cur, incoming} are resident per stream ----------------------------------------
void Tracking::assignSlot(const Frame::Ptr &f) {
    f->setDeviceSlot(pending_slot_);
    owned_slots_.push_back(pending_slot_);
    pending_slot_ = -1;
}

void Tracking::releaseUnusedSlots() {
    vector<int> keep;
    for (int s : owned_slots_) {
        bool used = (frame_cur_ && frame_cur_->deviceSlot() == s) || (frame_pre_ && frame_pre_->deviceSlot() == s) ||
                    (frame_ref_ && frame_ref_->deviceSlot() == s);
        if (used)
            keep.push_back(s);
        else
            device_->freeSlot(s);
    }
    owned_slots_.swap(keep);
}

// ---- stage 0: preprocessing (tracking.cc:107-142) ------------------------------------------------------------------
void Tracking::beginFrame(Frame::Ptr frame, StageBatch &next) {
    t_start_       = std::chrono::steady_clock::now(); // timecost_.restart() :147
    done_          = false;
    result_        = TRACK_PASSED;
    isnewkeyframe_ = false; // :108
    mode_          = M_NONE;
    det_job_       = -1;
    rs_set_        = -1;
    tri_queued_    = false;
    lk_map_n_ = lk_ref_n_ = 0;
    ref_tracked_   = false;
    pending_frame_ = std::move(frame);
    pending_slot_  = device_->allocSlot();
    Mat &img       = pending_frame_->image();
    next.pre_slots.push_back(pending_slot_);
    next.pre_imgs.push_back(img.data);
    next.pre_stride   = (int) img.step;
    next.pre_channels = img.channels(); // BGR->gray on device (:111-113)
    next.pre_device   = img.device;
    if (cfg_.track_check_histogram) next.pre_want_hist = true;
    // remember which job is ours
    det_job_ = (int) next.pre_slots.size() - 1; // reused as "preprocess job index" until stage 1
}

void Tracking::advance(int stage, StageBatch &done, StageBatch &next) {
    if (done_) return;
    switch (stage) {
    case 1: onPreprocessDone(done, next); break;
    case 2: onDetectADone(done, next); break;
    case 3: onLKDone(done, next); break;
    case 4: onRansacDone(done, next); break;
    case 5: onTriangulateDone(done, next); break;
    case 6: onDetectBDone(done); break;
    default: break;
    }
}

void Tracking::finish(TrackState st) {
    result_ = st;
    done_   = true;
}

void Tracking::onPreprocessDone(StageBatch &done, StageBatch &next) {
    if (cfg_.track_check_histogram) { // :115-133
        double hist = done.pre_hist[(size_t) det_job_];
        if (histogram_ != 0) {
            double rate = fabs((hist - histogram_) / histogram_);
            if (rate > 0.1) {
                passed_cnt_++;
                if (passed_cnt_ > 1) histogram_ = 0;
                device_->freeSlot(pending_slot_);
                pending_slot_ = -1;
                pending_frame_.reset();
                finish(TRACK_PASSED);
                return;
            }
        }
        histogram_ = hist;
    }
    det_job_   = -1;
    frame_pre_ = frame_cur_; // :135
    frame_cur_ = std::move(pending_frame_);
    assignSlot(frame_cur_);
    releaseUnusedSlots();

    if (isinitializing_) {
        if (frame_ref_ == nullptr) { // :158-166
            doResetTracking();
            frame_ref_ = frame_cur_;
            mode_      = M_FIRST;
            queueDetection(frame_ref_, false, next);
            return;
        }
        mode_ = M_INIT;
        if (pts2d_ref_.empty()) queueDetection(frame_ref_, false, next); // :168-170
    } else {
        mode_ = M_TRACK;
    }
}

void Tracking::onDetectADone(StageBatch &done, StageBatch &next) {
    if (det_job_ >= 0) integrateDetection(done);
    if (mode_ == M_FIRST) {
        releaseUnusedSlots();
        finish(TRACK_FIRST_FRAME);
        return;
    }
    if (mode_ == M_TRACK) queueTrackMappoint(next); // :206
    queueTrackReference(next);                      // :173 / :209
}

void Tracking::onLKDone(StageBatch &done, StageBatch &next) {
    if (mode_ == M_TRACK) finishTrackMappoint(done);
    ref_tracked_ = midTrackReference(done, next);
}

void Tracking::onRansacDone(StageBatch &done, StageBatch &next) {
    if (ref_tracked_) finishTrackReference(done);
    if (mode_ == M_INIT) {
        if (parallax_ref_ < cfg_.track_min_parallax) { // :175-178
            showTracking();
            finish(TRACK_INITIALIZING);
            return;
        }
        queueTriangulation(next); // :182
        return;
    }
    kf_state_ = checkKeyFrameSate(); // :212
    if ((kf_state_ == KEYFRAME_NORMAL) || (kf_state_ == KEYFRAME_REMOVE_OLDEST)) queueTriangulation(next); // :215-217
}

void Tracking::onTriangulateDone(StageBatch &done, StageBatch &next) {
    if (tri_queued_) finishTriangulation(done);
    if (mode_ == M_INIT) {
        if (doResetTracking()) { // :184-190
            showTracking();
            lost_reset_ = 1;
            makeNewFrameQueue(KEYFRAME_NORMAL, next);
            return;
        }
        lost_reset_ = 0;
        frame_ref_->setKeyFrame(KEYFRAME_NORMAL); // :193
        makeNewFrameQueue(KEYFRAME_NORMAL, next); // :196
        last_keyframe_  = frame_cur_;
        isinitializing_ = false;
        return;
    }
    // tracking mode (:214-239). featuresDetection never touches frame features, so the lost test (:224) can be
    // evaluated first: when it fires the results of the :220 detection would be wiped by doResetTracking anyway.
    lost_reset_ = 0;
    if (!frame_cur_->numFeatures()) {
        doResetTracking();
        lost_reset_ = 2;
        makeNewFrameQueue(KEYFRAME_NORMAL, next); // :225
        return;
    }
    if ((kf_state_ == KEYFRAME_NORMAL) || (kf_state_ == KEYFRAME_REMOVE_OLDEST)) {
        makeNewFrameQueue(kf_state_, next); // :230-232
    } else {
        queueDetection(frame_cur_, true, next); // :220
        if (kf_state_ != KEYFRAME_NONE) makeNewFrameQueue(kf_state_, next); // REMOVE_SECOND_NEW: flags only
    }
}

void Tracking::onDetectBDone(StageBatch &done) {
    if (det_job_ >= 0) integrateDetection(done);
    releaseUnusedSlots();
    if (mode_ == M_INIT) {
        if (lost_reset_ == 1) {
            finish(TRACK_FIRST_FRAME);
            return;
        }
        showTracking();
        finish(TRACK_TRACKING);
        return;
    }
    if (lost_reset_ == 2) {
        finish(TRACK_LOST);
        return;
    }
    if (kf_state_ != KEYFRAME_NONE) writeLoggingMessage(); // :236-238
    showTracking();
    finish(TRACK_TRACKING);
}

// makeNewFrame (:251-261) with its detection deferred to the batched stage
void Tracking::makeNewFrameQueue(int state, StageBatch &next) {
    frame_cur_->setKeyFrame(state);
    isnewkeyframe_ = true;
    if ((state == KEYFRAME_NORMAL) || (state == KEYFRAME_REMOVE_OLDEST)) {
        frame_ref_ = frame_cur_;
        queueDetection(frame_ref_, true, next);
    }
}

// ---- featuresDetection (:576-688) ------------------------------------------------------------------------------------
bool Tracking::queueDetection(Frame::Ptr &frame, bool ismask, StageBatch &next) {
    det_job_ = -1;
    int num_features = static_cast<int>(frame->numFeatures() + pts2d_ref_.size()); // :579
    if (num_features > (cfg_.track_max_features - 5)) return false;                     // :580
    vector<int> features_cnts((size_t) block_cnts_, 0);
    auto count = [&](float x, float y) {
        int col = int(x / (float) block_w_); // :598
        int row = int(y / (float) block_h_);
        // SURVEY.md hazard H5: the reference indexes features_cnts[row * block_cols_ + col] with the UNCLAMPED column.  The key
        // points counted here are undistorted, so near the right edge x can reach block_cols_ * block_w_ and beyond
        // (col == block_cols_): the reference then counts the feature in the FIRST block of the NEXT row.  That effective
        // behaviour is reproduced (pinned by tests/golden/tracking_ref_*.npz); only indices outside the array — undefined
        // behaviour in the reference (stack VLA overrun) — are dropped.
        const long idx = (long) row * block_cols_ + col;
        if (idx >= 0 && idx < (long) block_cnts_) features_cnts[(size_t) idx]++;
    };
    frame->forEachFeaturePipelined([&](ulong, const Feature::Ptr &feature) { count(feature->keyPoint().x, feature->keyPoint().y); });
    for (auto &pts2d : pts2d_new_) count(pts2d.x, pts2d.y);
    det_job_     = (int) next.det_slots.size();
    det_ismask_  = ismask;
    det_frame_   = frame;
    next.det_slots.push_back(frame->deviceSlot());
    if (ismask) { // :610-620
        frame_cur_->forEachFeaturePipelined([&](ulong, const Feature::Ptr &pt) {
            next.det_mask_pts.push_back(pt->keyPoint().x);
            next.det_mask_pts.push_back(pt->keyPoint().y);
        });
        for (const auto &pts2d : pts2d_new_) {
            next.det_mask_pts.push_back(pts2d.x);
            next.det_mask_pts.push_back(pts2d.y);
        }
    }
    next.det_mask_off.push_back((int32_t) (next.det_mask_pts.size() / 2));
    for (int k = 0; k < block_cnts_; k++) next.det_quota.push_back(track_max_block_features_ - features_cnts[(size_t) k]); // :629
    return true;
}

void Tracking::integrateDetection(StageBatch &done) { // :659-685
    if (!det_ismask_) {
        pts2d_new_.clear();
        pts2d_ref_.clear();
        pts2d_ref_undis_.clear();
        pts2d_new_undis_.clear();
        pts2d_ref_frame_.clear();
        velocity_ref_.clear();
    }
    const int max_per_job = maxFeaturesPerJob();
    const int n           = done.det_count[(size_t) det_job_];
    const float *p        = done.det_out.data() + (size_t) det_job_ * max_per_job * 2;
    vector<Point2f> fresh((size_t) n);
    for (int i = 0; i < n; i++) fresh[(size_t) i] = Point2f(p[2 * i], p[2 * i + 1]);
    vector<Point2f> fresh_undis = fresh;
    camera_->undistortPoints(fresh_undis); // the one undistortion a detected corner ever needs
    for (int i = 0; i < n; i++) {
        pts2d_ref_.push_back(fresh[(size_t) i]);
        pts2d_new_.push_back(fresh[(size_t) i]);
        pts2d_ref_undis_.push_back(fresh_undis[(size_t) i]);
        pts2d_new_undis_.push_back(fresh_undis[(size_t) i]);
        pts2d_ref_frame_.push_back(det_frame_);
        velocity_ref_.emplace_back(0, 0);
    }
    if (debugTriangulation()) fprintf(stderr, "[det] Add %d new features (ref list now %zu)\n", n, pts2d_ref_.size());
    det_job_ = -1;
    det_frame_.reset();
}

// ---- trackMappoint (:351-455) ---------------------------------------------------------------------------------------
void Tracking::queueTrackMappoint(StageBatch &next) {
    mappoint_matched_.clear();
    tm_pts2d_map_.clear();
    tm_pts2d_map_undis_.clear();
    tm_type_.clear();
    vector<Point2f> pts2d_matched;
    Pose pose_cur = frame_cur_->pose();
    pts2d_matched.reserve(frame_pre_->numFeatures());
    frame_pre_->forEachFeaturePipelined([&](ulong, const Feature::Ptr &feature) {
        auto mappoint = feature->getMapPoint();
        Vector3d pos;
        MapPointType type;
        if (mappoint && mappoint->trackingView(pos, type)) { // !isOutlier(), pos(), mapPointType() in one critical section
            tm_pts2d_map_undis_.push_back(feature->keyPoint());
            tm_pts2d_map_.push_back(feature->distortedKeyPoint());
            tm_type_.push_back(type);
            pts2d_matched.emplace_back(camera_->world2pixel(pos, pose_cur)); // INS-aided prediction :367
            mappoint_matched_.push_back(std::move(mappoint));
        }
    });
    lk_map_begin_ = (int) next.lk_prev_slot.size();
    lk_map_n_     = (int) pts2d_matched.size();
    if (pts2d_matched.empty()) return; // :372-375
    camera_->distortPoints(pts2d_matched); // :378
    for (int k = 0; k < lk_map_n_; k++) {
        next.lk_prev_slot.push_back(frame_pre_->deviceSlot());
        next.lk_next_slot.push_back(frame_cur_->deviceSlot());
        next.lk_prev.push_back(tm_pts2d_map_[k].x);
        next.lk_prev.push_back(tm_pts2d_map_[k].y);
        next.lk_guess.push_back(pts2d_matched[k].x);
        next.lk_guess.push_back(pts2d_matched[k].y);
    }
}

bool Tracking::finishTrackMappoint(StageBatch &done) {
    if (lk_map_n_ == 0) return false;
    const int n = lk_map_n_;
    vector<uint8_t> status(done.lk_status.begin() + lk_map_begin_, done.lk_status.begin() + lk_map_begin_ + n);
    vector<Point2f> pts2d_matched((size_t) n), pts2d_matched_undis_all((size_t) n);
    for (int k = 0; k < n; k++) {
        pts2d_matched[k]           = Point2f(done.lk_out[2 * (size_t) (lk_map_begin_ + k)], done.lk_out[2 * (size_t) (lk_map_begin_ + k) + 1]);
        pts2d_matched_undis_all[k] = Point2f(done.lk_undist[2 * (size_t) (lk_map_begin_ + k)], done.lk_undist[2 * (size_t) (lk_map_begin_ + k) + 1]);
    }
    // the device already fused status && status_reverse && !isOnBorder && ||bwd-orig|| < 0.5 (:396-403)
    reduceVector(tm_pts2d_map_, status);
    reduceVector(pts2d_matched, status);
    reduceVector(mappoint_matched_, status);
    reduceVector(tm_type_, status);
    reduceVector(tm_pts2d_map_undis_, status);
    auto pts2d_matched_undis = pts2d_matched_undis_all; // undistortPoints(:423) was applied per point on device
    reduceVector(pts2d_matched_undis, status);

    if (pts2d_matched.empty()) { // :410-419
        if (cfg_.is_use_visualization) drawer_->updateTrackedMapPoints({}, {}, {});
        parallax_map_        = 0;
        parallax_map_counts_ = 0;
        return false;
    }
    {
    hostprof::Scope hp_feat(hostprof::LK_MAP_FEATURES);
    frame_cur_->clearFeatures(); // :426 (no bucket reservation: the container must grow exactly as the reference's does, its iteration
                                 // order is the summation order of the parallax average, :873-905)
    tracked_mappoint_.clear();
    double dt = frame_cur_->stamp() - frame_pre_->stamp();
    for (size_t k = 0; k < pts2d_matched_undis.size(); k++) {
        // two-stage software pipeline over the map points of the list: the object (and its reference counts) 12 ahead, then — once it
        // has arrived — the slot of its observation list that addObservation() will write, 5 ahead
        if (k + 12 < mappoint_matched_.size()) prefetchShared(mappoint_matched_[k + 12].get());
        if (k + 5 < mappoint_matched_.size()) mappoint_matched_[k + 5]->prefetchObservationSlot();
        const MapPoint::Ptr &mappoint = mappoint_matched_[k];
        Vector3d velocity = (camera_->pixel2cam(pts2d_matched_undis[k]) - camera_->pixel2cam(tm_pts2d_map_undis_[k])) / dt;
        auto feature = Feature::createFeature(frame_cur_, Vector2d(velocity.x(), velocity.y()), pts2d_matched_undis[k],
                                              pts2d_matched[k], FEATURE_MATCHED);
        mappoint->addObservation(feature);
        feature->addMapPoint(mappoint);
        frame_cur_->addFeature(mappoint->id(), feature);
    }
    tracked_mappoint_.swap(mappoint_matched_); // every matched map point is a tracked one (:446): the list changes hands, no copies
    mappoint_matched_.clear();
    }
    if (cfg_.is_use_visualization) drawer_->updateTrackedMapPoints(tm_pts2d_map_, pts2d_matched, tm_type_);
    {
        hostprof::Scope hp_par(hostprof::LK_MAP_PARALLAX);
        parallax_map_counts_ = parallaxFromReferenceMapPoints(parallax_map_); // :450
    }
    return true;
}

// ---- trackReferenceFrame (:457-574) -----------------------------------------------------------------------------------
void Tracking::queueTrackReference(StageBatch &next) {
    lk_ref_begin_ = (int) next.lk_prev_slot.size();
    lk_ref_n_     = 0;
    if (pts2d_ref_.empty()) return; // :459-462
    Matrix3d r_cur_pre = frame_cur_->pose().R.transpose() * frame_pre_->pose().R; // :465
    checkCarriedUndistortion("trackReferenceFrame");
    pts2d_cur_.clear();
    for (const auto &pp_pre : pts2d_new_undis_) { // :469 (carried), :472-479
        Vector3d pc_pre = camera_->pixel2cam(pp_pre);
        Vector3d pc_cur = r_cur_pre * pc_pre;
        pts2d_cur_.emplace_back(camera_->distortCameraPoint(pc_cur));
    }
    lk_ref_n_ = (int) pts2d_new_.size();
    for (int k = 0; k < lk_ref_n_; k++) {
        next.lk_prev_slot.push_back(frame_pre_->deviceSlot());
        next.lk_next_slot.push_back(frame_cur_->deviceSlot());
        next.lk_prev.push_back(pts2d_new_[k].x);
        next.lk_prev.push_back(pts2d_new_[k].y);
        next.lk_guess.push_back(pts2d_cur_[k].x);
        next.lk_guess.push_back(pts2d_cur_[k].y);
    }
}

bool Tracking::midTrackReference(StageBatch &done, StageBatch &next) {
    hostprof::Scope hp_ref(hostprof::LK_REF);
    rs_set_ = -1;
    if (lk_ref_n_ == 0) return false;
    const int n = lk_ref_n_;
    vector<uint8_t> status(done.lk_status.begin() + lk_ref_begin_, done.lk_status.begin() + lk_ref_begin_ + n);
    vector<Point2f> cur_undis_all((size_t) n);
    for (int k = 0; k < n; k++) {
        pts2d_cur_[k]    = Point2f(done.lk_out[2 * (size_t) (lk_ref_begin_ + k)], done.lk_out[2 * (size_t) (lk_ref_begin_ + k) + 1]);
        cur_undis_all[k] = Point2f(done.lk_undist[2 * (size_t) (lk_ref_begin_ + k)], done.lk_undist[2 * (size_t) (lk_ref_begin_ + k) + 1]);
    }
    reduceVector(pts2d_ref_, status); // :507-511
    reduceVector(pts2d_cur_, status);
    reduceVector(pts2d_new_, status);
    reduceVector(pts2d_ref_frame_, status);
    reduceVector(velocity_ref_, status);
    reduceVector(cur_undis_all, status);
    reduceVector(pts2d_ref_undis_, status);
    reduceVector(pts2d_new_undis_, status);
    if (pts2d_ref_.empty()) { // :513-517
        drawer_->updateTrackedRefPoints({}, {});
        return false;
    }
    tr_new_undis_ = pts2d_new_undis_; // :520-524 (carried)
    tr_cur_undis_.swap(cur_undis_all);

    velocity_cur_.clear(); // :527-539
    double dt = frame_cur_->stamp() - frame_pre_->stamp();
    for (size_t k = 0; k < tr_cur_undis_.size(); k++) {
        Vector3d vel = (camera_->pixel2cam(tr_cur_undis_[k]) - camera_->pixel2cam(tr_new_undis_[k])) / dt;
        Vector2d velocity(vel.x(), vel.y());
        velocity_cur_.push_back(velocity);
        if (pts2d_ref_frame_[k]->id() > frame_ref_->id()) velocity_ref_[k] = velocity;
    }
    parallax_ref_counts_ = parallaxFromReferenceKeyPoints(pts2d_ref_undis_, tr_cur_undis_, parallax_ref_); // :542-544

    if (pts2d_cur_.size() >= 15) { // :547-548
        rs_set_        = (int) next.rs_off.size() - 1;
        next.rs_thresh = cfg_.reprojection_error_std;
        for (size_t k = 0; k < tr_new_undis_.size(); k++) {
            next.rs_p1.push_back(tr_new_undis_[k].x);
            next.rs_p1.push_back(tr_new_undis_[k].y);
            next.rs_p2.push_back(tr_cur_undis_[k].x);
            next.rs_p2.push_back(tr_cur_undis_[k].y);
        }
        next.rs_off.push_back((int32_t) (next.rs_p1.size() / 2));
    }
    return true;
}

bool Tracking::finishTrackReference(StageBatch &done) {
    if (rs_set_ >= 0) { // :550-554
        vector<uint8_t> status(done.rs_mask.begin() + done.rs_off[(size_t) rs_set_], done.rs_mask.begin() + done.rs_off[(size_t) rs_set_ + 1]);
        reduceVector(pts2d_ref_, status);
        reduceVector(pts2d_cur_, status);
        reduceVector(pts2d_ref_frame_, status);
        reduceVector(velocity_cur_, status);
        reduceVector(velocity_ref_, status);
        reduceVector(pts2d_ref_undis_, status);
        reduceVector(tr_cur_undis_, status);
        rs_set_ = -1;
    }
    if (pts2d_cur_.empty()) { // :557-561
        drawer_->updateTrackedRefPoints({}, {});
        return false;
    }
    if (cfg_.is_use_visualization) drawer_->updateTrackedRefPoints(pts2d_ref_, pts2d_cur_);
    pts2d_new_       = pts2d_cur_; // :569
    pts2d_new_undis_ = tr_cur_undis_;
    return !pts2d_new_.empty();
}

// ---- triangulation (:690-798) -------------------------------------------------------------------------------------------
bool Tracking::queueTriangulation(StageBatch &next) {
    tri_queued_ = false;
    if (pts2d_cur_.empty()) return false; // :692-694
    tri_queued_ = true;
    Pose pose1  = frame_cur_->pose();
    if (tr_cur_undis_.size() != pts2d_cur_.size()) { // no reference tracking ran this frame: derive them on the host
        tr_cur_undis_ = pts2d_cur_;
        camera_->undistortPoints(tr_cur_undis_);
    }
    if (pts2d_ref_undis_.size() != pts2d_ref_.size()) {
        pts2d_ref_undis_ = pts2d_ref_;
        camera_->undistortPoints(pts2d_ref_undis_);
    }
    checkCarriedUndistortion("triangulation");
    tri_ref_undis_ = pts2d_ref_undis_; // :712-713 (carried)
    tri_cur_undis_ = tr_cur_undis_;
    tri_status_.assign(pts2d_cur_.size(), 0);
    tri_action_.assign(pts2d_cur_.size(), 0);
    tri_point_index_.clear();
    tri_begin_ = (int) next.tri_T0.size();

    // one Tcw per distinct reference frame + the current frame, appended to the shared pose table
    const int T_cur = (int) (next.tri_Tcw.size() / 12);
    Matrix4d T1     = pose2Tcw(pose1);
    {
        double t12[12];
        toRowMajor3x4(T1, t12);
        next.tri_Tcw.insert(next.tri_Tcw.end(), t12, t12 + 12);
    }
    std::unordered_map<Frame *, int> T_of;

    for (size_t k = 0; k < pts2d_cur_.size(); k++) {
        auto frame_ref = pts2d_ref_frame_[k];
        if (frame_ref->id() > frame_ref_->id()) { // :723-730 feature added after the reference keyframe: re-anchor
            pts2d_ref_frame_[k] = frame_cur_;
            pts2d_ref_[k]       = pts2d_cur_[k];
            pts2d_ref_undis_[k] = tri_cur_undis_[k];
            tri_status_[k]      = 1;
            if (debugTriangulation()) fprintf(stderr, "[tri] k=%zu reset\n", k);
            continue;
        }
        if (map_->isWindowNormal() && !map_->isKeyFrameInMap(frame_ref)) { // :733-737
            if (debugTriangulation()) fprintf(stderr, "[tri] k=%zu outtime (ref frame id %lu kf %lu iskf %d)\n", k, frame_ref->id(), frame_ref->keyFrameId(), (int) frame_ref->isKeyFrame());
            tri_status_[k] = 0;
            continue;
        }
        Pose pose0      = frame_ref->pose();
        double parallax = keyPointParallax(tri_ref_undis_[k], tri_cur_undis_[k], pose0, pose1); // :741
        if (debugTriangulation()) fprintf(stderr, "[tri] k=%zu parallax=%.17g\n", k, parallax);
        if (parallax < TRACK_MIN_PARALLAX) {
            tri_status_[k] = 1;
            continue;
        }
        auto it = T_of.find(frame_ref.get());
        int T0;
        if (it == T_of.end()) {
            T0 = (int) (next.tri_Tcw.size() / 12);
            Matrix4d T = pose2Tcw(pose0);
            double t12[12];
            toRowMajor3x4(T, t12);
            next.tri_Tcw.insert(next.tri_Tcw.end(), t12, t12 + 12);
            T_of[frame_ref.get()] = T0;
        } else
            T0 = it->second;
        Vector3d pc0 = camera_->pixel2cam(tri_ref_undis_[k]); // :750-751
        Vector3d pc1 = camera_->pixel2cam(tri_cur_undis_[k]);
        next.tri_T0.push_back(T0);
        next.tri_T1.push_back(T_cur);
        for (int c = 0; c < 3; c++) next.tri_pc0.push_back(pc0[c]);
        for (int c = 0; c < 3; c++) next.tri_pc1.push_back(pc1[c]);
        tri_action_[k] = 1;
        tri_point_index_.push_back((int) k);
    }
    return true;
}

void Tracking::finishTriangulation(StageBatch &done) {
    tri_queued_ = false;
    Pose pose1  = frame_cur_->pose();
    for (size_t q = 0; q < tri_point_index_.size(); q++) {
        const size_t k = (size_t) tri_point_index_[q];
        const double *p = &done.tri_pw[3 * (size_t) (tri_begin_ + (int) q)];
        Vector3d pw(p[0], p[1], p[2]);
        auto frame_ref = pts2d_ref_frame_[k];
        Pose pose0     = frame_ref->pose();
        auto pp0 = tri_ref_undis_[k], pp1 = tri_cur_undis_[k];
        if (debugTriangulation()) {
            Vector3d pc0d = Camera::world2cam(pw, pose0), pc1d = Camera::world2cam(pw, pose1);
            fprintf(stderr, "[tri] k=%zu pw=(%.17g %.17g %.17g) z0=%.17g z1=%.17g e0=%.17g e1=%.17g good=%d/%d\n", k, pw[0], pw[1], pw[2],
                    pc0d[2], pc1d[2], camera_->reprojectionError(pose0, pw, pp0).norm(), camera_->reprojectionError(pose1, pw, pp1).norm(),
                    (int) isGoodToTrack(pp0, pose0, pw, 1.0, 3.0), (int) isGoodToTrack(pp1, pose1, pw, 1.0, 3.0));
        }
        if (!isGoodToTrack(pp0, pose0, pw, 1.0, 3.0) || !isGoodToTrack(pp1, pose1, pw, 1.0, 3.0)) { // :756-760
            tri_status_[k] = 0;
            continue;
        }
        tri_status_[k] = 0; // :761 consumed: becomes a map point
        auto pc        = Camera::world2cam(pw, frame_ref->pose());
        double depth   = pc.z();
        auto mappoint  = MapPoint::createMapPoint(frame_ref, pw, tri_ref_undis_[k], depth, MAPPOINT_TRIANGULATED, ids_);
        auto feature = Feature::createFeature(frame_cur_, velocity_cur_[k], tri_cur_undis_[k], pts2d_cur_[k], FEATURE_TRIANGULATED);
        mappoint->addObservation(feature);
        feature->addMapPoint(mappoint);
        frame_cur_->addFeature(mappoint->id(), feature);
        mappoint->increaseUsedTimes();
        feature = Feature::createFeature(frame_ref, velocity_ref_[k], tri_ref_undis_[k], pts2d_ref_[k], FEATURE_TRIANGULATED);
        mappoint->addObservation(feature);
        feature->addMapPoint(mappoint);
        frame_ref->addFeature(mappoint->id(), feature);
        mappoint->increaseUsedTimes();
        frame_cur_->addNewUnupdatedMappoint(mappoint); // :784
    }
    reduceVector(pts2d_ref_, tri_status_); // :788-793
    reduceVector(pts2d_ref_frame_, tri_status_);
    reduceVector(pts2d_cur_, tri_status_);
    reduceVector(velocity_ref_, tri_status_);
    reduceVector(pts2d_ref_undis_, tri_status_);
    reduceVector(tr_cur_undis_, tri_status_);
    pts2d_new_       = pts2d_cur_;
    pts2d_new_undis_ = tr_cur_undis_;
}

// ---- single-stream synchronous API (tracking.cc:144) ------------------------------------------------------------------
TrackState Tracking::track(Frame::Ptr frame) {
    StageBatch a, b;
    StageBatch *done = &a, *next = &b;
    next->clear();
    beginFrame(std::move(frame), *next);
    for (int stage = 1; stage < N_STAGES; stage++) {
        std::swap(done, next);
        device_->execute(*done, grid_, maxFeaturesPerJob());
        next->clear();
        advance(stage, *done, *next);
        if (done_) break;
    }
    return result_;
}

// ---- sliding-window stand-in ---------------------------------------------------------------------------------------
void WindowKeeper::onFrame(Tracking &tracking, const Frame::Ptr &frame, TrackState st) {
    if (!(tracking.isNewKeyFrame() || st == TRACK_FIRST_FRAME || st == TRACK_LOST)) return; // ic_gvins.cc:542
    {
        hostprof::Scope hp(hostprof::KEEP_INSERT);
        map_->insertKeyFrame(frame); // ic_gvins.cc:743
    }
    hostprof::Scope hp_rm(hostprof::KEEP_REMOVE);
    // gvinsRemoveAllSecondNewFrame (ic_gvins.cc:1391-1410)
    vector<ulong> ids = map_->orderedKeyFrames();
    for (auto id : ids) {
        auto it = map_->keyframes().find(id);
        if (it == map_->keyframes().end()) continue;
        auto f = it->second;
        if ((f->keyFrameState() == KEYFRAME_REMOVE_SECOND_NEW) || ((f->numFeatures() == 0) && (id != ids.back()))) {
            f->resetKeyFrame();
            map_->removeKeyFrame(f, false);
        }
    }
    // marginalization side effect on the map (ic_gvins.cc:445-448, 1675): drop the oldest keyframe and its landmarks
    while (map_->isMaximumKeframes()) {
        ids     = map_->orderedKeyFrames();
        auto f  = map_->keyframes().find(ids[0])->second;
        map_->removeKeyFrame(f, true);
    }
}

} // namespace icg
