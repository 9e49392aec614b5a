// INS-aided KLT visual front-end on MI355X: the reference's `Tracking` API re-built on the C ABI of
// include/icgvins_hip.h (reference: ic_gvins/ic_gvins/tracking/tracking.h:46-169, tracking.cc).
//
// Public surface kept from the reference: Tracking(camera, map, drawer, configfile, outputpath), track(frame),
// isNewKeyFrame(), isGoodToTrack(...), pose2Tcw(pose), TrackState.  New: the same per-frame algorithm is exposed as
// *stages* separated by the device calls, so that `TrackingBatch` can run many independent camera streams in
// lock-step and hand ONE batched launch per stage to the GPU (preprocess -> [detect] -> LK fwd/bwd -> RANSAC ->
// triangulate -> detect).  track(frame) simply runs the stages for a single stream.
#pragma once
#include <array>
#include <chrono>
#include <cstdio>
#include <string>

#include "model.h"

namespace icg {

typedef enum TrackState { TRACK_FIRST_FRAME, TRACK_INITIALIZING, TRACK_TRACKING, TRACK_PASSED, TRACK_LOST } TrackState;

// tracker keys of config/gvins.yaml:48-57 (+ is_use_visualization :44)
struct TrackingConfig {
    bool track_check_histogram{false};
    double track_min_parallax{20};
    int track_max_features{200};
    double track_max_interval{0.5};
    bool is_use_visualization{false};
    double reprojection_error_std{1.5};
    static bool fromYamlFile(const std::string &path, TrackingConfig &cfg, std::string *err = nullptr);
};

// One stage worth of device work gathered from any number of streams.
struct StageBatch {
    // F1 preprocess
    vector<int32_t> pre_slots;
    vector<const uint8_t *> pre_imgs;
    int pre_stride{0}, pre_channels{1};
    bool pre_device{false};
    bool pre_want_hist{false};
    vector<double> pre_hist;
    // F7 detect
    vector<int32_t> det_slots, det_mask_off{0}, det_quota;
    vector<float> det_mask_pts;
    vector<float> det_out;
    vector<int32_t> det_count;
    // F2/F3 LK forward-backward (+F4 undistort)
    vector<int32_t> lk_prev_slot, lk_next_slot;
    vector<float> lk_prev, lk_guess, lk_out, lk_undist;
    vector<uint8_t> lk_status;
    // lk_base: where this stream's points start in the concatenated call (set when results are split back)
    int lk_base{0};
    // F6 RANSAC
    vector<int32_t> rs_off{0};
    vector<float> rs_p1, rs_p2;
    vector<uint8_t> rs_mask;
    double rs_thresh{1.5}, rs_conf{0.99}; // reprojection_error_std_, 0.99 (tracking.cc:548)
    // F8 triangulate
    vector<int32_t> tri_T0, tri_T1;
    vector<double> tri_Tcw, tri_pc0, tri_pc1, tri_pw;
    void clear();
};

// Thin RAII wrapper over icg_ctx with the stage executor.
class DeviceContext {
public:
    typedef std::shared_ptr<DeviceContext> Ptr;
    DeviceContext(int device, int width, int height, int n_streams, int max_features);
    ~DeviceContext();
    icg_ctx *ctx() { return ctx_; }
    void setCamera(const Camera &cam);
    // runs whatever the batch contains, one ABI call per kind of work; throws std::runtime_error on ABI failure
    void execute(StageBatch &b, const icg_detect_grid &grid, int max_per_job);
    int allocSlot();
    void freeSlot(int s);
    // Kernel-only replay (profiles/r03_kernel_ceiling.json): while recording, every execute() keeps a copy of its work lists; replay()
    // issues the recorded calls again, back to back, with no tracker logic in between — under icg_prof_enable that yields the exclusive
    // device time of every kernel of a step.  Valid until the next frame is tracked (the frame slots still hold the recorded images).
    void record(bool on) {
        recording_ = on;
        if (on) recorded_.clear();
    }
    size_t recorded() const { return recorded_.size(); }
    // `first`: index of the recorded call to start with (the calls of a step only depend on the frame slots, not on each other, so a replay
    // may start anywhere: concurrent replays of several contexts are staggered that way, as free-running groups are)
    void replay(const icg_detect_grid &grid, int max_per_job, size_t first = 0) {
        for (size_t k = 0; k < recorded_.size(); k++) {
            StageBatch copy = recorded_[(first + k) % recorded_.size()];
            const bool was = recording_;
            recording_     = false;
            execute(copy, grid, max_per_job);
            recording_ = was;
        }
    }

private:
    bool recording_{false};
    vector<StageBatch> recorded_;
    icg_ctx *ctx_{nullptr};
    vector<int> free_slots_;
    std::mutex slot_mutex_;
};

class Tracking {
public:
    typedef std::shared_ptr<Tracking> Ptr;

    // reference signature (tracking.h:51); configfile is the flat YAML of config/gvins.yaml
    Tracking(Camera::Ptr camera, Map::Ptr map, Drawer::Ptr drawer, const std::string &configfile, const std::string &outputpath);
    // same with an explicit configuration, a shared device context (several streams on one GPU) and an id space
    Tracking(Camera::Ptr camera, Map::Ptr map, Drawer::Ptr drawer, const TrackingConfig &config, const std::string &outputpath,
             DeviceContext::Ptr device, std::shared_ptr<IdSpace> ids = IdSpace::global());
    ~Tracking();

    TrackState track(Frame::Ptr frame);
    bool isNewKeyFrame() const { return isnewkeyframe_; }
    bool isGoodToTrack(const Point2f &pp, const Pose &pose, const Vector3d &pw, double scale, double depth_scale = 1.0);
    static Matrix4d pose2Tcw(const Pose &pose);

    // ---- staged interface used by TrackingBatch (and by track()) ----
    enum { N_STAGES = 7 };
    void beginFrame(Frame::Ptr frame, StageBatch &next);           // stage 0: queue preprocess
    void advance(int stage, StageBatch &done, StageBatch &next);   // stage 1..6
    bool frameDone() const { return done_; }
    TrackState result() const { return result_; }
    const icg_detect_grid &grid() const { return grid_; }
    int maxFeaturesPerJob() const { return grid_.max_per_block * block_cnts_; }
    const std::shared_ptr<IdSpace> &ids() const { return ids_; }
    size_t numTrackedRefPoints() const { return pts2d_new_.size(); }
    // the un-triangulated candidates carried to the next frame (tracking.h:129 pts2d_new_ / pts2d_ref_), for tests
    const vector<Point2f> &trackedRefPoints() const { return pts2d_new_; }
    const vector<Point2f> &referencePoints() const { return pts2d_ref_; }
    const Frame::Ptr &currentFrame() const { return frame_cur_; }
    const Frame::Ptr &referenceFrame() const { return frame_ref_; }
    // tracker state for engine-vs-engine tests (TableTracker::dumpObjects)
    const Frame::Ptr &previousFrame() const { return frame_pre_; }
    const Frame::Ptr &lastKeyFrame() const { return last_keyframe_; }
    const vector<Frame::Ptr> &referencePointFrames() const { return pts2d_ref_frame_; }
    const vector<Vector2d> &referenceVelocities() const { return velocity_ref_; }
    bool initializing() const { return isinitializing_; }
    double parallaxMap() const { return parallax_map_; }
    double parallaxRef() const { return parallax_ref_; }
    int parallaxMapCounts() const { return parallax_map_counts_; }
    int parallaxRefCounts() const { return parallax_ref_counts_; }

public:
    static constexpr double ASSOCIATE_MAXIUM_DISTANCE      = 1.0;
    static constexpr double ASSOCIATE_MAXIUM_DISTANCE_RATE = 0.05;
    static constexpr double ASSOCIATE_DEPTH_STD            = 0.1;

private:
    void init(const std::string &outputpath);
    // stage bodies
    void onPreprocessDone(StageBatch &done, StageBatch &next);
    void onDetectADone(StageBatch &done, StageBatch &next);
    void onLKDone(StageBatch &done, StageBatch &next);
    void onRansacDone(StageBatch &done, StageBatch &next);
    void onTriangulateDone(StageBatch &done, StageBatch &next);
    void onDetectBDone(StageBatch &done);
    void finish(TrackState st);

    // pieces of the reference algorithm
    bool queueDetection(Frame::Ptr &frame, bool ismask, StageBatch &next);
    void integrateDetection(StageBatch &done);
    void queueTrackMappoint(StageBatch &next);
    bool finishTrackMappoint(StageBatch &done);
    void queueTrackReference(StageBatch &next);
    bool midTrackReference(StageBatch &done, StageBatch &next);
    bool finishTrackReference(StageBatch &done);
    bool queueTriangulation(StageBatch &next);
    void finishTriangulation(StageBatch &done);
    void makeNewFrameQueue(int state, StageBatch &next);
    keyFrameState checkKeyFrameSate();
    void writeLoggingMessage();
    bool doResetTracking();
    void showTracking();
    static bool isGoodDepth(double depth, double scale = 1.0);
    double relativeTranslation();
    double relativeRotation();
    int parallaxFromReferenceKeyPoints(const vector<Point2f> &ref, const vector<Point2f> &cur, double &parallax);
    int parallaxFromReferenceMapPoints(double &parallax);
    double keyPointParallax(const Point2f &pp0, const Point2f &pp1, const Pose &pose0, const Pose &pose1);
    double keyPointParallax(const Point2f &pp0, const Point2f &pp1, const Matrix3d &R10); // R10 = pose1.R^T * pose0.R
    void checkCarriedUndistortion(const char *where);
    bool isOnBorder(const Point2f &pts);
    template <typename T> static void reduceVector(T &vec, const vector<uint8_t> &status);
    void assignSlot(const Frame::Ptr &f);
    void releaseUnusedSlots();

private:
    const double TRACK_BLOCK_SIZE   = 200.0; // tracking.h:112
    const int TRACK_PYRAMID_LEVEL   = 3;     // tracking.h:113
    const double TRACK_MIN_PARALLAX = 10.0;  // tracking.h:114
    const double TRACK_MIN_INTERVAl = 0.08;  // tracking.h:115

    Frame::Ptr frame_cur_, frame_ref_, frame_pre_, last_keyframe_;
    Camera::Ptr camera_;
    Map::Ptr map_;
    Drawer::Ptr drawer_;
    DeviceContext::Ptr device_;
    std::shared_ptr<IdSpace> ids_;

    vector<Point2f> pts2d_cur_, pts2d_new_, pts2d_ref_;
    // undistorted twins of pts2d_ref_ / pts2d_new_, carried through every reduceVector instead of re-running
    // Camera::undistortPoints on unchanged points each frame (tracking.cc:469,520-524,542-544,712-713): a reference point
    // is undistorted once when detected / re-anchored, and a tracked point's undistorted position comes back from the LK
    // kernel (same iteration, bit-identical; ICG_HOST_CHECK=1 re-derives them on the host and compares)
    vector<Point2f> pts2d_ref_undis_, pts2d_new_undis_;
    vector<Frame::Ptr> pts2d_ref_frame_;
    vector<Vector2d> velocity_ref_, velocity_cur_;
    vector<MapPoint::Ptr> tracked_mappoint_, mappoint_matched_;

    int block_cols_, block_rows_, block_cnts_;
    int block_w_, block_h_;
    int track_max_block_features_;
    icg_detect_grid grid_{};

    double parallax_map_{0}, parallax_ref_{0};
    int parallax_map_counts_{0}, parallax_ref_counts_{0};

    bool isnewkeyframe_{false};
    bool isinitializing_{true};
    double histogram_{0};
    int passed_cnt_{0};

    TrackingConfig cfg_;
    int track_min_pixel_distance_;
    double track_max_interval_;

    FILE *logfile_{nullptr};
    std::chrono::steady_clock::time_point t_start_;
    vector<double> logging_data_;

    // ---- per-frame staged state ----
    bool done_{true};
    TrackState result_{TRACK_PASSED};
    Frame::Ptr pending_frame_;
    int pending_slot_{-1};
    vector<int> owned_slots_; // slots currently holding pre/cur/ref images
    enum Mode { M_NONE, M_FIRST, M_INIT, M_TRACK } mode_{M_NONE};
    // detection job bookkeeping
    int det_job_{-1};
    bool det_ismask_{true};
    Frame::Ptr det_frame_;
    // LK bookkeeping
    int lk_map_begin_{0}, lk_map_n_{0}, lk_ref_begin_{0}, lk_ref_n_{0};
    vector<Point2f> tm_pts2d_map_, tm_pts2d_map_undis_;
    vector<MapPointType> tm_type_;
    bool ref_tracked_{false};
    // RANSAC bookkeeping
    int rs_set_{-1};
    vector<Point2f> tr_new_undis_, tr_cur_undis_;
    // triangulation bookkeeping
    keyFrameState kf_state_{KEYFRAME_NONE};
    bool tri_queued_{false};
    int tri_begin_{0};
    vector<int> tri_point_index_;   // k (index into pts2d_cur_) per queued point
    vector<uint8_t> tri_status_;    // final reduceVector status (size pts2d_cur_)
    vector<int> tri_action_;        // per k: 0 = decided early, 1 = needs device result
    vector<Point2f> tri_ref_undis_, tri_cur_undis_;
    int lost_reset_{0};
};

// Stand-in for the part of GVINS that owns the sliding window (ic_gvins.cc:724-747, 1391-1410, 440-448, 1675):
// inserts keyframes, drops REMOVE_SECOND_NEW / empty frames, removes the oldest keyframe with its landmarks when
// the window overflows.  Used by the replay/bench harness so the tracker sees a live window.
class WindowKeeper {
public:
    explicit WindowKeeper(Map::Ptr map) : map_(std::move(map)) {}
    void onFrame(Tracking &tracking, const Frame::Ptr &frame, TrackState st);

private:
    Map::Ptr map_;
};

} // namespace icg
