// Data model kept as the drop-in API surface of the front-end:
//   Camera   reference ic_gvins/ic_gvins/tracking/camera.h:36-95, camera.cc:24-157
//   Feature  reference tracking/feature.h:41-118
//   Frame    reference tracking/frame.h:43-175, frame.cc:25-53
//   MapPoint reference tracking/mappoint.h:46-180, mappoint.cc:25-82
//   Map      reference tracking/map.h:33-103, map.cc:25-149
//   Drawer   reference tracking/drawer.h:31-63 (null object: the tracker calls it unconditionally, tracking.cc:515,559)
// Semantics (container types, insertion order, id factories, depth clamps) follow the reference so that indices and
// track ids come out the same.  One deliberate extension: id factories are per `IdSpace` so that several independent
// camera streams can live in one process with placement-invariant ids; IdSpace::global() reproduces
// the reference's process-wide static counters.
#pragma once
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/icgvins_hip.h"
#include "object_pool.h"
#include "types.h"

namespace icg {

using std::vector;

// The reference guards every accessor of Frame / MapPoint / Map with a std::mutex (tracking vs optimisation thread).  The
// critical sections are a handful of loads/stores and the front-end takes ~10 of them per feature per frame, so the same
// protection is provided by a test-and-set lock (one atomic exchange to acquire, one store to release; yields under
// contention) instead of a pthread mutex.
class SpinLock {
public:
    void lock() {
        int spins = 0;
        while (flag_.exchange(true, std::memory_order_acquire)) {
            while (flag_.load(std::memory_order_relaxed)) {
                if (++spins > 64) {
                    std::this_thread::yield();
                    spins = 0;
                } else
                    __builtin_ia32_pause();
            }
        }
    }
    bool try_lock() { return !flag_.exchange(true, std::memory_order_acquire); }
    void unlock() { flag_.store(false, std::memory_order_release); }

private:
    std::atomic<bool> flag_{false};
};
typedef std::lock_guard<SpinLock> ModelLock;

struct IdSpace {
    ulong frame_id{0}, keyframe_id{0}, mappoint_id{0};
    static std::shared_ptr<IdSpace> global();
};

// ---- Camera ------------------------------------------------------------------------------------------------
class Camera {
public:
    typedef std::shared_ptr<Camera> Ptr;
    Camera(const vector<double> &intrinsic, const vector<double> &distortion, const vector<int> &size);
    static Camera::Ptr createCamera(const vector<double> &intrinsic, const vector<double> &distortion,
                                    const vector<int> &size) {
        return std::make_shared<Camera>(intrinsic, distortion, size);
    }
    void undistortPoints(vector<Point2f> &pts) const;
    void distortPoints(vector<Point2f> &pts) const;
    // The per-point maps are defined here (inline): the front-end evaluates them a few thousand times per frame.
    void distortPoint(Point2f &pp) const { // camera.cc:91-102
        Vector3d pc = pixel2cam(pp);
        double x    = pc.x();
        double y    = pc.y();
        double r2   = x * x + y * y;
        double rr   = (1 + k1_ * r2 + k2_ * r2 * r2 + k3_ * r2 * r2 * r2);
        pc[0]       = x * rr + 2 * p1_ * x * y + p2_ * (r2 + 2 * x * x);
        pc[1]       = y * rr + p1_ * (r2 + 2 * y * y) + 2 * p2_ * x * y;
        pp          = cam2pixel(pc);
    }
    Point2f distortCameraPoint(const Vector3d &pc) const { // camera.cc:104-117
        double x  = pc.x() / pc.z();
        double y  = pc.y() / pc.z();
        double r2 = x * x + y * y;
        double rr = (1 + k1_ * r2 + k2_ * r2 * r2 + k3_ * r2 * r2 * r2);
        Vector3d pc1;
        pc1[0] = static_cast<float>(x * rr + 2 * p1_ * x * y + p2_ * (r2 + 2 * x * x));
        pc1[1] = static_cast<float>(y * rr + p1_ * (r2 + 2 * y * y) + 2 * p2_ * x * y);
        pc1[2] = 1.0;
        return cam2pixel(pc1);
    }
    Vector2d reprojectionError(const Pose &pose, const Vector3d &pw, const Point2f &pp) const { // camera.cc:153-157
        Point2f ppp = world2pixel(pw, pose);
        return {ppp.x - pp.x, ppp.y - pp.y};
    }
    static Vector3d world2cam(const Vector3d &world, const Pose &pose) { // camera.cc:145-147: pose.R^T * (world - pose.t)
        Vector3d d        = world - pose.t;
        const Matrix3d &R = pose.R;
        return {R(0, 0) * d[0] + R(1, 0) * d[1] + R(2, 0) * d[2], R(0, 1) * d[0] + R(1, 1) * d[1] + R(2, 1) * d[2],
                R(0, 2) * d[0] + R(1, 2) * d[1] + R(2, 2) * d[2]};
    }
    static Vector3d cam2world(const Vector3d &cam, const Pose &pose) { return pose.R * cam + pose.t; }
    Vector3d pixel2cam(const Point2f &pixel) const { // camera.cc:123-127
        double y = (pixel.y - cy_) / fy_;
        double x = (pixel.x - cx_ - skew_ * y) / fx_;
        return {x, y, 1.0};
    }
    Point2f cam2pixel(const Vector3d &cam) const { // camera.cc:129-131
        return Point2f((float) ((fx_ * cam[0] + skew_ * cam[1]) / cam[2] + cx_), (float) (fy_ * cam[1] / cam[2] + cy_));
    }
    Vector3d pixel2world(const Point2f &pixel, const Pose &pose) const { return cam2world(pixel2cam(pixel), pose); }
    Point2f world2pixel(const Vector3d &world, const Pose &pose) const { return cam2pixel(world2cam(world, pose)); }
    int width() const { return width_; }
    int height() const { return height_; }
    double focalLength() const { return (fx_ + fy_) * 0.5; }
    icg_camera abi() const { return icg_camera{fx_, fy_, cx_, cy_, skew_, k1_, k2_, p1_, p2_, k3_}; }

private:
    double fx_, fy_, cx_, cy_, skew_;
    double k1_, k2_, k3_, p1_, p2_;
    int width_, height_;
};

// ---- Feature -----------------------------------------------------------------------------------------------
class Frame;
class MapPoint;

// prefetch of an object created by make_shared (control block in front of it) that is about to be locked / read: its first lines
// The control block a shared_ptr / weak_ptr refers to, as a PREFETCH ADDRESS ONLY: both libstdc++ and libc++ lay a smart pointer out
// as {object pointer, control-block pointer}; a wrong guess costs a useless prefetch, never a fault.
template <typename P> inline const void *controlBlockHint(const P &smart_pointer) {
    static_assert(sizeof(P) == 2 * sizeof(void *), "unexpected smart-pointer layout");
    const void *words[2];
    memcpy(words, &smart_pointer, sizeof words);
    return words[1];
}
// only the reference counts of such an object (it is about to be released, not read)
inline void prefetchCounts(const void *object) {
    if (object) __builtin_prefetch(static_cast<const char *>(object) - 16, 1);
}
inline void prefetchShared(const void *object) {
    if (!object) return;
    const char *p = static_cast<const char *>(object);
    __builtin_prefetch(p - 16); // use / weak counts
    __builtin_prefetch(p + 48);
    __builtin_prefetch(p + 112);
}

enum FeatureType { FEATURE_NONE = -1, FEATURE_MATCHED = 0, FEATURE_TRIANGULATED = 1, FEATURE_DEPTH_ASSOCIATED = 2 };

class Feature {
public:
    typedef std::shared_ptr<Feature> Ptr;
    Feature(const std::shared_ptr<Frame> &frame, const Vector2d &velocity, Point2f keypoint, Point2f distorted, FeatureType type)
        : frame_(frame), keypoint_(keypoint), distorted_keypoint_(distorted), isoutlier_(false), type_(type) {
        velocity_ = Vector3d(velocity[0], velocity[1], 0);
    }
    static Ptr createFeature(const std::shared_ptr<Frame> &frame, const Vector2d &velocity, const Point2f &keypoint,
                             const Point2f &distorted, FeatureType type) {
        return std::allocate_shared<Feature>(PoolAllocator<Feature>(), frame, velocity, keypoint, distorted, type);
    }
    std::shared_ptr<Frame> getFrame() { return frame_.lock(); }
    std::shared_ptr<MapPoint> getMapPoint() { return mappoint_.lock(); }
    const Point2f &keyPoint() { return keypoint_; }
    const Point2f &distortedKeyPoint() { return distorted_keypoint_; }
    void addMapPoint(const std::shared_ptr<MapPoint> &mappoint) {
        mappoint_      = mappoint;
        mappoint_hint_ = mappoint.get();
    }
    // address of the map point this feature was attached to, for cache prefetching ONLY (never dereferenced: the object may be gone;
    // ownership questions go through getMapPoint()).  The walks over a frame's features chase feature -> map point -> observation
    // through cold memory (hundreds of streams per host), which is what the host layer's time goes into.
    const void *mapPointHint() const { return mappoint_hint_; }
    void setOutlier(bool isoutlier) { isoutlier_ = isoutlier; }
    bool isOutlier() const { return isoutlier_; }
    FeatureType featureType() { return type_; }
    const Vector3d &velocityInPixel() { return velocity_; }
    void setVelocityInPixel(const Point2f &velocity) { velocity_ = Vector3d(velocity.x, velocity.y, 0); }

private:
    std::weak_ptr<Frame> frame_;
    std::weak_ptr<MapPoint> mappoint_;
    const void *mappoint_hint_{nullptr};
    Point2f keypoint_, distorted_keypoint_;
    Vector3d velocity_{0, 0, 0};
    bool isoutlier_;
    FeatureType type_;
};

// ---- Frame -------------------------------------------------------------------------------------------------
enum keyFrameState { KEYFRAME_NONE = 0, KEYFRAME_REMOVE_SECOND_NEW = 1, KEYFRAME_NORMAL = 2, KEYFRAME_REMOVE_OLDEST = 3 };

class Frame {
public:
    typedef std::shared_ptr<Frame> Ptr;
    Frame(ulong id, double stamp, Mat image, std::shared_ptr<IdSpace> ids);
    static Frame::Ptr createFrame(double stamp, const Mat &image, const std::shared_ptr<IdSpace> &ids = IdSpace::global());
    void setKeyFrame(int state);
    void resetKeyFrame() {
        ModelLock lock(frame_mutex_);
        iskeyframe_     = false;
        keyframe_state_ = KEYFRAME_NONE;
    }
    // materialization hook (TableTracker::materialize): keyframe flag / id / state as recorded, no id is drawn from the id space
    void restoreKeyFrame(bool iskeyframe, ulong keyframe_id, int state) {
        ModelLock lock(frame_mutex_);
        iskeyframe_     = iskeyframe;
        keyframe_id_    = keyframe_id;
        keyframe_state_ = state;
    }
    Mat &image() { return image_; }
    // frame.cc:34 deep-copies every incoming image into raw_image_ for the drawer.  Here the copy (0.9 MB per C2 frame on the
    // ingest thread) is made only when someone asked for it — Tracking does when is_use_visualization is set; otherwise
    // rawImage() aliases the caller's buffer and is only valid while that buffer is.
    static void retainRawImages(bool on);
    Mat &rawImage() { return raw_image_; }
    Pose pose() {
        ModelLock lock(frame_mutex_);
        return pose_;
    }
    void setPose(Pose pose) {
        ModelLock lock(frame_mutex_);
        pose_ = pose;
    }
    // the reference's container (frame.h:80-83: unordered_map<ulong, Feature::Ptr>, same iteration order) with pooled nodes
    typedef std::unordered_map<ulong, Feature::Ptr, std::hash<ulong>, std::equal_to<ulong>, PoolAllocator<std::pair<const ulong, Feature::Ptr>>> FeatureMap;
    FeatureMap features() {
        ModelLock lock(frame_mutex_);
        return features_;
    }
    // the entries of features() in the same (container) order, without copying the hash table: the front-end walks a
    // frame's features several times per frame and only needs a stable view of the shared_ptrs
    typedef vector<std::pair<ulong, Feature::Ptr>> FeatureList;
    void featureSnapshot(FeatureList &out) {
        ModelLock lock(frame_mutex_);
        // walking the hash table is a dependent pointer chase through cold nodes; the list is rebuilt only after the feature set
        // changed (once per frame in practice) and copied from contiguous memory afterwards (independent loads: they overlap)
        refreshSnapshotLocked();
        out = snapshot_;
    }
    // Visits (map-point id, feature) in container order, software-pipelined: the feature object 16 ahead and the map point (object +
    // reference counts) of the feature 8 ahead are requested while feature k is visited.
    // The visitor runs on a copy of the list, WITHOUT the frame lock: visitors take map-point locks, grow vectors and drop
    // shared_ptrs (possibly destroying map points / frames), none of which may happen under a non-recursive spin lock that the estimator
    // and drawer threads also take (round-2 review).  The throughput path no longer walks object graphs at all (track_table.h).
    template <typename F> void forEachFeaturePipelined(F &&f) {
        FeatureList view;
        featureSnapshot(view);
        const size_t n = view.size();
        for (size_t k = 0; k < n; k++) {
            if (k + 16 < n) prefetchShared(view[k + 16].second.get());
            if (k + 8 < n) prefetchShared(view[k + 8].second->mapPointHint());
            f(view[k].first, view[k].second);
        }
    }
    // bucket space for n more features up front (no incremental rehashing while a frame is being filled)
    void reserveFeatures(size_t n) {
        ModelLock lock(frame_mutex_);
        features_.reserve(features_.size() + n); // (a rehash changes the iteration order: the cached list is rebuilt)
        snapshot_valid_ = false;
    }
    // visit (map-point id, feature) pairs in place (the lock is held: the visitor must not call back into this frame)
    template <typename F> void forEachFeature(F &&f) {
        ModelLock lock(frame_mutex_);
        for (const auto &kv : features_) f(kv.first, *kv.second);
    }
    void clearFeatures() {
        ModelLock lock(frame_mutex_);
        // destroying ~300 features of a frame that left the window ten keyframes ago touches cold memory three levels deep (hash node ->
        // feature -> its map point's weak count), through locked decrements that do not overlap their misses.  The cached list drops its
        // references first, software-pipelined (feature 16 ahead, its map point's counts 8 ahead); the container's own clear() then finds
        // every line in cache.
        if (snapshot_valid_ && snapshot_.size() == features_.size()) {
            const size_t n = snapshot_.size();
            for (size_t k = 0; k < n; k++) {
                if (k + 16 < n) prefetchShared(snapshot_[k + 16].second.get());
                if (k + 8 < n) prefetchCounts(snapshot_[k + 8].second->mapPointHint());
                snapshot_[k].second.reset();
            }
        }
        features_.clear();
        unupdated_mappoints_.clear();
        snapshot_.clear();
        snapshot_valid_ = false;
    }
    size_t numFeatures() {
        ModelLock lock(frame_mutex_);
        return features_.size();
    }
    const vector<std::shared_ptr<MapPoint>> &unupdatedMappoints() {
        ModelLock lock(frame_mutex_);
        return unupdated_mappoints_;
    }
    void addNewUnupdatedMappoint(const std::shared_ptr<MapPoint> &mappoint) {
        ModelLock lock(frame_mutex_);
        unupdated_mappoints_.push_back(mappoint);
    }
    void addFeature(ulong mappointid, const Feature::Ptr &feature) {
        ModelLock lock(frame_mutex_);
        features_.insert(std::make_pair(mappointid, feature));
        snapshot_valid_ = false;
    }
    double stamp() const { return stamp_; }
    void setStamp(double stamp) { stamp_ = stamp; }
    double timeDelay() const { return td_; }
    void setTimeDelay(double td) { td_ = td; }
    bool isKeyFrame() const { return iskeyframe_; }
    ulong id() const { return id_; }
    ulong keyFrameId() const { return keyframe_id_; }
    void setKeyFrameState(int state) {
        ModelLock lock(frame_mutex_);
        keyframe_state_ = state;
    }
    int keyFrameState() {
        ModelLock lock(frame_mutex_);
        return keyframe_state_;
    }
    // device residency (new): slot of the CLAHE image + pyramid inside the stream's icg_ctx, -1 when not resident
    int deviceSlot() const { return device_slot_; }
    void setDeviceSlot(int slot) { device_slot_ = slot; }

private:
    void refreshSnapshotLocked() {
        if (snapshot_valid_) return;
        snapshot_.clear();
        snapshot_.reserve(features_.size());
        for (const auto &kv : features_) snapshot_.emplace_back(kv.first, kv.second);
        snapshot_valid_ = true;
    }
    int keyframe_state_{KEYFRAME_NORMAL};
    SpinLock frame_mutex_;
    ulong id_, keyframe_id_;
    double stamp_;
    double td_{0};
    Pose pose_;
    Mat image_, raw_image_;
    bool iskeyframe_;
    FeatureMap features_;
    FeatureList snapshot_; // features_ in its iteration order (valid while snapshot_valid_)
    bool snapshot_valid_{false};
    vector<std::shared_ptr<MapPoint>> unupdated_mappoints_;
    std::shared_ptr<IdSpace> ids_;
    int device_slot_{-1};
};

// ---- MapPoint ----------------------------------------------------------------------------------------------
enum MapPointType {
    MAPPOINT_NONE = -1,
    MAPPOINT_TRIANGULATED = 0,
    MAPPOINT_DEPTH_ASSOCIATED = 1,
    MAPPOINT_DEPTH_INITIALIZED = 2,
    MAPPOINT_FIXED = 3
};

class MapPoint {
public:
    typedef std::shared_ptr<MapPoint> Ptr;
    static constexpr double DEFAULT_DEPTH  = 10.0;
    static constexpr double NEAREST_DEPTH  = 1;
    static constexpr double FARTHEST_DEPTH = 200;

    MapPoint(ulong id, const std::shared_ptr<Frame> &ref_frame, Vector3d pos, Point2f keypoint, double depth, MapPointType type);
    static MapPoint::Ptr createMapPoint(std::shared_ptr<Frame> &ref_frame, Vector3d &pos, Point2f &keypoint, double depth,
                                        MapPointType type, const std::shared_ptr<IdSpace> &ids = IdSpace::global());
    Vector3d &pos() {
        ModelLock lock(mappoint_mutex_);
        return pos_;
    }
    void setPos(const Vector3d &p) { // the optimizer's write-back path (ic_gvins.cc:1299)
        ModelLock lock(mappoint_mutex_);
        pos_ = p;
    }
    int observedTimes() const { return observed_times_; }
    ulong id() const { return id_; }
    void addObservation(const Feature::Ptr &feature);
    void increaseUsedTimes() {
        ModelLock lock(mappoint_mutex_);
        used_times_++;
    }
    void decreaseUsedTimes() {
        ModelLock lock(mappoint_mutex_);
        if (used_times_) used_times_--;
    }
    int usedTimes() {
        ModelLock lock(mappoint_mutex_);
        return used_times_;
    }
    void addOptimizedTimes() {
        ModelLock lock(mappoint_mutex_);
        optimized_times_++;
    }
    int optimizedTimes() {
        ModelLock lock(mappoint_mutex_);
        return optimized_times_;
    }
    // materialization hook (TableTracker::materialize): the counters as recorded (addObservation counted the live observations only)
    void restoreCounters(int used, int observed, int optimized, bool outlier) {
        ModelLock lock(mappoint_mutex_);
        used_times_      = used;
        observed_times_  = observed;
        optimized_times_ = optimized;
        isoutlier_       = outlier;
    }
    void removeAllObservations() {
        ModelLock lock(mappoint_mutex_);
        // releasing a weak_ptr is a locked decrement of the observing feature's control block: one cold line per observation, spread
        // over every frame of the window, and locked instructions do not overlap their misses.  Request the lines first.
        for (const auto &w : observations_) __builtin_prefetch(controlBlockHint(w), 1);
        observations_.clear();
    }
    vector<std::weak_ptr<Feature>> observations() {
        ModelLock lock(mappoint_mutex_);
        return observations_;
    }
    // The slot the next addObservation() will write (the list's storage is its own heap block, cold when a stream is touched again): a
    // store that misses is drained by the next locked instruction, i.e. it stalls the reference-count traffic that follows it.
    // Unlocked read of the end pointer, used as a prefetch address only.
    void prefetchObservationSlot() {
        ModelLock lock(mappoint_mutex_); // (the end pointer is written under this lock: no unlocked read, not even for a prefetch address)
        const std::weak_ptr<Feature> *end = observations_.data() + observations_.size();
        __builtin_prefetch(end, 1);
    }
    // observations().back() without copying the whole list (same result; the list grows with every tracked frame)
    bool lastObservation(std::shared_ptr<Feature> &out) {
        ModelLock lock(mappoint_mutex_);
        if (observations_.empty()) return false;
        out = observations_.back().lock();
        return true;
    }
    // !isOutlier() together with lastObservation(), one critical section (parallaxFromReferenceMapPoints)
    bool lastObservationUnlessOutlier(std::shared_ptr<Feature> &out) {
        ModelLock lock(mappoint_mutex_);
        if (isoutlier_ || observations_.empty()) return false;
        out = observations_.back().lock();
        return true;
    }
    // !isOutlier() together with pos() and mapPointType(), one critical section (trackMappoint's candidate scan)
    bool trackingView(Vector3d &pos, MapPointType &type) {
        ModelLock lock(mappoint_mutex_);
        if (isoutlier_) return false;
        pos  = pos_;
        type = mappoint_type_;
        return true;
    }
    void setOutlier(bool isoutlier) {
        ModelLock lock(mappoint_mutex_);
        isoutlier_ = isoutlier;
    }
    bool isOutlier() {
        ModelLock lock(mappoint_mutex_);
        return isoutlier_;
    }
    void setReferenceFrame(const std::shared_ptr<Frame> &frame, Vector3d pos, Point2f keypoint, double depth, MapPointType type);
    double depth() {
        ModelLock lock(mappoint_mutex_);
        return depth_;
    }
    void updateDepth(double depth) {
        ModelLock lock(mappoint_mutex_);
        depth_ = depth;
    }
    ulong referenceFrameId();
    MapPointType &mapPointType() {
        ModelLock lock(mappoint_mutex_);
        return mappoint_type_;
    }
    std::shared_ptr<Frame> referenceFrame() {
        ModelLock lock(mappoint_mutex_);
        return ref_frame_.lock();
    }
    const Point2f &referenceKeypoint() {
        ModelLock lock(mappoint_mutex_);
        return ref_frame_keypoint_;
    }
    bool isNeedUpdate() {
        ModelLock lock(mappoint_mutex_);
        return isneedupdate_;
    }

private:
    vector<std::weak_ptr<Feature>> observations_;
    SpinLock mappoint_mutex_;
    bool isneedupdate_{false};
    Vector3d pos_, pos_tmp_;
    double depth_{DEFAULT_DEPTH}, depth_tmp_{DEFAULT_DEPTH};
    Point2f ref_frame_keypoint_, ref_frame_keypoint_tmp_;
    std::weak_ptr<Frame> ref_frame_, ref_frame_tmp_;
    int optimized_times_, used_times_, observed_times_;
    bool isoutlier_;
    ulong id_;
    MapPointType mappoint_type_{MAPPOINT_NONE}, mappoint_type_tmp_{MAPPOINT_NONE};
};

// ---- Map ---------------------------------------------------------------------------------------------------
class Map {
public:
    typedef std::shared_ptr<Map> Ptr;
    typedef std::unordered_map<ulong, Frame::Ptr> KeyFrames;
    typedef std::unordered_map<ulong, MapPoint::Ptr, std::hash<ulong>, std::equal_to<ulong>, PoolAllocator<std::pair<const ulong, MapPoint::Ptr>>> LandMarks;
    explicit Map(size_t size) : window_size_(size) {}
    void resetWindowSize(size_t size) { window_size_ = size; }
    size_t windowSize() const { return window_size_; }
    void insertKeyFrame(const Frame::Ptr &frame);
    const KeyFrames &keyframes() { return keyframes_; }
    const LandMarks &landmarks() { return landmarks_; }
    vector<ulong> orderedKeyFrames();
    Frame::Ptr oldestKeyFrame(); // NOTE: the reference self-deadlocks here (map.cc:65-69); fixed, no callers upstream
    const Frame::Ptr &latestKeyFrame();
    void removeMappoint(MapPoint::Ptr &mappoint);
    void removeKeyFrame(Frame::Ptr &frame, bool isremovemappoint);
    double mappointObservedRate(const MapPoint::Ptr &mappoint);
    bool isMaximumKeframes() {
        ModelLock lock(map_mutex_);
        return keyframes_.size() > window_size_;
    }
    bool isKeyFrameInMap(const Frame::Ptr &frame) {
        ModelLock lock(map_mutex_);
        return keyframes_.find(frame->keyFrameId()) != keyframes_.end();
    }
    bool isWindowFull() {
        ModelLock lock(map_mutex_);
        return is_window_full_;
    }
    bool isWindowNormal() {
        ModelLock lock(map_mutex_);
        return keyframes_.size() == window_size_;
    }
    // materialization hook (TableTracker::view): keyframes under their keys, and the landmarks so that landmarks_ ITERATES in the order
    // given with `buckets` buckets.  libstdc++ puts a new node at the front of its bucket, or at the front of the whole list when the
    // bucket is empty, and keeps the nodes of one bucket adjacent; inserting back to front into a table that does not rehash on the way
    // therefore rebuilds any order an insert / erase history left behind.
    void restore(const vector<std::pair<ulong, Frame::Ptr>> &keyframes, const vector<MapPoint::Ptr> &landmarks, size_t buckets,
                 const Frame::Ptr &latest, bool is_window_full) {
        ModelLock lock(map_mutex_);
        keyframes_.clear();
        landmarks_ = LandMarks();
        for (const auto &k : keyframes) keyframes_[k.first] = k.second;
        if (buckets > 1) landmarks_.rehash(buckets);
        for (size_t k = landmarks.size(); k-- > 0;) landmarks_.insert(std::make_pair(landmarks[k]->id(), landmarks[k]));
        latest_keyframe_ = latest;
        is_window_full_  = is_window_full;
    }

private:
    SpinLock map_mutex_;
    KeyFrames keyframes_;
    LandMarks landmarks_;
    Frame::Ptr latest_keyframe_;
    size_t window_size_{20};
    bool is_window_full_{false};
};

// ---- Drawer (tracking/drawer.h:31-64; here a null object: every hook has an empty default) ----------------------------------------
class Drawer {
public:
    typedef std::shared_ptr<Drawer> Ptr;
    virtual ~Drawer() = default;
    virtual void run() {}
    virtual void setFinished() {}
    void setMap(Map::Ptr map) { map_ = std::move(map); }
    virtual void addNewFixedMappoint(Vector3d) {}
    virtual void updateMap(const Matrix4d &) {}
    virtual void updateFrame(Frame::Ptr) {}
    virtual void updateTrackedMapPoints(vector<Point2f>, vector<Point2f>, vector<MapPointType>) {}
    virtual void updateTrackedRefPoints(vector<Point2f>, vector<Point2f>) {}

protected:
    Map::Ptr map_;
};

} // namespace icg
