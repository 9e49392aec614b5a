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
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/icgvins_hip.h"
#include "types.h"

namespace icg {

using std::vector;

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
    void distortPoint(Point2f &pp) const;
    Point2f distortCameraPoint(const Vector3d &pc) const;
    Vector2d reprojectionError(const Pose &pose, const Vector3d &pw, const Point2f &pp) const;
    static Vector3d world2cam(const Vector3d &world, const Pose &pose);
    static Vector3d cam2world(const Vector3d &cam, const Pose &pose);
    Vector3d pixel2cam(const Point2f &pixel) const;
    Point2f cam2pixel(const Vector3d &cam) const;
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
        return std::make_shared<Feature>(frame, velocity, keypoint, distorted, type);
    }
    std::shared_ptr<Frame> getFrame() { return frame_.lock(); }
    std::shared_ptr<MapPoint> getMapPoint() { return mappoint_.lock(); }
    const Point2f &keyPoint() { return keypoint_; }
    const Point2f &distortedKeyPoint() { return distorted_keypoint_; }
    void addMapPoint(const std::shared_ptr<MapPoint> &mappoint) { mappoint_ = mappoint; }
    void setOutlier(bool isoutlier) { isoutlier_ = isoutlier; }
    bool isOutlier() const { return isoutlier_; }
    FeatureType featureType() { return type_; }
    const Vector3d &velocityInPixel() { return velocity_; }
    void setVelocityInPixel(const Point2f &velocity) { velocity_ = Vector3d(velocity.x, velocity.y, 0); }

private:
    std::weak_ptr<Frame> frame_;
    std::weak_ptr<MapPoint> mappoint_;
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
        std::unique_lock<std::mutex> lock(frame_mutex_);
        iskeyframe_     = false;
        keyframe_state_ = KEYFRAME_NONE;
    }
    Mat &image() { return image_; }
    Mat &rawImage() { return raw_image_; }
    Pose pose() {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        return pose_;
    }
    void setPose(Pose pose) {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        pose_ = pose;
    }
    std::unordered_map<ulong, Feature::Ptr> features() {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        return features_;
    }
    void clearFeatures() {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        features_.clear();
        unupdated_mappoints_.clear();
    }
    size_t numFeatures() {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        return features_.size();
    }
    const vector<std::shared_ptr<MapPoint>> &unupdatedMappoints() {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        return unupdated_mappoints_;
    }
    void addNewUnupdatedMappoint(const std::shared_ptr<MapPoint> &mappoint) {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        unupdated_mappoints_.push_back(mappoint);
    }
    void addFeature(ulong mappointid, const Feature::Ptr &feature) {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        features_.insert(std::make_pair(mappointid, feature));
    }
    double stamp() const { return stamp_; }
    void setStamp(double stamp) { stamp_ = stamp; }
    double timeDelay() const { return td_; }
    void setTimeDelay(double td) { td_ = td; }
    bool isKeyFrame() const { return iskeyframe_; }
    ulong id() const { return id_; }
    ulong keyFrameId() const { return keyframe_id_; }
    void setKeyFrameState(int state) {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        keyframe_state_ = state;
    }
    int keyFrameState() {
        std::unique_lock<std::mutex> lock(frame_mutex_);
        return keyframe_state_;
    }
    // device residency (new): slot of the CLAHE image + pyramid inside the stream's icg_ctx, -1 when not resident
    int deviceSlot() const { return device_slot_; }
    void setDeviceSlot(int slot) { device_slot_ = slot; }

private:
    int keyframe_state_{KEYFRAME_NORMAL};
    std::mutex frame_mutex_;
    ulong id_, keyframe_id_;
    double stamp_;
    double td_{0};
    Pose pose_;
    Mat image_, raw_image_;
    bool iskeyframe_;
    std::unordered_map<ulong, Feature::Ptr> features_;
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
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return pos_;
    }
    void setPos(const Vector3d &p) { // the optimizer's write-back path (ic_gvins.cc:1299)
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        pos_ = p;
    }
    int observedTimes() const { return observed_times_; }
    ulong id() const { return id_; }
    void addObservation(const Feature::Ptr &feature);
    void increaseUsedTimes() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        used_times_++;
    }
    void decreaseUsedTimes() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        if (used_times_) used_times_--;
    }
    int usedTimes() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return used_times_;
    }
    void addOptimizedTimes() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        optimized_times_++;
    }
    int optimizedTimes() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return optimized_times_;
    }
    void removeAllObservations() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        observations_.clear();
    }
    vector<std::weak_ptr<Feature>> observations() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return observations_;
    }
    // observations().back() without copying the whole list (same result; the list grows with every tracked frame)
    bool lastObservation(std::shared_ptr<Feature> &out) {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        if (observations_.empty()) return false;
        out = observations_.back().lock();
        return true;
    }
    void setOutlier(bool isoutlier) {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        isoutlier_ = isoutlier;
    }
    bool isOutlier() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return isoutlier_;
    }
    void setReferenceFrame(const std::shared_ptr<Frame> &frame, Vector3d pos, Point2f keypoint, double depth, MapPointType type);
    double depth() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return depth_;
    }
    void updateDepth(double depth) {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        depth_ = depth;
    }
    ulong referenceFrameId();
    MapPointType &mapPointType() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return mappoint_type_;
    }
    std::shared_ptr<Frame> referenceFrame() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return ref_frame_.lock();
    }
    const Point2f &referenceKeypoint() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return ref_frame_keypoint_;
    }
    bool isNeedUpdate() {
        std::unique_lock<std::mutex> lock(mappoint_mutex_);
        return isneedupdate_;
    }

private:
    vector<std::weak_ptr<Feature>> observations_;
    std::mutex mappoint_mutex_;
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
    typedef std::unordered_map<ulong, MapPoint::Ptr> LandMarks;
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
        std::unique_lock<std::mutex> lock(map_mutex_);
        return keyframes_.size() > window_size_;
    }
    bool isKeyFrameInMap(const Frame::Ptr &frame) {
        std::unique_lock<std::mutex> lock(map_mutex_);
        return keyframes_.find(frame->keyFrameId()) != keyframes_.end();
    }
    bool isWindowFull() {
        std::unique_lock<std::mutex> lock(map_mutex_);
        return is_window_full_;
    }
    bool isWindowNormal() {
        std::unique_lock<std::mutex> lock(map_mutex_);
        return keyframes_.size() == window_size_;
    }

private:
    std::mutex map_mutex_;
    KeyFrames keyframes_;
    LandMarks landmarks_;
    Frame::Ptr latest_keyframe_;
    size_t window_size_{20};
    bool is_window_full_{false};
};

// ---- Drawer (null object) ------------------------------------------------------------------------------------
class Drawer {
public:
    typedef std::shared_ptr<Drawer> Ptr;
    virtual ~Drawer() = default;
    virtual void updateFrame(Frame::Ptr) {}
    virtual void updateTrackedMapPoints(vector<Point2f>, vector<Point2f>, vector<MapPointType>) {}
    virtual void updateTrackedRefPoints(vector<Point2f>, vector<Point2f>) {}
};

} // namespace icg
