// Host-side data model implementation (see model.h for the reference lines each class mirrors).
#include "model.h"

#include <atomic>

namespace icg {

std::shared_ptr<IdSpace> IdSpace::global() {
    static std::shared_ptr<IdSpace> g = std::make_shared<IdSpace>();
    return g;
}

// ---- Camera (tracking/camera.cc) -----------------------------------------------------------------------------
Camera::Camera(const vector<double> &intrinsic, const vector<double> &distortion, const vector<int> &size) {
    fx_   = intrinsic[0];
    fy_   = intrinsic[1];
    cx_   = intrinsic[2];
    cy_   = intrinsic[3];
    skew_ = intrinsic.size() == 5 ? intrinsic[4] : 0.0; // camera.cc:51-57
    k1_   = distortion[0];
    k2_   = distortion[1];
    p1_   = distortion[2];
    p2_   = distortion[3];
    k3_   = distortion.size() == 5 ? distortion[4] : 0.0; // camera.cc:60-66
    width_  = size[0];
    height_ = size[1];
}

// cv::undistortPoints(pts, pts, K, D, noArray(), K): 5 fixed-point iterations
void Camera::undistortPoints(vector<Point2f> &pts) const {
    const double ifx = 1. / fx_, ify = 1. / fy_;
    for (auto &pt : pts) {
        double x = pt.x, y = pt.y;
        const double u = x, v = y;
        x = (x - cx_) * ifx;
        y = (y - cy_) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2     = x * x + y * y;
            double icdist = 1. / (1 + ((k3_ * r2 + k2_) * r2 + k1_) * r2);
            if (icdist < 0) {
                x = (u - cx_) * ifx;
                y = (v - cy_) * ify;
                break;
            }
            double deltaX = 2 * p1_ * x * y + p2_ * (r2 + 2 * x * x);
            double deltaY = p1_ * (r2 + 2 * y * y) + 2 * p2_ * x * y;
            x             = (x0 - deltaX) * icdist;
            y             = (y0 - deltaY) * icdist;
        }
        pt.x = (float) (fx_ * x + skew_ * y + cx_);
        pt.y = (float) (fy_ * y + cy_);
    }
}

void Camera::distortPoints(vector<Point2f> &pts) const { // camera.cc:76-89
    for (auto &pt : pts) distortPoint(pt);
}

// ---- Frame (tracking/frame.cc) -------------------------------------------------------------------------------
namespace {
std::atomic<bool> g_retain_raw_images{false};
}
void Frame::retainRawImages(bool on) { g_retain_raw_images.store(on); }

Frame::Frame(ulong id, double stamp, Mat image, std::shared_ptr<IdSpace> ids)
    : id_(id), keyframe_id_(0), stamp_(stamp), image_(std::move(image)), iskeyframe_(false), ids_(std::move(ids)) {
    if (g_retain_raw_images.load(std::memory_order_relaxed))
        image_.copyTo(raw_image_);
    else
        raw_image_ = image_;
}

Frame::Ptr Frame::createFrame(double stamp, const Mat &image, const std::shared_ptr<IdSpace> &ids) {
    return std::make_shared<Frame>(ids->frame_id++, stamp, image, ids);
}

void Frame::setKeyFrame(int state) {
    ModelLock lock(frame_mutex_);
    if (!iskeyframe_) {
        iskeyframe_     = true;
        keyframe_id_    = ids_->keyframe_id++;
        keyframe_state_ = state;
    }
}

// ---- MapPoint (tracking/mappoint.cc) -------------------------------------------------------------------------
MapPoint::MapPoint(ulong id, const std::shared_ptr<Frame> &ref_frame, Vector3d pos, Point2f keypoint, double depth,
                   MapPointType type)
    : pos_(pos), depth_(depth), ref_frame_keypoint_(keypoint), ref_frame_(ref_frame), optimized_times_(0), used_times_(0),
      observed_times_(0), isoutlier_(false), id_(id), mappoint_type_(type) {
    if ((depth_ < NEAREST_DEPTH) || (depth_ > FARTHEST_DEPTH)) depth_ = DEFAULT_DEPTH;
    observations_.reserve(16); // one observation per tracked frame: skip the 1-2-4-8 reallocation ladder
}

MapPoint::Ptr MapPoint::createMapPoint(std::shared_ptr<Frame> &ref_frame, Vector3d &pos, Point2f &feature, double depth,
                                       MapPointType type, const std::shared_ptr<IdSpace> &ids) {
    return std::allocate_shared<MapPoint>(PoolAllocator<MapPoint>(), ids->mappoint_id++, ref_frame, pos, feature, depth, type);
}

void MapPoint::addObservation(const Feature::Ptr &feature) {
    ModelLock lock(mappoint_mutex_);
    observations_.push_back(feature);
    observed_times_++;
}

void MapPoint::setReferenceFrame(const std::shared_ptr<Frame> &frame, Vector3d pos, Point2f keypoint, double depth,
                                 MapPointType type) {
    ModelLock lock(mappoint_mutex_);
    depth_tmp_ = depth;
    if (depth_tmp_ < 1.0) depth_tmp_ = DEFAULT_DEPTH;
    pos_tmp_                = pos;
    ref_frame_tmp_          = frame;
    ref_frame_keypoint_tmp_ = keypoint;
    mappoint_type_tmp_      = type;
    isneedupdate_           = true;
}

ulong MapPoint::referenceFrameId() {
    ModelLock lock(mappoint_mutex_);
    auto frame = ref_frame_.lock();
    return frame ? frame->id() : 0;
}

// ---- Map (tracking/map.cc) -----------------------------------------------------------------------------------
void Map::insertKeyFrame(const Frame::Ptr &frame) {
    ModelLock lock(map_mutex_);
    latest_keyframe_ = frame;
    if (keyframes_.find(frame->keyFrameId()) == keyframes_.end())
        keyframes_.insert(std::make_pair(frame->keyFrameId(), frame));
    else
        keyframes_[frame->keyFrameId()] = frame;
    auto &unupdated_mappoints = frame->unupdatedMappoints();
    for (const auto &mappoint : unupdated_mappoints) {
        if (landmarks_.find(mappoint->id()) == landmarks_.end())
            landmarks_.insert(std::make_pair(mappoint->id(), mappoint));
        else
            landmarks_[mappoint->id()] = mappoint;
    }
    if (keyframes_.size() > window_size_) is_window_full_ = true;
}

vector<ulong> Map::orderedKeyFrames() {
    ModelLock lock(map_mutex_);
    vector<ulong> keyframeid;
    for (auto &keyframe : keyframes_) keyframeid.push_back(keyframe.first);
    std::sort(keyframeid.begin(), keyframeid.end());
    return keyframeid;
}

Frame::Ptr Map::oldestKeyFrame() {
    auto ids = orderedKeyFrames();
    ModelLock lock(map_mutex_);
    return ids.empty() ? nullptr : keyframes_.at(ids[0]);
}

const Frame::Ptr &Map::latestKeyFrame() {
    ModelLock lock(map_mutex_);
    return latest_keyframe_;
}

void Map::removeMappoint(MapPoint::Ptr &mappoint) {
    ModelLock lock(map_mutex_);
    mappoint->setOutlier(true);
    mappoint->removeAllObservations();
    if (landmarks_.find(mappoint->id()) != landmarks_.end()) landmarks_.erase(mappoint->id());
    mappoint.reset();
}

void Map::removeKeyFrame(Frame::Ptr &frame, bool isremovemappoint) {
    ModelLock lock(map_mutex_);
    if (isremovemappoint) {
        vector<ulong> mappointid;
        const Frame *self = frame.get();
        frame->forEachFeaturePipelined([&](ulong, const Feature::Ptr &feature) {
            auto mappoint = feature->getMapPoint();
            if (mappoint && mappoint->referenceFrame().get() == self) mappointid.push_back(mappoint->id());
        });
        for (auto id : mappointid) {
            auto landmark = landmarks_.find(id);
            if (landmark != landmarks_.end()) {
                auto mappoint = landmark->second;
                if (mappoint) {
                    mappoint->removeAllObservations();
                    mappoint->setOutlier(true);
                    landmarks_.erase(id);
                }
            }
        }
        frame->clearFeatures();
    }
    keyframes_.erase(frame->keyFrameId());
    frame.reset();
}

double Map::mappointObservedRate(const MapPoint::Ptr &mappoint) {
    ModelLock lock(map_mutex_);
    size_t num_keyframes = keyframes_.size();
    size_t num_observed  = 0;
    auto features        = mappoint->observations();
    for (auto &feature : features) {
        auto feat = feature.lock();
        if (!feat) continue;
        auto frame = feat->getFrame();
        if (!frame) continue;
        if (keyframes_.find(frame->keyFrameId()) != keyframes_.end()) num_observed += 1;
    }
    return static_cast<double>(num_observed) / static_cast<double>(num_keyframes);
}

} // namespace icg
