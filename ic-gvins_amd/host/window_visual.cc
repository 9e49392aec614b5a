// VisualWindow: the map <-> optimizer glue of GVINS for the visual factors.  See window_visual.h.
#include "window_visual.h"

#include <algorithm>
#include <cmath>

namespace icg {

namespace {
// Eigen Quaterniond(Matrix3d) (Rotation::matrix2quaternion), then normalized as ic_gvins.cc:1741 does
void matrixToQuat(const Matrix3d &m, double *q /* x y z w */) {
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t    = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t    = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t;
        q[1] = (m(0, 2) - m(2, 0)) * t;
        q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t    = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[i] = 0.5 * t;
        t    = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; c++) q[c] /= n;
}
Matrix3d quatToMatrix(const double *q) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Matrix3d r;
    r(0, 0) = 1 - (tyy + tzz), r(0, 1) = txy - twz, r(0, 2) = txz + twy;
    r(1, 0) = txy + twz, r(1, 1) = 1 - (txx + tzz), r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy, r(2, 1) = tyz + twx, r(2, 2) = 1 - (txx + tyy);
    return r;
}
} // namespace

void VisualWindow::cameraToBody(const Pose &camera_pose, const Pose &pose_b_c, double *pose7) {
    Matrix3d Rwb = camera_pose.R * pose_b_c.R.transpose();
    Vector3d p   = camera_pose.t - Rwb * pose_b_c.t;
    pose7[0] = p[0], pose7[1] = p[1], pose7[2] = p[2];
    matrixToQuat(Rwb, pose7 + 3);
}

Pose VisualWindow::bodyToCamera(const double *pose7, const Pose &pose_b_c) { // misc.cc:102-108
    Matrix3d R = quatToMatrix(pose7 + 3);
    Pose pose;
    pose.t = Vector3d(pose7[0], pose7[1], pose7[2]) + R * pose_b_c.t;
    pose.R = R * pose_b_c.R;
    return pose;
}

VisualWindow::VisualWindow(Camera::Ptr camera, Map::Ptr map, const Pose &pose_b_c, double td_b_c, double reprojection_error_std)
    : camera_(std::move(camera)), map_(std::move(map)), pose_b_c_(pose_b_c), td_b_c_(td_b_c),
      std_(reprojection_error_std / camera_->focalLength()) /* optimize_reprojection_error_std_, ic_gvins.cc:141 */ {}

void VisualWindow::build() {
    frames_.clear(), index_of_.clear(), poses_.clear(), invdepthlist_.clear(), factors_.clear(), factor_blocks_.clear();
    batch_.reset(new ReprojectionBatch(0));
    for (ulong id : map_->orderedKeyFrames()) { // statedatalist_ is time ordered like the keyframe ids
        auto frame = map_->keyframes().at(id);
        index_of_[frame.get()] = (int) frames_.size();
        frames_.push_back(frame);
        double p7[7];
        cameraToBody(frame->pose(), pose_b_c_, p7);
        poses_.insert(poses_.end(), p7, p7 + 7);
    }
    // addReprojectionParameters (:1702-1733)
    for (const auto &landmark : map_->landmarks()) {
        const auto &mappoint = landmark.second;
        if (!mappoint || mappoint->isOutlier()) continue;
        auto frame = mappoint->referenceFrame();
        if (!frame || !map_->isKeyFrameInMap(frame)) continue;
        double inverse_depth = 1.0 / mappoint->depth();
        if (std::isnan(inverse_depth)) {
            mappoint->setOutlier(true);
            continue;
        }
        invdepthlist_[mappoint->id()] = inverse_depth;
        mappoint->addOptimizedTimes();
    }
    // extrinsic (:1735-1759)
    extrinsic_[0] = pose_b_c_.t[0], extrinsic_[1] = pose_b_c_.t[1], extrinsic_[2] = pose_b_c_.t[2];
    matrixToQuat(pose_b_c_.R, extrinsic_ + 3);
    extrinsic_[7] = td_b_c_;
    // addReprojectionFactors (:1763-1837)
    for (const auto &landmark : map_->landmarks()) {
        const auto &mappoint = landmark.second;
        if (!mappoint || mappoint->isOutlier()) continue;
        auto it = invdepthlist_.find(mappoint->id());
        if (it == invdepthlist_.end()) continue;
        auto ref_frame = mappoint->referenceFrame();
        if (!ref_frame || !map_->isKeyFrameInMap(ref_frame)) continue;
        auto ri = index_of_.find(ref_frame.get());
        if (ri == index_of_.end()) continue; // the reference's getStateDataIndex < 0 check (never true there: hazard H4)
        Vector3d ref_frame_pc = camera_->pixel2cam(mappoint->referenceKeypoint());
        double *invdepth      = &it->second;
        if (*invdepth == 0) *invdepth = 1.0 / MapPoint::DEFAULT_DEPTH;
        auto ref_features = ref_frame->features();
        auto rf           = ref_features.find(mappoint->id());
        if (rf == ref_features.end()) continue; // the reference dereferences end() here; skipped instead
        auto ref_feature = rf->second;
        for (auto &observation : mappoint->observations()) {
            auto obs_feature = observation.lock();
            if (!obs_feature || obs_feature->isOutlier()) continue;
            auto obs_frame = obs_feature->getFrame();
            if (!obs_frame || !obs_frame->isKeyFrame() || !map_->isKeyFrameInMap(obs_frame) || (obs_frame == ref_frame)) continue;
            auto oi = index_of_.find(obs_frame.get());
            if (oi == index_of_.end() || oi->second == ri->second) continue;
            Vector3d obs_frame_pc = camera_->pixel2cam(obs_feature->keyPoint());
            factors_.emplace_back(new ReprojectionFactor(ref_frame_pc, obs_frame_pc, ref_feature->velocityInPixel(), obs_feature->velocityInPixel(),
                                                         ref_frame->timeDelay(), obs_frame->timeDelay(), std_));
            batch_->add(factors_.back().get(), pose(ri->second), pose(oi->second), extrinsic_, invdepth, &extrinsic_[7]);
            factor_blocks_.push_back({pose(ri->second), pose(oi->second), invdepth});
        }
    }
    batch_->finalize();
}

void VisualWindow::addTo(WindowSolver &solver, bool estimate_extrinsic, bool estimate_td) {
    for (int k = 0; k < numKeyFrames(); k++) solver.addParameterBlock(pose(k), 7, true);
    for (auto &kv : invdepthlist_) solver.addParameterBlock(&kv.second, 1);
    solver.addParameterBlock(extrinsic_, 7, true);
    solver.addParameterBlock(&extrinsic_[7], 1);
    if (!estimate_extrinsic) solver.setParameterBlockConstant(extrinsic_);
    if (!estimate_td) solver.setParameterBlockConstant(&extrinsic_[7]);
}

void VisualWindow::addTo(WindowSolverBatch &solver, int w, bool estimate_extrinsic, bool estimate_td) {
    for (int k = 0; k < numKeyFrames(); k++) solver.addParameterBlock(w, pose(k), 7, true);
    for (auto &kv : invdepthlist_) solver.addParameterBlock(w, &kv.second, 1);
    solver.addParameterBlock(w, extrinsic_, 7, true);
    solver.addParameterBlock(w, &extrinsic_[7], 1);
    if (!estimate_extrinsic) solver.setParameterBlockConstant(w, extrinsic_);
    if (!estimate_td) solver.setParameterBlockConstant(w, &extrinsic_[7]);
    for (size_t f = 0; f < factors_.size(); f++)
        solver.addReprojectionFactor(w, factors_[f].get(), factor_blocks_[f].pose_i, factor_blocks_[f].pose_j, extrinsic_, factor_blocks_[f].invdepth,
                                     &extrinsic_[7]);
}

void VisualWindow::updateParametersFromOptimizer() { // :1347-1391
    for (int k = 0; k < numKeyFrames(); k++) frames_[(size_t) k]->setPose(bodyToCamera(pose(k), pose_b_c_));
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

} // namespace icg
