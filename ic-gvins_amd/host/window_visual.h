// The visual half of GVINS's window bookkeeping, per camera stream — the glue between the tracker's map and the optimizer
// (the caller side of rows R1 / f1 / f3):
//   addReprojectionParameters  ic_gvins.cc:1702-1760   inverse depths of the landmarks whose reference frame is in the window
//   addReprojectionFactors     ic_gvins.cc:1763-1837   one ReprojectionFactor per (landmark, observing keyframe != reference)
//   updateParametersFromOptimizer  ic_gvins.cc:1347-1391   keyframe poses, landmark positions / depths written back to the map
// States are the BODY poses of the keyframes (the reference optimizes IMU poses, frames hold camera poses: MISC::stateToCameraPose
// and its inverse with the body->camera extrinsic).  This class serves the keyframe-only windows of the multi-stream refinement
// (TrackingBatch + WindowSolverBatch); the complete estimator with IMU / GNSS time nodes, marginalization and the initialization
// state machine is icg::GVINS (gvins_hip.h), which builds the same factors on its own state list.
#pragma once
#include <memory>
#include <unordered_map>
#include <vector>

#include "factors.h"
#include "model.h"
#include "solver_batch_hip.h"
#include "solver_hip.h"

namespace icg {

class VisualWindow {
public:
    VisualWindow(Camera::Ptr camera, Map::Ptr map, const Pose &pose_b_c, double td_b_c, double reprojection_error_std);

    // keyframe states (ordered by keyframe id), inverse depths and reprojection factors from the current map
    void build();
    // parameter blocks + reprojection batch into a solver; extrinsic and td constant unless asked otherwise (:1748-1759)
    void addTo(WindowSolver &solver, bool estimate_extrinsic = false, bool estimate_td = false);
    // the same into window w of a WindowSolverBatch (many streams per launch): blocks and reprojection factors
    void addTo(WindowSolverBatch &solver, int w, bool estimate_extrinsic = false, bool estimate_td = false);
    ReprojectionBatch *batch() { return batch_.get(); }
    // write the optimized states back into the map (:1347-1391)
    void updateParametersFromOptimizer();

    int numKeyFrames() const { return (int) frames_.size(); }
    int numFactors() const { return (int) factors_.size(); }
    double *pose(int k) { return &poses_[7 * (size_t) k]; } // body pose of keyframe k: p3, q4 xyzw
    const Frame::Ptr &frame(int k) const { return frames_[(size_t) k]; }
    double *extrinsic() { return extrinsic_; }              // 7 + td
    std::unordered_map<ulong, double> &invdepthlist() { return invdepthlist_; }

    // body pose <-> camera pose with the extrinsic (MISC::stateToCameraPose misc.cc:102-108 and its inverse)
    static void cameraToBody(const Pose &camera_pose, const Pose &pose_b_c, double *pose7);
    static Pose bodyToCamera(const double *pose7, const Pose &pose_b_c);

private:
    Camera::Ptr camera_;
    Map::Ptr map_;
    Pose pose_b_c_;
    double td_b_c_, std_;
    std::vector<Frame::Ptr> frames_;
    std::unordered_map<Frame *, int> index_of_;
    std::vector<double> poses_;
    double extrinsic_[8];
    std::unordered_map<ulong, double> invdepthlist_;
    std::vector<std::unique_ptr<ReprojectionFactor>> factors_;
    struct FactorBlocks {
        double *pose_i, *pose_j, *invdepth;
    };
    std::vector<FactorBlocks> factor_blocks_; // the blocks each factor was registered with (for WindowSolverBatch)
    std::unique_ptr<ReprojectionBatch> batch_;
};

} // namespace icg
