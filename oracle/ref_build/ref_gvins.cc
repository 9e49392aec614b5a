// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own estimator: GVINS (/root/reference/ic_gvins/ic_gvins/ic_gvins.{h,cc}) with
// misc.cc, the four preintegration variants, the factors and the tracker sources, all compiled unmodified from where they lie, behind one
// C entry point that plays a recorded sequence into it the way the ROS shell does (ROS/fusion_ros.cc:123-234).  What the build replaces is
// stated per shim: Eigen / yaml-cpp / glog / absl / tbb interfaces (shim/), the OpenCV image primitives (ref_tracking.cc: forwarded to the
// CPU restatement, so they stay unpinned) and Ceres — shim/ceres/problem_shim.h restates the published Levenberg-Marquardt loop on dense
// normal equations, independently of the product's solver.  The estimator's three threads run as in the reference, so the result depends
// on thread timing; the driver paces the input slower than real time so that tracking and optimization finish between events, which is
// the schedule the product's deterministic event loop follows.  SURVEY.md §8 row f2.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "ref_tracking.cc" // the tracker sources + the OpenCV entry points on the oracle primitives (and its own C entry points)

#include "preintegration/preintegration_base.cc"
#include "preintegration/preintegration_earth.cc"
#include "preintegration/preintegration_earth_odo.cc"
#include "preintegration/preintegration_normal.cc"
#include "preintegration/preintegration_odo.cc"

#include "misc.cc"

#include "ic_gvins.cc"

static Frame::Ptr ref_make_frame(double stamp, const uint8_t *gray, int w, int h) {
    Mat image(h, w, CV_8UC1);
    memcpy(image.data, gray, (size_t) w * h);
    return Frame::createFrame(stamp, image);
}
#define REF_GVINS_RUN_NAME ref_gvins_run
#include "ref_gvins_driver.inc"

// ---- f1 cross-check: one sliding window of reprojection factors + pose priors through the REFERENCE's own factor code
// (factors/reprojection_factor.h, factors/pose_parameterization.h, preintegration/imu_pose_prior_factor.h, ceres::HuberLoss) and the shim's
// Levenberg-Marquardt (shim/ceres/problem_shim.h), following GVINS::gvinsOptimization's two solves with the chi-square removal in between
// (ic_gvins.cc:1178-1221, 1269-1297).  Same argument layout as icgh_backend_solve (ic-gvins_amd/host/capi.cc).
extern "C" int ref_window_solve(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm, int n_poses, double *poses,
                                double *ext, int n_lm, double *invdepth, double *td, const double *prior_poses, double prior_weight, double huber,
                                int ext_constant, int td_constant, int iters1, int iters2, double chi2, double *summary8, uint8_t *active_out) {
    ceres::Problem::Options problem_options;
    problem_options.enable_fast_removal = true;
    ceres::Problem problem(problem_options);
    for (int k = 0; k < n_poses; k++) problem.AddParameterBlock(poses + 7 * (size_t) k, 7, new PoseParameterization());
    problem.AddParameterBlock(ext, 7, new PoseParameterization());
    for (int l = 0; l < n_lm; l++) problem.AddParameterBlock(invdepth + l, 1);
    problem.AddParameterBlock(td, 1);
    if (ext_constant) problem.SetParameterBlockConstant(ext);
    if (td_constant) problem.SetParameterBlockConstant(td);
    ceres::LossFunction *loss = huber > 0 ? new ceres::HuberLoss(huber) : nullptr;
    std::vector<ceres::ResidualBlockId> ids;
    for (int k = 0; k < n; k++) {
        auto o       = [&](int c) { return obs_soa[(size_t) c * n + k]; };
        auto *factor = new ReprojectionFactor(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)), Vector3d(o(9), o(10), o(11)),
                                              o(12), o(13), o(14));
        ids.push_back(problem.AddResidualBlock(factor, loss, poses + 7 * (size_t) idx_i[k], poses + 7 * (size_t) idx_j[k], ext, invdepth + idx_lm[k], td));
    }
    double std6[6];
    for (int c = 0; c < 6; c++) std6[c] = 1.0 / prior_weight;
    for (int k = 0; k < n_poses; k++) {
        double prior[7];
        memcpy(prior, prior_poses + 7 * (size_t) k, sizeof prior);
        problem.AddResidualBlock(new ImuPosePriorFactor(prior, std6), nullptr, poses + 7 * (size_t) k);
    }
    ceres::Solver solver;
    ceres::Solver::Options options;
    ceres::Solver::Summary s1, s2;
    options.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
    options.linear_solver_type         = ceres::DENSE_SCHUR;
    options.max_num_iterations         = iters1;
    solver.Solve(options, &problem, &s1);
    summary8[0] = s1.initial_cost, summary8[1] = s1.final_cost, summary8[2] = s1.final_cost, summary8[3] = s1.num_successful_steps,
    summary8[4] = s1.num_unsuccessful_steps, summary8[5] = summary8[6] = summary8[7] = 0;
    for (int k = 0; k < n; k++) active_out[k] = 1;
    if (chi2 > 0) {
        int removed = 0;
        std::vector<int> out;
        for (int k = 0; k < n; k++) { // judge first, remove later (ic_gvins.cc:1276-1292)
            double cost;
            problem.EvaluateResidualBlock(ids[(size_t) k], false, &cost, nullptr, nullptr);
            if (cost * 2.0 > chi2) out.push_back(k);
        }
        for (int k : out) {
            problem.RemoveResidualBlock(ids[(size_t) k]);
            active_out[k] = 0;
            removed++;
        }
        options.max_num_iterations = iters2;
        solver.Solve(options, &problem, &s2);
        summary8[2] = s2.final_cost, summary8[5] = s2.num_successful_steps, summary8[6] = s2.num_unsuccessful_steps, summary8[7] = removed;
    }
    return 0;
}
