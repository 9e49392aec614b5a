// ORACLE / TEST INFRASTRUCTURE ONLY — the drop-in proof of boundary B1: the REFERENCE's own estimator with the visual factors
// of its window optimization (GVINS::gvinsOptimization, ic_gvins.cc:1130-1239; factors created at :1826-1831) served by the PRODUCT's back-end:
//     icg::ReprojectionFactor (same constructor as factors/reprojection_factor.h:42-53, a ceres::SizedCostFunction<2,7,7,7,1,1>) and
//     icg::ReprojectionBatch : ceres::EvaluationCallback — ONE batched evaluation of every factor per evaluation point, each Evaluate() a copy.
// ic_gvins.cc is compiled from a temporary copy (oracle/_ref/ic_gvins_b1.cc, made by the Makefile with sed, never committed) that carries
// exactly the registration INTEGRATION.md section 2 documents:
//     (1) after `problem_options.enable_fast_removal = true;` (:1137)   the batch is created and set as problem_options.evaluation_callback
//     (2) at :1826  `new ReprojectionFactor(` -> `new icg::ReprojectionFactor(`, and after the AddResidualBlock of :1830-1831
//         `icg_b1->add(factor, <the same five parameter blocks>)`
//     (3) before each `solver.Solve(options, &problem, &summary);` of the function (:1183, :1217)   `icg_b1->finalize();`
// Everything else — the tracker (icg::Tracking, as in the B2 proof), marginalization with the reference's own ReprojectionFactor, IMU / GNSS
// factors, the three threads — is the reference's code, unmodified.  ceres::Problem / Solver are shim/ceres/problem_shim.h (with
// Problem::Options::evaluation_callback honoured as Ceres >= 2.0 does); the C ABI underneath is the CPU shim (../abi_shim.o on liboracle.so).
#define ICG_REFERENCE_TYPES 1
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include <Eigen/Geometry>
#include <opencv2/opencv.hpp>

// the product's front-end host layer, in drop-in mode (the same sources that build libicgvins_host.so)
#include "../../ic-gvins_amd/host/model.cc"
#include "../../ic-gvins_amd/host/tracking_hip.cc"
// the product's back-end factors (the same source that builds libicgvins_host.so), in drop-in mode
#include "../../ic-gvins_amd/host/factors.cc"
static icg::ReprojectionBatch *icg_b1 = nullptr; // the batch of the window being optimized (set by the patched gvinsOptimization)

// ---- the reference's names -------------------------------------------------------------------------------------------------------
#define GVINS_CAMERA_H
#define GVINS_DRAWER_H
#define GVINS_FEATURE_H
#define GVINS_FRAME_H
#define GVINS_MAP_H
#define GVINS_MAPPOINT_H
#define GVINS_TRACKING_H
using std::string;
using std::vector;
using cv::Mat; // (the reference's frame.h: `using cv::Mat;` — only named by the drawer here; frames enter through ref_make_frame below)
using icg::Camera;
using icg::Drawer;
using icg::Feature;
using icg::Frame;
using icg::Map;
using icg::MapPoint;
using icg::Tracking;
// enumerations with their unscoped enumerators (tracking.h:38-44, frame.h:36-41, mappoint.h:33-39, feature.h:33-38)
using icg::TrackState;
using icg::TRACK_FIRST_FRAME;
using icg::TRACK_INITIALIZING;
using icg::TRACK_LOST;
using icg::TRACK_PASSED;
using icg::TRACK_TRACKING;
using icg::keyFrameState;
using icg::KEYFRAME_NONE;
using icg::KEYFRAME_NORMAL;
using icg::KEYFRAME_REMOVE_OLDEST;
using icg::KEYFRAME_REMOVE_SECOND_NEW;
using icg::MapPointType;
using icg::MAPPOINT_DEPTH_ASSOCIATED;
using icg::MAPPOINT_DEPTH_INITIALIZED;
using icg::MAPPOINT_FIXED;
using icg::MAPPOINT_NONE;
using icg::MAPPOINT_TRIANGULATED;
using icg::FeatureType;
using icg::FEATURE_DEPTH_ASSOCIATED;
using icg::FEATURE_MATCHED;
using icg::FEATURE_NONE;
using icg::FEATURE_TRIANGULATED;

// ---- the reference's estimator, unmodified -----------------------------------------------------------------------------------------
#include "fileio/filesaver.cc"
#include "preintegration/preintegration_base.cc"
#include "preintegration/preintegration_earth.cc"
#include "preintegration/preintegration_earth_odo.cc"
#include "preintegration/preintegration_normal.cc"
#include "preintegration/preintegration_odo.cc"

#include "misc.cc"

#include "ic_gvins_b1.cc" // oracle/_ref/ic_gvins_b1.cc: the reference's ic_gvins.cc + the three registration edits (see the Makefile)

static Frame::Ptr ref_make_frame(double stamp, const uint8_t *gray, int w, int h) {
    icg::Mat image(h, w, 1); // the product's image handle (host memory here)
    memcpy(image.data, gray, (size_t) w * h);
    return Frame::createFrame(stamp, image);
}
#define REF_GVINS_RUN_NAME ref_gvins_b1_run
#include "ref_gvins_driver.inc"
