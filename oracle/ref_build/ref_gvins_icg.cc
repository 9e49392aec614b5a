// ORACLE / TEST INFRASTRUCTURE ONLY — the drop-in proof of boundary B2: the REFERENCE's own estimator
// (/root/reference/ic_gvins/ic_gvins/ic_gvins.{h,cc}, misc.cc, the preintegration variants and the factor headers, all compiled unmodified
// from where they lie) with its `tracking_`, `map_`, `camera_` members bound to the PRODUCT's front-end classes:
//     icg::Tracking / Frame / Feature / MapPoint / Map / Camera / Drawer   (ic-gvins_amd/host/{tracking.h,model.h})
// compiled in drop-in mode (ICG_REFERENCE_TYPES: their value types ARE Eigen / cv::Point2f / the reference's Pose), so that
//     Tracking(Camera::Ptr, Map::Ptr, Drawer::Ptr, const string &configfile, const string &outputpath)      tracking/tracking.h:51
//     TrackState track(Frame::Ptr), bool isNewKeyFrame() const, bool isGoodToTrack(...), static Matrix4d pose2Tcw(Pose)     :53-61
// and every Frame / MapPoint / Map / Camera call ic_gvins.cc makes resolve to the product's code with the reference's argument types.
// How the reference's tracking/*.h are replaced WITHOUT touching the sources: this file pulls the product's headers in first, exports the
// reference's global names as aliases and defines the include guards of tracking/{tracking,frame,feature,mappoint,map,camera,drawer}.h —
// when ic_gvins.h includes them they are empty.  The device layer underneath is the CPU shim of the C ABI (../abi_shim.cc on the oracle), so
// this runs without a GPU; Eigen / OpenCV / Ceres / yaml-cpp / glog / tbb are the interface shims of shim/ as in ref_gvins.cc.
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

#include "ic_gvins.cc"

static Frame::Ptr ref_make_frame(double stamp, const uint8_t *gray, int w, int h) {
    icg::Mat image(h, w, 1); // the product's image handle (host memory here)
    memcpy(image.data, gray, (size_t) w * h);
    return Frame::createFrame(stamp, image);
}
#define REF_GVINS_RUN_NAME ref_gvins_icg_run
#include "ref_gvins_driver.inc"
