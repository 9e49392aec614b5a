// Outlier culling and window statistics of the MI355X host layer — SURVEY.md §8 row f3.
//
// Mirrors GVINS::gvinsOutlierCulling (ic_gvins.cc:1035-1128) and the reprojection part of GVINS::parametersStatistic
// (ic_gvins.cc:930-1033).  The reference walks map -> landmark -> observation (weak_ptr) -> frame under the model locks and
// evaluates Camera::reprojectionError + Tracking::isGoodToTrack per observation.  Here the walk only FLATTENS the window's
// observations (same iteration order, same filters), one icg_reproj_error_batch launch evaluates all of them — for all the
// streams handed in — and the decisions are then replayed on the returned arrays in the reference's order.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "model.h"
#include "tracking.h"

namespace icg {

struct CullingResult {
    int outlier_mappoints{0}, outlier_features{0}; // outliers_[0], outliers_[1]
    int by_reference_frame{0}, by_observation_count{0}, by_mean_error{0}; // num1, num2, num3 of the reference's log line
};

struct ReprojectionStatistics { // parameters[4..7] of parametersStatistic: min / max / mean / rms of the per-landmark mean error
    double min_error{0}, max_error{0}, avg_error{0}, rms_error{0};
    int landmarks{0};
};

class WindowCulling {
public:
    // one entry per stream: the map, the ids of the landmarks that took part in the optimization (invdepthlist_) and the
    // reprojection std (pixels); `camera` is the one set on ctx (icg_set_camera)
    struct Stream {
        Map::Ptr map;
        const std::unordered_map<ulong, double> *invdepthlist;
    };
    // gvinsOutlierCulling for every stream with ONE device launch; results[s] as the reference's counters
    static bool gvinsOutlierCulling(icg_ctx *ctx, const std::vector<Stream> &streams, double reprojection_error_std,
                                    std::vector<CullingResult> &results, std::string *err = nullptr);
    // the line parametersStatistic appends to statistics.txt (ic_gvins.cc:949-1032), in the reference's column order:
    // stamp, dt to the previous keyframe, frame-id difference, feature count, min / max / mean / rms reprojection error,
    // iterations[2], timecosts[3], outliers[2].  Empty when the map holds fewer than two keyframes (:938-940).
    static std::vector<double> statisticsRow(const Map::Ptr &map, const ReprojectionStatistics &stats, const int iterations[2],
                                             const double timecosts[3], const int outliers[2]);
    // the reprojection-error block of parametersStatistic for every stream with ONE device launch
    static bool reprojectionStatistics(icg_ctx *ctx, const std::vector<Stream> &streams, std::vector<ReprojectionStatistics> &stats,
                                       std::string *err = nullptr);
};

} // namespace icg
