// WindowCulling: gvinsOutlierCulling / parametersStatistic with the per-observation arithmetic batched on the device.  See culling_hip.h.
#include "culling_hip.h"

#include <cmath>
#include <numeric>

#include "../../include/icgvins_hip.h"

namespace icg {

namespace {
// the reference's walk (ic_gvins.cc:1046-1069 / :966-983) flattened: per stream, per eligible landmark, its eligible observations in
// list order
struct FlatObs {
    Feature::Ptr feat;
    Frame::Ptr frame;
};
struct FlatLandmark {
    MapPoint::Ptr mappoint;
    int first, count; // range in the flat observation arrays
};
struct Flat {
    std::vector<std::vector<FlatLandmark>> landmarks; // per stream
    std::vector<FlatObs> obs;
    std::vector<int32_t> pose_idx, lm_idx;
    std::vector<double> poses12, pw;
    std::vector<float> pix;
};

void flatten(const std::vector<WindowCulling::Stream> &streams, Flat &F) {
    F.landmarks.resize(streams.size());
    for (size_t s = 0; s < streams.size(); s++) {
        const auto &S = streams[s];
        if (S.map->keyframes().empty()) continue; // :1036-1038
        std::unordered_map<Frame *, int32_t> pose_of;
        for (auto &landmark : S.map->landmarks()) { // unordered_map iteration order, as the reference's loop
            auto mappoint = landmark.second;
            if (!mappoint || mappoint->isOutlier()) continue;
            if (S.invdepthlist->find(mappoint->id()) == S.invdepthlist->end()) continue;
            FlatLandmark L{mappoint, (int) F.obs.size(), 0};
            const int32_t lm = (int32_t) (F.pw.size() / 3);
            const Vector3d pos = mappoint->pos();
            F.pw.insert(F.pw.end(), {pos[0], pos[1], pos[2]});
            for (auto &observation : mappoint->observations()) {
                auto feat = observation.lock();
                if (!feat || feat->isOutlier()) continue;
                auto frame = feat->getFrame();
                if (!frame || !frame->isKeyFrame() || !S.map->isKeyFrameInMap(frame)) continue;
                auto it = pose_of.find(frame.get());
                if (it == pose_of.end()) {
                    it = pose_of.emplace(frame.get(), (int32_t) (F.poses12.size() / 12)).first;
                    const Pose p = frame->pose();
                    double p12[12];
                    poseToArray12(p, p12);
                    F.poses12.insert(F.poses12.end(), p12, p12 + 12);
                }
                const Point2f pp = feat->keyPoint();
                F.pose_idx.push_back(it->second);
                F.lm_idx.push_back(lm);
                F.pix.push_back(pp.x);
                F.pix.push_back(pp.y);
                F.obs.push_back({feat, frame});
                L.count++;
            }
            F.landmarks[s].push_back(L);
        }
    }
}

bool evaluate(icg_ctx *ctx, const Flat &F, double max_error, std::vector<double> &err, std::vector<uint8_t> &good, std::string *e) {
    const int n = (int) F.obs.size();
    err.assign((size_t) n, 0.0);
    good.assign((size_t) n, 0);
    if (n == 0) return true;
    if (icg_reproj_error_batch(ctx, n, F.pose_idx.data(), F.lm_idx.data(), (int) (F.poses12.size() / 12), F.poses12.data(), (int) (F.pw.size() / 3),
                               F.pw.data(), F.pix.data(), max_error, MapPoint::NEAREST_DEPTH, MapPoint::FARTHEST_DEPTH * 1.0, err.data(),
                               good.data()) != ICG_OK) {
        if (e) *e = icg_last_error(ctx);
        return false;
    }
    return true;
}
} // namespace

bool WindowCulling::gvinsOutlierCulling(icg_ctx *ctx, const std::vector<Stream> &streams, double reprojection_error_std,
                                        std::vector<CullingResult> &results, std::string *err) {
    Flat F;
    flatten(streams, F);
    std::vector<double> error;
    std::vector<uint8_t> good;
    // isGoodToTrack(pp, pose, pos, 3.0): three times the threshold (:1078), default depth scale
    if (!evaluate(ctx, F, reprojection_error_std * 3.0, error, good, err)) return false;
    results.assign(streams.size(), CullingResult());
    for (size_t s = 0; s < streams.size(); s++) {
        CullingResult &R = results[s];
        std::vector<MapPoint::Ptr> mappoints; // found first, removed later (:1040-1042)
        for (const FlatLandmark &L : F.landmarks[s]) {
            auto &mappoint = L.mappoint;
            std::vector<double> errors;
            bool broke = false;
            for (int k = L.first; k < L.first + L.count; k++) {
                const FlatObs &O = F.obs[(size_t) k];
                if (!good[(size_t) k]) { // feature outlier (:1078-1091)
                    O.feat->setOutlier(true);
                    mappoint->decreaseUsedTimes();
                    if (O.frame->id() == mappoint->referenceFrameId()) {
                        mappoint->setOutlier(true);
                        mappoints.push_back(mappoint);
                        R.outlier_mappoints++;
                        R.by_reference_frame++;
                        broke = true;
                        break;
                    }
                    R.outlier_features++;
                } else {
                    errors.push_back(error[(size_t) k]);
                }
            }
            (void) broke; // like the reference, the checks below run even after the break
            if (errors.size() < 2) { // :1098-1103
                mappoint->setOutlier(true);
                mappoints.push_back(mappoint);
                R.outlier_mappoints++;
                R.by_observation_count++;
            } else {
                double avg_error = std::accumulate(errors.begin(), errors.end(), 0.0) / static_cast<double>(errors.size());
                if (avg_error > reprojection_error_std) {
                    mappoint->setOutlier(true);
                    mappoints.push_back(mappoint);
                    R.outlier_mappoints++;
                    R.by_mean_error++;
                }
            }
        }
        for (auto &mappoint : mappoints) streams[s].map->removeMappoint(mappoint); // :1113-1117
    }
    return true;
}

std::vector<double> WindowCulling::statisticsRow(const Map::Ptr &map, const ReprojectionStatistics &stats, const int iterations[2],
                                                 const double timecosts[3], const int outliers[2]) {
    std::vector<double> parameters;
    std::vector<ulong> keyframeids = map->orderedKeyFrames();
    size_t size                    = keyframeids.size();
    if (size < 2) return parameters;
    auto keyframes = map->keyframes();
    auto frame_cur = keyframes.at(keyframeids[size - 1]);
    auto frame_pre = keyframes.at(keyframeids[size - 2]);
    parameters.push_back(frame_cur->stamp());
    parameters.push_back(frame_cur->stamp() - frame_pre->stamp());
    parameters.push_back(static_cast<double>(frame_cur->id() - frame_pre->id()));
    parameters.push_back(static_cast<double>(frame_cur->numFeatures()));
    parameters.push_back(stats.min_error);
    parameters.push_back(stats.max_error);
    parameters.push_back(stats.avg_error);
    parameters.push_back(stats.rms_error);
    parameters.push_back(iterations[0]);
    parameters.push_back(iterations[1]);
    parameters.push_back(timecosts[0]);
    parameters.push_back(timecosts[1]);
    parameters.push_back(timecosts[2]);
    parameters.push_back(outliers[0]);
    parameters.push_back(outliers[1]);
    return parameters;
}

bool WindowCulling::reprojectionStatistics(icg_ctx *ctx, const std::vector<Stream> &streams, std::vector<ReprojectionStatistics> &stats,
                                           std::string *err) {
    Flat F;
    flatten(streams, F);
    std::vector<double> error;
    std::vector<uint8_t> good;
    if (!evaluate(ctx, F, 0.0, error, good, err)) return false;
    stats.assign(streams.size(), ReprojectionStatistics());
    for (size_t s = 0; s < streams.size(); s++) {
        std::vector<double> reprojection_errors;
        for (const FlatLandmark &L : F.landmarks[s]) {
            if (L.count == 0) continue; // "Mappoint with zero observation" (:988-991)
            double sum = 0.0;
            for (int k = L.first; k < L.first + L.count; k++) sum += error[(size_t) k];
            reprojection_errors.emplace_back(sum / static_cast<double>(L.count));
        }
        stats[s].landmarks = (int) reprojection_errors.size();
        if (reprojection_errors.empty()) reprojection_errors.push_back(0); // :996-998
        ReprojectionStatistics &R = stats[s];
        R.min_error = *std::min_element(reprojection_errors.begin(), reprojection_errors.end());
        R.max_error = *std::max_element(reprojection_errors.begin(), reprojection_errors.end());
        R.avg_error = std::accumulate(reprojection_errors.begin(), reprojection_errors.end(), 0.0) / static_cast<double>(reprojection_errors.size());
        double sq_sum = std::inner_product(reprojection_errors.begin(), reprojection_errors.end(), reprojection_errors.begin(), 0.0);
        R.rms_error   = std::sqrt(sq_sum / static_cast<double>(reprojection_errors.size()));
    }
    return true;
}

} // namespace icg
