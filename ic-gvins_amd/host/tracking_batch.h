// Lock-step executor for many independent camera streams on one MI355X: every stream runs the same
// staged front-end (tracking.h); between stages the per-stream work lists are concatenated so that each stage costs ONE
// batched ABI call (one kernel launch set) for all streams.  Streams never exchange data; results are split back by
// offset, so per-stream outputs are identical to running the streams one at a time (shard invariance).
#pragma once
#include <thread>

#include "tracking.h"

namespace icg {

class TrackingBatch {
public:
    struct Stream {
        Camera::Ptr camera;
        Map::Ptr map;
        Tracking::Ptr tracking;
        std::shared_ptr<WindowKeeper> keeper;
        std::shared_ptr<IdSpace> ids;
        StageBatch box[2];
        // statistics / digest
        uint64_t frames{0}, keyframes{0}, tracked_sum{0}, digest{1469598103934665603ull};
        TrackState last_state{TRACK_PASSED};
    };

    TrackingBatch(int device, int n_streams, const vector<double> &intrinsic, const vector<double> &distortion,
                  const vector<int> &size, const TrackingConfig &cfg, int window_size, int host_threads = 1);

    // one frame per stream (frames[i] may be null to idle a stream); returns per-stream states
    void step(const vector<Frame::Ptr> &frames, vector<TrackState> &states);
    Stream &stream(int i) { return streams_[(size_t) i]; }
    int size() const { return (int) streams_.size(); }
    DeviceContext::Ptr device() { return device_; }

private:
    void gather(int cur, StageBatch &global, vector<std::array<int, 8>> &bases);
    void scatter(int cur, const StageBatch &global, const vector<std::array<int, 8>> &bases);
    template <typename F> void forEachStream(F &&f);

    DeviceContext::Ptr device_;
    vector<Stream> streams_;
    int host_threads_;
    icg_detect_grid grid_{};
    int max_per_job_{0};
};

} // namespace icg
