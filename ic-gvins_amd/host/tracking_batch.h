// Lock-step executor for many independent camera streams on one MI355X: every stream runs the same
// staged front-end (tracking.h); between stages the per-stream work lists are concatenated so that each stage costs ONE
// batched ABI call (one kernel launch set) for all streams.  Streams never exchange data; results are split back by
// offset, so per-stream outputs are identical to running the streams one at a time (shard invariance).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>

#include "host_pool.h"
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
    // wall-clock breakdown of step(): [0] begin+advance (host logic), [1] gather, [2] device execute, [3] scatter,
    // [4] finalize (digest + window keeper); seconds, accumulated
    double timing[5]{0, 0, 0, 0, 0};
    // work counters of the batched device calls: [0] LK points, [1] LK calls, [2] detection jobs, [3] detection calls,
    // [4] RANSAC point sets, [5] RANSAC calls, [6] preprocessed frames, [7] triangulated points
    uint64_t counters[8]{0, 0, 0, 0, 0, 0, 0, 0};
    // per-step log (bounded): {steady-clock time at the end of step(), host-logic seconds, device-execute seconds} of each
    // call since the last clear — the bench derives per-step median / p95 and the start-up transient from it
    struct StepLog {
        double t_end, host_logic, device_execute;
    };
    vector<StepLog> step_log;
    static constexpr size_t kStepLogCap = 1 << 14;

private:
    void gather(int cur, StageBatch &global, vector<std::array<int, 8>> &bases);
    void scatter(int cur, const StageBatch &global, const vector<std::array<int, 8>> &bases);
    template <typename F> void forEachStream(F &&f);

    DeviceContext::Ptr device_;
    vector<Stream> streams_;
    int host_threads_;
    std::unique_ptr<HostPool> pool_;
    icg_detect_grid grid_{};
    int max_per_job_{0};
};

// Several TrackingBatch groups, each with its own icg_ctx (own HIP stream) and its own persistent host thread: while one
// group's kernels run, the other groups' host stages and kernels proceed, so host bookkeeping overlaps device work and
// latency-bound kernels of different groups overlap on the GPU.  Streams stay independent; results are identical to a
// single batch (and to one-by-one execution).
class StreamGroups {
public:
    StreamGroups(int device, int n_streams, int n_groups, const vector<double> &intrinsic, const vector<double> &distortion,
                 const vector<int> &size, const TrackingConfig &cfg, int window_size, int host_threads_per_group);
    ~StreamGroups();
    void step(const vector<Frame::Ptr> &frames, vector<TrackState> &states);
    // K consecutive steps without a cross-group barrier in between: every group walks through its own K frames at its
    // own pace (streams are independent, so the per-stream results equal K calls of step()).
    // (each group releases its reference to a frame as soon as the frame has been processed)
    void stepMany(vector<vector<Frame::Ptr>> &frames, vector<vector<TrackState>> &states);
    int size() const { return n_streams_; }
    int groups() const { return (int) groups_.size(); }
    TrackingBatch &group(int g) { return *groups_[(size_t) g]; }
    // global stream index -> (group, local index)
    TrackingBatch::Stream &stream(int i) { return groups_[(size_t) group_of_[(size_t) i]]->stream(local_of_[(size_t) i]); }
    std::string error() const { return error_; }

private:
    void workerLoop(int g);
    int n_streams_;
    vector<std::unique_ptr<TrackingBatch>> groups_;
    vector<int> group_of_, local_of_, group_begin_;
    vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_go_, cv_done_;
    uint64_t generation_{0};
    int pending_{0};
    bool stop_{false};
    vector<vector<Frame::Ptr>> *frames_{nullptr};
    vector<vector<TrackState>> *states_{nullptr};
    std::string error_;
};

} // namespace icg
