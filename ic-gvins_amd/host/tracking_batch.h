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
#include "track_table.h"

namespace icg {

// One incoming frame of one stream: what FusionROS::imageCallback + GVINS::runTracking hand to Tracking::track (ROS/fusion_ros.cc:201-234,
// ic_gvins.cc:531): stamp, image (host or device memory, not copied) and the INS pose prior.  valid == false idles the stream that step.
struct FrameInput {
    bool valid{false};
    double stamp{0};
    Mat image;
    Pose pose;
};

class TrackingBatch {
public:
    // Two engines run the same per-frame algorithm behind the same stages: the track table (track_table.h; default, the throughput
    // path) and the reference-shaped object graph (icg::Tracking + Map + WindowKeeper; ICG_TRACK_ENGINE=object).  Per-stream results are
    // identical (tests/test_host_engines_cpu.py).
    // ENGINE_DEVICE (round 4, ICG_TRACK_ENGINE=device): the device-resident tracker (icg_tracker_*, csrc/tracker.hip) — the streams' state lives in
    // HBM, a step is one chain of launches and one wait, the host reads a stream's block only when an accessor below asks for it.
    enum Engine { ENGINE_TABLE = 0, ENGINE_OBJECT = 1, ENGINE_CORE = 2 /* the table interface on track_core.h */, ENGINE_DEVICE = 3 };
    struct Stream {
        Camera::Ptr camera;
        std::shared_ptr<IdSpace> ids;
        TableTracker::Ptr table; // ENGINE_TABLE
        Map::Ptr map;            // ENGINE_OBJECT
        Tracking::Ptr tracking;
        std::shared_ptr<WindowKeeper> keeper;
        Frame::Ptr frame; // the object engine's frame of the current step
        StageBatch box[2];
        // (ENGINE_DEVICE: the step's results are the stream's state — the table's copy of the block may be absent or older)
        bool frameDone() const { return tracker ? true : table ? table->frameDone() : tracking->frameDone(); }
        TrackState result() const { return tracker ? (TrackState) last.state : table ? table->result() : tracking->result(); }
        bool isNewKeyFrame() const { return tracker ? last.is_new_keyframe != 0 : table ? table->isNewKeyFrame() : tracking->isNewKeyFrame(); }
        const vector<Point2f> &trackedRefPoints() const { return syncDevice(), table ? table->trackedRefPoints() : tracking->trackedRefPoints(); }
        const vector<Point2f> &referencePoints() const { return syncDevice(), table ? table->referencePoints() : tracking->referencePoints(); }
        size_t windowKeyFrames() const { return tracker ? (size_t) last.window_keyframes : table ? table->windowKeyFrames() : map->keyframes().size(); }
        size_t landmarks() const { return tracker ? (size_t) last.landmarks : table ? table->landmarks() : map->landmarks().size(); }
        // ENGINE_DEVICE: the group's tracker, this stream's index in it, the results of its last step, and whether the table's copy of the
        // block (TableTracker core mode, device-resident) is older than the device's
        icg_tracker *tracker{nullptr};
        int tracker_index{0};
        icg_tracker_result last{};
        mutable bool device_stale{false};
        void syncDevice() const; // downloads the block if the device's is newer (no-op for the host engines)
        // (map-point id, distorted key point) of the features of the stream's current frame, unordered
        void currentFeatures(vector<std::pair<ulong, Point2f>> &out) const;
        std::string dump(int kind) const; // 0: engine state (canonical text), 1: table map part, 2: the same from materialize()
        // The stream's map as reference-shaped objects: the object engine's own map, or a view of the track table (built on first use
        // and kept until commitMap()).  Code that writes to it (optimizer write-back, culling) calls commitMap() when it is done: the
        // table engine takes the changes over (TableTracker::absorb) and drops the view; a no-op for the object engine.
        // discardMap() drops the view WITHOUT taking anything over (error paths: a half-modified view must not be seen again).
        Map::Ptr objectMap();
        void commitMap();
        void discardMap() { view_.reset(); }
        std::shared_ptr<TableTracker::ObjectView> view_;
        // statistics / digest
        uint64_t frames{0}, keyframes{0}, tracked_sum{0}, digest{1469598103934665603ull};
        TrackState last_state{TRACK_PASSED};
    };

    TrackingBatch(int device, int n_streams, const vector<double> &intrinsic, const vector<double> &distortion,
                  const vector<int> &size, const TrackingConfig &cfg, int window_size, int host_threads = 1, int engine = -1);
    ~TrackingBatch();

    // one frame per stream (frames[i].valid == false idles a stream); returns per-stream states
    void step(const FrameInput *frames, vector<TrackState> &states);
    Engine engine() const { return engine_; }
    double lastStepSeconds() const { return last_step_s_; }
    // kernel-only replay of the device calls of the steps run while recording (DeviceContext::record / replay)
    void record(bool on) { device_->record(on); }
    void replay(int reps, size_t first = 0) {
        for (int r = 0; r < reps; r++) device_->replay(grid_, max_per_job_, first);
    }
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
    void stepDevice(const FrameInput *frames, vector<TrackState> &states);
    icg_tracker *tracker_{nullptr};
    double last_step_s_{0};
    vector<const uint8_t *> dev_images_;
    vector<double> dev_stamps_, dev_poses_;
    vector<icg_tracker_result> dev_results_;
    vector<int32_t> dev_drain_, dev_drain_n_;
    vector<tc::LmLog> dev_log_;
    void gather(int cur, StageBatch &global, vector<std::array<int, 8>> &bases);
    void scatter(int cur, const StageBatch &global, const vector<std::array<int, 8>> &bases);
    template <typename F> void forEachStream(F &&f);

    DeviceContext::Ptr device_;
    vector<Stream> streams_;
    Engine engine_{ENGINE_TABLE};
    StageBatch global_; // the concatenated work lists of a stage (kept: their capacity is the steady-state size)
    vector<std::array<int, 8>> bases_;
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
    void step(const vector<FrameInput> &frames, vector<TrackState> &states);
    // K consecutive steps without a cross-group barrier in between: every group walks through its own K frames at its
    // own pace (streams are independent, so the per-stream results equal K calls of step()).
    void stepMany(const vector<vector<FrameInput>> &frames, vector<vector<TrackState>> &states);
    // device-only run: every group's thread issues its recorded device calls (TrackingBatch::record) `reps` times, all groups at once, no
    // tracker logic — the rate the kernels and the launch structure allow under the same concurrency as a real run
    void replayAll(int reps);
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
    bool stagger_{true};
    int replay_reps_{0}; // > 0: the pending job is a device-only replay
    const vector<vector<FrameInput>> *frames_{nullptr};
    vector<vector<TrackState>> *states_{nullptr};
    std::string error_;
};

} // namespace icg
