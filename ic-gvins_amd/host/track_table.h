// Table engine of the front-end: the reference's per-frame algorithm (ic_gvins/ic_gvins/tracking/tracking.cc:144-245, :351-455,
// :457-574, :576-688, :690-798, :263-307) on an arena-resident structure-of-arrays track table instead of the shared_ptr / weak_ptr /
// unordered_map object graph of tracking/{frame,feature,mappoint,map}.h.
//
// Why: with hundreds of camera streams per host the object graph costs ~140 us of host CPU per frame (pointer chasing through cold
// memory, ~10 locked reference-count operations per feature), which bounds both the one-GPU rate and the GPUs one host can feed.
// Here a frame is a handful of contiguous arrays (map-point handle, undistorted / distorted pixel, normalized-plane velocity per
// feature row), a map point is a row of a per-stream pool, an observation is a (frame, row) pair, and the window bookkeeping of
// GVINS (ic_gvins.cc:724-747, 1391-1410, 440-448) — `WindowKeeper` for the object engine — works on frame handles.
//
// The result is the same, bit for bit: ids, states, key-point bits, candidate lists, the tracking.txt rows.  In particular the
// features of a frame are visited in the order std::unordered_map<ulong, Feature::Ptr> (frame.h:134) would yield them (HashOrder
// emulates libstdc++'s bucket list), because the parallax averages (tracking.cc:873-905) are floating-point sums in that order.
// The B2 surface survives as a view: materialize() builds the reference-shaped icg::Map / Frame / Feature / MapPoint objects of a
// stream on demand (tests compare them with what icg::Tracking built on the same frames).
#pragma once
#include <cstdio>
#include <string>
#include <unordered_map>

#include "tracking.h"
#include "track_core.h"

namespace icg {

// Iteration order of a libstdc++ std::unordered_map<ulong, T> with unique keys that has only ever been inserted into (the only use
// a Frame makes of features_): singly linked node list + per-bucket "node before the bucket's first node" (bits/hashtable.h
// _M_insert_bucket_begin, _M_rehash_aux(unique keys)), identity hash, bucket = key % bucket_count, bucket counts taken from a real
// std::unordered_map of this standard library (growthTable()).  Nodes are row indices 0..n-1 inserted in that order.
class HashOrder {
public:
    void clear() {
        next_.clear();
        key_.clear();
        head_ = -1;
        // (std::unordered_map::clear() keeps its buckets; a Frame's map is only ever cleared while empty, so this is a fresh table)
        bucket_.assign(1, kEmpty);
        magic_ = 0; // (M for one bucket is 2^64: wraps to 0, and the fastmod of any key is 0 — the right bucket)
    }
    void reserveRows(size_t n) {
        next_.reserve(n);
        key_.reserve(n);
    }
    size_t size() const { return next_.size(); }
    // appends row index size() with this key; returns false (and adds nothing) if the key is already present
    bool insert(ulong key) { return contains(key) ? false : (insertUnique(key), true); }
    // the same for a key the caller knows to be absent (ids of one frame's rows carried to the next frame, freshly drawn ids)
    void insertUnique(ulong key);
    int head() const { return head_; }
    int next(int i) const { return next_[(size_t) i]; }
    bool contains(ulong key) const;
    // bucket count after k insertions into a fresh std::unordered_map<ulong, char> of this libstdc++
    static size_t bucketsAfter(size_t k);
    // n pseudo-random distinct keys inserted one by one into a HashOrder and into a real std::unordered_map<ulong, int>, the iteration
    // orders compared after every `check_every` insertions: 0 when they always agree, k > 0 = first disagreement after k insertions
    static int selfTest(uint64_t seed, int n, int check_every, bool dense_ids);
    // run once per process before the table engine is used: throws when this standard library's container does not iterate the way
    // HashOrder assumes (the parallax sums and landmark order of the engine would silently leave the reference's order)
    static void verifyOnce();
    // the container order as flat arrays (the tracker core's representation, track_core.h): read access, and the way back
    int bucketCount() const { return (int) bucket_.size(); }
    const int *bucketData() const { return bucket_.data(); }
    const int *nextData() const { return next_.data(); }
    uint64_t magic() const { return magic_; }
    void restore(const int *next, const uint64_t *keys, size_t stride_keys_bytes, int n, const int *bucket, int n_buckets, int head, uint64_t magic) {
        next_.assign(next, next + n);
        key_.resize((size_t) n);
        for (int i = 0; i < n; i++) key_[(size_t) i] = *reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(keys) + (size_t) i * stride_keys_bytes);
        bucket_.assign(bucket, bucket + n_buckets);
        head_  = head;
        magic_ = magic;
    }

private:
    static constexpr int kEmpty = -1, kBeforeBegin = -2;
    void rehash(size_t n);
    int nextOf(int prev) const { return prev == kBeforeBegin ? head_ : next_[(size_t) prev]; }
    void setNext(int prev, int i) {
        if (prev == kBeforeBegin)
            head_ = i;
        else
            next_[(size_t) prev] = i;
    }
    vector<int> next_, bucket_{kEmpty}, scratch_;
    uint64_t magic_{0};
    vector<ulong> key_;
    int head_{-1};
};

class TableTracker {
public:
    typedef std::shared_ptr<TableTracker> Ptr;
    struct Input {
        double stamp{0};
        Mat image;
        Pose pose;
    };

    TableTracker(Camera::Ptr camera, size_t window_size, const TrackingConfig &config, const std::string &outputpath,
                 DeviceContext::Ptr device, std::shared_ptr<IdSpace> ids);
    ~TableTracker();

    // ---- staged interface (same stages as icg::Tracking) ----
    void beginFrame(const Input &in, StageBatch &next);
    void advance(int stage, StageBatch &done, StageBatch &next);
    bool frameDone() const { return core_ ? core_->done != 0 : done_; }
    TrackState result() const { return core_ ? (TrackState) core_->result : result_; }
    bool isNewKeyFrame() const { return core_ ? core_->isnewkeyframe != 0 : isnewkeyframe_; }

    // ---- core mode (round 4): the stream's state lives in a tc::Stream block (track_core.h) and the stage bodies are the tracker core's —
    // the code the device-resident tracker runs inside its stage kernels, compiled for the host.  The table members above are then a
    // lazily refreshed IMAGE of the block (syncTable(): importCore when the block has changed), so that every accessor, the object view,
    // absorb() and the canonical dumps work unchanged; absorb() writes the image back (exportCore).  ICG_TRACK_ENGINE=core selects it for
    // the host executor; the device executor (tracking_device.h) keeps the blocks in HBM and hands a downloaded copy to attachCore().
    void enableCore(bool device_resident = false);
    // the device-resident tracker's line of tracking.txt (tracking.cc:309-315): the five numbers of the keyframe decision and the feature
    // count come with the step's result (icg_tracker_result), the time cost is the executor's
    void writeTrackingLog(const double data5[5], int features, double cost_ms);
    bool coreMode() const { return core_ != nullptr; }
    bool coreDeviceResident() const { return core_device_resident_; }
    // device-resident block: exportCore() ran since the last call (the executor uploads the block then) / the executor restarted the
    // device's landmark history after this copy replayed it
    bool takeCoreChanged() {
        const bool c  = core_changed_;
        core_changed_ = false;
        return c;
    }
    // entries [0, n) of the device's landmark history, fetched without the block (icg_tracker_fetch_logs): replayed into map_lm_ from where
    // this copy had got to; the device's history restarts at 0
    void coreApplyFetchedLog(const tc::LmLog *log, int n) {
        for (int k = core_log_applied_; k < n; k++) {
            if (log[k].op)
                map_lm_.insert(std::make_pair((ulong) log[k].id, log[k].mp));
            else
                map_lm_.erase((ulong) log[k].id);
        }
        core_log_applied_ = 0;
        if (core_) core_->n_log = 0;
    }
    void coreLogRestarted() {
        core_->n_log      = 0;
        core_log_applied_ = 0;
    }
    void setCoreImageFormat(const Mat &m) {
        core_image_format_      = m;
        core_image_format_.data = nullptr;
        core_image_format_.storage.reset();
    }
    tc::Stream *core() { return core_.get(); }
    const tc::Cfg &coreCfg() const { return core_cfg_; }
    void markCoreChanged() { core_dirty_ = true; }
    void syncTable() const {
        if (core_ && core_dirty_) const_cast<TableTracker *>(this)->importCore();
    }
    void importCore(); // tc::Stream -> table members
    void exportCore(); // table members -> tc::Stream
    static tc::Cfg makeCoreCfg(const Camera &camera, const TrackingConfig &cfg, size_t window_size);
    static const uint32_t *bucketsAfterTable(); // HashOrder::bucketsAfter(k) for k = 0..tc::MAX_ROWS + 1
    // sliding-window side effects of GVINS on the map after a frame (WindowKeeper::onFrame) + release of dead frames
    void endFrame();

    const icg_detect_grid &grid() const { return grid_; }
    int maxFeaturesPerJob() const { return grid_.max_per_block * block_cnts_; }
    size_t numTrackedRefPoints() const { return core_ ? (size_t) core_->n_new : pts2d_new_.size(); }
    const vector<Point2f> &trackedRefPoints() const { return syncTable(), pts2d_new_; }
    const vector<Point2f> &referencePoints() const { return syncTable(), pts2d_ref_; }

    // ---- results of the current frame ----
    ulong currentFrameId() const;
    ulong lastInputFrameId() const { return core_ ? (ulong) core_->last_input_fid : last_input_fid_; } // id of the frame handed to the last beginFrame (also when it was skipped)
    size_t numCurrentFeatures() const;
    // visits (map-point id, distorted key point) of the current frame's features
    template <typename F> void forEachCurrentFeature(F &&f) const {
        syncTable();
        if (cur_ < 0) return;
        const Frame_ &fr = frames_[(size_t) cur_];
        for (const Row &r : fr.row) f(r.id, r.kpd); // (any order: the digest does not depend on it)
    }
    size_t windowKeyFrames() const { return core_ ? (size_t) core_->n_map_kf : map_kf_.size(); }
    size_t landmarks() const { return core_ ? (size_t) core_->n_landmarks : n_landmarks_; }

    // ---- B2 view: the reference-shaped object graph of this stream, built on demand, and the way back ----
    // What code written against the reference's types does to a map between two frames — the optimizer's write-back (keyframe poses,
    // landmark positions and depths: ic_gvins.cc:1347-1391), outlier culling (feature / map-point outlier flags, used-times, removal of
    // landmarks: ic_gvins.cc:1035-1128) — is done on the objects of a view and taken over by absorb().
    struct ObjectView {
        Map::Ptr map;
        vector<Frame::Ptr> frame;          // by frame handle (null: not alive at the time)
        vector<vector<Feature::Ptr>> feat; // by frame handle, row
        vector<MapPoint::Ptr> mappoint;    // by map-point pool index (null: not live at the time)
        vector<uint32_t> frame_gen, mp_gen;
    };
    std::shared_ptr<ObjectView> view() const;
    void absorb(const ObjectView &v);
    // keyframes of the window (+ the tracker's current / previous / reference frames when they are not keyframes) with their features,
    // the landmarks with position / reference frame / depth / counters / observation lists.  `extra` receives the non-keyframe frames.
    Map::Ptr materialize(vector<Frame::Ptr> *extra = nullptr) const;
    // canonical text dump of the state (sorted by ids; floats as bit patterns), for engine-vs-engine tests
    std::string dump() const;
    // the same dump computed from an object graph (icg::Tracking + Map)
    static std::string dumpObjects(Tracking &tracking, Map &map);
    std::string dumpMap() const;          // the map part of dump()
    std::string dumpMaterialized() const; // the same text computed from materialize(): must equal dumpMap()

private:
    struct Row {           // one Feature (feature.h:41-118) of a frame
        ulong id;          // map-point id (the key of the reference's features_ container)
        uint32_t mp, mpgen; // map-point handle
        Point2f kp, kpd;   // undistorted / distorted key point
        Vector2d vel;      // velocity on the normalized plane
        double pcx, pcy;   // Camera::pixel2cam(kp), kept: the next frame's velocity and every parallax need it again
        int32_t lk_idx;    // index of the LK point that produced this key point in the group's LK call of that frame (-1: not from LK)
        int8_t type;
        uint8_t outlier;   // Feature::isOutlier (set by the optimizer's culling, read by the parallax of tracking.cc:873-905)
    };
    static_assert(sizeof(Row) == 72, "a feature row is 72 bytes (DESIGN.md section 1)");
    struct Frame_ {
        bool alive{false};
        uint32_t gen{0};
        ulong fid{0}, kf_id{0};
        double stamp{0};
        Pose pose;
        bool is_kf{false};
        int kf_state{KEYFRAME_NORMAL}; // frame.h default
        int slot{-1};
        Mat image;
        // feature rows (insertion order); `order` is the container order of the reference's features_
        vector<Row> row;
        HashOrder order;
        vector<uint32_t> unupdated, unupdated_gen; // map points created with this frame as the current one (frame.h unupdated_mappoints_)
        size_t rows() const { return row.size(); }
        void clearRows();
    };
    struct LastObs {
        int32_t frame{-1};
        uint32_t gen{0};
        int32_t row{-1};
    };
    struct alignas(64) MapPointHot { // what the per-feature loops touch: one cache line per map point (aligned: a record never straddles two)
        uint32_t gen{0};
        uint8_t live{0}, outlier{0}, in_map{0};
        int8_t type{0};
        ulong id{0};
        Vector3d pos;
        int32_t observed{0}, used{0};
        LastObs last;
    };
    static_assert(sizeof(MapPointHot) == 64, "the hot record of a map point is one cache line");
    struct MapPointCold {
        ulong born_fid{0};
        int32_t ref_frame{-1};
        uint32_t ref_gen{0};
        Point2f ref_kp;
        double depth{0};
        int32_t optimized{0};
    };
    struct MapPoints { // per-stream pool; handle = (index, gen)
        vector<MapPointHot> hot;
        vector<MapPointCold> cold;
        vector<uint32_t> free_list;
        uint32_t alloc();
        void release(uint32_t i) {
            hot[i].live = 0;
            hot[i].gen++;
            free_list.push_back(i);
        }
        bool valid(uint32_t i, uint32_t g) const { return hot[i].live && hot[i].gen == g; }
        size_t size() const { return hot.size(); }
    };

    // frames
    int allocFrame();
    void freeFrame(int h);
    void sweepFrames();
    void setKeyFrame(int h, int state);
    int addRow(int h, ulong id, uint32_t mp, const Point2f &kp, const Point2f &kpd, const Vector2d &vel, FeatureType type, double pcx,
               double pcy, bool unique_key, int32_t lk_idx = -1);
    vector<ulong> observationFrames(uint32_t mp, const vector<int> &alive_by_fid) const;
    size_t listContainerOrder(const Frame_ &f);
    bool frameValid(int h, uint32_t g) const { return h >= 0 && frames_[(size_t) h].alive && frames_[(size_t) h].gen == g; }

    // map (tracking/map.cc) on handles
    void mapInsertKeyFrame(int h);
    void mapRemoveKeyFrame(int h, bool isremovemappoint);
    bool mapIsWindowFull() const { return is_window_full_; }
    bool mapIsWindowNormal() const { return map_kf_.size() == window_size_; }
    bool mapIsKeyFrameInMap(int h) const;
    int mapFind(ulong kf_id) const;

    // stage bodies (tracking_hip.cc has the object-graph twins)
    void onPreprocessDone(StageBatch &done, StageBatch &next);
    void onDetectADone(StageBatch &done, StageBatch &next);
    void onLKDone(StageBatch &done, StageBatch &next);
    void onRansacDone(StageBatch &done, StageBatch &next);
    void onTriangulateDone(StageBatch &done, StageBatch &next);
    void onDetectBDone(StageBatch &done);
    void finish(TrackState st);
    bool queueDetection(int frame, bool ismask, StageBatch &next);
    void integrateDetection(StageBatch &done);
    void queueTrackMappoint(StageBatch &next);
    bool finishTrackMappoint(StageBatch &done);
    void queueTrackReference(StageBatch &next);
    bool midTrackReference(StageBatch &done, StageBatch &next);
    bool finishTrackReference(StageBatch &done);
    bool queueTriangulation(StageBatch &next);
    void finishTriangulation(StageBatch &done);
    void makeNewFrameQueue(int state, StageBatch &next);
    keyFrameState checkKeyFrameSate();
    void writeLoggingMessage();
    bool doResetTracking();
    double relativeTranslation() const;
    double relativeRotation() const;
    int parallaxFromReferenceKeyPoints(const vector<Point2f> &ref, const vector<Point2f> &cur, double &parallax);
    int parallaxFromReferenceMapPoints(double &parallax);
    double keyPointParallax(const Point2f &pp0, const Point2f &pp1, const Matrix3d &R10) const;
    bool isGoodToTrack(const Point2f &pp, const Pose &pose, const Vector3d &pw, double scale, double depth_scale) const;
    bool isOnBorder(const Point2f &pts) const;
    void checkCarriedUndistortion(const char *where);
    void assignSlot(int h);
    void releaseUnusedSlots();
    template <typename T> static void reduceVector(T &vec, const vector<uint8_t> &status);

    const double TRACK_BLOCK_SIZE   = 200.0; // tracking.h:112
    const double TRACK_MIN_PARALLAX = 10.0;  // tracking.h:114
    const double TRACK_MIN_INTERVAl = 0.08;  // tracking.h:115

    Camera::Ptr camera_;
    DeviceContext::Ptr device_;
    std::shared_ptr<IdSpace> ids_;
    TrackingConfig cfg_;

    vector<Frame_> frames_;
    vector<int> free_frames_;
    MapPoints mps_;
    int cur_{-1}, ref_{-1}, pre_{-1}, last_keyframe_{-1}, pending_{-1};

    // map
    size_t window_size_;
    struct MapKf {
        ulong key;
        int frame;
    };
    vector<MapKf> map_kf_; // unordered in the reference; every use sorts or searches by key
    int latest_keyframe_{-1};
    bool is_window_full_{false};
    size_t n_landmarks_{0};
    // Map::landmarks_ by operation history: the same container type (key, hasher, bucket policy) fed the same insert / erase sequence
    // iterates in the same order, which is the order VisualWindow walks the landmarks in (window_visual.cc) — and so the order of the
    // optimizer's residual blocks and of every floating-point sum over them.  Value: the map point's pool index.
    typedef std::unordered_map<ulong, uint32_t, std::hash<ulong>, std::equal_to<ulong>, PoolAllocator<std::pair<const ulong, uint32_t>>> LandmarkOrder;
    LandmarkOrder map_lm_;

    // candidates (tracking.h:129-136)
    vector<Point2f> pts2d_cur_, pts2d_new_, pts2d_ref_, pts2d_ref_undis_, pts2d_new_undis_;
    vector<int> pts2d_ref_frame_;
    vector<int32_t> cand_lk_idx_; // per candidate: index of the LK point that produced pts2d_new_[k] in the group's last LK call (-1: detected)
    vector<Vector2d> velocity_ref_, velocity_cur_;
    struct MpRef {
        uint32_t i, g;
    };
    vector<MpRef> tracked_mappoint_, mappoint_matched_;

    int block_cols_, block_rows_, block_cnts_, block_w_, block_h_, track_max_block_features_;
    icg_detect_grid grid_{};
    double parallax_map_{0}, parallax_ref_{0};
    int parallax_map_counts_{0}, parallax_ref_counts_{0};
    bool isnewkeyframe_{false}, isinitializing_{true};
    double histogram_{0};
    int passed_cnt_{0};
    int track_min_pixel_distance_;
    double track_max_interval_;
    FILE *logfile_{nullptr};
    std::chrono::steady_clock::time_point t_start_;
    vector<double> logging_data_;

    // per-frame staged state
    bool done_{true};
    TrackState result_{TRACK_PASSED};
    int pending_slot_{-1};
    vector<int> owned_slots_;
    enum Mode { M_NONE, M_FIRST, M_INIT, M_TRACK } mode_{M_NONE};
    int det_job_{-1};
    bool det_ismask_{true};
    int det_frame_{-1};
    int lk_map_begin_{0}, lk_map_n_{0}, lk_ref_begin_{0}, lk_ref_n_{0};
    vector<Point2f> tm_pts2d_map_, tm_pred_;
    vector<double> tm_pc_;
    vector<int32_t> tm_hint_;
    vector<int> order_idx_; // rows of a frame in container order (scratch)
    static constexpr size_t kAhead = 8; // prefetch distance of the per-row loops, in rows
    bool ref_tracked_{false};
    int rs_set_{-1};
    vector<Point2f> tr_new_undis_, tr_cur_undis_;
    keyFrameState kf_state_{KEYFRAME_NONE};
    bool tri_queued_{false};
    int tri_begin_{0};
    vector<int> tri_point_index_;
    vector<uint8_t> tri_status_, status_;
    vector<Point2f> tri_ref_undis_, tri_cur_undis_, scratch_a_, scratch_b_;
    int lost_reset_{0};
    ulong last_input_fid_{0};
    vector<uint8_t> mark_;

    // core mode
    struct CoreFree {
        void operator()(tc::Stream *p) const { free(p); }
    };
    std::unique_ptr<tc::Stream, CoreFree> core_;
    tc::Cfg core_cfg_{};
    std::unique_ptr<tc::Scratch> core_scratch_;
    bool core_dirty_{false};
    bool core_changed_{false};          // exportCore() ran: a device-resident copy of the block is stale (tracking_device.h uploads it)
    bool core_device_resident_{false};  // the authoritative block lives in HBM; core_ is a downloaded copy
    int core_log_applied_{0};      // entries of core_->log already replayed into map_lm_
    vector<int> core_slots_;       // the device slots reserved for this stream (core slot pool)
    Mat core_image_format_;        // rows / cols / channels / step / device flag of the stream's frames (the block keeps addresses only)
    struct CoreArena {             // this stream's segment of the primitives' work lists (track_core.h Io)
        int32_t pre_slot{-1}, lk_count{0}, rs_count{0}, tri_count{0}, tri_n_tcw{0}, det_slot{-1}, det_mask_count{0}, det_count{0};
        double pre_hist{0};
        vector<int32_t> lk_prev_slot, lk_next_slot, tri_T0, tri_T1, det_quota;
        vector<tc::P2f> lk_prev, lk_guess, lk_out, lk_undist, rs_p1, rs_p2, det_mask_pts, det_out;
        vector<uint8_t> lk_status, rs_mask;
        vector<double> tri_Tcw, tri_pc0, tri_pc1, tri_pw;
    } arena_;
    int core_pre_job_{-1}, core_det_job_{-1};
    tc::Io coreIo(int lk_base);
    void coreBeginFrame(const Input &in, StageBatch &next);
    void coreAdvance(int stage, StageBatch &done, StageBatch &next);
    void coreQueueOutputs(StageBatch &next, bool pre, bool det, bool lk, bool rs, bool tri);
};

} // namespace icg
