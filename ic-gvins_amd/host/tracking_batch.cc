#include "tracking_batch.h"
#include "hostprof.h"

#include <cerrno>
#include <cstdlib>
#include <string>
#include <sys/stat.h>
#include <stdexcept>

#include <atomic>

namespace icg {

TrackingBatch::TrackingBatch(int device, int n_streams, const vector<double> &intrinsic, const vector<double> &distortion,
                             const vector<int> &size, const TrackingConfig &cfg, int window_size, int host_threads, int engine)
    : host_threads_(host_threads < 1 ? 1 : host_threads) {
    if (engine < 0) { // ICG_TRACK_ENGINE=object|table|core; the drawer hooks only exist in the object engine
        const char *e = getenv("ICG_TRACK_ENGINE");
        engine        = (e && e[0] == 'o') || cfg.is_use_visualization ? ENGINE_OBJECT : (e && e[0] == 'c') ? ENGINE_CORE : (e && e[0] == 'd') ? ENGINE_DEVICE : ENGINE_TABLE;
    }
    // ENGINE_CORE: the track table's interface on the tracker core (track_core.h) — the stage bodies of the device-resident tracker, here
    // compiled for the host and run between the same batched device calls as the table engine
    const bool core = engine == ENGINE_CORE;
    engine_ = engine == ENGINE_OBJECT ? ENGINE_OBJECT : (core ? ENGINE_CORE : engine == ENGINE_DEVICE ? ENGINE_DEVICE : ENGINE_TABLE);
    if (engine_ != ENGINE_OBJECT) HashOrder::verifyOnce(); // fails loudly if the container order cannot be reproduced here
    device_ = std::make_shared<DeviceContext>(device, size[0], size[1], n_streams, cfg.track_max_features);
    streams_.resize((size_t) n_streams);
    for (int i = 0; i < n_streams; i++) {
        Stream &s  = streams_[(size_t) i];
        s.camera   = Camera::createCamera(intrinsic, distortion, size);
        s.ids      = std::make_shared<IdSpace>();
        // tracking.txt (tracking.cc:309-315) per stream under $ICG_TRACKING_LOG_DIR/stream<i>/ when that is set
        std::string outputpath;
        if (const char *dir = getenv("ICG_TRACKING_LOG_DIR")) {
            outputpath = std::string(dir) + "/stream" + std::to_string(i);
            if (mkdir(outputpath.c_str(), 0755) != 0 && errno != EEXIST)
                throw std::runtime_error("TrackingBatch: cannot create " + outputpath);
        }
        if (engine_ != ENGINE_OBJECT) {
            s.table = std::make_shared<TableTracker>(s.camera, (size_t) window_size, cfg, outputpath, device_, s.ids);
            if (core) s.table->enableCore();
        } else {
            s.map      = std::make_shared<Map>((size_t) window_size);
            s.tracking = std::make_shared<Tracking>(s.camera, s.map, nullptr, cfg, outputpath, device_, s.ids);
            s.keeper   = std::make_shared<WindowKeeper>(s.map);
        }
    }
    if (host_threads_ > 1 && n_streams > 1) pool_.reset(new HostPool(std::min(host_threads_, n_streams)));
    device_->setCamera(*streams_[0].camera);
    if (engine_ == ENGINE_DEVICE) {
        const tc::Cfg C = TableTracker::makeCoreCfg(*streams_[0].camera, cfg, (size_t) window_size);
        icg_tracker_config tcfg;
        static_assert(sizeof(tcfg) == sizeof(C), "icg_tracker_config mirrors tc::Cfg");
        memcpy(&tcfg, &C, sizeof tcfg);
        const int rc = icg_tracker_create(device_->ctx(), n_streams, &tcfg, TableTracker::bucketsAfterTable(), tc::MAX_ROWS + 2, &tracker_);
        if (rc != ICG_OK) throw std::runtime_error(std::string("icg_tracker_create failed: ") + icg_last_error(device_->ctx()));
        for (int i = 0; i < n_streams; i++) {
            streams_[(size_t) i].tracker       = tracker_;
            streams_[(size_t) i].tracker_index = i;
        }
    }
    grid_        = engine_ != ENGINE_OBJECT ? streams_[0].table->grid() : streams_[0].tracking->grid();
    max_per_job_ = engine_ != ENGINE_OBJECT ? streams_[0].table->maxFeaturesPerJob() : streams_[0].tracking->maxFeaturesPerJob();
}

TrackingBatch::~TrackingBatch() {
    if (tracker_) icg_tracker_destroy(tracker_);
}

void TrackingBatch::Stream::syncDevice() const {
    if (!tracker || !device_stale) return;
    if (!table->coreMode()) table->enableCore(true); // the host copy of the block (2.7 MB) exists only for streams somebody looked at
    const int rc = icg_tracker_download(tracker, tracker_index, table->core());
    if (rc != ICG_OK) throw std::runtime_error("icg_tracker_download failed");
    table->markCoreChanged();
    device_stale = false;
}

// ---- ENGINE_DEVICE: one chain of launches per step, one wait (csrc/tracker.hip) ---------------------------------------------------------
void TrackingBatch::stepDevice(const FrameInput *frames, vector<TrackState> &states) {
    const int n = (int) streams_.size();
    hostprof::Scope hp_total(hostprof::STEP_TOTAL);
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    states.assign((size_t) n, TRACK_PASSED);
    dev_images_.assign((size_t) n, nullptr);
    dev_stamps_.assign((size_t) n, 0.0);
    dev_poses_.resize(12 * (size_t) n);
    dev_results_.resize((size_t) n);
    int stride = 0, channels = 1, on_device = 0;
    bool any = false;
    for (int i = 0; i < n; i++) {
        const FrameInput &in = frames[(size_t) i];
        Stream &s            = streams_[(size_t) i];
        s.view_.reset(); // (an object view is a snapshot between two frames)
        if (!in.valid) continue;
        if (any && ((int) in.image.step != stride || in.image.channels() != channels || (in.image.device ? 1 : 0) != on_device))
            throw std::runtime_error("TrackingBatch (device engine): the frames of one step must share stride, channels and residency");
        any = true, stride = (int) in.image.step, channels = in.image.channels(), on_device = in.image.device ? 1 : 0;
        dev_images_[(size_t) i] = in.image.data;
        dev_stamps_[(size_t) i] = in.stamp;
        poseToArray12(in.pose, dev_poses_.data() + 12 * (size_t) i);
        s.table->setCoreImageFormat(in.image);
    }
    if (any) {
        hostprof::Scope hp(hostprof::DEV_LK);
        const int rc = icg_tracker_step(tracker_, dev_images_.data(), stride, channels, on_device, dev_stamps_.data(), dev_poses_.data(), dev_results_.data());
        if (rc != ICG_OK) throw std::runtime_error(std::string("icg_tracker_step failed: ") + icg_last_error(device_->ctx()));
    }
    const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    timing[2] += t1 - t0;
    uint64_t lk = 0, det = 0, rs = 0, tri = 0, pre = 0;
    for (int i = 0; any && i < n; i++) {
        Stream &s                   = streams_[(size_t) i];
        const icg_tracker_result &r = dev_results_[(size_t) i];
        if (!r.active) continue;
        s.last         = r;
        s.device_stale = true;
        states[(size_t) i] = (TrackState) r.state;
        s.last_state = (TrackState) r.state, s.frames = r.frames, s.keyframes = r.keyframes, s.tracked_sum = r.tracked_sum, s.digest = r.digest;
        s.ids->frame_id = r.frame_id, s.ids->keyframe_id = r.keyframe_id, s.ids->mappoint_id = r.mappoint_id;
        lk += (uint64_t) r.lk_points, det += (uint64_t) r.detect_jobs, rs += (uint64_t) r.ransac_sets, tri += (uint64_t) r.tri_points, pre++;
        if (r.log_valid && s.table) s.table->writeTrackingLog(r.log_data, r.log_features, (t1 - t0) * 1e3); // (time cost: the group's step)
        // (ICG_TRACKER_LOG_DRAIN: drain threshold in operations, for tests; default half the block's capacity)
        static const int drain_at = getenv("ICG_TRACKER_LOG_DRAIN") ? std::max(1, atoi(getenv("ICG_TRACKER_LOG_DRAIN"))) : tc::LOG_CAP / 2;
        if (r.n_log > drain_at) dev_drain_.push_back(i), dev_drain_n_.push_back(r.n_log);
    }
    if (!dev_drain_.empty()) {
        // the landmark-container histories that are half full are fetched (the log only, 16 bytes per operation — not the 2.7 MB block) and
        // replayed into the streams' Map::landmarks_ twins before they can wrap; one wait for all of them
        dev_log_.resize((size_t) tc::LOG_CAP * dev_drain_.size());
        if (icg_tracker_fetch_logs(tracker_, (int) dev_drain_.size(), dev_drain_.data(), dev_drain_n_.data(), dev_log_.data(), tc::LOG_CAP) != ICG_OK)
            throw std::runtime_error(std::string("icg_tracker_fetch_logs failed: ") + icg_last_error(device_->ctx()));
        for (size_t k = 0; k < dev_drain_.size(); k++) {
            Stream &s = streams_[(size_t) dev_drain_[k]];
            s.table->coreApplyFetchedLog(dev_log_.data() + (size_t) tc::LOG_CAP * k, dev_drain_n_[k]);
            s.last.n_log = 0;
        }
        dev_drain_.clear(), dev_drain_n_.clear();
    }
    counters[0] += lk, counters[1] += lk ? 1 : 0, counters[2] += det, counters[3] += det ? 1 : 0, counters[4] += rs, counters[5] += rs ? 1 : 0;
    counters[6] += pre, counters[7] += tri;
    const double t2 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    timing[4] += t2 - t1;
    if (step_log.size() < kStepLogCap) step_log.push_back({t2, t2 - t1, t1 - t0});
    if (any) last_step_s_ = t2 - t0;
}

void TrackingBatch::Stream::currentFeatures(vector<std::pair<ulong, Point2f>> &out) const {
    syncDevice();
    out.clear();
    if (table) {
        table->forEachCurrentFeature([&](ulong id, const Point2f &kp) { out.emplace_back(id, kp); });
    } else if (auto f = tracking->currentFrame()) {
        f->forEachFeature([&](ulong id, Feature &ft) { out.emplace_back(id, ft.distortedKeyPoint()); });
    }
}

Map::Ptr TrackingBatch::Stream::objectMap() {
    if (!table) return map;
    syncDevice();
    if (!view_) view_ = table->view();
    return view_->map;
}

void TrackingBatch::Stream::commitMap() {
    if (table && view_) table->absorb(*view_);
    view_.reset();
    if (tracker && table->coreMode() && table->takeCoreChanged()) { // the write-back goes into the device's block
        if (icg_tracker_upload(tracker, tracker_index, table->core()) != ICG_OK) throw std::runtime_error("icg_tracker_upload failed");
        table->coreLogRestarted(); // (exportCore emptied the history: map_lm_ is current)
        // culling removes landmarks, refinement removes keyframes: the counts of the last step's results follow the uploaded block
        last.landmarks        = (int32_t) table->landmarks();
        last.window_keyframes = (int32_t) table->windowKeyFrames();
    }
}

std::string TrackingBatch::Stream::dump(int kind) const {
    syncDevice();
    if (table) return kind == 0 ? table->dump() : kind == 1 ? table->dumpMap() : table->dumpMaterialized();
    return kind == 0 ? TableTracker::dumpObjects(*tracking, *map) : std::string();
}

template <typename F> void TrackingBatch::forEachStream(F &&f) {
    const int n = (int) streams_.size();
    if (!pool_ || n < 2) {
        for (int i = 0; i < n; i++) f(i);
        return;
    }
    const std::function<void(int)> fn = std::ref(f);
    pool_->parallelFor(n, fn);
}

template <typename T> static void append(vector<T> &dst, const vector<T> &src) { dst.insert(dst.end(), src.begin(), src.end()); }

// bases[i] = {pre, det, mask_pts, lk, rs_set, rs_pts, tri, tri_T}
void TrackingBatch::gather(int cur, StageBatch &g, vector<std::array<int, 8>> &bases) {
    g.clear();
    bases.assign(streams_.size(), {});
    for (size_t i = 0; i < streams_.size(); i++) {
        const StageBatch &b = streams_[i].box[cur];
        auto &B             = bases[i];
        B[0] = (int) g.pre_slots.size();
        B[1] = (int) g.det_slots.size();
        B[2] = (int) (g.det_mask_pts.size() / 2);
        B[3] = (int) g.lk_prev_slot.size();
        B[4] = (int) g.rs_off.size() - 1;
        B[5] = (int) (g.rs_p1.size() / 2);
        B[6] = (int) g.tri_T0.size();
        B[7] = (int) (g.tri_Tcw.size() / 12);
        if (!b.pre_slots.empty()) {
            append(g.pre_slots, b.pre_slots);
            append(g.pre_imgs, b.pre_imgs);
            g.pre_stride   = b.pre_stride;
            g.pre_channels = b.pre_channels;
            g.pre_device   = b.pre_device;
            g.pre_want_hist |= b.pre_want_hist;
        }
        if (!b.det_slots.empty()) {
            append(g.det_slots, b.det_slots);
            append(g.det_quota, b.det_quota);
            append(g.det_mask_pts, b.det_mask_pts);
            for (size_t k = 1; k < b.det_mask_off.size(); k++) g.det_mask_off.push_back(b.det_mask_off[k] + B[2]);
        }
        if (!b.lk_prev_slot.empty()) {
            append(g.lk_prev_slot, b.lk_prev_slot);
            append(g.lk_next_slot, b.lk_next_slot);
            append(g.lk_prev, b.lk_prev);
            append(g.lk_guess, b.lk_guess);
        }
        if (b.rs_off.size() > 1) {
            append(g.rs_p1, b.rs_p1);
            append(g.rs_p2, b.rs_p2);
            for (size_t k = 1; k < b.rs_off.size(); k++) g.rs_off.push_back(b.rs_off[k] + B[5]);
            g.rs_thresh = b.rs_thresh;
            g.rs_conf   = b.rs_conf;
        }
        if (!b.tri_T0.empty()) {
            for (int v : b.tri_T0) g.tri_T0.push_back(v + B[7]);
            for (int v : b.tri_T1) g.tri_T1.push_back(v + B[7]);
            append(g.tri_pc0, b.tri_pc0);
            append(g.tri_pc1, b.tri_pc1);
        }
        append(g.tri_Tcw, b.tri_Tcw);
    }
}

void TrackingBatch::scatter(int cur, const StageBatch &g, const vector<std::array<int, 8>> &bases) {
    for (size_t i = 0; i < streams_.size(); i++) {
        StageBatch &b = streams_[i].box[cur];
        const auto &B = bases[i];
        if (!b.pre_slots.empty() && !g.pre_hist.empty())
            b.pre_hist.assign(g.pre_hist.begin() + B[0], g.pre_hist.begin() + B[0] + (long) b.pre_slots.size());
        if (!b.det_slots.empty()) {
            size_t n = b.det_slots.size();
            b.det_count.assign(g.det_count.begin() + B[1], g.det_count.begin() + B[1] + (long) n);
            b.det_out.assign(g.det_out.begin() + (size_t) B[1] * max_per_job_ * 2,
                             g.det_out.begin() + ((size_t) B[1] + n) * max_per_job_ * 2);
        }
        if (!b.lk_prev_slot.empty()) {
            size_t n  = b.lk_prev_slot.size();
            b.lk_base = B[3];
            b.lk_status.assign(g.lk_status.begin() + B[3], g.lk_status.begin() + B[3] + (long) n);
            b.lk_out.assign(g.lk_out.begin() + 2 * (size_t) B[3], g.lk_out.begin() + 2 * ((size_t) B[3] + n));
            b.lk_undist.assign(g.lk_undist.begin() + 2 * (size_t) B[3], g.lk_undist.begin() + 2 * ((size_t) B[3] + n));
        }
        if (b.rs_off.size() > 1) {
            size_t n = (size_t) b.rs_off.back();
            b.rs_mask.assign(g.rs_mask.begin() + B[5], g.rs_mask.begin() + B[5] + (long) n);
        }
        if (!b.tri_T0.empty()) {
            size_t n = b.tri_T0.size();
            b.tri_pw.assign(g.tri_pw.begin() + 3 * (size_t) B[6], g.tri_pw.begin() + 3 * ((size_t) B[6] + n));
        }
    }
}

static inline void fnv(uint64_t &h, const void *p, size_t n) {
    const unsigned char *c = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) {
        h ^= c[i];
        h *= 1099511628211ull;
    }
}

// per-feature hash of (map-point id, key-point bits): two rounds of a 64-bit finalizer (the digest visits ~300 features per frame
// of every stream; byte-wise FNV over the 16 bytes was 5 us of host time per frame)
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

static inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void TrackingBatch::step(const FrameInput *frames, vector<TrackState> &states) {
    if (engine_ == ENGINE_DEVICE) return stepDevice(frames, states);
    const int n = (int) streams_.size();
    hostprof::Scope hp_total(hostprof::STEP_TOTAL);
    double t0 = now_s(), t1;
    const double log_host0 = timing[0], log_dev0 = timing[2];
    states.assign((size_t) n, TRACK_PASSED);
    vector<char> active((size_t) n, 0);
    int cur = 0;
    forEachStream([&](int i) {
        Stream &s = streams_[(size_t) i];
        s.box[0].clear();
        s.box[1].clear();
        const FrameInput &in = frames[(size_t) i];
        if (in.valid) {
            active[(size_t) i] = 1;
            hostprof::Scope hp(hostprof::BEGIN_FRAME);
            s.view_.reset(); // (an object view of the table is a snapshot between two frames)
            if (s.table) {
                TableTracker::Input ti;
                ti.stamp = in.stamp, ti.image = in.image, ti.pose = in.pose;
                s.table->beginFrame(ti, s.box[0]);
            } else {
                s.frame = Frame::createFrame(in.stamp, in.image, s.ids);
                s.frame->setPose(in.pose);
                s.tracking->beginFrame(s.frame, s.box[0]);
            }
        }
    });
    t1 = now_s();
    timing[0] += t1 - t0;
    t0 = t1;
    StageBatch &global = global_;
    vector<std::array<int, 8>> &bases = bases_;
    for (int stage = 1; stage < Tracking::N_STAGES; stage++) {
        gather(cur, global, bases);
        t1 = now_s();
        timing[1] += t1 - t0;
        t0 = t1;
        if (!global.lk_prev_slot.empty()) {
            counters[0] += global.lk_prev_slot.size();
            counters[1]++;
        }
        if (!global.det_slots.empty()) {
            counters[2] += global.det_slots.size();
            counters[3]++;
        }
        if (global.rs_off.size() > 1) {
            counters[4] += global.rs_off.size() - 1;
            counters[5]++;
        }
        counters[6] += global.pre_slots.size();
        counters[7] += global.tri_T0.size();
        device_->execute(global, grid_, max_per_job_);
        t1 = now_s();
        timing[2] += t1 - t0;
        t0 = t1;
        scatter(cur, global, bases);
        t1 = now_s();
        timing[3] += t1 - t0;
        t0 = t1;
        const int nxt = cur ^ 1;
        bool any      = false;
        forEachStream([&](int i) {
            Stream &s = streams_[(size_t) i];
            s.box[nxt].clear();
            if (active[(size_t) i] && !s.frameDone()) {
                hostprof::Scope hp(stage);
                if (s.table)
                    s.table->advance(stage, s.box[cur], s.box[nxt]);
                else
                    s.tracking->advance(stage, s.box[cur], s.box[nxt]);
            }
        });
        for (int i = 0; i < n; i++)
            if (active[(size_t) i] && !streams_[(size_t) i].frameDone()) any = true;
        cur = nxt;
        t1  = now_s();
        timing[0] += t1 - t0;
        t0 = t1;
        if (!any) break;
    }
    forEachStream([&](int i) {
        if (!active[(size_t) i]) return;
        Stream &s     = streams_[(size_t) i];
        TrackState st = s.result();
        states[(size_t) i] = st;
        if (s.table && s.table->coreMode()) { // statistics, digest and the window keeper are part of the core's end-of-frame stage
            hostprof::Scope hpk(hostprof::KEEPER);
            s.table->endFrame();
            const tc::Stream &C = *s.table->core();
            s.last_state = st, s.frames = C.frames, s.keyframes = C.keyframes, s.tracked_sum = C.tracked_sum, s.digest = C.digest;
            s.ids->frame_id = C.frame_id, s.ids->keyframe_id = C.keyframe_id, s.ids->mappoint_id = C.mappoint_id;
            return;
        }
        s.last_state       = st;
        s.frames++;
        if (s.isNewKeyFrame() || st == TRACK_FIRST_FRAME || st == TRACK_LOST) s.keyframes++;
        // digest of everything index-like this frame produced: state, frame id, (map-point id, pixel bits) per feature and the
        // number of surviving un-triangulated reference points
        {
            hostprof::Scope hpd(hostprof::DIGEST);
            int sti = (int) st;
            fnv(s.digest, &sti, sizeof sti);
            ulong fid = s.table ? s.table->lastInputFrameId() : s.frame->id();
            fnv(s.digest, &fid, sizeof fid);
            if (st != TRACK_PASSED) {
                // order-independent combination of per-feature hashes (the container order is not part of the result)
                uint64_t acc = 0, cnt = 0;
                auto one = [&](ulong id, const Point2f &kp) {
                    uint64_t bits;
                    memcpy(&bits, &kp, sizeof bits);
                    const uint64_t hf = mix64(mix64((uint64_t) id) ^ bits);
                    acc += hf;
                    cnt++;
                };
                if (s.table)
                    s.table->forEachCurrentFeature(one);
                else
                    s.frame->forEachFeature([&](ulong id, Feature &f) { one(id, f.distortedKeyPoint()); });
                fnv(s.digest, &acc, sizeof acc);
                fnv(s.digest, &cnt, sizeof cnt);
                s.tracked_sum += cnt;
                uint64_t nref = s.trackedRefPoints().size();
                fnv(s.digest, &nref, sizeof nref);
            }
        }
        hostprof::Scope hpk(hostprof::KEEPER);
        if (s.table) {
            s.table->endFrame();
        } else {
            s.keeper->onFrame(*s.tracking, s.frame, st);
            s.frame.reset();
        }
    });
    t1 = now_s();
    timing[4] += t1 - t0;
    if (step_log.size() < kStepLogCap) step_log.push_back({t1, timing[0] - log_host0, timing[2] - log_dev0});
}

// ---- StreamGroups -----------------------------------------------------------------------------------------------------
StreamGroups::StreamGroups(int device, int n_streams, int n_groups, const vector<double> &intrinsic,
                           const vector<double> &distortion, const vector<int> &size, const TrackingConfig &cfg,
                           int window_size, int host_threads_per_group)
    : n_streams_(n_streams) {
    if (n_groups < 1) n_groups = 1;
    if (n_groups > n_streams) n_groups = n_streams;
    group_of_.resize((size_t) n_streams);
    local_of_.resize((size_t) n_streams);
    int begin = 0;
    for (int g = 0; g < n_groups; g++) {
        int cnt = n_streams / n_groups + (g < n_streams % n_groups ? 1 : 0);
        group_begin_.push_back(begin);
        groups_.emplace_back(new TrackingBatch(device, cnt, intrinsic, distortion, size, cfg, window_size, host_threads_per_group));
        for (int k = 0; k < cnt; k++) {
            group_of_[(size_t) (begin + k)] = g;
            local_of_[(size_t) (begin + k)] = k;
        }
        begin += cnt;
    }
    group_begin_.push_back(begin);
    stagger_ = !(getenv("ICG_GROUP_STAGGER") && getenv("ICG_GROUP_STAGGER")[0] == '0');
    if (n_groups > 1) {
        // many contexts share the host cores: waits must not spin (a spinning wait burns the core another group needs).
        // 100 us between stream queries: measured on MI355X with 32 groups / 16 usable cores, 20 us and 100 us give the same
        // throughput (the other groups keep the GPU busy while one sleeps) and the longer sleep leaves ~2 more cores idle
        for (auto &g : groups_) (void) icg_ctx_set_wait_mode(g->device()->ctx(), ICG_WAIT_POLL, 100);
        for (int g = 0; g < n_groups; g++) workers_.emplace_back(&StreamGroups::workerLoop, this, g);
    }
}

StreamGroups::~StreamGroups() {
    {
        std::unique_lock<std::mutex> lock(m_);
        stop_ = true;
        generation_++;
    }
    cv_go_.notify_all();
    for (auto &t : workers_) t.join();
}

void StreamGroups::workerLoop(int g) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lock(m_);
            cv_go_.wait(lock, [&] { return generation_ != seen; });
            seen = generation_;
            if (stop_) return;
        }
        const int b = group_begin_[(size_t) g], e = group_begin_[(size_t) g + 1];
        std::string err;
        const double tw0 = now_s();
        try {
            vector<TrackState> st;
            if (replay_reps_ > 0) groups_[(size_t) g]->replay(replay_reps_, (size_t) g); // staggered: group g starts at its g-th recorded call
            // Device engine, several steps in one call: the groups leave the call's start line one G-th of a step apart.  Their launch chains
            // are identical, so groups that start together stay in phase for ~20 steps — all in their stage kernels (the chip idles), then
            // all in LK (the chip is contended): 7.7 ms per step instead of 6.9 until they drift apart.  (ICG_GROUP_STAGGER=0 turns it off.)
            if (replay_reps_ == 0 && frames_->size() > 1 && groups_[(size_t) g]->engine() == TrackingBatch::ENGINE_DEVICE && stagger_) {
                const double T = groups_[(size_t) g]->lastStepSeconds();
                const double d = T > 0 && T < 0.1 ? T * g / (double) groups_.size() : 0.0;
                if (d > 0) std::this_thread::sleep_for(std::chrono::duration<double>(d));
            }
            for (size_t k = 0; replay_reps_ == 0 && k < frames_->size(); k++) {
                groups_[(size_t) g]->step((*frames_)[k].data() + b, st);
                for (int i = b; i < e; i++) (*states_)[k][(size_t) i] = st[(size_t) (i - b)];
            }
        } catch (const std::exception &ex) {
            err = ex.what();
        }
        if (getenv("ICG_DEBUG_TIMING") && replay_reps_ == 0 && frames_->size() > 1) {
            const double *t = groups_[(size_t) g]->timing;
            fprintf(stderr, "[group %d] worker wall %.2f ms, in-step accumulated %.2f ms\n", g, 1e3 * (now_s() - tw0),
                    1e3 * (t[0] + t[1] + t[2] + t[3] + t[4]));
        }
        {
            std::unique_lock<std::mutex> lock(m_);
            if (!err.empty()) error_ = err;
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
}

void StreamGroups::step(const vector<FrameInput> &frames, vector<TrackState> &states) {
    vector<vector<FrameInput>> f1(1, frames);
    vector<vector<TrackState>> s1;
    stepMany(f1, s1);
    states = s1[0];
}

void StreamGroups::replayAll(int reps) {
    if (reps <= 0) return;
    if (workers_.empty()) {
        groups_[0]->replay(reps);
        return;
    }
    {
        std::unique_lock<std::mutex> lock(m_);
        replay_reps_ = reps;
        pending_     = (int) groups_.size();
        error_.clear();
        generation_++;
    }
    cv_go_.notify_all();
    std::unique_lock<std::mutex> lock(m_);
    cv_done_.wait(lock, [&] { return pending_ == 0; });
    replay_reps_ = 0;
    if (!error_.empty()) throw std::runtime_error(error_);
}

void StreamGroups::stepMany(const vector<vector<FrameInput>> &frames, vector<vector<TrackState>> &states) {
    states.assign(frames.size(), vector<TrackState>((size_t) n_streams_, TRACK_PASSED));
    if (workers_.empty()) {
        for (size_t k = 0; k < frames.size(); k++) groups_[0]->step(frames[k].data(), states[k]);
        return;
    }
    {
        std::unique_lock<std::mutex> lock(m_);
        frames_  = &frames;
        states_  = &states;
        pending_ = (int) groups_.size();
        error_.clear();
        generation_++;
    }
    cv_go_.notify_all();
    std::unique_lock<std::mutex> lock(m_);
    cv_done_.wait(lock, [&] { return pending_ == 0; });
    if (!error_.empty()) throw std::runtime_error(error_);
}

} // namespace icg
