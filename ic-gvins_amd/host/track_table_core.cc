// TableTracker in core mode (track_table.h): the staged interface on the tracker core (track_core.h, the stage bodies of the device-resident
// tracker compiled for the host), and the two-way conversion between the core's flat block and the table members.
#include <cstdlib>
#include <mutex>
#include <stdexcept>

#include "track_table.h"

namespace icg {

static_assert(sizeof(tc::P2f) == sizeof(Point2f), "tc::P2f mirrors Point2f");
static_assert(sizeof(tc::Pose) == 12 * sizeof(double), "tc::Pose is R (row-major) | t");

const uint32_t *TableTracker::bucketsAfterTable() {
    static const vector<uint32_t> table = [] {
        vector<uint32_t> t((size_t) tc::MAX_ROWS + 2);
        for (size_t k = 0; k < t.size(); k++) t[k] = (uint32_t) HashOrder::bucketsAfter(k);
        if (t.back() > (uint32_t) tc::MAX_BUCKETS) throw std::runtime_error("tracker core: MAX_BUCKETS too small for this standard library");
        return t;
    }();
    return table.data();
}

tc::Cfg TableTracker::makeCoreCfg(const Camera &camera, const TrackingConfig &cfg, size_t window_size) {
    tc::Cfg C;
    memset(&C, 0, sizeof C);
    const icg_camera a = camera.abi();
    C.cam.fx = a.fx, C.cam.fy = a.fy, C.cam.cx = a.cx, C.cam.cy = a.cy, C.cam.skew = a.skew;
    C.cam.k1 = a.k1, C.cam.k2 = a.k2, C.cam.p1 = a.p1, C.cam.p2 = a.p2, C.cam.k3 = a.k3;
    C.cam.width = camera.width(), C.cam.height = camera.height();
    C.track_max_features     = cfg.track_max_features;
    C.check_histogram        = cfg.track_check_histogram ? 1 : 0;
    C.window_size            = (int) window_size;
    C.track_min_parallax     = cfg.track_min_parallax;
    C.reprojection_error_std = cfg.reprojection_error_std;
    C.track_max_interval     = cfg.track_max_interval * 0.95; // tracking.cc:57
    // tracking.cc:66-85
    const double TRACK_BLOCK_SIZE = 200.0;
    C.block_cols         = static_cast<int>(lround(camera.width() / TRACK_BLOCK_SIZE));
    C.block_rows         = static_cast<int>(lround(camera.height() / TRACK_BLOCK_SIZE));
    C.block_cnts         = C.block_cols * C.block_rows;
    C.block_h            = camera.height() / C.block_rows;
    C.block_w            = camera.width() / C.block_cols;
    C.max_block_features = static_cast<int>(lround(static_cast<double>(cfg.track_max_features) / static_cast<double>(C.block_cnts)));
    C.min_pixel_distance = static_cast<int>(round(TRACK_BLOCK_SIZE / sqrt(C.max_block_features * 1.5)));
    C.max_per_job        = C.max_block_features * C.block_cnts;
    if (C.block_cnts > tc::MAX_BLOCKS) throw std::runtime_error("tracker core: more detection blocks than MAX_BLOCKS");
    if ((int) window_size + 2 > tc::MAX_WINDOW) throw std::runtime_error("tracker core: window larger than MAX_WINDOW");
    if (C.max_per_job + 64 > tc::MAX_ROWS) throw std::runtime_error("tracker core: feature budget larger than MAX_ROWS");
    return C;
}

void TableTracker::enableCore(bool device_resident) {
    if (core_) return;
    core_device_resident_ = device_resident;
    HashOrder::verifyOnce();
    core_cfg_ = makeCoreCfg(*camera_, cfg_, window_size_);
    (void) bucketsAfterTable();
    tc::Stream *S = static_cast<tc::Stream *>(calloc(1, sizeof(tc::Stream)));
    if (!S) throw std::bad_alloc();
    core_.reset(S);
    tc::stream_init(*S, 0);
    if (!device_resident) {
        // the stream's own slot pool: MAX_SLOTS slots of the context, reserved for its lifetime
        for (int k = 0; k < tc::MAX_SLOTS; k++) core_slots_.push_back(device_->allocSlot());
        for (int k = 0; k < tc::MAX_SLOTS; k++) S->free_slots[k] = core_slots_[(size_t) (tc::MAX_SLOTS - 1 - k)];
        S->n_free_slots = tc::MAX_SLOTS;
    }
    arena_.lk_prev_slot.resize(tc::MAX_ROWS), arena_.lk_next_slot.resize(tc::MAX_ROWS);
    arena_.lk_prev.resize(tc::MAX_ROWS), arena_.lk_guess.resize(tc::MAX_ROWS), arena_.lk_out.resize(tc::MAX_ROWS), arena_.lk_undist.resize(tc::MAX_ROWS);
    arena_.lk_status.resize(tc::MAX_ROWS), arena_.rs_mask.resize(tc::MAX_ROWS);
    arena_.rs_p1.resize(tc::MAX_ROWS), arena_.rs_p2.resize(tc::MAX_ROWS);
    arena_.tri_T0.resize(tc::MAX_ROWS), arena_.tri_T1.resize(tc::MAX_ROWS);
    arena_.tri_Tcw.resize(12 * tc::MAX_TCW), arena_.tri_pc0.resize(3 * tc::MAX_ROWS), arena_.tri_pc1.resize(3 * tc::MAX_ROWS), arena_.tri_pw.resize(3 * tc::MAX_ROWS);
    arena_.det_quota.resize(tc::MAX_BLOCKS), arena_.det_mask_pts.resize(tc::MAX_ROWS), arena_.det_out.resize(tc::MAX_ROWS);
    core_scratch_.reset(new tc::Scratch);
    core_dirty_       = true;
    core_log_applied_ = 0;
}

tc::Io TableTracker::coreIo(int lk_base) {
    tc::Io io;
    memset(&io, 0, sizeof io);
    io.pre_slot = &arena_.pre_slot, io.pre_hist = &arena_.pre_hist;
    io.lk_count = &arena_.lk_count, io.lk_prev_slot = arena_.lk_prev_slot.data(), io.lk_next_slot = arena_.lk_next_slot.data();
    io.lk_prev = arena_.lk_prev.data(), io.lk_guess = arena_.lk_guess.data(), io.lk_out = arena_.lk_out.data(), io.lk_undist = arena_.lk_undist.data();
    io.lk_status = arena_.lk_status.data(), io.lk_base = lk_base;
    io.rs_count = &arena_.rs_count, io.rs_p1 = arena_.rs_p1.data(), io.rs_p2 = arena_.rs_p2.data(), io.rs_mask = arena_.rs_mask.data();
    io.tri_count = &arena_.tri_count, io.tri_n_tcw = &arena_.tri_n_tcw, io.tri_T0 = arena_.tri_T0.data(), io.tri_T1 = arena_.tri_T1.data();
    io.tri_Tcw = arena_.tri_Tcw.data(), io.tri_pc0 = arena_.tri_pc0.data(), io.tri_pc1 = arena_.tri_pc1.data(), io.tri_pw = arena_.tri_pw.data();
    io.det_slot = &arena_.det_slot, io.det_quota = arena_.det_quota.data(), io.det_mask_count = &arena_.det_mask_count;
    io.det_mask_pts = arena_.det_mask_pts.data(), io.det_count = &arena_.det_count, io.det_out = arena_.det_out.data();
    return io;
}

// what the stage body left in the stream's arena goes into the stage batch of the host executor (which concatenates the streams' lists)
void TableTracker::coreQueueOutputs(StageBatch &next, bool pre, bool det, bool lk, bool rs, bool tri) {
    (void) pre;
    if (det && arena_.det_slot >= 0) {
        core_det_job_ = (int) next.det_slots.size();
        next.det_slots.push_back(arena_.det_slot);
        const float *m = reinterpret_cast<const float *>(arena_.det_mask_pts.data());
        next.det_mask_pts.insert(next.det_mask_pts.end(), m, m + 2 * (size_t) arena_.det_mask_count);
        next.det_mask_off.push_back((int32_t) (next.det_mask_pts.size() / 2));
        next.det_quota.insert(next.det_quota.end(), arena_.det_quota.begin(), arena_.det_quota.begin() + core_cfg_.block_cnts);
    }
    if (lk && arena_.lk_count > 0) {
        const size_t n = (size_t) arena_.lk_count;
        next.lk_prev_slot.insert(next.lk_prev_slot.end(), arena_.lk_prev_slot.begin(), arena_.lk_prev_slot.begin() + (long) n);
        next.lk_next_slot.insert(next.lk_next_slot.end(), arena_.lk_next_slot.begin(), arena_.lk_next_slot.begin() + (long) n);
        const float *p = reinterpret_cast<const float *>(arena_.lk_prev.data()), *g = reinterpret_cast<const float *>(arena_.lk_guess.data());
        next.lk_prev.insert(next.lk_prev.end(), p, p + 2 * n);
        next.lk_guess.insert(next.lk_guess.end(), g, g + 2 * n);
    }
    if (rs && arena_.rs_count > 0) {
        const size_t m = (size_t) arena_.rs_count;
        next.rs_thresh = cfg_.reprojection_error_std;
        const float *a = reinterpret_cast<const float *>(arena_.rs_p1.data()), *b = reinterpret_cast<const float *>(arena_.rs_p2.data());
        next.rs_p1.insert(next.rs_p1.end(), a, a + 2 * m);
        next.rs_p2.insert(next.rs_p2.end(), b, b + 2 * m);
        next.rs_off.push_back((int32_t) (next.rs_p1.size() / 2));
    }
    if (tri) {
        const int base = (int) (next.tri_Tcw.size() / 12);
        next.tri_Tcw.insert(next.tri_Tcw.end(), arena_.tri_Tcw.begin(), arena_.tri_Tcw.begin() + 12 * (long) arena_.tri_n_tcw);
        for (int k = 0; k < arena_.tri_count; k++) {
            next.tri_T0.push_back(arena_.tri_T0[(size_t) k] + base);
            next.tri_T1.push_back(arena_.tri_T1[(size_t) k] + base);
        }
        next.tri_pc0.insert(next.tri_pc0.end(), arena_.tri_pc0.begin(), arena_.tri_pc0.begin() + 3 * (long) arena_.tri_count);
        next.tri_pc1.insert(next.tri_pc1.end(), arena_.tri_pc1.begin(), arena_.tri_pc1.begin() + 3 * (long) arena_.tri_count);
    }
}

void TableTracker::coreBeginFrame(const Input &in, StageBatch &next) {
    t_start_  = std::chrono::steady_clock::now();
    tc::Io io = coreIo(0);
    tc::Pose pose;
    poseToArray12(in.pose, pose.R); // R row-major then t: the 12 doubles of tc::Pose
    core_image_format_      = in.image;
    core_image_format_.data = nullptr;
    core_image_format_.storage.reset();
    tc::stage_begin_frame(*core_, io, in.stamp, pose, (tc::u64) (uintptr_t) in.image.data);
    core_dirty_ = true;
    next.pre_slots.push_back(arena_.pre_slot);
    next.pre_imgs.push_back(in.image.data);
    next.pre_stride   = (int) in.image.step;
    next.pre_channels = in.image.channels();
    next.pre_device   = in.image.device;
    if (cfg_.track_check_histogram) next.pre_want_hist = true;
    core_pre_job_ = (int) next.pre_slots.size() - 1;
    core_det_job_ = -1;
}

void TableTracker::coreAdvance(int stage, StageBatch &done, StageBatch &next) {
    tc::Stream &S = *core_;
    if (S.done) return;
    const uint32_t *ba = bucketsAfterTable();
    tc::Io io          = coreIo(done.lk_base);
    auto take_detection = [&] {
        if (core_det_job_ < 0) return;
        const int max_per_job = maxFeaturesPerJob();
        arena_.det_count      = done.det_count[(size_t) core_det_job_];
        if (arena_.det_count > 0)
            memcpy(arena_.det_out.data(), done.det_out.data() + (size_t) core_det_job_ * max_per_job * 2, sizeof(tc::P2f) * (size_t) arena_.det_count);
        core_det_job_ = -1;
    };
    switch (stage) {
    case 1:
        if (cfg_.track_check_histogram) arena_.pre_hist = done.pre_hist[(size_t) core_pre_job_];
        tc::stage_on_preprocess(S, core_cfg_, io, *core_scratch_);
        coreQueueOutputs(next, false, true, false, false, false);
        break;
    case 2:
        take_detection();
        tc::stage_on_detect_a(S, core_cfg_, io, *core_scratch_);
        coreQueueOutputs(next, false, false, true, false, false);
        break;
    case 3: {
        const size_t n = done.lk_status.size();
        if (n) {
            memcpy(arena_.lk_status.data(), done.lk_status.data(), n);
            memcpy(arena_.lk_out.data(), done.lk_out.data(), n * sizeof(tc::P2f));
            memcpy(arena_.lk_undist.data(), done.lk_undist.data(), n * sizeof(tc::P2f));
        }
        tc::stage_on_lk(S, core_cfg_, io, ba, *core_scratch_);
        coreQueueOutputs(next, false, false, false, true, false);
        break;
    }
    case 4:
        if (done.rs_off.size() > 1 && S.rs_set >= 0) memcpy(arena_.rs_mask.data(), done.rs_mask.data() + done.rs_off[0], (size_t) (done.rs_off[1] - done.rs_off[0]));
        tc::stage_on_ransac(S, core_cfg_, io, *core_scratch_);
        coreQueueOutputs(next, false, false, false, false, true);
        break;
    case 5:
        if (!done.tri_pw.empty()) memcpy(arena_.tri_pw.data(), done.tri_pw.data(), done.tri_pw.size() * sizeof(double));
        tc::stage_on_triangulate(S, core_cfg_, io, ba, *core_scratch_);
        coreQueueOutputs(next, false, true, false, false, false);
        break;
    case 6:
        take_detection();
        tc::stage_on_detect_b(S, core_cfg_, io);
        break;
    default: break;
    }
    core_dirty_ = true;
}

// ---- block -> table members -----------------------------------------------------------------------------------------------------------
static inline Pose poseOf(const tc::Pose &p) { return poseFromArray12(p.R); }

void TableTracker::importCore() {
    static_assert(sizeof(Row) == sizeof(tc::Row) && sizeof(MapPointHot) == sizeof(tc::MpHot), "records shared with the tracker core");
    const tc::Stream &S = *core_;
    frames_.resize((size_t) S.n_frames);
    for (int h = 0; h < S.n_frames; h++) {
        const tc::Frame &c = S.frame[h];
        Frame_ &f          = frames_[(size_t) h];
        f.alive    = c.alive != 0;
        f.gen      = c.gen;
        f.fid      = (ulong) c.fid;
        f.kf_id    = (ulong) c.kf_id;
        f.stamp    = c.stamp;
        f.pose     = poseOf(c.pose);
        f.is_kf    = c.is_kf != 0;
        f.kf_state = c.kf_state;
        f.slot     = c.slot;
        f.image    = core_image_format_;
        f.image.data = reinterpret_cast<uint8_t *>((uintptr_t) c.image);
        if (!c.alive) f.image = Mat();
        f.row.resize((size_t) c.n_rows);
        if (c.n_rows) memcpy((void *) f.row.data(), c.row, sizeof(Row) * (size_t) c.n_rows);
        f.order.restore(c.next, reinterpret_cast<const uint64_t *>(&c.row[0].id), sizeof(tc::Row), c.n_rows, c.bucket, c.n_buckets, c.head, c.magic);
        f.unupdated.assign(c.unupd, c.unupd + c.n_unupd);
        f.unupdated_gen.assign(c.unupd_gen, c.unupd_gen + c.n_unupd);
    }
    free_frames_.assign(S.free_frames, S.free_frames + S.n_free_frames);
    mps_.hot.resize((size_t) S.n_mps);
    mps_.cold.resize((size_t) S.n_mps);
    if (S.n_mps) memcpy((void *) mps_.hot.data(), S.hot, sizeof(MapPointHot) * (size_t) S.n_mps);
    for (int i = 0; i < S.n_mps; i++) {
        MapPointCold &d = mps_.cold[(size_t) i];
        const tc::MpCold &c = S.cold[i];
        d.born_fid = (ulong) c.born_fid, d.ref_frame = c.ref_frame, d.ref_gen = c.ref_gen, d.ref_kp = Point2f(c.ref_kp.x, c.ref_kp.y), d.depth = c.depth,
        d.optimized = c.optimized;
    }
    mps_.free_list.assign(S.free_mps, S.free_mps + S.n_free_mps);
    cur_ = S.cur, ref_ = S.ref, pre_ = S.pre, last_keyframe_ = S.last_keyframe, pending_ = S.pending, latest_keyframe_ = S.latest_keyframe;
    det_frame_ = S.det_frame;
    map_kf_.clear();
    for (int k = 0; k < S.n_map_kf; k++) map_kf_.push_back({(ulong) S.map_kf_key[k], S.map_kf_frame[k]});
    is_window_full_ = S.is_window_full != 0;
    n_landmarks_    = (size_t) S.n_landmarks;
    // Map::landmarks_ iteration order: replay the operations logged since the last import into the container itself
    for (int k = core_log_applied_; k < S.n_log; k++) {
        if (S.log[k].op)
            map_lm_.insert(std::make_pair((ulong) S.log[k].id, S.log[k].mp));
        else
            map_lm_.erase((ulong) S.log[k].id);
    }
    core_log_applied_ = S.n_log;
    if (S.n_log > tc::LOG_CAP / 2 && !core_device_resident_) { // drained: the host-resident block restarts its history
        core_->n_log      = 0;
        core_log_applied_ = 0;
    }
    auto pts = [](vector<Point2f> &v, const tc::P2f *p, int n) {
        v.resize((size_t) n);
        if (n) memcpy((void *) v.data(), p, sizeof(Point2f) * (size_t) n);
    };
    pts(pts2d_cur_, S.pts2d_cur, S.n_cur), pts(pts2d_new_, S.pts2d_new, S.n_new), pts(pts2d_ref_, S.pts2d_ref, S.n_ref);
    pts(pts2d_ref_undis_, S.pts2d_ref_undis, S.n_ref_undis), pts(pts2d_new_undis_, S.pts2d_new_undis, S.n_new_undis);
    pts(tr_new_undis_, S.tr_new_undis, S.n_tr_new_undis), pts(tr_cur_undis_, S.tr_cur_undis, S.n_tr_cur_undis);
    pts2d_ref_frame_.assign(S.pts2d_ref_frame, S.pts2d_ref_frame + S.n_ref_frame);
    cand_lk_idx_.assign(S.cand_lk_idx, S.cand_lk_idx + S.n_cand_lk);
    velocity_ref_.resize((size_t) S.n_vel_ref);
    for (int k = 0; k < S.n_vel_ref; k++) velocity_ref_[(size_t) k] = Vector2d(S.velocity_ref[k][0], S.velocity_ref[k][1]);
    velocity_cur_.resize((size_t) S.n_vel_cur);
    for (int k = 0; k < S.n_vel_cur; k++) velocity_cur_[(size_t) k] = Vector2d(S.velocity_cur[k][0], S.velocity_cur[k][1]);
    tracked_mappoint_.resize((size_t) S.n_tracked);
    for (int k = 0; k < S.n_tracked; k++) tracked_mappoint_[(size_t) k] = {S.tracked_mappoint[k].i, S.tracked_mappoint[k].g};
    parallax_map_ = S.parallax_map, parallax_ref_ = S.parallax_ref, parallax_map_counts_ = S.parallax_map_counts, parallax_ref_counts_ = S.parallax_ref_counts;
    isnewkeyframe_ = S.isnewkeyframe != 0, isinitializing_ = S.isinitializing != 0;
    histogram_ = S.histogram, passed_cnt_ = S.passed_cnt;
    done_ = S.done != 0, result_ = (TrackState) S.result;
    last_input_fid_ = (ulong) S.last_input_fid;
    ids_->frame_id = (ulong) S.frame_id, ids_->keyframe_id = (ulong) S.keyframe_id, ids_->mappoint_id = (ulong) S.mappoint_id;
    core_dirty_ = false;
}

// ---- table members -> block (after absorb(): poses, flags, landmark positions / counters / removals) ---------------------------------------
void TableTracker::exportCore() {
    tc::Stream &S = *core_;
    if ((int) frames_.size() > tc::MAX_FRAMES || (int) mps_.size() > tc::MAX_MPS) throw std::runtime_error("tracker core: table larger than the block");
    S.n_frames = (int) frames_.size();
    for (int h = 0; h < S.n_frames; h++) {
        tc::Frame &c    = S.frame[h];
        const Frame_ &f = frames_[(size_t) h];
        c.alive = f.alive ? 1 : 0, c.gen = f.gen, c.fid = f.fid, c.kf_id = f.kf_id, c.stamp = f.stamp;
        poseToArray12(f.pose, c.pose.R);
        c.is_kf = f.is_kf ? 1 : 0, c.kf_state = f.kf_state, c.slot = f.slot;
        c.image = (tc::u64) (uintptr_t) f.image.data;
        if ((int) f.rows() > tc::MAX_ROWS) throw std::runtime_error("tracker core: frame larger than MAX_ROWS");
        c.n_rows = (int) f.rows();
        if (c.n_rows) memcpy((void *) c.row, f.row.data(), sizeof(Row) * (size_t) c.n_rows);
        if (c.n_rows) memcpy(c.next, f.order.nextData(), sizeof(int) * (size_t) c.n_rows);
        c.n_buckets = f.order.bucketCount();
        if (c.n_buckets) memcpy(c.bucket, f.order.bucketData(), sizeof(int) * (size_t) c.n_buckets);
        c.head = f.order.head(), c.magic = f.order.magic();
        c.n_unupd = (int) f.unupdated.size();
        for (int k = 0; k < c.n_unupd; k++) c.unupd[k] = f.unupdated[(size_t) k], c.unupd_gen[k] = f.unupdated_gen[(size_t) k];
    }
    S.n_free_frames = (int) free_frames_.size();
    for (int k = 0; k < S.n_free_frames; k++) S.free_frames[k] = free_frames_[(size_t) k];
    S.n_mps = (int) mps_.size();
    if (S.n_mps) memcpy((void *) S.hot, mps_.hot.data(), sizeof(MapPointHot) * (size_t) S.n_mps);
    for (int i = 0; i < S.n_mps; i++) {
        const MapPointCold &d = mps_.cold[(size_t) i];
        tc::MpCold &c         = S.cold[i];
        c.born_fid = d.born_fid, c.ref_frame = d.ref_frame, c.ref_gen = d.ref_gen, c.ref_kp.x = d.ref_kp.x, c.ref_kp.y = d.ref_kp.y, c.depth = d.depth,
        c.optimized = d.optimized, c.pad_ = 0;
    }
    S.n_free_mps = (int) mps_.free_list.size();
    for (int k = 0; k < S.n_free_mps; k++) S.free_mps[k] = mps_.free_list[(size_t) k];
    S.cur = cur_, S.ref = ref_, S.pre = pre_, S.last_keyframe = last_keyframe_, S.pending = pending_, S.latest_keyframe = latest_keyframe_, S.det_frame = det_frame_;
    S.n_map_kf = (int) map_kf_.size();
    for (int k = 0; k < S.n_map_kf; k++) S.map_kf_key[k] = map_kf_[(size_t) k].key, S.map_kf_frame[k] = map_kf_[(size_t) k].frame;
    S.is_window_full = is_window_full_ ? 1 : 0;
    S.n_landmarks    = (int) n_landmarks_;
    // (map_lm_ is already current: absorb() erased from it directly; the block's history is spent)
    S.n_log           = 0;
    core_log_applied_ = 0;
    auto pts = [](tc::P2f *p, int32_t &n, const vector<Point2f> &v) {
        n = (int32_t) v.size();
        if (n) memcpy((void *) p, v.data(), sizeof(Point2f) * v.size());
    };
    pts(S.pts2d_cur, S.n_cur, pts2d_cur_), pts(S.pts2d_new, S.n_new, pts2d_new_), pts(S.pts2d_ref, S.n_ref, pts2d_ref_);
    pts(S.pts2d_ref_undis, S.n_ref_undis, pts2d_ref_undis_), pts(S.pts2d_new_undis, S.n_new_undis, pts2d_new_undis_);
    pts(S.tr_new_undis, S.n_tr_new_undis, tr_new_undis_), pts(S.tr_cur_undis, S.n_tr_cur_undis, tr_cur_undis_);
    S.n_ref_frame = (int) pts2d_ref_frame_.size();
    for (int k = 0; k < S.n_ref_frame; k++) S.pts2d_ref_frame[k] = pts2d_ref_frame_[(size_t) k];
    S.n_cand_lk = (int) cand_lk_idx_.size();
    for (int k = 0; k < S.n_cand_lk; k++) S.cand_lk_idx[k] = cand_lk_idx_[(size_t) k];
    S.n_vel_ref = (int) velocity_ref_.size();
    for (int k = 0; k < S.n_vel_ref; k++) S.velocity_ref[k][0] = velocity_ref_[(size_t) k][0], S.velocity_ref[k][1] = velocity_ref_[(size_t) k][1];
    S.n_vel_cur = (int) velocity_cur_.size();
    for (int k = 0; k < S.n_vel_cur; k++) S.velocity_cur[k][0] = velocity_cur_[(size_t) k][0], S.velocity_cur[k][1] = velocity_cur_[(size_t) k][1];
    S.n_tracked = (int) tracked_mappoint_.size();
    for (int k = 0; k < S.n_tracked; k++) S.tracked_mappoint[k].i = tracked_mappoint_[(size_t) k].i, S.tracked_mappoint[k].g = tracked_mappoint_[(size_t) k].g;
    S.parallax_map = parallax_map_, S.parallax_ref = parallax_ref_, S.parallax_map_counts = parallax_map_counts_, S.parallax_ref_counts = parallax_ref_counts_;
    S.isnewkeyframe = isnewkeyframe_ ? 1 : 0, S.isinitializing = isinitializing_ ? 1 : 0;
    S.histogram = histogram_, S.passed_cnt = passed_cnt_;
    S.frame_id = ids_->frame_id, S.keyframe_id = ids_->keyframe_id, S.mappoint_id = ids_->mappoint_id;
    core_dirty_   = false;
    core_changed_ = true;
}

} // namespace icg
