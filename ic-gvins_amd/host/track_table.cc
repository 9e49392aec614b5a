// Table engine of the front-end (see track_table.h).  Every function names the reference lines it follows; the object-graph twin
// of each stage body is in tracking_hip.cc.
#include "track_table.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "hostprof.h"

#if !defined(__GLIBCXX__)
#error "HashOrder reproduces libstdc++'s std::unordered_map node order (bits/hashtable.h); build the table engine with libstdc++"
#endif

namespace icg {

// ---- HashOrder -------------------------------------------------------------------------------------------------------------
int HashOrder::selfTest(uint64_t seed, int n, int check_every, bool dense_ids) {
    std::unordered_map<ulong, int> ref;
    HashOrder h;
    h.clear();
    uint64_t x    = seed * 0x9E3779B97F4A7C15ull + 1;
    ulong next_id = seed % 1000;
    for (int k = 0; k < n; k++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        ulong key = dense_ids ? (next_id += 1 + (x % 3)) : (ulong) (x % 1000003);
        if (ref.count(key)) {
            if (h.insert(key)) return k + 1; // duplicates must be refused
            continue;
        }
        ref.emplace(key, (int) h.size());
        if (!h.insert(key)) return k + 1;
        if ((k + 1) % check_every == 0 || k + 1 == n) {
            int r = h.head();
            for (const auto &kv : ref) {
                if (r < 0 || r != kv.second) return k + 1;
                r = h.next(r);
            }
            if (r >= 0) return k + 1;
        }
    }
    return 0;
}

void HashOrder::verifyOnce() {
    static std::once_flag once;
    static int bad = 0;
    std::call_once(once, [] {
        // a frame holds a few hundred features: 1500 insertions cross six rehashes; dense ids as the id factories hand them out, and
        // scattered ones (~0.2 ms altogether)
        bad = selfTest(1, 1500, 50, true);
        if (!bad) bad = selfTest(2, 1500, 50, false);
    });
    if (bad)
        throw std::runtime_error("HashOrder: this standard library's std::unordered_map iterates differently from the emulation (first "
                                 "difference after " + std::to_string(bad) + " insertions); use ICG_TRACK_ENGINE=object");
}

size_t HashOrder::bucketsAfter(size_t k) {
    static const vector<uint32_t> table = [] {
        // ask the standard library itself: bucket_count() after every insertion into a fresh map (insert-only history, which is
        // all a Frame's features_ ever sees); depends on the element count only
        const size_t N = 1 << 15;
        vector<uint32_t> t(N + 1);
        std::unordered_map<ulong, char> m;
        t[0] = (uint32_t) m.bucket_count();
        for (size_t i = 1; i <= N; i++) {
            m.emplace((ulong) i, 0);
            t[i] = (uint32_t) m.bucket_count();
        }
        return t;
    }();
    if (k >= table.size()) throw std::runtime_error("HashOrder: more than 32768 features in one frame");
    return table[k];
}

// key % n without a division for keys below 2^32 (the ids of this code base are small counters): Lemire's fastmod, exact for every
// 32-bit key and divisor; M = floor((2^64 - 1) / n) + 1 is computed once per bucket count
static inline uint64_t modMagic(size_t n) { return UINT64_C(0xFFFFFFFFFFFFFFFF) / (uint64_t) n + 1; }
static inline size_t bucketOf(ulong key, size_t n, uint64_t M) {
    if ((key >> 32) != 0 || (n >> 32) != 0) return (size_t) (key % n);
    const uint64_t low = M * (uint64_t) (uint32_t) key;
    return (size_t) (((__uint128_t) low * (uint64_t) n) >> 64);
}

void HashOrder::rehash(size_t n) { // bits/hashtable.h _M_rehash_aux(__n, true_type)
    vector<int> &nb = scratch_;
    nb.assign(n, kEmpty);
    const uint64_t M = modMagic(n);
    int p = head_;
    head_ = -1;
    size_t bbegin_bkt = 0;
    while (p >= 0) {
        const int nx   = next_[(size_t) p];
        const size_t b = bucketOf(key_[(size_t) p], n, M);
        if (nb[b] == kEmpty) {
            next_[(size_t) p] = head_;
            head_             = p;
            nb[b]             = kBeforeBegin;
            if (next_[(size_t) p] >= 0) nb[bbegin_bkt] = p;
            bbegin_bkt = b;
        } else {
            const int prev    = nb[b];
            next_[(size_t) p] = prev == kBeforeBegin ? head_ : next_[(size_t) prev];
            if (prev == kBeforeBegin)
                head_ = p;
            else
                next_[(size_t) prev] = p;
        }
        p = nx;
    }
    bucket_.swap(nb);
    magic_ = M;
}

bool HashOrder::contains(ulong key) const {
    const size_t n = bucket_.size();
    const size_t b = bucketOf(key, n, magic_);
    const int prev = bucket_[b];
    if (prev == kEmpty) return false;
    for (int p = nextOf(prev); p >= 0 && bucketOf(key_[(size_t) p], n, magic_) == b; p = next_[(size_t) p])
        if (key_[(size_t) p] == key) return true;
    return false;
}

void HashOrder::insertUnique(ulong key) { // _M_insert_unique_node + _M_insert_bucket_begin
    const size_t want = bucketsAfter(next_.size() + 1);
    if (want != bucket_.size()) rehash(want);
    const int i    = (int) next_.size();
    const size_t n = bucket_.size(), b = bucketOf(key, n, magic_);
    key_.push_back(key);
    next_.push_back(-1);
    if (bucket_[b] != kEmpty) {
        const int prev    = bucket_[b];
        next_[(size_t) i] = nextOf(prev);
        setNext(prev, i);
    } else {
        next_[(size_t) i] = head_;
        head_             = i;
        if (next_[(size_t) i] >= 0) bucket_[bucketOf(key_[(size_t) next_[(size_t) i]], n, magic_)] = i;
        bucket_[b] = kBeforeBegin;
    }
}

// ---- pools -------------------------------------------------------------------------------------------------------------------
void TableTracker::Frame_::clearRows() {
    row.clear();
    order.clear();
    unupdated.clear();
    unupdated_gen.clear();
}

uint32_t TableTracker::MapPoints::alloc() {
    uint32_t i;
    if (!free_list.empty()) {
        i = free_list.back();
        free_list.pop_back();
    } else {
        i = (uint32_t) hot.size();
        hot.emplace_back();
        cold.emplace_back();
    }
    const uint32_t g = hot[i].gen;
    hot[i]           = MapPointHot();
    hot[i].gen       = g;
    hot[i].live      = 1;
    cold[i]          = MapPointCold();
    return i;
}

int TableTracker::allocFrame() {
    int h;
    if (!free_frames_.empty()) {
        h = free_frames_.back();
        free_frames_.pop_back();
    } else {
        h = (int) frames_.size();
        frames_.emplace_back();
    }
    Frame_ &f  = frames_[(size_t) h];
    f.alive    = true;
    f.fid      = 0;
    f.kf_id    = 0;
    f.is_kf    = false;
    f.kf_state = KEYFRAME_NORMAL;
    f.slot     = -1;
    f.clearRows();
    return h;
}

void TableTracker::freeFrame(int h) {
    Frame_ &f = frames_[(size_t) h];
    // a map point lives as long as the map or a frame's unupdated list holds it; this frame's list goes away with it
    for (size_t k = 0; k < f.unupdated.size(); k++) {
        const uint32_t i = f.unupdated[k];
        if (mps_.valid(i, f.unupdated_gen[k]) && !mps_.hot[i].in_map) mps_.release(i);
    }
    f.alive = false;
    f.gen++;
    f.image = Mat();
    f.clearRows();
    free_frames_.push_back(h);
}

// A Frame of the reference dies with its last shared_ptr: the tracker's roles, the candidates' reference frames, the map.
void TableTracker::sweepFrames() {
    mark_.assign(frames_.size(), 0);
    auto mark = [&](int h) {
        if (h >= 0) mark_[(size_t) h] = 1;
    };
    mark(cur_), mark(pre_), mark(ref_), mark(last_keyframe_), mark(pending_), mark(latest_keyframe_), mark(det_frame_);
    for (const auto &k : map_kf_) mark(k.frame);
    for (int h : pts2d_ref_frame_) mark(h);
    for (size_t h = 0; h < frames_.size(); h++)
        if (frames_[h].alive && !mark_[h]) freeFrame((int) h);
}

void TableTracker::setKeyFrame(int h, int state) { // frame.cc:42-54
    Frame_ &f = frames_[(size_t) h];
    if (!f.is_kf) {
        f.is_kf    = true;
        f.kf_id    = ids_->keyframe_id++;
        f.kf_state = state;
    }
}

int TableTracker::addRow(int h, ulong id, uint32_t mp, const Point2f &kp, const Point2f &kpd, const Vector2d &vel, FeatureType type,
                         double pcx, double pcy, bool unique_key, int32_t lk_idx) {
    Frame_ &f = frames_[(size_t) h];
    if (unique_key)
        f.order.insertUnique(id);
    else if (!f.order.insert(id))
        return -1; // std::unordered_map::insert of an existing key adds nothing (frame.h:71-74)
    f.row.push_back(Row{id, mp, mps_.hot[mp].gen, kp, kpd, vel, pcx, pcy, lk_idx, (int8_t) type, 0});
    return (int) f.row.size() - 1;
}

// ---- map (tracking/map.cc) -----------------------------------------------------------------------------------------------------
int TableTracker::mapFind(ulong key) const {
    for (size_t k = 0; k < map_kf_.size(); k++)
        if (map_kf_[k].key == key) return (int) k;
    return -1;
}

bool TableTracker::mapIsKeyFrameInMap(int h) const { return mapFind(frames_[(size_t) h].kf_id) >= 0; } // map.h: find(frame->keyFrameId())

void TableTracker::mapInsertKeyFrame(int h) { // map.cc:27-61
    latest_keyframe_ = h;
    Frame_ &f        = frames_[(size_t) h];
    const int at     = mapFind(f.kf_id);
    if (at < 0)
        map_kf_.push_back({f.kf_id, h});
    else
        map_kf_[(size_t) at].frame = h;
    for (size_t k = 0; k < f.unupdated.size(); k++) {
        const uint32_t i = f.unupdated[k];
        if (!mps_.valid(i, f.unupdated_gen[k])) continue;
        if (!mps_.hot[i].in_map) {
            mps_.hot[i].in_map = 1;
            map_lm_.insert(std::make_pair(mps_.hot[i].id, i)); // map.cc:56-61
            n_landmarks_++;
        }
    }
    if (map_kf_.size() > window_size_) is_window_full_ = true;
}

void TableTracker::mapRemoveKeyFrame(int h, bool isremovemappoint) { // map.cc:89-127
    Frame_ &f = frames_[(size_t) h];
    if (isremovemappoint) {
        for (const Row &r : f.row) {
            const uint32_t i = r.mp;
            if (!mps_.valid(i, r.mpgen)) continue;
            if (mps_.cold[i].ref_frame == h && mps_.cold[i].ref_gen == f.gen && mps_.hot[i].in_map) {
                // removeAllObservations + setOutlier + landmarks_.erase: nothing can reach it any more
                mps_.hot[i].in_map  = 0;
                mps_.hot[i].outlier = 1;
                map_lm_.erase(mps_.hot[i].id);
                n_landmarks_--;
                mps_.release(i);
            }
        }
        // Frame::clearFeatures (frame.h:46-51): features and the unupdated list
        for (size_t k = 0; k < f.unupdated.size(); k++) {
            const uint32_t i = f.unupdated[k];
            if (mps_.valid(i, f.unupdated_gen[k]) && !mps_.hot[i].in_map) mps_.release(i);
        }
        f.clearRows();
    }
    const int at = mapFind(f.kf_id);
    if (at >= 0) map_kf_.erase(map_kf_.begin() + at);
}

void TableTracker::writeTrackingLog(const double data5[5], int features, double cost_ms) {
    if (!logfile_) return;
    for (int k = 0; k < 5; k++) fprintf(logfile_, "%-15.9lf ", data5[k]);
    fprintf(logfile_, "%-15.9lf %-15.9lf \n", static_cast<double>(features), cost_ms);
    fflush(logfile_);
}

// Sliding-window stand-in (WindowKeeper::onFrame; ic_gvins.cc:542, 743, 1391-1410, 445-448, 1675)
void TableTracker::endFrame() {
    if (core_) { // statistics, digest, window keeper and sweep are the core's (tc::stage_end_frame)
        // tracking.txt (:236-238, 309-315): the decision's numbers were kept by the core; the line is written when the frame ends as TRACKING
        if (logfile_ && core_->log_valid && core_->result == tc::TRACK_TRACKING && core_->mode == tc::M_TRACK && core_->lost_reset != 2) {
            logging_data_.assign(core_->log_data, core_->log_data + 5);
            logging_data_.push_back(static_cast<double>(core_->frame[core_->cur].n_rows));
            logging_data_.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start_).count());
            for (double v : logging_data_) fprintf(logfile_, "%-15.9lf ", v);
            fprintf(logfile_, "\n");
            fflush(logfile_);
        }
        tc::stage_end_frame(*core_, core_cfg_);
        core_dirty_ = true;
        if (core_->overflow) throw std::runtime_error("tracker core: capacity exceeded (overflow flags " + std::to_string(core_->overflow) + ")");
        if (core_->n_log > tc::LOG_CAP / 2) importCore(); // (drains the landmark history into map_lm_)
        return;
    }
    const TrackState st = result_;
    const int frame     = cur_;
    if (st != TRACK_PASSED && frame >= 0 && (isnewkeyframe_ || st == TRACK_FIRST_FRAME || st == TRACK_LOST)) {
        {
            hostprof::Scope hp(hostprof::KEEP_INSERT);
            mapInsertKeyFrame(frame);
        }
        hostprof::Scope hp_rm(hostprof::KEEP_REMOVE);
        vector<ulong> ids;
        for (const auto &k : map_kf_) ids.push_back(k.key);
        std::sort(ids.begin(), ids.end());
        for (ulong id : ids) {
            const int at = mapFind(id);
            if (at < 0) continue;
            const int h = map_kf_[(size_t) at].frame;
            Frame_ &f   = frames_[(size_t) h];
            if ((f.kf_state == KEYFRAME_REMOVE_SECOND_NEW) || ((f.rows() == 0) && (id != ids.back()))) {
                f.is_kf    = false; // resetKeyFrame (frame.h:58-63); the keyframe id stays
                f.kf_state = KEYFRAME_NONE;
                mapRemoveKeyFrame(h, false);
            }
        }
        while (map_kf_.size() > window_size_) {
            size_t oldest = 0;
            for (size_t k = 1; k < map_kf_.size(); k++)
                if (map_kf_[k].key < map_kf_[oldest].key) oldest = k;
            mapRemoveKeyFrame(map_kf_[oldest].frame, true);
        }
    }
    sweepFrames();
}

// ---- construction (tracking.cc:32-86) --------------------------------------------------------------------------------------------
TableTracker::TableTracker(Camera::Ptr camera, size_t window_size, const TrackingConfig &config, const std::string &outputpath,
                           DeviceContext::Ptr device, std::shared_ptr<IdSpace> ids)
    : camera_(std::move(camera)), device_(std::move(device)), ids_(std::move(ids)), cfg_(config), window_size_(window_size) {
    if (cfg_.is_use_visualization) throw std::runtime_error("TableTracker: the drawer hooks need the object engine (icg::Tracking)");
    if (!outputpath.empty()) {
        logfile_ = fopen((outputpath + "/tracking.txt").c_str(), "w");
        if (!logfile_) throw std::runtime_error("Tracking: failed to open " + outputpath + "/tracking.txt");
    }
    track_max_interval_ = cfg_.track_max_interval * 0.95; // :57
    block_cols_ = static_cast<int>(lround(camera_->width() / TRACK_BLOCK_SIZE));  // :66
    block_rows_ = static_cast<int>(lround(camera_->height() / TRACK_BLOCK_SIZE)); // :67
    block_cnts_ = block_cols_ * block_rows_;
    block_h_    = camera_->height() / block_rows_; // :71
    block_w_    = camera_->width() / block_cols_;  // :72
    track_max_block_features_ =
        static_cast<int>(lround(static_cast<double>(cfg_.track_max_features) / static_cast<double>(block_cnts_))); // :81
    track_min_pixel_distance_ = static_cast<int>(round(TRACK_BLOCK_SIZE / sqrt(track_max_block_features_ * 1.5))); // :85
    grid_.block_cols    = block_cols_;
    grid_.block_rows    = block_rows_;
    grid_.block_w       = block_w_;
    grid_.block_h       = block_h_;
    grid_.min_dist      = track_min_pixel_distance_;
    grid_.max_per_block = track_max_block_features_;
}

TableTracker::~TableTracker() {
    if (logfile_) fclose(logfile_);
    for (int s : core_slots_) device_->freeSlot(s);
    for (int s : owned_slots_) device_->freeSlot(s);
    if (pending_slot_ >= 0) device_->freeSlot(pending_slot_);
}

ulong TableTracker::currentFrameId() const {
    if (core_) return core_->cur >= 0 ? (ulong) core_->frame[core_->cur].fid : 0;
    return cur_ >= 0 ? frames_[(size_t) cur_].fid : 0;
}
size_t TableTracker::numCurrentFeatures() const {
    if (core_) return core_->cur >= 0 ? (size_t) core_->frame[core_->cur].n_rows : 0;
    return cur_ >= 0 ? frames_[(size_t) cur_].rows() : 0;
}

// ---- helpers -------------------------------------------------------------------------------------------------------------------
template <typename T> void TableTracker::reduceVector(T &vec, const vector<uint8_t> &status) { // :831-839
    size_t index = 0;
    for (size_t k = 0; k < vec.size(); k++)
        if (status[k]) {
            if (index != k) vec[index] = vec[k];
            index++;
        }
    vec.resize(index);
}

bool TableTracker::isOnBorder(const Point2f &pts) const { // :847-849
    return pts.x < 5.0 || pts.y < 5.0 || (pts.x > (camera_->width() - 5.0)) || (pts.y > (camera_->height() - 5.0));
}

bool TableTracker::isGoodToTrack(const Point2f &pp, const Pose &pose, const Vector3d &pw, double scale, double depth_scale) const { // :813-829
    Vector3d pc = Camera::world2cam(pw, pose);
    if (!((pc[2] > MapPoint::NEAREST_DEPTH) && (pc[2] < MapPoint::FARTHEST_DEPTH * depth_scale))) return false; // :247-249
    if (camera_->reprojectionError(pose, pw, pp).norm() > cfg_.reprojection_error_std * scale) return false;
    return true;
}

double TableTracker::keyPointParallax(const Point2f &pp0, const Point2f &pp1, const Matrix3d &R10) const { // :861-871
    Vector3d pc0  = camera_->pixel2cam(pp0);
    Vector3d pc1  = camera_->pixel2cam(pp1);
    Vector3d pc01 = R10 * pc0;
    return Vector2d(pc01[0] - pc1[0], pc01[1] - pc1[1]).norm() * camera_->focalLength();
}

// order_idx_ := the rows of f in container order; requests the rows and the map-point records of the first few (see queueTrackMappoint)
size_t TableTracker::listContainerOrder(const Frame_ &f) {
    order_idx_.clear();
    for (int q = f.order.head(); q >= 0; q = f.order.next(q)) {
        order_idx_.push_back(q);
        const char *line = reinterpret_cast<const char *>(&f.row[(size_t) q]);
        __builtin_prefetch(line);
        __builtin_prefetch(line + sizeof(Row) - 1);
    }
    const size_t nq = order_idx_.size();
    for (size_t k = 0; k < std::min(kAhead, nq); k++) __builtin_prefetch(&mps_.hot[f.row[(size_t) order_idx_[k]].mp]);
    return nq;
}

int TableTracker::parallaxFromReferenceMapPoints(double &parallax) { // :873-905
    parallax   = 0;
    int counts = 0;
    const Frame_ &fc   = frames_[(size_t) cur_];
    const Frame_ &fr   = frames_[(size_t) ref_];
    const Matrix3d R10 = fc.pose.R.transpose() * fr.pose.R;
    const double focal = camera_->focalLength();
    const size_t nq = listContainerOrder(fr);
    for (size_t k = 0; k < nq; k++) {
        if (k + kAhead < nq) __builtin_prefetch(&mps_.hot[fr.row[(size_t) order_idx_[k + kAhead]].mp]);
        const Row &r0    = fr.row[(size_t) order_idx_[k]];
        const uint32_t i = r0.mp;
        if (!mps_.valid(i, r0.mpgen) || mps_.hot[i].outlier) continue; // getMapPoint() && !isOutlier()
        const LastObs &lo = mps_.hot[i].last;                           // observations().back().lock()
        if (lo.frame != cur_ || lo.gen != fc.gen) continue;             // feat && feat->getFrame() == frame_cur_
        const Row &r1 = fc.row[(size_t) lo.row];
        if (r1.outlier) continue; // feat && !feat->isOutlier() (:884)
        // keyPointParallax (:861-871) on the rows' stored pixel2cam values
        const double x = R10(0, 0) * r0.pcx + R10(0, 1) * r0.pcy + R10(0, 2) * 1.0, y = R10(1, 0) * r0.pcx + R10(1, 1) * r0.pcy + R10(1, 2) * 1.0;
        parallax += Vector2d(x - r1.pcx, y - r1.pcy).norm() * focal;
        counts++;
    }
    if (counts != 0) parallax /= counts;
    return counts;
}

void TableTracker::checkCarriedUndistortion(const char *where) {
    static const bool on = getenv("ICG_HOST_CHECK") != nullptr;
    if (!on) return;
    auto same = [&](const vector<Point2f> &src, const vector<Point2f> &carried, const char *what) {
        if (src.size() != carried.size())
            throw std::runtime_error(std::string("carried undistortion size mismatch (") + what + ") at " + where);
        vector<Point2f> u = src;
        camera_->undistortPoints(u);
        for (size_t k = 0; k < u.size(); k++)
            if (memcmp(&u[k], &carried[k], sizeof(Point2f)) != 0)
                throw std::runtime_error(std::string("carried undistortion differs (") + what + ") at " + where);
    };
    same(pts2d_ref_, pts2d_ref_undis_, "ref");
    if (where[0] == 't' && where[2] == 'i')
        same(pts2d_cur_, tr_cur_undis_, "cur");
    else
        same(pts2d_new_, pts2d_new_undis_, "new");
}

int TableTracker::parallaxFromReferenceKeyPoints(const vector<Point2f> &ref, const vector<Point2f> &cur, double &parallax) { // :907-922
    parallax   = 0;
    int counts = 0;
    const Matrix3d R10 = frames_[(size_t) cur_].pose.R.transpose() * frames_[(size_t) ref_].pose.R;
    for (size_t k = 0; k < pts2d_ref_frame_.size(); k++) {
        if (pts2d_ref_frame_[k] == ref_) {
            parallax += keyPointParallax(ref[k], cur[k], R10);
            counts++;
        }
    }
    if (counts != 0) parallax /= counts;
    return counts;
}

double TableTracker::relativeTranslation() const { return (frames_[(size_t) cur_].pose.t - frames_[(size_t) ref_].pose.t).norm(); } // :331-333

double TableTracker::relativeRotation() const { // :335-341
    Matrix3d R   = frames_[(size_t) cur_].pose.R.transpose() * frames_[(size_t) ref_].pose.R;
    double pitch = atan(-R(2, 0) / sqrt(R(2, 1) * R(2, 1) + R(2, 2) * R(2, 2)));
    return fabs(pitch * (180.0 / M_PI));
}

bool TableTracker::doResetTracking() { // :317-329
    if (!frames_[(size_t) cur_].rows()) {
        isinitializing_ = true;
        ref_            = cur_;
        pts2d_new_.clear();
        pts2d_ref_.clear();
        pts2d_ref_undis_.clear();
        pts2d_new_undis_.clear();
        pts2d_ref_frame_.clear();
        velocity_ref_.clear();
        cand_lk_idx_.clear();
        return true;
    }
    return false;
}

void TableTracker::writeLoggingMessage() { // :309-315
    logging_data_.push_back(static_cast<double>(frames_[(size_t) cur_].rows()));
    logging_data_.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start_).count());
    if (logfile_) {
        for (double v : logging_data_) fprintf(logfile_, "%-15.9lf ", v);
        fprintf(logfile_, "\n");
        fflush(logfile_);
    }
}

keyFrameState TableTracker::checkKeyFrameSate() { // :263-307
    keyFrameState keyframe_state = KEYFRAME_NONE;
    double dt                    = frames_[(size_t) cur_].stamp - frames_[(size_t) last_keyframe_].stamp;
    if (dt < TRACK_MIN_INTERVAl) return keyframe_state;
    double parallax = (parallax_map_ * parallax_map_counts_ + parallax_ref_ * parallax_ref_counts_) /
                      (parallax_map_counts_ + parallax_ref_counts_);
    if (parallax > cfg_.track_min_parallax) {
        keyframe_state = mapIsWindowFull() ? KEYFRAME_REMOVE_OLDEST : KEYFRAME_NORMAL;
    } else if (dt > track_max_interval_) {
        keyframe_state = KEYFRAME_REMOVE_SECOND_NEW;
    }
    if (keyframe_state != KEYFRAME_NONE) {
        last_keyframe_ = cur_;
        for (const auto &m : tracked_mappoint_)
            if (mps_.valid(m.i, m.g)) mps_.hot[m.i].used++;
        logging_data_.clear();
        logging_data_.push_back(frames_[(size_t) cur_].stamp);
        logging_data_.push_back(dt);
        logging_data_.push_back(parallax);
        logging_data_.push_back(relativeTranslation());
        logging_data_.push_back(relativeRotation());
    }
    return keyframe_state;
}

// ---- device slots ------------------------------------------------------------------------------------------------------------
void TableTracker::assignSlot(int h) {
    frames_[(size_t) h].slot = pending_slot_;
    owned_slots_.push_back(pending_slot_);
    pending_slot_ = -1;
}

void TableTracker::releaseUnusedSlots() {
    size_t keep = 0;
    for (int s : owned_slots_) {
        bool used = (cur_ >= 0 && frames_[(size_t) cur_].slot == s) || (pre_ >= 0 && frames_[(size_t) pre_].slot == s) ||
                    (ref_ >= 0 && frames_[(size_t) ref_].slot == s);
        if (used)
            owned_slots_[keep++] = s;
        else
            device_->freeSlot(s);
    }
    owned_slots_.resize(keep);
}

// ---- stage 0: preprocessing (tracking.cc:107-142) -------------------------------------------------------------------------------
void TableTracker::beginFrame(const Input &in, StageBatch &next) {
    if (core_) return coreBeginFrame(in, next);
    t_start_       = std::chrono::steady_clock::now();
    done_          = false;
    result_        = TRACK_PASSED;
    isnewkeyframe_ = false; // :108
    mode_          = M_NONE;
    det_job_       = -1;
    rs_set_        = -1;
    tri_queued_    = false;
    lk_map_n_ = lk_ref_n_ = 0;
    ref_tracked_   = false;
    pending_       = allocFrame();
    Frame_ &f      = frames_[(size_t) pending_];
    f.fid          = ids_->frame_id++; // frame.cc:37-40
    last_input_fid_ = f.fid;
    f.stamp        = in.stamp;
    f.pose         = in.pose;
    f.image        = in.image;
    pending_slot_  = device_->allocSlot();
    next.pre_slots.push_back(pending_slot_);
    next.pre_imgs.push_back(in.image.data);
    next.pre_stride   = (int) in.image.step;
    next.pre_channels = in.image.channels();
    next.pre_device   = in.image.device;
    if (cfg_.track_check_histogram) next.pre_want_hist = true;
    det_job_ = (int) next.pre_slots.size() - 1; // "preprocess job index" until stage 1
}

void TableTracker::advance(int stage, StageBatch &done, StageBatch &next) {
    if (core_) return coreAdvance(stage, done, next);
    if (done_) return;
    switch (stage) {
    case 1: onPreprocessDone(done, next); break;
    case 2: onDetectADone(done, next); break;
    case 3: onLKDone(done, next); break;
    case 4: onRansacDone(done, next); break;
    case 5: onTriangulateDone(done, next); break;
    case 6: onDetectBDone(done); break;
    default: break;
    }
}

void TableTracker::finish(TrackState st) {
    result_ = st;
    done_   = true;
}

void TableTracker::onPreprocessDone(StageBatch &done, StageBatch &next) {
    if (cfg_.track_check_histogram) { // :115-133
        double hist = done.pre_hist[(size_t) det_job_];
        if (histogram_ != 0) {
            double rate = fabs((hist - histogram_) / histogram_);
            if (rate > 0.1) {
                passed_cnt_++;
                if (passed_cnt_ > 1) histogram_ = 0;
                device_->freeSlot(pending_slot_);
                pending_slot_ = -1;
                freeFrame(pending_);
                pending_ = -1;
                finish(TRACK_PASSED);
                return;
            }
        }
        histogram_ = hist;
    }
    det_job_ = -1;
    pre_     = cur_; // :135
    cur_     = pending_;
    pending_ = -1;
    assignSlot(cur_);
    releaseUnusedSlots();

    if (isinitializing_) {
        if (ref_ < 0) { // :158-166
            doResetTracking();
            ref_  = cur_;
            mode_ = M_FIRST;
            queueDetection(ref_, false, next);
            return;
        }
        mode_ = M_INIT;
        if (pts2d_ref_.empty()) queueDetection(ref_, false, next); // :168-170
    } else {
        mode_ = M_TRACK;
    }
}

void TableTracker::onDetectADone(StageBatch &done, StageBatch &next) {
    if (det_job_ >= 0) {
        hostprof::Scope hp(hostprof::DET_INTEGRATE);
        integrateDetection(done);
    }
    if (mode_ == M_FIRST) {
        releaseUnusedSlots();
        finish(TRACK_FIRST_FRAME);
        return;
    }
    if (mode_ == M_TRACK) { // :206
        hostprof::Scope hp(hostprof::QUEUE_MAP);
        queueTrackMappoint(next);
    }
    hostprof::Scope hp(hostprof::QUEUE_REF);
    queueTrackReference(next); // :173 / :209
}

void TableTracker::onLKDone(StageBatch &done, StageBatch &next) {
    if (mode_ == M_TRACK) finishTrackMappoint(done);
    ref_tracked_ = midTrackReference(done, next);
}

void TableTracker::onRansacDone(StageBatch &done, StageBatch &next) {
    if (ref_tracked_) finishTrackReference(done);
    if (mode_ == M_INIT) {
        if (parallax_ref_ < cfg_.track_min_parallax) { // :175-178
            finish(TRACK_INITIALIZING);
            return;
        }
        queueTriangulation(next); // :182
        return;
    }
    kf_state_ = checkKeyFrameSate(); // :212
    if ((kf_state_ == KEYFRAME_NORMAL) || (kf_state_ == KEYFRAME_REMOVE_OLDEST)) queueTriangulation(next); // :215-217
}

void TableTracker::onTriangulateDone(StageBatch &done, StageBatch &next) {
    if (tri_queued_) finishTriangulation(done);
    if (mode_ == M_INIT) {
        if (doResetTracking()) { // :184-190
            lost_reset_ = 1;
            makeNewFrameQueue(KEYFRAME_NORMAL, next);
            return;
        }
        lost_reset_ = 0;
        setKeyFrame(ref_, KEYFRAME_NORMAL);       // :193
        makeNewFrameQueue(KEYFRAME_NORMAL, next); // :196
        last_keyframe_  = cur_;
        isinitializing_ = false;
        return;
    }
    lost_reset_ = 0;
    if (!frames_[(size_t) cur_].rows()) { // :224 (see tracking_hip.cc for why it may be evaluated before the :220 detection)
        doResetTracking();
        lost_reset_ = 2;
        makeNewFrameQueue(KEYFRAME_NORMAL, next); // :225
        return;
    }
    if ((kf_state_ == KEYFRAME_NORMAL) || (kf_state_ == KEYFRAME_REMOVE_OLDEST)) {
        makeNewFrameQueue(kf_state_, next); // :230-232
    } else {
        queueDetection(cur_, true, next); // :220
        if (kf_state_ != KEYFRAME_NONE) makeNewFrameQueue(kf_state_, next); // REMOVE_SECOND_NEW: flags only
    }
}

void TableTracker::onDetectBDone(StageBatch &done) {
    if (det_job_ >= 0) integrateDetection(done);
    releaseUnusedSlots();
    if (mode_ == M_INIT) {
        if (lost_reset_ == 1) {
            finish(TRACK_FIRST_FRAME);
            return;
        }
        finish(TRACK_TRACKING);
        return;
    }
    if (lost_reset_ == 2) {
        finish(TRACK_LOST);
        return;
    }
    if (kf_state_ != KEYFRAME_NONE) writeLoggingMessage(); // :236-238
    finish(TRACK_TRACKING);
}

void TableTracker::makeNewFrameQueue(int state, StageBatch &next) { // :251-261
    setKeyFrame(cur_, state);
    isnewkeyframe_ = true;
    if ((state == KEYFRAME_NORMAL) || (state == KEYFRAME_REMOVE_OLDEST)) {
        ref_ = cur_;
        queueDetection(ref_, true, next);
    }
}

// ---- featuresDetection (:576-688) ------------------------------------------------------------------------------------------------
bool TableTracker::queueDetection(int frame, bool ismask, StageBatch &next) {
    det_job_         = -1;
    const Frame_ &f  = frames_[(size_t) frame];
    int num_features = static_cast<int>(f.rows() + pts2d_ref_.size()); // :579
    if (num_features > (cfg_.track_max_features - 5)) return false;    // :580
    int features_cnts[256];
    if (block_cnts_ > 256) throw std::runtime_error("TableTracker: more than 256 detection blocks");
    for (int k = 0; k < block_cnts_; k++) features_cnts[k] = 0;
    auto count = [&](float x, float y) {
        int col = int(x / (float) block_w_); // :598
        int row = int(y / (float) block_h_);
        // hazard H5 (unclamped column of an undistorted key point), reproduced as in tracking_hip.cc
        const long idx = (long) row * block_cols_ + col;
        if (idx >= 0 && idx < (long) block_cnts_) features_cnts[idx]++;
    };
    for (const Row &r : f.row) count(r.kp.x, r.kp.y);
    for (auto &pts2d : pts2d_new_) count(pts2d.x, pts2d.y);
    det_job_    = (int) next.det_slots.size();
    det_ismask_ = ismask;
    det_frame_  = frame;
    next.det_slots.push_back(f.slot);
    if (ismask) { // :610-620 (a union of discs: the order of the points is immaterial)
        const Frame_ &fc = frames_[(size_t) cur_];
        const size_t at  = next.det_mask_pts.size();
        next.det_mask_pts.resize(at + 2 * (fc.rows() + pts2d_new_.size()));
        float *p = next.det_mask_pts.data() + at;
        for (const Row &r : fc.row) *p++ = r.kp.x, *p++ = r.kp.y;
        memcpy(p, pts2d_new_.data(), pts2d_new_.size() * sizeof(Point2f));
    }
    next.det_mask_off.push_back((int32_t) (next.det_mask_pts.size() / 2));
    for (int k = 0; k < block_cnts_; k++) next.det_quota.push_back(track_max_block_features_ - features_cnts[k]); // :629
    return true;
}

void TableTracker::integrateDetection(StageBatch &done) { // :659-685
    if (!det_ismask_) {
        pts2d_new_.clear();
        pts2d_ref_.clear();
        pts2d_ref_undis_.clear();
        pts2d_new_undis_.clear();
        pts2d_ref_frame_.clear();
        velocity_ref_.clear();
        cand_lk_idx_.clear();
    }
    const int max_per_job = maxFeaturesPerJob();
    const int n           = done.det_count[(size_t) det_job_];
    const float *p        = done.det_out.data() + (size_t) det_job_ * max_per_job * 2;
    scratch_a_.resize((size_t) n);
    for (int i = 0; i < n; i++) scratch_a_[(size_t) i] = Point2f(p[2 * i], p[2 * i + 1]);
    scratch_b_ = scratch_a_;
    camera_->undistortPoints(scratch_b_); // the one undistortion a detected corner ever needs
    for (int i = 0; i < n; i++) {
        pts2d_ref_.push_back(scratch_a_[(size_t) i]);
        pts2d_new_.push_back(scratch_a_[(size_t) i]);
        pts2d_ref_undis_.push_back(scratch_b_[(size_t) i]);
        pts2d_new_undis_.push_back(scratch_b_[(size_t) i]);
        pts2d_ref_frame_.push_back(det_frame_);
        velocity_ref_.emplace_back(0, 0);
    }
    cand_lk_idx_.resize(pts2d_new_.size(), -1); // (a list that lost its alignment is padded / cut: hints are hints)
    det_job_   = -1;
    det_frame_ = -1;
}

// ---- trackMappoint (:351-455) ----------------------------------------------------------------------------------------------------
void TableTracker::queueTrackMappoint(StageBatch &next) {
    mappoint_matched_.clear();
    tm_pts2d_map_.clear();
    tm_pc_.clear();
    tm_pred_.clear();
    tm_hint_.clear();
    const Frame_ &fp    = frames_[(size_t) pre_];
    const Pose pose_cur = frames_[(size_t) cur_].pose;
    // The rows are visited in container order (random within the row array) and every row names a map point somewhere in the pool: with
    // hundreds of streams per GPU both are cold by the time a stream's turn comes round again.  Pass 1 lists the rows in container order
    // and requests them; pass 2 requests the map-point record a few rows ahead of the one it works on.
    const size_t nq = listContainerOrder(fp);
    for (size_t k = 0; k < nq; k++) {
        if (k + kAhead < nq) __builtin_prefetch(&mps_.hot[fp.row[(size_t) order_idx_[k + kAhead]].mp]);
        const Row &r     = fp.row[(size_t) order_idx_[k]];
        const uint32_t i = r.mp;
        if (!mps_.valid(i, r.mpgen) || mps_.hot[i].outlier) continue; // mappoint && !mappoint->isOutlier() (:360)
        tm_pc_.push_back(r.pcx); // pixel2cam of the previous undistorted key point (:434 needs it for the velocity)
        tm_pc_.push_back(r.pcy);
        tm_pts2d_map_.push_back(r.kpd);
        Point2f pp = camera_->world2pixel(mps_.hot[i].pos, pose_cur); // INS-aided prediction :367
        camera_->distortPoint(pp);                                    // :378
        tm_pred_.push_back(pp);
        mappoint_matched_.push_back({i, r.mpgen});
        tm_hint_.push_back(r.lk_idx);
    }
    lk_map_begin_ = (int) next.lk_prev_slot.size();
    lk_map_n_     = (int) tm_pred_.size();
    if (tm_pred_.empty()) return; // :372-375
    const size_t at = next.lk_prev_slot.size();
    next.lk_prev_slot.resize(at + (size_t) lk_map_n_, fp.slot);
    next.lk_next_slot.resize(at + (size_t) lk_map_n_, frames_[(size_t) cur_].slot);
    next.lk_prev.resize(2 * (at + (size_t) lk_map_n_));
    next.lk_guess.resize(2 * (at + (size_t) lk_map_n_));
    memcpy(next.lk_prev.data() + 2 * at, tm_pts2d_map_.data(), (size_t) lk_map_n_ * sizeof(Point2f));
    memcpy(next.lk_guess.data() + 2 * at, tm_pred_.data(), (size_t) lk_map_n_ * sizeof(Point2f));
}

bool TableTracker::finishTrackMappoint(StageBatch &done) {
    if (lk_map_n_ == 0) return false;
    const int n = lk_map_n_;
    // the device already fused status && status_reverse && !isOnBorder && ||bwd - orig|| < 0.5 (:396-403)
    const uint8_t *status = done.lk_status.data() + lk_map_begin_;
    const Point2f *out    = reinterpret_cast<const Point2f *>(done.lk_out.data()) + lk_map_begin_;
    const Point2f *undis  = reinterpret_cast<const Point2f *>(done.lk_undist.data()) + lk_map_begin_; // undistortPoints (:423), per point on the device
    int kept = 0;
    for (int k = 0; k < n; k++) kept += status[k] ? 1 : 0;
    if (kept == 0) { // :410-419
        parallax_map_        = 0;
        parallax_map_counts_ = 0;
        return false;
    }
    {
        hostprof::Scope hp_feat(hostprof::LK_MAP_FEATURES);
        Frame_ &fc = frames_[(size_t) cur_];
        fc.clearRows(); // :426 (a fresh frame: nothing to clear)
        fc.row.reserve((size_t) kept + 64);
        fc.order.reserveRows((size_t) kept + 64);
        tracked_mappoint_.clear();
        const double dt = fc.stamp - frames_[(size_t) pre_].stamp;
        for (int k = 0; k < n; k++) { // reduceVector (:404-408) and the feature loop (:430-444) in one pass
            if (k + 8 < n) __builtin_prefetch(&mps_.hot[mappoint_matched_[(size_t) k + 8].i], 1);
            if (!status[k]) continue;
            const MpRef m     = mappoint_matched_[(size_t) k];
            const Vector3d pc = camera_->pixel2cam(undis[k]);
            // (pixel2cam(cur) - pixel2cam(pre)) / dt (:434); the ids of the previous frame's rows are distinct keys
            const Vector2d velocity((pc[0] - tm_pc_[2 * (size_t) k]) / dt, (pc[1] - tm_pc_[2 * (size_t) k + 1]) / dt);
            const int row = addRow(cur_, mps_.hot[m.i].id, m.i, undis[k], out[k], velocity, FEATURE_MATCHED, pc[0], pc[1], true,
                                   done.lk_base + lk_map_begin_ + k);
            mps_.hot[m.i].observed++; // addObservation (mappoint.cc:58-62)
            mps_.hot[m.i].last = LastObs{cur_, fc.gen, row};
            tracked_mappoint_.push_back(m);
        }
    }
    {
        hostprof::Scope hp_par(hostprof::LK_MAP_PARALLAX);
        parallax_map_counts_ = parallaxFromReferenceMapPoints(parallax_map_); // :450
    }
    return true;
}

// ---- trackReferenceFrame (:457-574) --------------------------------------------------------------------------------------------------
void TableTracker::queueTrackReference(StageBatch &next) {
    lk_ref_begin_ = (int) next.lk_prev_slot.size();
    lk_ref_n_     = 0;
    if (pts2d_ref_.empty()) return; // :459-462
    const Frame_ &fc = frames_[(size_t) cur_], &fp = frames_[(size_t) pre_];
    Matrix3d r_cur_pre = fc.pose.R.transpose() * fp.pose.R; // :465
    checkCarriedUndistortion("trackReferenceFrame");
    pts2d_cur_.clear();
    for (const auto &pp_pre : pts2d_new_undis_) { // :469 (carried), :472-479
        Vector3d pc_pre = camera_->pixel2cam(pp_pre);
        Vector3d pc_cur = r_cur_pre * pc_pre;
        pts2d_cur_.emplace_back(camera_->distortCameraPoint(pc_cur));
    }
    lk_ref_n_       = (int) pts2d_new_.size();
    const size_t at = next.lk_prev_slot.size();
    next.lk_prev_slot.resize(at + (size_t) lk_ref_n_, fp.slot);
    next.lk_next_slot.resize(at + (size_t) lk_ref_n_, fc.slot);
    next.lk_prev.resize(2 * (at + (size_t) lk_ref_n_));
    next.lk_guess.resize(2 * (at + (size_t) lk_ref_n_));
    memcpy(next.lk_prev.data() + 2 * at, pts2d_new_.data(), (size_t) lk_ref_n_ * sizeof(Point2f));
    memcpy(next.lk_guess.data() + 2 * at, pts2d_cur_.data(), (size_t) lk_ref_n_ * sizeof(Point2f));
}

bool TableTracker::midTrackReference(StageBatch &done, StageBatch &next) {
    hostprof::Scope hp_ref(hostprof::LK_REF);
    rs_set_ = -1;
    if (lk_ref_n_ == 0) return false;
    const int n = lk_ref_n_;
    status_.assign(done.lk_status.begin() + lk_ref_begin_, done.lk_status.begin() + lk_ref_begin_ + n);
    scratch_a_.resize((size_t) n);
    memcpy((void *) pts2d_cur_.data(), done.lk_out.data() + 2 * (size_t) lk_ref_begin_, (size_t) n * sizeof(Point2f));
    memcpy((void *) scratch_a_.data(), done.lk_undist.data() + 2 * (size_t) lk_ref_begin_, (size_t) n * sizeof(Point2f));
    cand_lk_idx_.resize((size_t) n);
    for (int k = 0; k < n; k++) cand_lk_idx_[(size_t) k] = done.lk_base + lk_ref_begin_ + k; // pts2d_cur_[k] is this call's forward result k
    reduceVector(cand_lk_idx_, status_);
    reduceVector(pts2d_ref_, status_); // :507-511
    reduceVector(pts2d_cur_, status_);
    reduceVector(pts2d_new_, status_);
    reduceVector(pts2d_ref_frame_, status_);
    reduceVector(velocity_ref_, status_);
    reduceVector(scratch_a_, status_);
    reduceVector(pts2d_ref_undis_, status_);
    reduceVector(pts2d_new_undis_, status_);
    if (pts2d_ref_.empty()) return false; // :513-517
    tr_new_undis_ = pts2d_new_undis_;     // :520-524 (carried)
    tr_cur_undis_.swap(scratch_a_);

    velocity_cur_.clear(); // :527-539
    const Frame_ &fc = frames_[(size_t) cur_];
    const ulong ref_fid = frames_[(size_t) ref_].fid;
    double dt = fc.stamp - frames_[(size_t) pre_].stamp;
    for (size_t k = 0; k < tr_cur_undis_.size(); k++) {
        Vector3d vel = (camera_->pixel2cam(tr_cur_undis_[k]) - camera_->pixel2cam(tr_new_undis_[k])) / dt;
        Vector2d velocity(vel.x(), vel.y());
        velocity_cur_.push_back(velocity);
        if (frames_[(size_t) pts2d_ref_frame_[k]].fid > ref_fid) velocity_ref_[k] = velocity;
    }
    parallax_ref_counts_ = parallaxFromReferenceKeyPoints(pts2d_ref_undis_, tr_cur_undis_, parallax_ref_); // :542-544

    if (pts2d_cur_.size() >= 15) { // :547-548
        rs_set_        = (int) next.rs_off.size() - 1;
        next.rs_thresh = cfg_.reprojection_error_std;
        const size_t m = tr_new_undis_.size(), at = next.rs_p1.size();
        next.rs_p1.resize(at + 2 * m);
        next.rs_p2.resize(at + 2 * m);
        memcpy(next.rs_p1.data() + at, tr_new_undis_.data(), m * sizeof(Point2f));
        memcpy(next.rs_p2.data() + at, tr_cur_undis_.data(), m * sizeof(Point2f));
        next.rs_off.push_back((int32_t) (next.rs_p1.size() / 2));
    }
    return true;
}

bool TableTracker::finishTrackReference(StageBatch &done) {
    if (rs_set_ >= 0) { // :550-554
        status_.assign(done.rs_mask.begin() + done.rs_off[(size_t) rs_set_], done.rs_mask.begin() + done.rs_off[(size_t) rs_set_ + 1]);
        reduceVector(pts2d_ref_, status_);
        reduceVector(pts2d_cur_, status_);
        reduceVector(pts2d_ref_frame_, status_);
        reduceVector(velocity_cur_, status_);
        reduceVector(velocity_ref_, status_);
        reduceVector(pts2d_ref_undis_, status_);
        reduceVector(tr_cur_undis_, status_);
        reduceVector(cand_lk_idx_, status_);
        rs_set_ = -1;
    }
    if (pts2d_cur_.empty()) return false; // :557-561
    pts2d_new_       = pts2d_cur_;        // :569
    pts2d_new_undis_ = tr_cur_undis_;
    return !pts2d_new_.empty();
}

// ---- triangulation (:690-798) ----------------------------------------------------------------------------------------------------------
bool TableTracker::queueTriangulation(StageBatch &next) {
    tri_queued_ = false;
    if (pts2d_cur_.empty()) return false; // :692-694
    tri_queued_ = true;
    const Pose pose1 = frames_[(size_t) cur_].pose;
    if (tr_cur_undis_.size() != pts2d_cur_.size()) { // no reference tracking ran this frame: derive them on the host
        tr_cur_undis_ = pts2d_cur_;
        camera_->undistortPoints(tr_cur_undis_);
    }
    if (pts2d_ref_undis_.size() != pts2d_ref_.size()) {
        pts2d_ref_undis_ = pts2d_ref_;
        camera_->undistortPoints(pts2d_ref_undis_);
    }
    checkCarriedUndistortion("triangulation");
    tri_ref_undis_ = pts2d_ref_undis_; // :712-713 (carried)
    tri_cur_undis_ = tr_cur_undis_;
    tri_status_.assign(pts2d_cur_.size(), 0);
    tri_point_index_.clear();
    tri_begin_ = (int) next.tri_T0.size();

    const int T_cur = (int) (next.tri_Tcw.size() / 12);
    {
        double t12[12];
        toRowMajor3x4(Tracking::pose2Tcw(pose1), t12);
        next.tri_Tcw.insert(next.tri_Tcw.end(), t12, t12 + 12);
    }
    int T_frame[8], T_index[8], n_T = 0; // distinct reference frames of the candidates (a handful)
    const ulong ref_fid = frames_[(size_t) ref_].fid;
    const Matrix3d R1t  = pose1.R.transpose();
    for (size_t k = 0; k < pts2d_cur_.size(); k++) {
        const int frame_ref = pts2d_ref_frame_[k];
        const Frame_ &fr    = frames_[(size_t) frame_ref];
        if (fr.fid > ref_fid) { // :723-730 feature added after the reference keyframe: re-anchor
            pts2d_ref_frame_[k] = cur_;
            pts2d_ref_[k]       = pts2d_cur_[k];
            pts2d_ref_undis_[k] = tri_cur_undis_[k];
            tri_status_[k]      = 1;
            continue;
        }
        if (mapIsWindowNormal() && !mapIsKeyFrameInMap(frame_ref)) { // :733-737
            tri_status_[k] = 0;
            continue;
        }
        // keyPointParallax(pp0, pp1, pose0, pose1) (:861-871): (pose1.R^T * pose0.R) * pc0
        double parallax = keyPointParallax(tri_ref_undis_[k], tri_cur_undis_[k], R1t * fr.pose.R); // :741
        if (parallax < TRACK_MIN_PARALLAX) {
            tri_status_[k] = 1;
            continue;
        }
        int T0 = -1;
        for (int q = 0; q < n_T; q++)
            if (T_frame[q] == frame_ref) T0 = T_index[q];
        if (T0 < 0) {
            T0 = (int) (next.tri_Tcw.size() / 12);
            double t12[12];
            toRowMajor3x4(Tracking::pose2Tcw(fr.pose), t12);
            next.tri_Tcw.insert(next.tri_Tcw.end(), t12, t12 + 12);
            if (n_T < 8) T_frame[n_T] = frame_ref, T_index[n_T] = T0, n_T++;
        }
        Vector3d pc0 = camera_->pixel2cam(tri_ref_undis_[k]); // :750-751
        Vector3d pc1 = camera_->pixel2cam(tri_cur_undis_[k]);
        next.tri_T0.push_back(T0);
        next.tri_T1.push_back(T_cur);
        for (int c = 0; c < 3; c++) next.tri_pc0.push_back(pc0[c]);
        for (int c = 0; c < 3; c++) next.tri_pc1.push_back(pc1[c]);
        tri_point_index_.push_back((int) k);
    }
    return true;
}

void TableTracker::finishTriangulation(StageBatch &done) {
    tri_queued_      = false;
    const Pose pose1 = frames_[(size_t) cur_].pose;
    for (size_t q = 0; q < tri_point_index_.size(); q++) {
        const size_t k  = (size_t) tri_point_index_[q];
        const double *p = &done.tri_pw[3 * (size_t) (tri_begin_ + (int) q)];
        Vector3d pw(p[0], p[1], p[2]);
        const int frame_ref = pts2d_ref_frame_[k];
        const Pose pose0    = frames_[(size_t) frame_ref].pose;
        auto pp0 = tri_ref_undis_[k], pp1 = tri_cur_undis_[k];
        tri_status_[k] = 0; // :757 / :761: rejected or consumed, the candidate leaves the list either way
        if (!isGoodToTrack(pp0, pose0, pw, 1.0, 3.0) || !isGoodToTrack(pp1, pose1, pw, 1.0, 3.0)) continue; // :756-760
        auto pc      = Camera::world2cam(pw, pose0);
        double depth = pc.z();
        // MapPoint::createMapPoint (mappoint.cc:25-49)
        const uint32_t i = mps_.alloc();
        mps_.hot[i].id        = ids_->mappoint_id++;
        mps_.cold[i].born_fid  = frames_[(size_t) cur_].fid;
        mps_.hot[i].pos       = pw;
        mps_.cold[i].ref_frame = frame_ref;
        mps_.cold[i].ref_gen   = frames_[(size_t) frame_ref].gen;
        mps_.cold[i].ref_kp    = tri_ref_undis_[k];
        mps_.cold[i].depth     = ((depth < MapPoint::NEAREST_DEPTH) || (depth > MapPoint::FARTHEST_DEPTH)) ? MapPoint::DEFAULT_DEPTH : depth;
        mps_.hot[i].type      = (int8_t) MAPPOINT_TRIANGULATED;
        const Vector3d pcc = camera_->pixel2cam(tri_cur_undis_[k]), pcr = camera_->pixel2cam(tri_ref_undis_[k]);
        addRow(cur_, mps_.hot[i].id, i, tri_cur_undis_[k], pts2d_cur_[k], velocity_cur_[k], FEATURE_TRIANGULATED, pcc[0], pcc[1], true,
               k < cand_lk_idx_.size() ? cand_lk_idx_[k] : -1); // :769-774
        mps_.hot[i].observed++;
        mps_.hot[i].used++;
        const int row = addRow(frame_ref, mps_.hot[i].id, i, tri_ref_undis_[k], pts2d_ref_[k], velocity_ref_[k], FEATURE_TRIANGULATED, pcr[0], pcr[1],
                               true); // :776-781 (a freshly drawn id is in no frame yet)
        mps_.hot[i].observed++;
        mps_.hot[i].used++;
        mps_.hot[i].last = LastObs{frame_ref, frames_[(size_t) frame_ref].gen, row};
        Frame_ &fc   = frames_[(size_t) cur_];
        fc.unupdated.push_back(i); // :784
        fc.unupdated_gen.push_back(mps_.hot[i].gen);
    }
    reduceVector(pts2d_ref_, tri_status_); // :788-793
    reduceVector(pts2d_ref_frame_, tri_status_);
    reduceVector(pts2d_cur_, tri_status_);
    reduceVector(velocity_ref_, tri_status_);
    reduceVector(pts2d_ref_undis_, tri_status_);
    reduceVector(tr_cur_undis_, tri_status_);
    if (cand_lk_idx_.size() == tri_status_.size()) reduceVector(cand_lk_idx_, tri_status_);
    pts2d_new_       = pts2d_cur_;
    pts2d_new_undis_ = tr_cur_undis_;
}


// ---- B2 view and canonical dumps ---------------------------------------------------------------------------------------------------
namespace {
struct Dump {
    std::string s;
    void f(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        s += buf;
    }
    static unsigned fb(float v) {
        unsigned u;
        memcpy(&u, &v, 4);
        return u;
    }
    static unsigned long long db(double v) {
        unsigned long long u;
        memcpy(&u, &v, 8);
        return u;
    }
    void pose(const Pose &p) {
        double a[12];
        poseToArray12(p, a);
        for (double v : a) f(" %016llx", db(v));
    }
};
} // namespace

std::string TableTracker::dump() const {
    syncTable();
    Dump d;
    auto fid = [&](int h) { return h >= 0 ? (long) frames_[(size_t) h].fid : -1L; };
    d.f("T init=%d cur=%ld pre=%ld ref=%ld lastkf=%ld pmap=%016llx/%d pref=%016llx/%d ncand=%zu\n", (int) isinitializing_, fid(cur_), fid(pre_),
        fid(ref_), fid(last_keyframe_), Dump::db(parallax_map_), parallax_map_counts_, Dump::db(parallax_ref_), parallax_ref_counts_, pts2d_ref_.size());
    for (size_t k = 0; k < pts2d_ref_.size(); k++)
        d.f("C %zu ref=%08x,%08x new=%08x,%08x frame=%ld vref=%016llx,%016llx\n", k, Dump::fb(pts2d_ref_[k].x), Dump::fb(pts2d_ref_[k].y),
            Dump::fb(pts2d_new_[k].x), Dump::fb(pts2d_new_[k].y), fid(pts2d_ref_frame_[k]), Dump::db(velocity_ref_[k][0]), Dump::db(velocity_ref_[k][1]));
    d.f("M window=%zu full=%d nkf=%zu nlm=%zu latest=%ld\n", window_size_, (int) is_window_full_, map_kf_.size(), n_landmarks_, fid(latest_keyframe_));
    vector<MapKf> kfs = map_kf_;
    std::sort(kfs.begin(), kfs.end(), [](const MapKf &a, const MapKf &b) { return a.key < b.key; });
    for (const auto &k : kfs) d.f("K key=%lu fid=%ld\n", k.key, fid(k.frame));
    vector<int> alive;
    for (size_t h = 0; h < frames_.size(); h++)
        if (frames_[h].alive && (int) h != pending_) alive.push_back((int) h);
    std::sort(alive.begin(), alive.end(), [&](int a, int b) { return frames_[(size_t) a].fid < frames_[(size_t) b].fid; });
    for (int h : alive) {
        const Frame_ &f = frames_[(size_t) h];
        d.f("F fid=%lu kfid=%lu iskf=%d state=%d stamp=%016llx nrows=%zu pose", f.fid, f.kf_id, (int) f.is_kf, f.kf_state, Dump::db(f.stamp), f.rows());
        d.pose(f.pose);
        d.f("\n");
        for (int r = f.order.head(); r >= 0; r = f.order.next(r)) {
            const Row &q  = f.row[(size_t) r];
            const bool mp = mps_.valid(q.mp, q.mpgen) && !mps_.hot[q.mp].outlier;
            d.f("R id=%lu kp=%08x,%08x kpd=%08x,%08x vel=%016llx,%016llx type=%d mp=%d\n", q.id, Dump::fb(q.kp.x), Dump::fb(q.kp.y), Dump::fb(q.kpd.x),
                Dump::fb(q.kpd.y), Dump::db(q.vel[0]), Dump::db(q.vel[1]), (int) q.type, (int) mp);
        }
    }
    vector<uint32_t> lms;
    for (uint32_t i = 0; i < mps_.size(); i++)
        if (mps_.hot[i].live && mps_.hot[i].in_map) lms.push_back(i);
    std::sort(lms.begin(), lms.end(), [&](uint32_t a, uint32_t b) { return mps_.hot[a].id < mps_.hot[b].id; });
    for (uint32_t i : lms) {
        const long ref = frameValid(mps_.cold[i].ref_frame, mps_.cold[i].ref_gen) ? (long) frames_[(size_t) mps_.cold[i].ref_frame].fid : -1L;
        d.f("L id=%lu pos=%016llx,%016llx,%016llx depth=%016llx ref=%ld refkp=%08x,%08x type=%d used=%d observed=%d optimized=%d outlier=%d obs=", mps_.hot[i].id,
            Dump::db(mps_.hot[i].pos[0]), Dump::db(mps_.hot[i].pos[1]), Dump::db(mps_.hot[i].pos[2]), Dump::db(mps_.cold[i].depth), ref, Dump::fb(mps_.cold[i].ref_kp.x),
            Dump::fb(mps_.cold[i].ref_kp.y), (int) mps_.hot[i].type, mps_.hot[i].used, mps_.hot[i].observed, mps_.cold[i].optimized, (int) mps_.hot[i].outlier);
        for (ulong o : observationFrames(i, alive)) d.f("%lu,", o);
        d.f("\n");
    }
    d.f("O buckets=%zu order=", map_lm_.bucket_count());
    for (const auto &kv : map_lm_) d.f("%lu,", kv.first);
    d.f("\n");
    return d.s;
}

// Frames (ids) that hold a live observation of map point i, in the order of MapPoint::observations_ (mappoint.cc:58-62): the frame that
// was current when the point was triangulated, then its reference frame (tracking.cc:769-781), then every later frame that tracked it.
vector<ulong> TableTracker::observationFrames(uint32_t i, const vector<int> &alive_by_fid) const {
    vector<ulong> out;
    const ulong born = mps_.cold[i].born_fid;
    auto has = [&](int h) {
        const Frame_ &f = frames_[(size_t) h];
        if (!f.order.contains(mps_.hot[i].id)) return false;
        for (const Row &r : f.row)
            if (r.id == mps_.hot[i].id) return r.mp == i && r.mpgen == mps_.hot[i].gen;
        return false;
    };
    for (int h : alive_by_fid)
        if (frames_[(size_t) h].fid == born && has(h)) out.push_back(born);
    if (frameValid(mps_.cold[i].ref_frame, mps_.cold[i].ref_gen) && frames_[(size_t) mps_.cold[i].ref_frame].fid != born && has(mps_.cold[i].ref_frame))
        out.push_back(frames_[(size_t) mps_.cold[i].ref_frame].fid);
    for (int h : alive_by_fid)
        if (frames_[(size_t) h].fid > born && has(h)) out.push_back(frames_[(size_t) h].fid);
    return out;
}

static void dumpMapObjects(Dump &d, Map &map, const vector<Frame::Ptr> &roots) {
    auto fid = [&](const Frame::Ptr &f) { return f ? (long) f->id() : -1L; };
    d.f("M window=%zu full=%d nkf=%zu nlm=%zu latest=%ld\n", map.windowSize(), (int) map.isWindowFull(), map.keyframes().size(), map.landmarks().size(),
        fid(map.latestKeyFrame()));
    vector<std::pair<ulong, Frame::Ptr>> kfs(map.keyframes().begin(), map.keyframes().end());
    std::sort(kfs.begin(), kfs.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (const auto &k : kfs) d.f("K key=%lu fid=%ld\n", k.first, fid(k.second));
    vector<Frame::Ptr> alive;
    auto add = [&](const Frame::Ptr &f) {
        if (f && std::find(alive.begin(), alive.end(), f) == alive.end()) alive.push_back(f);
    };
    for (const auto &f : roots) add(f);
    for (const auto &k : kfs) add(k.second);
    add(map.latestKeyFrame());
    std::sort(alive.begin(), alive.end(), [](const Frame::Ptr &a, const Frame::Ptr &b) { return a->id() < b->id(); });
    for (const auto &f : alive) {
        Frame::FeatureList feats;
        f->featureSnapshot(feats);
        d.f("F fid=%lu kfid=%lu iskf=%d state=%d stamp=%016llx nrows=%zu pose", f->id(), f->keyFrameId(), (int) f->isKeyFrame(), f->keyFrameState(),
            Dump::db(f->stamp()), feats.size());
        d.pose(f->pose());
        d.f("\n");
        for (const auto &kv : feats) {
            const auto &ft = kv.second;
            auto mp        = ft->getMapPoint();
            d.f("R id=%lu kp=%08x,%08x kpd=%08x,%08x vel=%016llx,%016llx type=%d mp=%d\n", kv.first, Dump::fb(ft->keyPoint().x), Dump::fb(ft->keyPoint().y),
                Dump::fb(ft->distortedKeyPoint().x), Dump::fb(ft->distortedKeyPoint().y), Dump::db(ft->velocityInPixel()[0]),
                Dump::db(ft->velocityInPixel()[1]), (int) ft->featureType(), (int) (mp && !mp->isOutlier()));
        }
    }
    vector<std::pair<ulong, MapPoint::Ptr>> lms(map.landmarks().begin(), map.landmarks().end());
    std::sort(lms.begin(), lms.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (const auto &kv : lms) {
        const auto &m = kv.second;
        auto rf       = m->referenceFrame();
        Vector3d pos  = m->pos();
        Point2f rk    = m->referenceKeypoint();
        d.f("L id=%lu pos=%016llx,%016llx,%016llx depth=%016llx ref=%ld refkp=%08x,%08x type=%d used=%d observed=%d optimized=%d outlier=%d obs=", m->id(),
            Dump::db(pos[0]), Dump::db(pos[1]), Dump::db(pos[2]), Dump::db(m->depth()), fid(rf), Dump::fb(rk.x), Dump::fb(rk.y), (int) m->mapPointType(),
            m->usedTimes(), m->observedTimes(), m->optimizedTimes(), (int) m->isOutlier());
        for (auto &w : m->observations()) {
            auto ft = w.lock();
            if (!ft) continue;
            auto fr = ft->getFrame();
            if (fr) d.f("%lu,", fr->id());
        }
        d.f("\n");
    }
    d.f("O buckets=%zu order=", map.landmarks().bucket_count());
    for (const auto &kv : map.landmarks()) d.f("%lu,", kv.first);
    d.f("\n");
}

std::string TableTracker::dumpObjects(Tracking &t, Map &map) {
    Dump d;
    auto fid = [&](const Frame::Ptr &f) { return f ? (long) f->id() : -1L; };
    const auto &ref = t.referencePoints(), &nw = t.trackedRefPoints();
    d.f("T init=%d cur=%ld pre=%ld ref=%ld lastkf=%ld pmap=%016llx/%d pref=%016llx/%d ncand=%zu\n", (int) t.initializing(), fid(t.currentFrame()),
        fid(t.previousFrame()), fid(t.referenceFrame()), fid(t.lastKeyFrame()), Dump::db(t.parallaxMap()), t.parallaxMapCounts(), Dump::db(t.parallaxRef()),
        t.parallaxRefCounts(), ref.size());
    for (size_t k = 0; k < ref.size(); k++)
        d.f("C %zu ref=%08x,%08x new=%08x,%08x frame=%ld vref=%016llx,%016llx\n", k, Dump::fb(ref[k].x), Dump::fb(ref[k].y), Dump::fb(nw[k].x), Dump::fb(nw[k].y),
            fid(t.referencePointFrames()[k]), Dump::db(t.referenceVelocities()[k][0]), Dump::db(t.referenceVelocities()[k][1]));
    vector<Frame::Ptr> roots{t.currentFrame(), t.previousFrame(), t.referenceFrame(), t.lastKeyFrame()};
    for (const auto &f : t.referencePointFrames()) roots.push_back(f);
    dumpMapObjects(d, map, roots);
    return d.s;
}

std::string TableTracker::dumpMaterialized() const {
    vector<Frame::Ptr> extra;
    Map::Ptr map = materialize(&extra);
    Dump d;
    dumpMapObjects(d, *map, extra);
    return d.s;
}

std::string TableTracker::dumpMap() const {
    const std::string all = dump();
    return all.substr(all.find("M window="));
}

std::shared_ptr<TableTracker::ObjectView> TableTracker::view() const {
    syncTable();
    auto V   = std::make_shared<ObjectView>();
    auto map = std::make_shared<Map>(window_size_);
    V->map   = map;
    vector<int> alive;
    for (size_t h = 0; h < frames_.size(); h++)
        if (frames_[h].alive && (int) h != pending_) alive.push_back((int) h);
    std::sort(alive.begin(), alive.end(), [&](int a, int b) { return frames_[(size_t) a].fid < frames_[(size_t) b].fid; });
    vector<Frame::Ptr> &obj             = V->frame;
    vector<vector<Feature::Ptr>> &feat = V->feat;
    obj.assign(frames_.size(), nullptr);
    feat.assign(frames_.size(), {});
    V->frame_gen.assign(frames_.size(), 0);
    for (int h : alive) {
        const Frame_ &f = frames_[(size_t) h];
        auto fr         = std::make_shared<Frame>(f.fid, f.stamp, f.image, ids_);
        fr->setPose(f.pose);
        fr->restoreKeyFrame(f.is_kf, f.kf_id, f.kf_state);
        fr->setDeviceSlot(f.slot);
        obj[(size_t) h]          = fr;
        V->frame_gen[(size_t) h] = f.gen;
        feat[(size_t) h].resize(f.rows());
        for (size_t r = 0; r < f.rows(); r++) { // insertion order: the container of the object reproduces the iteration order
            feat[(size_t) h][r] = Feature::createFeature(fr, f.row[r].vel, f.row[r].kp, f.row[r].kpd, (FeatureType) f.row[r].type);
            feat[(size_t) h][r]->setOutlier(f.row[r].outlier != 0);
        }
    }
    // (frame handle, row) of every row that holds a live map point, per map point, frames in id order
    vector<vector<std::pair<int, int>>> seen(mps_.size());
    for (int h : alive) {
        const Frame_ &f = frames_[(size_t) h];
        for (size_t r = 0; r < f.rows(); r++)
            if (mps_.valid(f.row[r].mp, f.row[r].mpgen) && f.row[r].id == mps_.hot[f.row[r].mp].id) seen[f.row[r].mp].emplace_back(h, (int) r);
    }
    vector<MapPoint::Ptr> &mpo = V->mappoint;
    mpo.assign(mps_.size(), nullptr);
    V->mp_gen.assign(mps_.size(), 0);
    vector<uint32_t> lms;
    for (uint32_t i = 0; i < mps_.size(); i++)
        if (mps_.hot[i].live) lms.push_back(i);
    std::sort(lms.begin(), lms.end(), [&](uint32_t a, uint32_t b) { return mps_.hot[a].id < mps_.hot[b].id; });
    for (uint32_t i : lms) {
        Frame::Ptr rf = frameValid(mps_.cold[i].ref_frame, mps_.cold[i].ref_gen) ? obj[(size_t) mps_.cold[i].ref_frame] : nullptr;
        auto m        = std::allocate_shared<MapPoint>(PoolAllocator<MapPoint>(), mps_.hot[i].id, rf, mps_.hot[i].pos, mps_.cold[i].ref_kp, mps_.cold[i].depth,
                                                (MapPointType) mps_.hot[i].type);
        mpo[i]       = m;
        V->mp_gen[i] = mps_.hot[i].gen;
        // observations in list order (see observationFrames): the frame that was current at the triangulation, the reference frame, then
        // every later frame — from the (frame, row) pairs collected in one pass over the rows above
        const ulong born = mps_.cold[i].born_fid;
        const auto &at   = seen[i];
        for (const auto &o : at)
            if (frames_[(size_t) o.first].fid == born) m->addObservation(feat[(size_t) o.first][(size_t) o.second]);
        if (rf && frames_[(size_t) mps_.cold[i].ref_frame].fid != born)
            for (const auto &o : at)
                if (o.first == mps_.cold[i].ref_frame) m->addObservation(feat[(size_t) o.first][(size_t) o.second]);
        for (const auto &o : at)
            if (frames_[(size_t) o.first].fid > born) m->addObservation(feat[(size_t) o.first][(size_t) o.second]);
        m->restoreCounters(mps_.hot[i].used, mps_.hot[i].observed, mps_.cold[i].optimized, mps_.hot[i].outlier != 0);
    }
    for (int h : alive) {
        const Frame_ &f = frames_[(size_t) h];
        for (size_t r = 0; r < f.rows(); r++) {
            if (mps_.valid(f.row[r].mp, f.row[r].mpgen)) feat[(size_t) h][r]->addMapPoint(mpo[f.row[r].mp]);
            obj[(size_t) h]->addFeature(f.row[r].id, feat[(size_t) h][r]);
        }
        for (size_t k = 0; k < f.unupdated.size(); k++)
            if (mps_.valid(f.unupdated[k], f.unupdated_gen[k])) obj[(size_t) h]->addNewUnupdatedMappoint(mpo[f.unupdated[k]]);
    }
    vector<std::pair<ulong, Frame::Ptr>> kfs;
    for (const auto &k : map_kf_) kfs.emplace_back(k.key, obj[(size_t) k.frame]);
    vector<MapPoint::Ptr> in_map; // in Map::landmarks_' iteration order
    for (const auto &kv : map_lm_) in_map.push_back(mpo[kv.second]);
    map->restore(kfs, in_map, map_lm_.bucket_count(), latest_keyframe_ >= 0 ? obj[(size_t) latest_keyframe_] : nullptr, is_window_full_);
    return V;
}

Map::Ptr TableTracker::materialize(vector<Frame::Ptr> *extra) const {
    auto V = view();
    if (extra)
        for (const auto &f : V->frame)
            if (f) extra->push_back(f);
    if (extra) std::sort(extra->begin(), extra->end(), [](const Frame::Ptr &a, const Frame::Ptr &b) { return a->id() < b->id(); });
    return V->map;
}

// Takes over what was done to the objects of a view since it was built (see ObjectView).  Frames and map points that have gone in the
// table meanwhile (generation mismatch) are skipped; nothing else may have changed the table in between (no frame was tracked).
void TableTracker::absorb(const ObjectView &V) {
    syncTable();
    struct WriteBack { // core mode: the block follows the table image
        TableTracker *t;
        ~WriteBack() {
            if (t->core_) t->exportCore();
        }
    } write_back{this};
    for (size_t h = 0; h < V.frame.size() && h < frames_.size(); h++) {
        if (!V.frame[h] || !frames_[h].alive || frames_[h].gen != V.frame_gen[h]) continue;
        Frame_ &f = frames_[h];
        f.pose    = V.frame[h]->pose(); // updateParametersFromOptimizer (ic_gvins.cc:1347-1357)
        for (size_t r = 0; r < f.rows() && r < V.feat[h].size(); r++) f.row[r].outlier = V.feat[h][r]->isOutlier() ? 1 : 0; // :1078-1091
    }
    for (uint32_t i = 0; i < V.mappoint.size() && i < mps_.size(); i++) {
        const MapPoint::Ptr &m = V.mappoint[i];
        if (!m || !mps_.hot[i].live || mps_.hot[i].gen != V.mp_gen[i]) continue;
        mps_.hot[i].pos        = m->pos();
        mps_.cold[i].depth     = m->depth();
        mps_.hot[i].used       = m->usedTimes();
        mps_.cold[i].optimized = m->optimizedTimes();
        mps_.hot[i].outlier    = m->isOutlier() ? 1 : 0;
        if (mps_.hot[i].in_map && V.map->landmarks().find(m->id()) == V.map->landmarks().end()) {
            // Map::removeMappoint (map.cc:129-137): outlier, observations dropped, erased from the landmarks — nothing reaches it any more
            mps_.hot[i].in_map  = 0;
            mps_.hot[i].outlier = 1;
            map_lm_.erase(mps_.hot[i].id);
            n_landmarks_--;
            mps_.release(i);
        }
    }
}

} // namespace icg
