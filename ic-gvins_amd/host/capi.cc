// C entry points of libicgvins_host.so for harnesses that cannot speak C++ (tests, bench.py): drive a TrackingBatch.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <atomic>
#include <mutex>

#include "tracking_batch.h"
#include "hostprof.h"

using namespace icg;

struct icgh_batch {
    std::unique_ptr<StreamGroups> tb;
    int w, h;
};

static void set_err(char *err, int errlen, const char *msg) {
    if (err && errlen > 0) snprintf(err, (size_t) errlen, "%s", msg);
}

extern "C" {

icgh_batch *icgh_batch_create(int device, int n_streams, const double *cam10, int w, int h, int max_features,
                              double min_parallax, double max_interval, int check_hist, double reproj_std, int window,
                              int host_threads, int n_groups, char *err, int errlen) {
    try {
        TrackingConfig cfg;
        cfg.track_max_features     = max_features;
        cfg.track_min_parallax     = min_parallax;
        cfg.track_max_interval     = max_interval;
        cfg.track_check_histogram  = check_hist != 0;
        cfg.reprojection_error_std = reproj_std;
        vector<double> intr{cam10[0], cam10[1], cam10[2], cam10[3], cam10[4]};
        vector<double> dist{cam10[5], cam10[6], cam10[7], cam10[8], cam10[9]};
        auto *b = new icgh_batch();
        b->w    = w;
        b->h    = h;
        b->tb.reset(new StreamGroups(device, n_streams, n_groups, intr, dist, {w, h}, cfg, window, host_threads));
        return b;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return nullptr;
    }
}

void icgh_batch_destroy(icgh_batch *b) { delete b; }

int icgh_batch_groups(icgh_batch *b) { return b ? b->tb->groups() : 0; }
void *icgh_batch_ctx(icgh_batch *b, int group) {
    return (b && group >= 0 && group < b->tb->groups()) ? (void *) b->tb->group(group).device()->ctx() : nullptr;
}

// K lock-step frames for every stream in one call (K = 1: the classic per-frame step).
// images[k*n + i]: pointer to the frame of stream i at step k (host or device memory), NULL to idle the stream that step.
// stamps[k*n + i]; poses12[(k*n + i)*12 ..] = R (camera->world, row-major) | t, the INS prior the reference sets with
// frame->setPose().  states[k*n + i] receives the TrackState.  Groups do not wait for each other between the K steps.
int icgh_batch_run(icgh_batch *b, int K, const void *const *images, int stride, int channels, int on_device,
                   const double *stamps, const double *poses12, int32_t *states, char *err, int errlen) {
    try {
        const int n = b->tb->size();
        auto t0     = std::chrono::steady_clock::now();
        vector<vector<FrameInput>> frames((size_t) K, vector<FrameInput>((size_t) n));
        for (int k = 0; k < K; k++)
            for (int i = 0; i < n; i++) {
                const size_t j = (size_t) k * n + i;
                if (!images[j]) continue;
                FrameInput &f = frames[(size_t) k][(size_t) i];
                f.valid       = true;
                f.stamp       = stamps[j];
                f.image       = Mat::wrap((uint8_t *) images[j], b->h, b->w, channels, (size_t) stride, on_device != 0);
                f.pose        = poseFromArray12(poses12 + 12 * j);
            }
        vector<vector<TrackState>> st;
        auto t1 = std::chrono::steady_clock::now();
        b->tb->stepMany(frames, st);
        auto t2 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; k++)
            for (int i = 0; i < n; i++) states[(size_t) k * n + i] = (int32_t) st[(size_t) k][(size_t) i];
        frames.clear();
        auto t3 = std::chrono::steady_clock::now();
        if (getenv("ICG_DEBUG_TIMING"))
            fprintf(stderr, "[icgh_batch_run] K=%d create %.2f ms, stepMany %.2f ms, teardown %.2f ms\n", K,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                    std::chrono::duration<double, std::milli>(t3 - t2).count());
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

int icgh_batch_step(icgh_batch *b, const void *const *images, int stride, int channels, int on_device, const double *stamps,
                    const double *poses12, int32_t *states, char *err, int errlen) {
    return icgh_batch_run(b, 1, images, stride, channels, on_device, stamps, poses12, states, err, errlen);
}

// out: frames, keyframes, tracked_sum, digest, mappoints created, keyframes in window, landmarks in map, last state
int icgh_batch_stats(icgh_batch *b, int stream, uint64_t *out8) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    auto &s = b->tb->stream(stream);
    out8[0] = s.frames;
    out8[1] = s.keyframes;
    out8[2] = s.tracked_sum;
    out8[3] = s.digest;
    out8[4] = s.ids->mappoint_id;
    out8[5] = s.windowKeyFrames();
    out8[6] = s.landmarks();
    out8[7] = (uint64_t) s.last_state;
    return 0;
}

// the same for every stream in one call (n x 8): the bench reads the statistics of hundreds of streams between its warm-up and its timed
// region, where every millisecond the GPU idles costs clock state
int icgh_batch_stats_all(icgh_batch *b, uint64_t *out8n) {
    if (!b || !out8n) return -1;
    for (int i = 0; i < b->tb->size(); i++)
        if (icgh_batch_stats(b, i, out8n + 8 * (size_t) i) != 0) return -1;
    return 0;
}

int icgh_batch_timing(icgh_batch *b, double *out5, int reset) {
    if (!b) return -1;
    for (int i = 0; i < 5; i++) out5[i] = 0;
    for (int g = 0; g < b->tb->groups(); g++)
        for (int i = 0; i < 5; i++) {
            out5[i] += b->tb->group(g).timing[i] / b->tb->groups(); // mean over groups (they run concurrently)
            if (reset) b->tb->group(g).timing[i] = 0;
        }
    return 0;
}

// work counters summed over the groups (see TrackingBatch::counters); reset != 0 clears them
int icgh_batch_counters(icgh_batch *b, uint64_t *out8, int reset) {
    if (!b) return -1;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (int g = 0; g < b->tb->groups(); g++)
        for (int i = 0; i < 8; i++) {
            out8[i] += b->tb->group(g).counters[i];
            if (reset) b->tb->group(g).counters[i] = 0;
        }
    return 0;
}

// per-step log of group g since the last reset: out[3k..3k+2] = {steady-clock seconds at the end of the step, host-logic
// seconds, device-execute seconds}; returns the number of steps written (<= max_steps); reset != 0 clears the log
int icgh_batch_step_log(icgh_batch *b, int g, double *out, int max_steps, int reset) {
    if (!b || g < 0 || g >= b->tb->groups()) return -1;
    auto &log = b->tb->group(g).step_log;
    int n     = (int) std::min<size_t>(log.size(), (size_t) std::max(0, max_steps));
    for (int k = 0; k < n && out; k++) {
        out[3 * k]     = log[(size_t) k].t_end;
        out[3 * k + 1] = log[(size_t) k].host_logic;
        out[3 * k + 2] = log[(size_t) k].device_execute;
    }
    if (reset) log.clear();
    return n;
}

// steady-clock seconds on the clock icgh_batch_step_log reports
double icgh_now_s(void) { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int icgh_batch_timing_group(icgh_batch *b, int g, double *out5) {
    if (!b || g < 0 || g >= b->tb->groups()) return -1;
    for (int i = 0; i < 5; i++) out5[i] = b->tb->group(g).timing[i];
    return 0;
}

// section timers of the host layer (ICG_HOST_PROF=1): out[2k] = seconds, out[2k+1] = calls; names are ';'-separated
int icgh_hostprof(double *out, int max_sections, char *names, int names_len, int reset) {
    int n = std::min(max_sections, (int) icg::hostprof::N_SECTIONS);
    std::string nm;
    for (int k = 0; k < n; k++) {
        out[2 * k]     = 1e-9 * (double) icg::hostprof::ns()[k].load();
        out[2 * k + 1] = (double) icg::hostprof::calls()[k].load();
        nm += icg::hostprof::name(k);
        nm += ';';
        if (reset) {
            icg::hostprof::ns()[k]    = 0;
            icg::hostprof::calls()[k] = 0;
        }
    }
    if (names && names_len > 0) snprintf(names, (size_t) names_len, "%s", nm.c_str());
    return n;
}

// the tracker's un-triangulated candidate points in list order: cur[2k..] (pts2d_new_), ref[2k..] (pts2d_ref_)
int icgh_batch_candidates(icgh_batch *b, int stream, int max, float *cur, float *ref) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    const auto &pn = b->tb->stream(stream).trackedRefPoints();
    const auto &pr = b->tb->stream(stream).referencePoints();
    int n = (int) std::min(pn.size(), pr.size());
    if (n > max) n = max;
    for (int k = 0; k < n; k++) {
        cur[2 * k] = pn[(size_t) k].x, cur[2 * k + 1] = pn[(size_t) k].y;
        ref[2 * k] = pr[(size_t) k].x, ref[2 * k + 1] = pr[(size_t) k].y;
    }
    return (pn.size() == pr.size()) ? n : -2;
}

// features of the stream's current frame, sorted by map-point id: ids[k], px[2k..2k+1] (distorted keypoint)
int icgh_batch_features(icgh_batch *b, int stream, int max, uint64_t *ids, float *px) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    vector<std::pair<ulong, Point2f>> v;
    b->tb->stream(stream).currentFeatures(v);
    std::sort(v.begin(), v.end(), [](const auto &a, const auto &c) { return a.first < c.first; });
    int n = 0;
    for (const auto &kv : v) {
        if (n >= max) break;
        ids[n]        = kv.first;
        px[2 * n]     = kv.second.x;
        px[2 * n + 1] = kv.second.y;
        n++;
    }
    return n;
}

// Kernel-only ceiling: icgh_batch_record(b, 1); <one icgh_batch_run step>; icgh_batch_record(b, 0); then icgh_batch_replay(b, reps) issues the
// recorded device calls of every group again — one group after the other, nothing else on the GPU, no tracker logic — so that the HIP-event
// times of icg_prof_* are the EXCLUSIVE device times of the step's kernels.  Returns the number of recorded stage batches (all groups).
int icgh_batch_record(icgh_batch *b, int on) {
    if (!b) return -1;
    for (int g = 0; g < b->tb->groups(); g++) b->tb->group(g).record(on != 0);
    return 0;
}
int icgh_batch_replay(icgh_batch *b, int reps, char *err, int errlen) {
    if (!b) return -1;
    try {
        int n = 0;
        for (int g = 0; g < b->tb->groups(); g++) {
            b->tb->group(g).replay(reps);
            n += (int) b->tb->group(g).device()->recorded();
        }
        return n;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

// the same recorded calls issued by ALL groups at once from their own threads (the concurrency of a real run, no tracker logic)
int icgh_batch_replay_concurrent(icgh_batch *b, int reps, char *err, int errlen) {
    if (!b) return -1;
    try {
        b->tb->replayAll(reps);
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -2;
    }
}

// 0 = track table (default), 1 = object graph (ICG_TRACK_ENGINE=object)
int icgh_batch_engine(icgh_batch *b) { return b ? (int) b->tb->group(0).engine() : -1; }

// canonical text dump of a stream's tracker + map state (engine-vs-engine tests): kind 0 = the engine's state, 1 = the map part of the
// table engine's state, 2 = the same text computed from the table engine's materialized object graph (B2 view).  Returns the length
// of the text (the buffer receives at most len-1 characters), -1 on bad arguments.
long icgh_batch_dump(icgh_batch *b, int stream, int kind, char *out, long len) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    try {
        const std::string s = b->tb->stream(stream).dump(kind);
        if (out && len > 0) {
            const size_t n = std::min((size_t) len - 1, s.size());
            memcpy(out, s.data(), n);
            out[n] = 0;
        }
        return (long) s.size();
    } catch (const std::exception &) {
        return -2;
    }
}

// HashOrder (track_table.h) against a real std::unordered_map<ulong, int>: n random distinct keys inserted one by one, the iteration
// orders compared after every `check_every` insertions.  Returns 0 when they always agree, k > 0 = first disagreement after k insertions.
int icgh_hashorder_selftest(uint64_t seed, int n, int check_every, int dense_ids) {
    return HashOrder::selfTest(seed, n, check_every, dense_ids != 0);
}

// The tracker core's container order (tc::order_extend, track_core.h: node list + buckets in scratch memory, batched insertion of the rows
// [n_old, n_rows) — what the stage kernels run) against a real std::unordered_map<ulong, int>: `n_first` rows entered at once into an empty
// frame, then `rounds` times `n_more` further rows appended to the existing order (every extension starts from the stored head / buckets and
// crosses rehashes).  Returns 0 when the iteration orders agree after every step, else the step (1-based) of the first disagreement.
int icgh_core_order_selftest(uint64_t seed, int n_first, int n_more, int rounds) {
    if (n_first < 0 || n_more < 0 || rounds < 0 || n_first + (long) n_more * rounds > tc::MAX_ROWS) return -1;
    std::unique_ptr<tc::Frame> f(new tc::Frame);
    std::unique_ptr<tc::Scratch> X(new tc::Scratch);
    memset(f.get(), 0, sizeof(tc::Frame));
    tc::order_clear(*f);
    std::unordered_map<ulong, int> ref;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 7;
    ulong id   = seed % 977;
    const uint32_t *ba = TableTracker::bucketsAfterTable();
    auto append = [&](int count) {
        for (int k = 0; k < count; k++) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            id += 1 + (x % 5) * ((seed & 1) ? 1 : 131); // dense ids as the id factories hand them out, or scattered ones
            f->row[f->n_rows].id = id;
            ref.emplace(id, f->n_rows);
            f->n_rows++;
        }
    };
    auto same = [&]() {
        int r = f->head;
        for (const auto &kv : ref) {
            if (r < 0 || r != kv.second) return false;
            r = f->next[r];
        }
        return r < 0 && (size_t) f->n_buckets == ref.bucket_count();
    };
    int old = 0;
    append(n_first);
    tc::order_extend(*f, old, ba, *X);
    if (!same()) return 1;
    // the sort form (order_extend_parallel) on a copy of the same rows: same list, same buckets, at every step
    std::unique_ptr<tc::Frame> fp(new tc::Frame);
    memset(fp.get(), 0, sizeof(tc::Frame));
    tc::order_clear(*fp);
    auto mirror = [&]() {
        for (int k = fp->n_rows; k < f->n_rows; k++) fp->row[k].id = f->row[k].id;
        fp->n_rows = f->n_rows;
    };
    auto same_as_parallel = [&]() {
        if (fp->head != f->head || fp->n_buckets != f->n_buckets || fp->magic != f->magic) return false;
        for (int k = 0; k < f->n_rows; k++)
            if (fp->next[k] != f->next[k]) return false;
        for (int b = 0; b < f->n_buckets; b++)
            if (fp->bucket[b] != f->bucket[b]) return false;
        return true;
    };
    mirror();
    tc::order_extend_parallel(*fp, 0, ba, *X);
    if (!same_as_parallel()) return 500;
    for (int r = 0; r < rounds; r++) {
        old = f->n_rows;
        append(n_more);
        tc::order_extend(*f, old, ba, *X);
        if (!same()) return 2 + r;
        mirror();
        tc::order_extend_parallel(*fp, old, ba, *X);
        if (!same_as_parallel()) return 501 + r;
    }
    // and the one-by-one form (order_insert_unique: the path of rows added outside a batch) gives the same list
    std::unique_ptr<tc::Frame> g(new tc::Frame);
    memset(g.get(), 0, sizeof(tc::Frame));
    tc::order_clear(*g);
    vector<int32_t> scratch((size_t) tc::MAX_BUCKETS);
    for (int k = 0; k < f->n_rows; k++) {
        g->row[k].id = f->row[k].id;
        tc::order_insert_unique(*g, ba, scratch.data());
    }
    int a = f->head, b = g->head;
    while (a >= 0 && b >= 0 && a == b) a = f->next[a], b = g->next[b];
    return (a < 0 && b < 0) ? 0 : 1000;
}

} // extern "C"

// ---- back-end test/driver entry points -----------------------------------------------------------------------------------
#include "factors.h"
#include "object_pool.h"
#include "misc_hip.h"
#include "solver_hip.h"
#include "solver_batch_hip.h"
#include "marg_batch.h"
#include "culling_hip.h"
#include "window_visual.h"

namespace {
// simple generic host factor used to exercise the non-reprojection path of MarginalizationInfo:
// residual = w * [p - p0 ; 2 vec(q0^-1 q)] on one pose block (6 residuals, 7 parameters)
class PosePriorFactor : public ceres::SizedCostFunction<6, 7> {
public:
    PosePriorFactor(const double *pose0, double weight) : w_(weight) { memcpy(x0_, pose0, sizeof x0_); }
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        const double *x = parameters[0];
        const double n2 = x0_[3] * x0_[3] + x0_[4] * x0_[4] + x0_[5] * x0_[5] + x0_[6] * x0_[6];
        const double ax = -x0_[3] / n2, ay = -x0_[4] / n2, az = -x0_[5] / n2, aw = x0_[6] / n2;
        const double bx = x[3], by = x[4], bz = x[5], bw = x[6];
        const double dq[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                              aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz};
        for (int k = 0; k < 3; k++) {
            residuals[k]     = w_ * (x[k] - x0_[k]);
            residuals[3 + k] = w_ * 2.0 * dq[k];
        }
        if (jacobians && jacobians[0]) {
            memset(jacobians[0], 0, sizeof(double) * 42);
            for (int k = 0; k < 3; k++) {
                jacobians[0][k * 7 + k]           = w_;
                jacobians[0][(3 + k) * 7 + 3 + k] = w_ * dq[3]; // d(2 vec(dq * exp(phi/2)))/dphi ~ w I at dq ~ identity
            }
        }
        return true;
    }

private:
    double x0_[7], w_;
};
// a host factor on a one-dimensional block (an inverse depth): residual = w (x - x0).  On a landmark it breaks the structure the
// landmark-eliminated marginalization relies on (tests: that window then takes the dense M2 + M3)
class ScalarPriorFactor : public ceres::SizedCostFunction<1, 1> {
public:
    ScalarPriorFactor(double x0, double weight) : x0_(x0), w_(weight) {}
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        residuals[0] = w_ * (parameters[0][0] - x0_);
        if (jacobians && jacobians[0]) jacobians[0][0] = w_;
        return true;
    }

private:
    double x0_, w_;
};
} // namespace

extern "C" {
// symmetricEigen (factors.h) on a caller's matrix: A is n x n row-major, evals (n) ascending, evecs (n x n row-major, eigenvector k in
// column k).  A test hook: tests/test_host_backend_cpu.py pins the bit patterns of the restructured routine to those of the plain form.
int icgh_symmetric_eigen(int n, const double *A, double *evals, double *evecs) {
    if (n < 0 || (n > 0 && (!A || !evals || !evecs))) return -1;
    std::vector<double> a(A, A + (size_t) n * n), ev, V;
    symmetricEigen(n, a, ev, V);
    if (n) memcpy(evals, ev.data(), sizeof(double) * (size_t) n), memcpy(evecs, V.data(), sizeof(double) * (size_t) n * n);
    return 0;
}


// R1 through the ceres::CostFunction surface: factors + EvaluationCallback, one Evaluate() per factor.
// rc: 0 ok, 1 = an unprepared factor did NOT fail (contract violation), <0 = error.
int icgh_backend_reproj(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm,
                        int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth, double td,
                        double *out_r, double *out_J, char *err, int errlen) {
    try {
        vector<double> P(poses, poses + 7 * (size_t) n_poses), E(ext, ext + 7), D(invdepth, invdepth + n_lm);
        double TD = td;
        vector<std::unique_ptr<ReprojectionFactor>> factors;
        ReprojectionBatch batch(0);
        for (int k = 0; k < n; k++) {
            auto o = [&](int c) { return obs_soa[(size_t) c * n + k]; };
            factors.emplace_back(new ReprojectionFactor(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                        Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14)));
        }
        // before registration / preparation Evaluate must fail
        {
            double r[2];
            const double *params[5] = {&P[0], &P[0], E.data(), &D[0], &TD};
            if (n > 0 && factors[0]->Evaluate(params, r, nullptr)) return 1;
        }
        for (int k = 0; k < n; k++)
            batch.add(factors[(size_t) k].get(), &P[7 * (size_t) idx_i[k]], &P[7 * (size_t) idx_j[k]], E.data(), &D[(size_t) idx_lm[k]], &TD);
        batch.finalize();
        {
            double r[2];
            const double *params[5] = {&P[0], &P[0], E.data(), &D[0], &TD};
            if (n > 0 && factors[0]->Evaluate(params, r, nullptr)) return 1; // registered but not prepared
        }
        batch.PrepareForEvaluation(true, true);
        for (int k = 0; k < n; k++) {
            const double *params[5] = {&P[7 * (size_t) idx_i[k]], &P[7 * (size_t) idx_j[k]], E.data(), &D[(size_t) idx_lm[k]], &TD};
            double *J              = out_J + 46 * (size_t) k;
            double *jac[5]         = {J, J + 14, J + 28, J + 42, J + 44};
            if (!factors[(size_t) k]->Evaluate(params, out_r + 2 * (size_t) k, jac)) {
                set_err(err, errlen, batch.error().c_str());
                return -2;
            }
        }
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// M1-M4 through the reference's API: marginalize pose 0 and the landmarks it references.  Parameter ids: pose k -> k,
// landmark l -> 100000 + l, extrinsic -> 900000, td -> 900001.  estimate_ext/td = 0 keeps those blocks out (constant).
// Outputs: sizes[0..1] = marginalized, remained local sizes; rem_ids/rem_index/rem_size per retained block (caller
// allocates n_poses + n_lm + 2 entries); Hp, bp, J0, e0 sized by the caller to (6*n_poses + n_lm + 7)^2 etc.
// Then evaluates MarginalizationFactor at x = current parameters perturbed by `perturb` (applied as p += d, per block by
// id order of rem_ids) and writes residuals to marg_res.
int icgh_backend_marginalize(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm,
                             int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth, double td,
                             double huber_delta, double prior_weight, int estimate_ext, int estimate_td, int32_t *sizes,
                             int64_t *rem_ids, int32_t *rem_index, int32_t *rem_size, int32_t *n_rem, double *Hp, double *bp,
                             double *J0, double *e0, const double *x_eval /* concatenated by rem order, may be NULL */,
                             double *marg_res, char *err, int errlen) {
    try {
        vector<double> P(poses, poses + 7 * (size_t) n_poses), E(ext, ext + 7), D(invdepth, invdepth + n_lm);
        double TD = td;
        std::unordered_map<long, long> ids;
        std::unordered_map<long, double *> address;
        auto reg = [&](double *p, long id) {
            ids[reinterpret_cast<long>(p)] = id;
            address[id]                     = p;
        };
        for (int k = 0; k < n_poses; k++) reg(&P[7 * (size_t) k], k);
        for (int l = 0; l < n_lm; l++) reg(&D[(size_t) l], 100000 + l);
        reg(E.data(), 900000);
        reg(&TD, 900001);
        (void) estimate_ext;
        (void) estimate_td;

        auto info = std::make_shared<MarginalizationInfo>();
        info->updateParamtersIds(ids);
        ReprojectionBatch batch(0);
        info->setReprojectionBatch(&batch);
        auto loss = huber_delta > 0 ? std::make_shared<HuberLossHip>(huber_delta) : nullptr;
        for (int k = 0; k < n; k++) {
            auto o = [&](int c) { return obs_soa[(size_t) c * n + k]; };
            auto f = std::make_shared<ReprojectionFactor>(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                          Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14));
            double *pi = &P[7 * (size_t) idx_i[k]], *pj = &P[7 * (size_t) idx_j[k]], *lm = &D[(size_t) idx_lm[k]];
            batch.add(f.get(), pi, pj, E.data(), lm, &TD);
            // marginalize {pose_ref, invdepth} as ic_gvins.cc:1600-1606 does
            info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(f, loss, vector<double *>{pi, pj, E.data(), lm, &TD}, vector<int>{0, 3}));
        }
        batch.finalize();
        // host-evaluated generic factor on the marginalized pose (stands in for prior/IMU factors of the real window)
        vector<double> pose0_prior(P.begin(), P.begin() + 7);
        pose0_prior[0] += 0.01; // non-zero residual
        info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(std::make_shared<PosePriorFactor>(pose0_prior.data(), prior_weight),
                                                                       nullptr, vector<double *>{&P[0]}, vector<int>{0}));
        if (!info->marginalization()) {
            set_err(err, errlen, ("marginalization failed: " + batch.error()).c_str());
            return -2;
        }
        auto blocks = info->getParamterBlocks(address);
        sizes[0]    = info->marginalizedSize();
        sizes[1]    = info->remainedSize();
        *n_rem      = (int32_t) blocks.size();
        for (size_t b = 0; b < blocks.size(); b++) {
            rem_ids[b]   = ids[reinterpret_cast<long>(blocks[b])];
            rem_index[b] = info->remainedBlockIndex()[b];
            rem_size[b]  = info->remainedBlockSize()[b];
        }
        const size_t r = (size_t) info->remainedSize();
        memcpy(Hp, info->Hp().data(), sizeof(double) * r * r);
        memcpy(bp, info->bp().data(), sizeof(double) * r);
        memcpy(J0, info->linearizedJacobians().data(), sizeof(double) * r * r);
        memcpy(e0, info->linearizedResiduals().data(), sizeof(double) * r);
        if (x_eval && marg_res) {
            MarginalizationFactor factor(info);
            vector<const double *> params;
            size_t off = 0;
            for (size_t b = 0; b < blocks.size(); b++) {
                params.push_back(x_eval + off);
                off += (size_t) rem_size[b];
            }
            if (!factor.Evaluate(params.data(), marg_res, nullptr)) return -3;
        }
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// The marginalizations of n_windows streams (M1-M4 of each: the window of icgh_backend_marginalize, window w > 0 with its poses and inverse
// depths moved by a deterministic jitter of relative size `jitter`), mode 0: one MarginalizationBatch (marg_batch.h: the windows share
// their device launches), mode 1: one MarginalizationInfo::marginalization() after the other on a ReprojectionBatch (what a stream on its
// own does).  dense_window >= 0: that window gets a host factor on one of its inverse depths, which takes it off the landmark-eliminated
// path in both modes.  The whole set is marginalized `reps` times on the SAME batch object (as a group of streams does keyframe after
// keyframe: the problems are rebuilt each time, outside the clock).  Outputs per window of the last repetition (r = sizes[1], equal for
// all windows): Hp (r x r), bp, J0 (r x r), e0; counts = windows on the structured / dense path, seconds = wall time of the
// marginalizations alone in the fastest repetition (problem construction excluded).
int icgh_backend_marginalize_batch(int mode, int n_windows, int dense_window, double jitter, int reps, int n, const double *obs_soa,
                                   const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm, int n_poses, const double *poses,
                                   const double *ext, int n_lm, const double *invdepth, double td, double huber_delta, double prior_weight,
                                   int host_threads, int32_t *sizes, double *Hp, double *bp, double *J0, double *e0, int32_t *counts,
                                   double *seconds, char *err, int errlen) {
    try {
        struct Win {
            vector<double> P, E, D;
            double TD;
            std::unordered_map<long, long> ids;
            std::shared_ptr<MarginalizationInfo> info;
            vector<std::shared_ptr<ReprojectionFactor>> factors;
            vector<double> pose0_prior;
        };
        auto loss = huber_delta > 0 ? std::make_shared<HuberLossHip>(huber_delta) : nullptr;
        auto build = [&](vector<std::unique_ptr<Win>> &wins) {
            wins.clear();
            uint64_t lcg = 0x9E3779B97F4A7C15ull;
            auto rnd     = [&] { // uniform in [-1, 1)
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                return (double) ((lcg >> 11) & ((1ull << 53) - 1)) / (double) (1ull << 52) - 1.0;
            };
            for (int w = 0; w < n_windows; w++) {
                std::unique_ptr<Win> W(new Win);
                W->P.assign(poses, poses + 7 * (size_t) n_poses), W->E.assign(ext, ext + 7), W->D.assign(invdepth, invdepth + n_lm), W->TD = td;
                if (w > 0) {
                    for (int k = 0; k < n_poses; k++)
                        for (int c = 0; c < 3; c++) W->P[7 * (size_t) k + c] += jitter * rnd();
                    for (int l = 0; l < n_lm; l++) W->D[(size_t) l] *= 1.0 + jitter * rnd();
                }
                for (int k = 0; k < n_poses; k++) W->ids[reinterpret_cast<long>(&W->P[7 * (size_t) k])] = k;
                for (int l = 0; l < n_lm; l++) W->ids[reinterpret_cast<long>(&W->D[(size_t) l])] = 100000 + l;
                W->ids[reinterpret_cast<long>(W->E.data())] = 900000;
                W->ids[reinterpret_cast<long>(&W->TD)]       = 900001;
                W->info = std::make_shared<MarginalizationInfo>();
                W->info->updateParamtersIds(W->ids);
                for (int k = 0; k < n; k++) {
                    auto o = [&](int c) { return obs_soa[(size_t) c * n + k]; };
                    W->factors.push_back(std::make_shared<ReprojectionFactor>(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)),
                                                                              Vector3d(o(6), o(7), o(8)), Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14)));
                    double *pi = &W->P[7 * (size_t) idx_i[k]], *pj = &W->P[7 * (size_t) idx_j[k]], *lm = &W->D[(size_t) idx_lm[k]];
                    W->info->addResidualBlockInfo(
                        std::make_shared<ResidualBlockInfo>(W->factors.back(), loss, vector<double *>{pi, pj, W->E.data(), lm, &W->TD}, vector<int>{0, 3}));
                }
                W->pose0_prior.assign(W->P.begin(), W->P.begin() + 7);
                W->pose0_prior[0] += 0.01;
                W->info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(std::make_shared<PosePriorFactor>(W->pose0_prior.data(), prior_weight),
                                                                                  nullptr, vector<double *>{&W->P[0]}, vector<int>{0}));
                if (w == dense_window && n > 0) {
                    double *lm = &W->D[(size_t) idx_lm[0]];
                    W->info->addResidualBlockInfo(std::make_shared<ResidualBlockInfo>(std::make_shared<ScalarPriorFactor>(*lm * 1.01, 0.5 * prior_weight),
                                                                                      nullptr, vector<double *>{lm}, vector<int>{0}));
                }
                wins.push_back(std::move(W));
            }
        };
        vector<std::unique_ptr<Win>> wins;
        vector<char> ok((size_t) n_windows, 0);
        std::string what;
        double best = -1.0;
        std::unique_ptr<MarginalizationBatch> mb;
        std::unique_ptr<ReprojectionBatch> batch;
        if (mode == 0)
            mb.reset(new MarginalizationBatch(0, huber_delta, host_threads));
        else
            batch.reset(new ReprojectionBatch(0));
        for (int rep = 0; rep < std::max(1, reps); rep++) {
            if (mb) mb->clear(); // (before the infos of the last repetition go)
            if (batch) batch->clear();
            build(wins);
            counts[0] = counts[1] = 0;
            double took = 0;
            if (mode == 0) {
                for (auto &W : wins) {
                    const int w = mb->addWindow(W->info);
                    for (int k = 0; k < n; k++)
                        mb->addReprojectionFactor(w, W->factors[(size_t) k].get(), &W->P[7 * (size_t) idx_i[k]], &W->P[7 * (size_t) idx_j[k]], W->E.data(),
                                                  &W->D[(size_t) idx_lm[k]], &W->TD);
                }
                auto t0         = std::chrono::steady_clock::now();
                const bool good = mb->marginalize(&ok);
                took            = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (!good) {
                    set_err(err, errlen, ("batched marginalization failed: " + mb->error()).c_str());
                    return -2;
                }
                what = mb->windowError();
                counts[0] = mb->structuredWindows(), counts[1] = mb->denseWindows();
            } else {
                for (size_t w = 0; w < wins.size(); w++) {
                    Win &W = *wins[w];
                    batch->clear();
                    for (int k = 0; k < n; k++)
                        batch->add(W.factors[(size_t) k].get(), &W.P[7 * (size_t) idx_i[k]], &W.P[7 * (size_t) idx_j[k]], W.E.data(), &W.D[(size_t) idx_lm[k]], &W.TD);
                    W.info->setReprojectionBatch(batch.get());
                    auto a = std::chrono::steady_clock::now();
                    batch->finalize();
                    ok[w] = W.info->marginalization() ? 1 : 0;
                    took += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
                    if (!ok[w]) what = batch->error();
                    counts[MarginalizationInfo::lastWasStructured() ? 0 : 1] += ok[w] ? 1 : 0;
                }
            }
            for (int w = 0; w < n_windows; w++)
                if (!ok[(size_t) w]) {
                    set_err(err, errlen, ("marginalization of window " + std::to_string(w) + " failed: " + what).c_str());
                    return -2;
                }
            if (best < 0 || took < best) best = took;
        }
        *seconds = best;
        const size_t r = (size_t) wins[0]->info->remainedSize();
        sizes[0] = wins[0]->info->marginalizedSize(), sizes[1] = (int32_t) r;
        for (int w = 0; w < n_windows; w++) {
            const MarginalizationInfo &I = *wins[(size_t) w]->info;
            if ((size_t) I.remainedSize() != r) {
                set_err(err, errlen, "windows of different remained size");
                return -3;
            }
            memcpy(Hp + (size_t) w * r * r, I.Hp().data(), sizeof(double) * r * r);
            memcpy(bp + (size_t) w * r, I.bp().data(), sizeof(double) * r);
            memcpy(J0 + (size_t) w * r * r, I.linearizedJacobians().data(), sizeof(double) * r * r);
            memcpy(e0 + (size_t) w * r, I.linearizedResiduals().data(), sizeof(double) * r);
        }
        if (mb) mb->clear(); // (the infos die with `wins` before the batch does)
        if (batch) batch->clear();
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// phases of the last MarginalizationInfo::marginalization() of this process: evaluate, construct, Schur, linearize [ms]
void icgh_backend_marginalization_phases(double *out4) { memcpy(out4, MarginalizationInfo::lastPhaseMs(), sizeof(double) * 4); }
int icgh_backend_marginalization_structured(void) { return MarginalizationInfo::lastWasStructured() ? 1 : 0; }
void icgh_backend_marginalization_force_dense(int on) { MarginalizationInfo::forceDense(on != 0); }

// P1 (device batch) + P2 (host evaluate) through the Preintegration / PreintegrationFactor classes.
// imu: total x 8; offsets: n+1; state0: n x 16; params9 as icg_preint_batch; pose/mix: the evaluation point per interval
// (pose0[7], mix0[9], pose1[7], mix1[9] concatenated = 32 doubles per interval).  Outputs: cur_state n x 16, residuals
// n x 15, jacobians n x 480 (15x7 | 15x9 | 15x7 | 15x9).
int icgh_backend_preint(int variant, int n, const int32_t *offsets, const double *imu, const double *state0, const double *params9,
                        const double *eval_point, double *cur_state, double *residuals, double *jacobians, char *err, int errlen) {
    try {
        auto P           = std::make_shared<IntegrationParameters>();
        P->gyr_arw       = params9[0];
        P->acc_vrw       = params9[1];
        P->gyr_bias_std  = params9[2];
        P->acc_bias_std  = params9[3];
        P->corr_time     = params9[4];
        P->gravity       = params9[5];
        P->iewn          = Vector3d(params9[6], params9[7], params9[8]);
        icg_ctx_config cfg{};
        cfg.device = 0, cfg.width = 64, cfg.height = 64, cfg.n_slots = 1, cfg.max_batch = 1, cfg.max_points = 64;
        icg_ctx *ctx = nullptr;
        if (icg_ctx_create(&cfg, &ctx) != ICG_OK) {
            set_err(err, errlen, icg_last_error(nullptr));
            return -1;
        }
        auto mk_imu = [&](int row) {
            const double *p = imu + 8 * (size_t) row;
            IMU s;
            s.time = p[0], s.dt = p[1];
            s.dtheta = Vector3d(p[2], p[3], p[4]);
            s.dvel   = Vector3d(p[5], p[6], p[7]);
            return s;
        };
        vector<std::shared_ptr<Preintegration>> pre;
        vector<Preintegration *> raw;
        for (int k = 0; k < n; k++) {
            const double *s = state0 + 16 * (size_t) k;
            IntegrationState st;
            st.p = Vector3d(s[0], s[1], s[2]);
            st.q = Quaterniond{s[3], s[4], s[5], s[6]};
            st.v = Vector3d(s[7], s[8], s[9]), st.bg = Vector3d(s[10], s[11], s[12]), st.ba = Vector3d(s[13], s[14], s[15]);
            auto p = std::make_shared<Preintegration>(P, mk_imu(offsets[k]), st, variant ? Preintegration::EARTH : Preintegration::NORMAL);
            for (int row = offsets[k] + 1; row < offsets[k + 1]; row++) p->addNewImu(mk_imu(row));
            pre.push_back(p);
            raw.push_back(p.get());
        }
        // unintegrated factors must fail
        {
            PreintegrationFactor f(pre[0]);
            double r[15];
            const double *pp[4] = {eval_point, eval_point + 7, eval_point + 16, eval_point + 23};
            if (f.Evaluate(pp, r, nullptr)) {
                icg_ctx_destroy(ctx);
                return 1;
            }
        }
        std::string e;
        if (!Preintegration::integrateBatch(ctx, raw, &e)) {
            set_err(err, errlen, e.c_str());
            icg_ctx_destroy(ctx);
            return -2;
        }
        for (int k = 0; k < n; k++) {
            const IntegrationState &c = pre[(size_t) k]->currentState();
            double *o = cur_state + 16 * (size_t) k;
            o[0] = c.p[0], o[1] = c.p[1], o[2] = c.p[2], o[3] = c.q.x, o[4] = c.q.y, o[5] = c.q.z, o[6] = c.q.w;
            for (int i = 0; i < 3; i++) o[7 + i] = c.v[i], o[10 + i] = c.bg[i], o[13 + i] = c.ba[i];
            PreintegrationFactor f(pre[(size_t) k]);
            const double *ep   = eval_point + 32 * (size_t) k;
            const double *pp[4] = {ep, ep + 7, ep + 16, ep + 23};
            double *J           = jacobians + 480 * (size_t) k;
            double *jj[4]       = {J, J + 105, J + 240, J + 345};
            if (!f.Evaluate(pp, residuals + 15 * (size_t) k, jj)) {
                icg_ctx_destroy(ctx);
                return -3;
            }
        }
        icg_ctx_destroy(ctx);
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// ---- f3: outlier culling / statistics (culling_hip.h) on the maps the tracker built -------------------------------------------
// Raw dump of a stream's landmark graph, NO filtering (the test re-derives the reference's filters and decisions from it):
// landmarks sorted by id: lm_id, lm_pos[3], lm_flags (bit0 outlier), lm_ref_frame (frame id), lm_obs_off[n+1];
// observations in list order: obs_frame (frame id, ~0 = expired), obs_flags (bit0 feature expired, bit1 feature outlier,
// bit2 frame is keyframe, bit3 keyframe in map), obs_pose12, obs_pix[2] (undistorted key point).  Returns the landmark count.
int icgh_batch_landmark_table(icgh_batch *b, int stream, int max_lm, int max_obs, uint64_t *lm_id, double *lm_pos, int32_t *lm_flags,
                              uint64_t *lm_ref_frame, int32_t *lm_obs_off, uint64_t *obs_frame, int32_t *obs_flags, double *obs_pose12,
                              float *obs_pix) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    auto &S       = b->tb->stream(stream);
    Map::Ptr Smap = S.objectMap(); // (track-table engine: a view of the table as reference-shaped objects; read-only use here)
    vector<ulong> ids;
    for (auto &kv : Smap->landmarks()) ids.push_back(kv.first);
    std::sort(ids.begin(), ids.end());
    if ((int) ids.size() > max_lm) return -2;
    int no = 0;
    for (size_t k = 0; k < ids.size(); k++) {
        auto mp      = Smap->landmarks().at(ids[k]);
        lm_id[k]     = ids[k];
        Vector3d pos = mp->pos();
        for (int c = 0; c < 3; c++) lm_pos[3 * k + c] = pos[c];
        lm_flags[k]     = mp->isOutlier() ? 1 : 0;
        lm_ref_frame[k] = mp->referenceFrameId();
        lm_obs_off[k]   = no;
        for (auto &w : mp->observations()) {
            if (no >= max_obs) return -3;
            auto feat = w.lock();
            int fl    = 0;
            obs_frame[no] = ~0ull;
            for (int c = 0; c < 12; c++) obs_pose12[12 * (size_t) no + c] = 0;
            obs_pix[2 * no] = obs_pix[2 * no + 1] = 0;
            if (!feat) {
                fl |= 1;
            } else {
                if (feat->isOutlier()) fl |= 2;
                obs_pix[2 * no] = feat->keyPoint().x, obs_pix[2 * no + 1] = feat->keyPoint().y;
                auto frame = feat->getFrame();
                if (frame) {
                    obs_frame[no] = frame->id();
                    if (frame->isKeyFrame()) fl |= 4;
                    if (frame->isKeyFrame() && Smap->isKeyFrameInMap(frame)) fl |= 8;
                    Pose p = frame->pose();
                    poseToArray12(p, obs_pose12 + 12 * (size_t) no);
                }
            }
            obs_flags[no] = fl;
            no++;
        }
    }
    lm_obs_off[ids.size()] = no;
    return (int) ids.size();
}

// moves landmarks (by id) to new positions: stands in for the optimizer's write-back (ic_gvins.cc:1299-1357) in the tests
int icgh_batch_set_landmark_pos(icgh_batch *b, int stream, int n, const uint64_t *ids, const double *pos3) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    auto &S       = b->tb->stream(stream);
    Map::Ptr Smap = S.objectMap();
    int rc        = 0;
    for (int k = 0; k < n && rc == 0; k++) {
        auto it = Smap->landmarks().find(ids[k]);
        if (it == Smap->landmarks().end())
            rc = -2;
        else
            it->second->setPos(Vector3d(pos3[3 * k], pos3[3 * k + 1], pos3[3 * k + 2]));
    }
    S.commitMap(); // (track-table engine: the new positions go into the table)
    return rc;
}

namespace {
// Views of the streams' maps (TrackingBatch::Stream::objectMap) that an entry point mutates: committed to the track tables only when the
// entry point ran to its end (commit()); on an exception or an early error return every view is dropped unabsorbed, so the tables keep the
// state they had and no later objectMap() sees a half-modified view.
struct MapViewsGuard {
    explicit MapViewsGuard(icgh_batch *batch) : b(batch) {}
    ~MapViewsGuard() {
        if (done) return;
        for (int s = 0; s < b->tb->size(); s++) b->tb->stream(s).discardMap();
    }
    void commit() {
        for (int s = 0; s < b->tb->size(); s++) b->tb->stream(s).commitMap();
        done = true;
    }
    icgh_batch *b;
    bool done{false};
};
} // namespace

// WindowCulling over ALL streams of the batch with one device launch.  in_list: per stream the landmark ids that "took part in
// the optimization" (invdepthlist_), concatenated, list_off[n_streams+1].  mode 0: gvinsOutlierCulling -> out5[s*5..] =
// outlier mappoints, outlier features, num1, num2, num3;  mode 1: reprojectionStatistics -> stats5[s*5..] = min, max, avg, rms, count
int icgh_batch_culling(icgh_batch *b, int mode, const int32_t *list_off, const uint64_t *in_list, double reprojection_error_std, int32_t *out5,
                       double *stats5, char *err, int errlen) {
    try {
        const int n = b->tb->size();
        MapViewsGuard views(b);
        vector<std::unordered_map<ulong, double>> lists((size_t) n);
        vector<WindowCulling::Stream> streams;
        for (int s = 0; s < n; s++) {
            for (int k = list_off[s]; k < list_off[s + 1]; k++) lists[(size_t) s][in_list[k]] = 0.0;
            streams.push_back({b->tb->stream(s).objectMap(), &lists[(size_t) s]});
        }
        icg_ctx *ctx = b->tb->group(0).device()->ctx();
        std::string e;
        if (mode == 0) {
            vector<CullingResult> R;
            if (!WindowCulling::gvinsOutlierCulling(ctx, streams, reprojection_error_std, R, &e)) {
                set_err(err, errlen, e.c_str());
                return -2;
            }
            for (int s = 0; s < n; s++) {
                const CullingResult &r = R[(size_t) s];
                const int32_t v[5]     = {r.outlier_mappoints, r.outlier_features, r.by_reference_frame, r.by_observation_count, r.by_mean_error};
                memcpy(out5 + 5 * s, v, sizeof v);
            }
        } else {
            vector<ReprojectionStatistics> R;
            if (!WindowCulling::reprojectionStatistics(ctx, streams, R, &e)) {
                set_err(err, errlen, e.c_str());
                return -2;
            }
            for (int s = 0; s < n; s++) {
                const ReprojectionStatistics &r = R[(size_t) s];
                const double v[5]               = {r.min_error, r.max_error, r.avg_error, r.rms_error, (double) r.landmarks};
                memcpy(stats5 + 5 * s, v, sizeof v);
            }
        }
        views.commit(); // (track-table engine: flags, counters and removals go into the table)
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// ---- map -> optimizer -> map on the windows the tracker built (window_visual.h + solver_hip.h + culling_hip.h) ----------------
// For every stream: VisualWindow::build (addReprojectionParameters / addReprojectionFactors), pose priors at the current keyframe
// poses (weight prior_weight; they stand in for the IMU / GNSS / marginalization factors), LM solve - chi-square culling - LM solve,
// updateParametersFromOptimizer, gvinsOutlierCulling.  out7[s*7..] = keyframes, factors, initial cost, final cost, removed by chi2,
// culled map points, culled features.  kf_out (optional, max_kf rows of 14 per stream): keyframe stamp, frame id, camera pose12.
int icgh_batch_refine_windows(icgh_batch *b, const double *pose_b_c12, double td, double reprojection_error_std, double prior_weight, int iters1,
                              int iters2, double chi2, double *out7, int max_kf, double *kf_out, char *err, int errlen) {
    try {
        const int n = b->tb->size();
        MapViewsGuard views(b);
        Pose pbc;
        pbc = poseFromArray12(pose_b_c12);
        icg_ctx *ctx = b->tb->group(0).device()->ctx();
        const bool lockstep = getenv("ICG_REFINE_PER_STREAM") == nullptr; // default: all streams' windows in ONE WindowSolverBatch
        vector<std::unique_ptr<VisualWindow>> wins;
        vector<vector<vector<double>>> priors((size_t) n);
        vector<int> slot((size_t) n, -1);
        WindowSolverBatch batch(0, 1.0);
        for (int s = 0; s < n; s++) {
            auto &S = b->tb->stream(s);
            double *o = out7 + 7 * (size_t) s;
            for (int k = 0; k < 7; k++) o[k] = 0;
            wins.emplace_back(new VisualWindow(S.camera, S.objectMap(), pbc, td, reprojection_error_std));
            VisualWindow &win = *wins.back();
            win.build();
            o[0] = win.numKeyFrames(), o[1] = win.numFactors();
            if (win.numKeyFrames() < 2 || win.numFactors() == 0) continue;
            priors[(size_t) s].resize((size_t) win.numKeyFrames());
            for (int k = 0; k < win.numKeyFrames(); k++) priors[(size_t) s][(size_t) k].assign(win.pose(k), win.pose(k) + 7);
            if (lockstep) {
                slot[(size_t) s] = batch.addWindow();
                win.addTo(batch, slot[(size_t) s]);
                for (int k = 0; k < win.numKeyFrames(); k++)
                    batch.addResidualBlock(slot[(size_t) s], std::make_shared<PosePriorFactor>(priors[(size_t) s][(size_t) k].data(), prior_weight), nullptr,
                                           {win.pose(k)});
            } else {
                WindowSolver solver(win.batch(), 1.0);
                win.addTo(solver);
                for (int k = 0; k < win.numKeyFrames(); k++)
                    solver.addResidualBlock(std::make_shared<PosePriorFactor>(priors[(size_t) s][(size_t) k].data(), prior_weight), nullptr, {win.pose(k)});
                WindowSolver::Options opt;
                WindowSolver::Summary s1, s2;
                opt.max_num_iterations = iters1;
                if (!solver.solve(opt, &s1)) {
                    set_err(err, errlen, solver.error().c_str());
                    return -2;
                }
                o[2] = s1.initial_cost, o[3] = s1.final_cost;
                if (chi2 > 0) {
                    o[4] = solver.removeReprojectionFactorsByChi2(chi2);
                    opt.max_num_iterations = iters2;
                    if (!solver.solve(opt, &s2)) {
                        set_err(err, errlen, solver.error().c_str());
                        return -3;
                    }
                    o[3] = s2.final_cost;
                }
            }
        }
        if (lockstep && batch.numWindows() > 0) {
            WindowSolverBatch::Options opt;
            vector<WindowSolverBatch::Summary> s1, s2;
            opt.max_num_iterations = iters1;
            if (!batch.solve(opt, &s1)) {
                set_err(err, errlen, batch.error().c_str());
                return -2;
            }
            vector<int> removed((size_t) batch.numWindows(), 0);
            if (chi2 > 0) {
                removed                = batch.removeReprojectionFactorsByChi2(chi2);
                opt.max_num_iterations = iters2;
                if (!batch.solve(opt, &s2)) {
                    set_err(err, errlen, batch.error().c_str());
                    return -3;
                }
            }
            for (int s = 0; s < n; s++) {
                if (slot[(size_t) s] < 0) continue;
                double *o = out7 + 7 * (size_t) s;
                const size_t w = (size_t) slot[(size_t) s];
                o[2] = s1[w].initial_cost, o[3] = chi2 > 0 ? s2[w].final_cost : s1[w].final_cost, o[4] = removed[w];
            }
        }
        // write-back and culling (all streams' observations in one launch)
        vector<WindowCulling::Stream> cull;
        vector<int> cull_stream;
        for (int s = 0; s < n; s++) {
            if (wins[(size_t) s]->numKeyFrames() < 2 || wins[(size_t) s]->numFactors() == 0) continue;
            wins[(size_t) s]->updateParametersFromOptimizer();
            cull.push_back({b->tb->stream(s).objectMap(), &wins[(size_t) s]->invdepthlist()});
            cull_stream.push_back(s);
        }
        vector<CullingResult> R;
        std::string e;
        if (!cull.empty() && !WindowCulling::gvinsOutlierCulling(ctx, cull, reprojection_error_std, R, &e)) {
            set_err(err, errlen, e.c_str());
            return -4;
        }
        for (size_t k = 0; k < cull_stream.size(); k++) {
            double *o = out7 + 7 * (size_t) cull_stream[k];
            o[5] = R[k].outlier_mappoints, o[6] = R[k].outlier_features;
        }
        if (kf_out)
            for (int s = 0; s < n; s++)
                for (int k = 0; k < std::min(max_kf, wins[(size_t) s]->numKeyFrames()); k++) {
                    double *r = kf_out + 14 * ((size_t) s * max_kf + k);
                    r[0] = wins[(size_t) s]->frame(k)->stamp(), r[1] = (double) wins[(size_t) s]->frame(k)->id();
                    Pose p = wins[(size_t) s]->frame(k)->pose();
                    poseToArray12(p, r + 2);
                }
        views.commit(); // (track-table engine: the write-back and the culling go into the table)
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// BlockPool / PoolAllocator (object_pool.h) under cross-thread traffic, for tests: `threads` workers each allocate `iters` blocks of two
// size classes, stamp them, hand every second one to the next worker through a mailbox (freed on a thread other than the allocating one:
// the spill / refill path of the per-thread lists) and free the rest themselves; every block is checked for its stamp before it is freed.
// Returns the number of corrupted blocks (0 = pass), -1 on an internal error.
int icgh_pool_selftest(int threads, int iters) {
    struct Small {
        uint64_t tag, a;
    };
    struct Large {
        uint64_t tag, pad[11];
    };
    if (threads < 1 || iters < 1) return -1;
    std::vector<std::mutex> box_m((size_t) threads);
    std::vector<std::vector<std::pair<void *, int>>> box((size_t) threads); // (block, size class)
    std::atomic<int> bad{0}, live{0};
    auto check_free = [&](void *p, int cls) {
        if (cls == 0) {
            Small *s = static_cast<Small *>(p);
            if (s->tag != (0xabcdef0000000000ull ^ (uint64_t) (uintptr_t) p) || s->a != ~s->tag) bad++;
            PoolAllocator<Small>().deallocate(s, 1);
        } else {
            Large *l = static_cast<Large *>(p);
            if (l->tag != (0x1234560000000000ull ^ (uint64_t) (uintptr_t) p) || l->pad[10] != ~l->tag) bad++;
            PoolAllocator<Large>().deallocate(l, 1);
        }
        live--;
    };
    auto worker = [&](int t) {
        std::vector<std::pair<void *, int>> mine;
        for (int i = 0; i < iters; i++) {
            const int cls = (i + t) & 1;
            void *p;
            if (cls == 0) {
                Small *s = PoolAllocator<Small>().allocate(1);
                s->tag   = 0xabcdef0000000000ull ^ (uint64_t) (uintptr_t) s;
                s->a     = ~s->tag;
                p        = s;
            } else {
                Large *l   = PoolAllocator<Large>().allocate(1);
                l->tag     = 0x1234560000000000ull ^ (uint64_t) (uintptr_t) l;
                l->pad[10] = ~l->tag;
                p          = l;
            }
            live++;
            if (i & 1) {
                std::lock_guard<std::mutex> lock(box_m[(size_t) ((t + 1) % threads)]);
                box[(size_t) ((t + 1) % threads)].emplace_back(p, cls);
            } else {
                mine.emplace_back(p, cls);
            }
            if ((i & 63) == 63) { // drain the mailbox and half of the own blocks (LIFO reuse follows)
                std::vector<std::pair<void *, int>> got;
                {
                    std::lock_guard<std::mutex> lock(box_m[(size_t) t]);
                    got.swap(box[(size_t) t]);
                }
                for (auto &g : got) check_free(g.first, g.second);
                for (size_t k = mine.size() / 2; k < mine.size(); k++) check_free(mine[k].first, mine[k].second);
                mine.resize(mine.size() / 2);
            }
        }
        for (auto &m : mine) check_free(m.first, m.second);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
    for (int t = 0; t < threads; t++)
        for (auto &g : box[(size_t) t]) check_free(g.first, g.second);
    return live.load() == 0 ? bad.load() : -1;
}

// the dense helpers of the window solvers (dense_kernels.cc), for tests: in-place Cholesky solve of A x = b (row-major, lower triangle
// read; A is overwritten with the factor, b with x; -1 = not positive definite) and T (upper triangle) += J^T J, g += J^T r
int icgh_dense_cholesky_solve(int n, double *A, double *b) {
    vector<double> Av(A, A + (size_t) n * n), bv(b, b + n);
    if (!solver_detail::choleskySolve(n, Av, bv)) return -1;
    memcpy(A, Av.data(), sizeof(double) * (size_t) n * n);
    memcpy(b, bv.data(), sizeof(double) * (size_t) n);
    return 0;
}
void icgh_dense_accumulate_jtj(int nr, int nf, const double *J, const double *r, double *T, double *g) {
    solver_detail::accumulateJtJ(nr, nf, J, r, T, g);
}

// ---- f1: the window optimization flow of GVINS::gvinsOptimization (ic_gvins.cc:1130-1239) on WindowSolver -----------------
// Reprojection factors from flat arrays (as icgh_backend_reproj) + one PosePriorFactor per pose (weight prior_weight, target
// prior_poses: fixes the gauge like the reference's marginalization prior / GNSS factors do).  Two solves with the chi-square
// culling pass in between (chi2 <= 0: one solve of iters1 iterations).  All parameter arrays are updated in place.
// summary10: initial cost, cost after solve 1, final cost, successful steps 1, unsuccessful 1, successful 2, unsuccessful 2, removed,
// ms spent in solve + culling, ms spent building the problem (context, factor upload)
int icgh_backend_solve(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm, int n_poses,
                       double *poses, double *ext, int n_lm, double *invdepth, double *td, const double *prior_poses, double prior_weight,
                       double huber, int ext_constant, int td_constant, int iters1, int iters2, double chi2, double *summary8,
                       uint8_t *active_out, char *err, int errlen) {
    try {
        auto t_begin = std::chrono::steady_clock::now();
        vector<std::unique_ptr<ReprojectionFactor>> factors;
        ReprojectionBatch batch(0);
        for (int k = 0; k < n; k++) {
            auto o = [&](int c) { return obs_soa[(size_t) c * n + k]; };
            factors.emplace_back(new ReprojectionFactor(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                        Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14)));
            batch.add(factors.back().get(), poses + 7 * (size_t) idx_i[k], poses + 7 * (size_t) idx_j[k], ext, invdepth + idx_lm[k], td);
        }
        batch.finalize();
        auto t_built = std::chrono::steady_clock::now();
        WindowSolver solver(&batch, huber);
        for (int k = 0; k < n_poses; k++) solver.addParameterBlock(poses + 7 * (size_t) k, 7, true);
        solver.addParameterBlock(ext, 7, true);
        for (int l = 0; l < n_lm; l++) solver.addParameterBlock(invdepth + l, 1);
        solver.addParameterBlock(td, 1);
        if (ext_constant) solver.setParameterBlockConstant(ext);
        if (td_constant) solver.setParameterBlockConstant(td);
        for (int k = 0; k < n_poses; k++)
            solver.addResidualBlock(std::make_shared<PosePriorFactor>(prior_poses + 7 * (size_t) k, prior_weight), nullptr,
                                    {poses + 7 * (size_t) k});
        WindowSolver::Options opt;
        WindowSolver::Summary s1, s2;
        opt.max_num_iterations = iters1;
        if (!solver.solve(opt, &s1)) {
            set_err(err, errlen, solver.error().c_str());
            return -2;
        }
        summary8[0] = s1.initial_cost, summary8[1] = s1.final_cost, summary8[2] = s1.final_cost;
        summary8[3] = s1.num_successful_steps, summary8[4] = s1.num_unsuccessful_steps;
        summary8[5] = summary8[6] = summary8[7] = 0;
        if (chi2 > 0) {
            int removed = solver.removeReprojectionFactorsByChi2(chi2);
            if (removed < 0) {
                set_err(err, errlen, solver.error().c_str());
                return -3;
            }
            opt.max_num_iterations = iters2;
            if (!solver.solve(opt, &s2)) {
                set_err(err, errlen, solver.error().c_str());
                return -4;
            }
            summary8[2] = s2.final_cost, summary8[5] = s2.num_successful_steps, summary8[6] = s2.num_unsuccessful_steps, summary8[7] = removed;
        }
        summary8[8] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_built).count();
        summary8[9] = std::chrono::duration<double, std::milli>(t_built - t_begin).count();
        if (active_out) memcpy(active_out, solver.activeReprojectionFactors().data(), (size_t) n);
        if (getenv("ICG_SOLVER_DEBUG")) fprintf(stderr, "%s\n%s\n", s1.BriefReport().c_str(), s2.BriefReport().c_str());
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// ---- f4: MISC (misc_hip.h) driven through flat arrays: imu rows of 8, state rows of 23 (see include/icgvins_hip.h) ----------
namespace {
icg::IMU ins_imu(const double *p) {
    icg::IMU s;
    s.time = p[0], s.dt = p[1];
    s.dtheta = icg::Vector3d(p[2], p[3], p[4]);
    s.dvel   = icg::Vector3d(p[5], p[6], p[7]);
    return s;
}
void ins_put_imu(const icg::IMU &m, double *r) {
    r[0] = m.time, r[1] = m.dt;
    for (int k = 0; k < 3; k++) r[2 + k] = m.dtheta[k], r[5 + k] = m.dvel[k];
}
icg::IntegrationState ins_state(const double *r) {
    icg::IntegrationState s;
    s.time = r[0];
    for (int k = 0; k < 3; k++) s.p[k] = r[1 + k], s.v[k] = r[8 + k], s.bg[k] = r[11 + k], s.ba[k] = r[14 + k], s.sg[k] = r[17 + k], s.sa[k] = r[20 + k];
    s.q = icg::Quaterniond{r[4], r[5], r[6], r[7]};
    return s;
}
void ins_put_state(const icg::IntegrationState &s, double *r) {
    r[0] = s.time;
    for (int k = 0; k < 3; k++) r[1 + k] = s.p[k], r[8 + k] = s.v[k], r[11 + k] = s.bg[k], r[14 + k] = s.ba[k], r[17 + k] = s.sg[k], r[20 + k] = s.sa[k];
    r[4] = s.q.x, r[5] = s.q.y, r[6] = s.q.z, r[7] = s.q.w;
}
icg::IntegrationConfiguration ins_config(const double *c) {
    icg::IntegrationConfiguration cfg;
    cfg.gravity     = icg::Vector3d(c[0], c[1], c[2]);
    cfg.iewn        = icg::Vector3d(c[3], c[4], c[5]);
    cfg.iswithearth = c[6] != 0, cfg.iswithscale = c[7] != 0;
    return cfg;
}
icg::InsWindow ins_window(int n, const double *imu, const double *states) {
    icg::InsWindow w;
    for (int k = 0; k < n; k++) w.emplace_back(ins_imu(imu + 8 * (size_t) k), states ? ins_state(states + 23 * (size_t) k) : icg::IntegrationState());
    return w;
}
struct TempCtx {
    icg_ctx *ctx = nullptr;
    explicit TempCtx(int device) {
        icg_ctx_config cfg{};
        cfg.device = device, cfg.width = 64, cfg.height = 64, cfg.n_slots = 1, cfg.max_batch = 1, cfg.max_points = 64;
        if (icg_ctx_create(&cfg, &ctx) != ICG_OK) throw std::runtime_error(icg_last_error(nullptr));
    }
    ~TempCtx() { icg_ctx_destroy(ctx); }
};
} // namespace

int icgh_ins_mechanize(int n_streams, const int32_t *offsets, const double *imu, const double *cfg8, double *states23, double *traj23,
                       char *err, int errlen) {
    try {
        TempCtx T(0);
        std::vector<std::vector<icg::IMU>> series((size_t) n_streams);
        std::vector<icg::IntegrationState> st((size_t) n_streams);
        std::vector<const std::vector<icg::IMU> *> sp;
        std::vector<icg::IntegrationState *> stp;
        for (int s = 0; s < n_streams; s++) {
            for (int r = offsets[s]; r < offsets[s + 1]; r++) series[(size_t) s].push_back(ins_imu(imu + 8 * (size_t) r));
            st[(size_t) s] = ins_state(states23 + 23 * (size_t) s);
            sp.push_back(&series[(size_t) s]);
            stp.push_back(&st[(size_t) s]);
        }
        std::vector<std::vector<icg::IntegrationState>> traj;
        std::string e;
        if (!icg::MISC::insMechanizationBatch(T.ctx, ins_config(cfg8), sp, stp, traj23 ? &traj : nullptr, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        for (int s = 0; s < n_streams; s++) {
            ins_put_state(st[(size_t) s], states23 + 23 * (size_t) s);
            if (traj23)
                for (size_t k = 0; k < traj[(size_t) s].size(); k++) ins_put_state(traj[(size_t) s][k], traj23 + 23 * ((size_t) offsets[s] + 1 + k));
        }
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// one (window, time) query per stream; windows are concatenated, stream s owns rows [win_offsets[s], win_offsets[s+1])
int icgh_ins_camera_pose(int n_streams, const int32_t *win_offsets, const double *imu, const double *states, const double *pose_b_c12,
                         const double *times, double *pose12_out, uint8_t *found_out, char *err, int errlen) {
    try {
        TempCtx T(0);
        std::vector<icg::InsWindow> w;
        std::vector<const icg::InsWindow *> wp;
        for (int s = 0; s < n_streams; s++)
            w.push_back(ins_window(win_offsets[s + 1] - win_offsets[s], imu + 8 * (size_t) win_offsets[s], states + 23 * (size_t) win_offsets[s]));
        for (auto &x : w) wp.push_back(&x);
        icg::Pose pbc;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) pbc.R(i, j) = pose_b_c12[3 * i + j];
            pbc.t[i] = pose_b_c12[9 + i];
        }
        std::vector<icg::Pose> poses;
        std::vector<uint8_t> found;
        std::string e;
        if (!icg::MISC::getCameraPoseFromInsWindowBatch(T.ctx, wp, pbc, std::vector<double>(times, times + n_streams), poses, found, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        for (int s = 0; s < n_streams; s++) {
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) pose12_out[12 * (size_t) s + 3 * i + j] = poses[(size_t) s].R(i, j);
                pose12_out[12 * (size_t) s + 9 + i] = poses[(size_t) s].t[i];
            }
            found_out[s] = found[(size_t) s];
        }
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// MISC::writeNavResult into <dir>/nav.txt, err.txt, traj.txt: `calls` consecutive calls with the same state (a row every 10th call)
int icgh_ins_write_nav_result(const double *cfg8, const double *origin3, const double *state23, double sodo, const char *dir, int calls) {
    icg::IntegrationConfiguration cfg = ins_config(cfg8);
    cfg.origin                        = icg::Vector3d(origin3[0], origin3[1], origin3[2]);
    icg::IntegrationState st          = ins_state(state23);
    st.sodo                           = sodo;
    std::string d(dir);
    auto nav = icg::FileSaver::create(d + "/nav.txt", 11), errf = icg::FileSaver::create(d + "/err.txt", 7), traj = icg::FileSaver::create(d + "/traj.txt", 8);
    if (!nav->isOpen() || !errf->isOpen() || !traj->isOpen()) return -1;
    for (int k = 0; k < calls; k++) icg::MISC::writeNavResult(cfg, st, nav, errf, traj);
    return 0;
}

long icgh_ins_window_index(int n_win, const double *imu, double time) {
    return (long) icg::MISC::getInsWindowIndex(ins_window(n_win, imu, nullptr), time);
}

// MISC::getImuSeriesFromTo: number of samples written, -1 on failure, -2 when cap is too small
int icgh_ins_imu_series(int n_win, const double *imu, double start, double end, int cap, double *series) {
    std::vector<icg::IMU> out;
    if (!icg::MISC::getImuSeriesFromTo(ins_window(n_win, imu, nullptr), start, end, out)) return -1;
    if ((int) out.size() > cap) return -2;
    for (size_t k = 0; k < out.size(); k++) ins_put_imu(out[k], series + 8 * k);
    return (int) out.size();
}

// MISC::redoInsMechanizationBatch: windows concatenated like icgh_ins_camera_pose; states updated in place, new_len[s] = the
// window length after the expired front entries were dropped (rows compacted to the front of each stream's slice)
int icgh_ins_redo(int n_streams, const double *cfg8, const double *updated23, int reserved, const int32_t *win_offsets, double *imu,
                  double *states, int32_t *new_len, char *err, int errlen) {
    try {
        TempCtx T(0);
        std::vector<icg::InsWindow> w;
        std::vector<icg::InsWindow *> wp;
        std::vector<icg::IntegrationState> upd;
        for (int s = 0; s < n_streams; s++) {
            w.push_back(ins_window(win_offsets[s + 1] - win_offsets[s], imu + 8 * (size_t) win_offsets[s], states + 23 * (size_t) win_offsets[s]));
            upd.push_back(ins_state(updated23 + 23 * (size_t) s));
        }
        for (auto &x : w) wp.push_back(&x);
        std::string e;
        if (!icg::MISC::redoInsMechanizationBatch(T.ctx, ins_config(cfg8), upd, (size_t) reserved, wp, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        for (int s = 0; s < n_streams; s++) {
            new_len[s] = (int32_t) w[(size_t) s].size();
            for (size_t k = 0; k < w[(size_t) s].size(); k++) {
                ins_put_imu(w[(size_t) s][k].first, imu + 8 * ((size_t) win_offsets[s] + k));
                ins_put_state(w[(size_t) s][k].second, states + 23 * ((size_t) win_offsets[s] + k));
            }
        }
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// icgh_backend_solve for W windows at once on WindowSolverBatch (lock-step LM, one launch per phase for all windows).  Arrays are
// concatenated window-major: window w owns factors [fac_off[w], fac_off[w+1]) (obs_soa is 15 x n_total, idx_* LOCAL to the window),
// poses [pose_off[w], ..), inverse depths [lm_off[w], ..); ext is W x 7, td has W entries.  summary8 is W x 8 as in icgh_backend_solve.
// Returns the wall time of the two solves + culling in ms through *solve_ms.
int icgh_backend_solve_batch(int W, const int32_t *fac_off, const int32_t *pose_off, const int32_t *lm_off, const double *obs_soa, const int32_t *idx_i,
                             const int32_t *idx_j, const int32_t *idx_lm, double *poses, double *ext, double *invdepth, double *td,
                             const double *prior_poses, double prior_weight, double huber, int ext_constant, int td_constant, int iters1, int iters2,
                             double chi2, double *summary8, double *solve_ms, char *err, int errlen) {
    try {
        const int n = fac_off[W];
        vector<std::unique_ptr<ReprojectionFactor>> factors;
        WindowSolverBatch solver(0, huber);
        for (int w = 0; w < W; w++) {
            const int ww = solver.addWindow();
            double *P = poses + 7 * (size_t) pose_off[w], *E = ext + 7 * (size_t) w, *D = invdepth + lm_off[w], *TD = td + w;
            const int K = pose_off[w + 1] - pose_off[w], L = lm_off[w + 1] - lm_off[w];
            for (int k = 0; k < K; k++) solver.addParameterBlock(ww, P + 7 * (size_t) k, 7, true);
            solver.addParameterBlock(ww, E, 7, true);
            for (int l = 0; l < L; l++) solver.addParameterBlock(ww, D + l, 1);
            solver.addParameterBlock(ww, TD, 1);
            if (ext_constant) solver.setParameterBlockConstant(ww, E);
            if (td_constant) solver.setParameterBlockConstant(ww, TD);
            for (int f = fac_off[w]; f < fac_off[w + 1]; f++) {
                auto o = [&](int c) { return obs_soa[(size_t) c * n + f]; };
                factors.emplace_back(new ReprojectionFactor(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                            Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14)));
                solver.addReprojectionFactor(ww, factors.back().get(), P + 7 * (size_t) idx_i[f], P + 7 * (size_t) idx_j[f], E, D + idx_lm[f], TD);
            }
            for (int k = 0; k < K; k++)
                solver.addResidualBlock(ww, std::make_shared<PosePriorFactor>(prior_poses + 7 * ((size_t) pose_off[w] + k), prior_weight), nullptr,
                                        {P + 7 * (size_t) k});
        }
        if (!solver.prepare()) {
            set_err(err, errlen, solver.error().c_str());
            return -5;
        }
        auto t0 = std::chrono::steady_clock::now();
        WindowSolverBatch::Options opt;
        vector<WindowSolverBatch::Summary> s1, s2;
        opt.max_num_iterations = iters1;
        if (!solver.solve(opt, &s1)) {
            set_err(err, errlen, solver.error().c_str());
            return -2;
        }
        vector<int> removed((size_t) W, 0);
        if (chi2 > 0) {
            removed                = solver.removeReprojectionFactorsByChi2(chi2);
            opt.max_num_iterations = iters2;
            if (!solver.solve(opt, &s2)) {
                set_err(err, errlen, solver.error().c_str());
                return -4;
            }
        }
        if (solve_ms) *solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int w = 0; w < W; w++) {
            double *o = summary8 + 8 * (size_t) w;
            o[0] = s1[(size_t) w].initial_cost, o[1] = s1[(size_t) w].final_cost, o[2] = s1[(size_t) w].final_cost;
            o[3] = s1[(size_t) w].num_successful_steps, o[4] = s1[(size_t) w].num_unsuccessful_steps, o[5] = o[6] = o[7] = 0;
            if (chi2 > 0)
                o[2] = s2[(size_t) w].final_cost, o[5] = s2[(size_t) w].num_successful_steps, o[6] = s2[(size_t) w].num_unsuccessful_steps,
                o[7] = removed[(size_t) w];
        }
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// Aggregate solve throughput with many windows in flight: `threads` host threads, each with its own ReprojectionBatch (own icg_ctx
// and HIP stream, like the stream groups of the front-end) and its own copy of the problem, each solving it `repeat` times from
// the same start (problem construction outside the timed region).  Returns the wall time in seconds for threads x repeat solves,
// < 0 on error.  Same flow as icgh_backend_solve (two solves with the chi-square pass).
double icgh_backend_solve_throughput(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm, int n_poses,
                                     const double *poses, const double *ext, int n_lm, const double *invdepth, double td,
                                     const double *prior_poses, double prior_weight, double huber, int iters1, int iters2, double chi2, int threads,
                                     int repeat, char *err, int errlen) {
    try {
        struct Job {
            vector<double> P, E, D;
            double TD;
            vector<std::unique_ptr<ReprojectionFactor>> factors;
            std::unique_ptr<ReprojectionBatch> batch;
        };
        vector<std::unique_ptr<Job>> jobs;
        for (int t = 0; t < threads; t++) {
            std::unique_ptr<Job> J(new Job);
            J->P.assign(poses, poses + 7 * (size_t) n_poses), J->E.assign(ext, ext + 7), J->D.assign(invdepth, invdepth + n_lm), J->TD = td;
            J->batch.reset(new ReprojectionBatch(0));
            if (threads > 4) J->batch->setWaitMode(ICG_WAIT_POLL, 5); // more solvers than spare host cores: do not spin on completion
            for (int k = 0; k < n; k++) {
                auto o = [&](int c) { return obs_soa[(size_t) c * n + k]; };
                J->factors.emplace_back(new ReprojectionFactor(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                               Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14)));
                J->batch->add(J->factors.back().get(), &J->P[7 * (size_t) idx_i[k]], &J->P[7 * (size_t) idx_j[k]], J->E.data(), &J->D[(size_t) idx_lm[k]],
                              &J->TD);
            }
            J->batch->finalize();
            jobs.push_back(std::move(J));
        }
        std::atomic<int> failed{0};
        auto work = [&](int t) {
            Job &J = *jobs[(size_t) t];
            for (int r = 0; r < repeat; r++) {
                J.P.assign(poses, poses + 7 * (size_t) n_poses), J.E.assign(ext, ext + 7), J.D.assign(invdepth, invdepth + n_lm), J.TD = td;
                WindowSolver solver(J.batch.get(), huber);
                for (int k = 0; k < n_poses; k++) solver.addParameterBlock(&J.P[7 * (size_t) k], 7, true);
                solver.addParameterBlock(J.E.data(), 7, true);
                for (int l = 0; l < n_lm; l++) solver.addParameterBlock(&J.D[(size_t) l], 1);
                solver.addParameterBlock(&J.TD, 1);
                for (int k = 0; k < n_poses; k++)
                    solver.addResidualBlock(std::make_shared<PosePriorFactor>(prior_poses + 7 * (size_t) k, prior_weight), nullptr, {&J.P[7 * (size_t) k]});
                WindowSolver::Options opt;
                WindowSolver::Summary s1;
                opt.max_num_iterations = iters1;
                if (!solver.solve(opt, &s1)) failed++;
                if (chi2 > 0) {
                    solver.removeReprojectionFactorsByChi2(chi2);
                    opt.max_num_iterations = iters2;
                    if (!solver.solve(opt, &s1)) failed++;
                }
            }
        };
        auto t0 = std::chrono::steady_clock::now();
        vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (failed.load()) {
            set_err(err, errlen, "a solve failed");
            return -2.0;
        }
        return sec;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1.0;
    }
}

namespace {
// r = w (x - x0) on one 9-vector block (velocity, gyroscope bias, accelerometer bias): stands in for the part of the
// marginalization prior that anchors the first state's velocity and biases
class MixPriorFactor : public ceres::SizedCostFunction<9, 9> {
public:
    MixPriorFactor(const double *x0, double weight) : w_(weight) { memcpy(x0_, x0, sizeof x0_); }
    bool Evaluate(const double *const *parameters, double *residuals, double **jacobians) const override {
        for (int k = 0; k < 9; k++) residuals[k] = w_ * (parameters[0][k] - x0_[k]);
        if (jacobians && jacobians[0]) {
            memset(jacobians[0], 0, sizeof(double) * 81);
            for (int k = 0; k < 9; k++) jacobians[0][k * 9 + k] = w_;
        }
        return true;
    }

private:
    double x0_[9], w_;
};
} // namespace

// f1 with the factor mix of the real window: K preintegration factors (device P1 + host P2) between K+1 states, reprojection
// factors on the same pose blocks (device), a pose prior and a velocity/bias prior on state 0 (what the marginalization prior
// provides in the real window).  states: (K+1) x 16 (p3, q4 xyzw, v3, bg3, ba3) in/out;
// imu rows of 8, interval k owns rows [offsets[k], offsets[k+1]).  summary4: initial cost, final cost, successful, unsuccessful steps.
int icgh_backend_solve_vio(int n_intervals, const int32_t *offsets, const double *imu, const double *params9, double *states16, int n,
                           const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm, double *ext, int n_lm,
                           double *invdepth, double *td, const double *prior_pose0, const double *prior_mix0, double prior_weight, double huber,
                           int iters, double *summary4, char *err, int errlen) {
    try {
        const int K = n_intervals + 1;
        vector<double> pose((size_t) K * 7), mix((size_t) K * 9);
        for (int k = 0; k < K; k++) {
            memcpy(&pose[7 * (size_t) k], states16 + 16 * (size_t) k, sizeof(double) * 7);
            memcpy(&mix[9 * (size_t) k], states16 + 16 * (size_t) k + 7, sizeof(double) * 9);
        }
        vector<std::unique_ptr<ReprojectionFactor>> factors;
        ReprojectionBatch batch(0);
        for (int f = 0; f < n; f++) {
            auto o = [&](int c) { return obs_soa[(size_t) c * n + f]; };
            factors.emplace_back(new ReprojectionFactor(Vector3d(o(0), o(1), o(2)), Vector3d(o(3), o(4), o(5)), Vector3d(o(6), o(7), o(8)),
                                                        Vector3d(o(9), o(10), o(11)), o(12), o(13), o(14)));
            batch.add(factors.back().get(), &pose[7 * (size_t) idx_i[f]], &pose[7 * (size_t) idx_j[f]], ext, invdepth + idx_lm[f], td);
        }
        batch.finalize();
        // preintegration of every interval from its start state: one icg_preint_batch launch on a context of its own
        auto P           = std::make_shared<IntegrationParameters>();
        P->gyr_arw = params9[0], P->acc_vrw = params9[1], P->gyr_bias_std = params9[2], P->acc_bias_std = params9[3], P->corr_time = params9[4];
        P->gravity = params9[5];
        P->iewn    = Vector3d(params9[6], params9[7], params9[8]);
        TempCtx T(0);
        vector<std::shared_ptr<Preintegration>> pre;
        vector<Preintegration *> raw;
        for (int k = 0; k < n_intervals; k++) {
            const double *s0 = states16 + 16 * (size_t) k;
            IntegrationState st;
            st.p = Vector3d(s0[0], s0[1], s0[2]);
            st.q = Quaterniond{s0[3], s0[4], s0[5], s0[6]};
            st.v = Vector3d(s0[7], s0[8], s0[9]), st.bg = Vector3d(s0[10], s0[11], s0[12]), st.ba = Vector3d(s0[13], s0[14], s0[15]);
            auto p = std::make_shared<Preintegration>(P, ins_imu(imu + 8 * (size_t) offsets[k]), st, Preintegration::NORMAL);
            for (int row = offsets[k] + 1; row < offsets[k + 1]; row++) p->addNewImu(ins_imu(imu + 8 * (size_t) row));
            pre.push_back(p);
            raw.push_back(p.get());
        }
        std::string e;
        if (!Preintegration::integrateBatch(T.ctx, raw, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        WindowSolver solver(&batch, huber);
        for (int k = 0; k < K; k++) {
            solver.addParameterBlock(&pose[7 * (size_t) k], 7, true);
            solver.addParameterBlock(&mix[9 * (size_t) k], 9);
        }
        solver.addParameterBlock(ext, 7, true);
        for (int l = 0; l < n_lm; l++) solver.addParameterBlock(invdepth + l, 1);
        solver.addParameterBlock(td, 1);
        solver.setParameterBlockConstant(ext); // estimated off-line in the default configuration (optimize_estimate_extrinsic: false)
        solver.setParameterBlockConstant(td);
        for (int k = 0; k < n_intervals; k++)
            solver.addResidualBlock(std::make_shared<PreintegrationFactor>(pre[(size_t) k]), nullptr,
                                    {&pose[7 * (size_t) k], &mix[9 * (size_t) k], &pose[7 * (size_t) (k + 1)], &mix[9 * (size_t) (k + 1)]});
        solver.addResidualBlock(std::make_shared<PosePriorFactor>(prior_pose0, prior_weight), nullptr, {&pose[0]});
        solver.addResidualBlock(std::make_shared<MixPriorFactor>(prior_mix0, prior_weight), nullptr, {&mix[0]});
        WindowSolver::Options opt;
        WindowSolver::Summary sum;
        opt.max_num_iterations = iters;
        if (!solver.solve(opt, &sum)) {
            set_err(err, errlen, solver.error().c_str());
            return -3;
        }
        summary4[0] = sum.initial_cost, summary4[1] = sum.final_cost, summary4[2] = sum.num_successful_steps, summary4[3] = sum.num_unsuccessful_steps;
        for (int k = 0; k < K; k++) {
            memcpy(states16 + 16 * (size_t) k, &pose[7 * (size_t) k], sizeof(double) * 7);
            memcpy(states16 + 16 * (size_t) k + 7, &mix[9 * (size_t) k], sizeof(double) * 9);
        }
        if (getenv("ICG_SOLVER_DEBUG")) fprintf(stderr, "%s\n", sum.BriefReport().c_str());
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

} // extern "C"
