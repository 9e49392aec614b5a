// C entry points of libicgvins_host.so for harnesses that cannot speak C++ (tests, bench.py): drive a TrackingBatch.
#include <cstring>
#include <stdexcept>

#include "tracking_batch.h"

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
        vector<vector<Frame::Ptr>> frames((size_t) K, vector<Frame::Ptr>((size_t) n));
        for (int k = 0; k < K; k++)
            for (int i = 0; i < n; i++) {
                const size_t j = (size_t) k * n + i;
                if (!images[j]) continue;
                Mat img = Mat::wrap((uint8_t *) images[j], b->h, b->w, channels, (size_t) stride, on_device != 0);
                auto f  = Frame::createFrame(stamps[j], img, b->tb->stream(i).ids);
                Pose p;
                memcpy(p.R.m, poses12 + 12 * j, sizeof(double) * 9);
                memcpy(p.t.v, poses12 + 12 * j + 9, sizeof(double) * 3);
                f->setPose(p);
                frames[(size_t) k][(size_t) i] = f;
            }
        vector<vector<TrackState>> st;
        b->tb->stepMany(frames, st);
        for (int k = 0; k < K; k++)
            for (int i = 0; i < n; i++) states[(size_t) k * n + i] = (int32_t) st[(size_t) k][(size_t) i];
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
    out8[5] = s.map->keyframes().size();
    out8[6] = s.map->landmarks().size();
    out8[7] = (uint64_t) s.last_state;
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

// features of the stream's current frame, sorted by map-point id: ids[k], px[2k..2k+1] (distorted keypoint)
int icgh_batch_features(icgh_batch *b, int stream, int max, uint64_t *ids, float *px) {
    if (!b || stream < 0 || stream >= b->tb->size()) return -1;
    auto frame = b->tb->stream(stream).tracking->currentFrame();
    if (!frame) return 0;
    auto feats = frame->features();
    vector<ulong> v;
    for (auto &kv : feats) v.push_back(kv.first);
    std::sort(v.begin(), v.end());
    int n = 0;
    for (ulong id : v) {
        if (n >= max) break;
        ids[n]        = id;
        px[2 * n]     = feats[id]->distortedKeyPoint().x;
        px[2 * n + 1] = feats[id]->distortedKeyPoint().y;
        n++;
    }
    return n;
}

} // extern "C"
