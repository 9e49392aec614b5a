// Device-resident tracker of a stream group (include/icgvins_hip.h, icg_tracker_*): the stage bodies of the tracker core
// (../host/track_core.h — the reference's per-frame algorithm between its image primitives, tracking/tracking.cc:144-245, 263-307, 351-574,
// 576-688 (lists), 690-798, 831-922 and map.cc:27-127) compiled for gfx950 and run by one wave per stream, between the primitives'
// segmented launches.  One step of a group:
//
//   k_trk_begin     new frame: id, pose, stamp, a slot from the stream's own pool                       -> slot per stream
//   preprocess      CLAHE + pyramid into that slot (image.hip, slots read from device memory)
//   k_trk_stage 12  roles rotate, state machine; INS-aided prediction of every map point of the previous frame in the reference
//                   container's order + rotation-compensated prediction of the candidates                 -> LK points per stream
//   LK              forward / backward / cull / undistort of every stream's segment (lk.hip)
//   k_trk_stage 3   stable compaction, new feature rows + container order, observation bookkeeping, the parallax sums in container
//                   order (sequential float adds: the reference's summation order survives), velocities  -> RANSAC set per stream
//   RANSAC          the whole findFundamentalMat run of every set in one launch (ransac.hip k_fm_ransac_sets)
//   k_trk_stage 4   inlier mask applied, keyframe decision, triangulation list                            -> points + camera matrices
//   triangulate     segmented DLT (ransac.hip)
//   k_trk_stage 5   gates, new map points, rows in both frames; detection job: block quotas, disc centres  -> ROI table
//   detect          k_min_eig_nms + k_select_subpix on the dense ROI table (detect.hip)
//   k_trk_stage 6   corners in block order -> candidates; slots released; statistics, digest, window keeper, frame sweep -> results
//
// (stage 1 -> detection -> stage 2 replaces stage 12 in the rare steps in which a stream starts with a detection: first frame,
// initialization without candidates — the host knows from the previous step's results.)  The host issues the launches and waits ONCE.
//
// Parallelism: streams are independent — a wave per stream, all streams of the group in one launch.  Inside a stream the data-parallel loops
// (prediction of every map point, feature rows, velocities, parallax terms, the compactions of reduceVector, detection lists, the digest)
// are split over the 64 lanes with ballot / rank so that list order is kept; what is sequential by nature — the walk of the reference
// container's node list, the hash-table insertions that give a new frame its iteration order, the order-dependent float sums — runs
// through LDS (tc::Scratch) and is executed redundantly by all lanes in lockstep (track_core.h "execution model").
#include <algorithm>
#include <chrono>
#include <ctime>
#include <mutex>

#include "icg_internal.h"
#include "../host/track_core.h"
#if defined(TC_TIMING)
namespace tc {
__device__ unsigned long long g_tc_marks[256];
}
#endif

static_assert(sizeof(icg_tracker_config) == sizeof(tc::Cfg), "icg_tracker_config mirrors tc::Cfg");
static_assert(offsetof(icg_tracker_config, track_min_parallax) == offsetof(tc::Cfg, track_min_parallax), "icg_tracker_config mirrors tc::Cfg");
static_assert(offsetof(icg_tracker_config, max_per_job) == offsetof(tc::Cfg, max_per_job), "icg_tracker_config mirrors tc::Cfg");

namespace {

struct TrkInput { // per stream and step, written by the host into pinned memory, read once by k_trk_begin
    double stamp;
    double pose[12];
    unsigned long long image;
    int32_t valid, pad;
};

// the group's arenas: per field one array of n_streams segments (track_core.h Io, SoA across streams so that the primitives index flat)
struct TrkArena {
    int32_t *active;    // [n] the stream has a frame in this step
    int32_t *pre_slot;  // [n]
    double *pre_hist;   // [n]
    int32_t *lk_count, *lk_prev_slot, *lk_next_slot;
    float2 *lk_prev, *lk_guess, *lk_out, *lk_undist;
    uint8_t *lk_status;
    int32_t *rs_count;
    float2 *rs_p1, *rs_p2;
    uint8_t *rs_mask;
    int32_t *tri_count, *tri_n_tcw, *tri_T0, *tri_T1;
    double *tri_Tcw, *tri_pc0, *tri_pc1, *tri_pw;
    int32_t *det_slot, *det_quota, *det_mask_begin, *det_mask_count, *det_count;
    float2 *det_mask_pts, *det_out;
    int32_t *work;        // [n x 4] lk points, detection jobs, RANSAC sets, triangulated points of this step
    det_roi *rois;        // [n x block_cnts]
    float2 *corners;      // [n x block_cnts x max_block_features]
    int32_t *corner_cnt;  // [n x block_cnts]
    float2 *picks;        // [n x block_cnts x max_block_features] integer corners between k_select and k_subpix
    int32_t *pick_cnt;    // [n x block_cnts]
};

__device__ __forceinline__ tc::Io io_of(const TrkArena &A, const tc::Cfg &C, int s) {
    tc::Io io;
    const size_t r = (size_t) s * tc::MAX_ROWS;
    io.pre_slot = A.pre_slot + s, io.pre_hist = A.pre_hist + s;
    io.lk_count = A.lk_count + s, io.lk_prev_slot = A.lk_prev_slot + r, io.lk_next_slot = A.lk_next_slot + r;
    io.lk_prev = (tc::P2f *) (A.lk_prev + r), io.lk_guess = (tc::P2f *) (A.lk_guess + r);
    io.lk_out = (const tc::P2f *) (A.lk_out + r), io.lk_undist = (const tc::P2f *) (A.lk_undist + r);
    io.lk_status = A.lk_status + r, io.lk_base = (int32_t) r;
    io.rs_count = A.rs_count + s, io.rs_p1 = (tc::P2f *) (A.rs_p1 + r), io.rs_p2 = (tc::P2f *) (A.rs_p2 + r), io.rs_mask = A.rs_mask + r;
    io.tri_count = A.tri_count + s, io.tri_n_tcw = A.tri_n_tcw + s, io.tri_T0 = A.tri_T0 + r, io.tri_T1 = A.tri_T1 + r;
    io.tri_Tcw = A.tri_Tcw + (size_t) s * tc::MAX_TCW * 12, io.tri_pc0 = A.tri_pc0 + 3 * r, io.tri_pc1 = A.tri_pc1 + 3 * r, io.tri_pw = A.tri_pw + 3 * r;
    io.det_slot = A.det_slot + s, io.det_quota = A.det_quota + (size_t) s * tc::MAX_BLOCKS, io.det_mask_count = A.det_mask_count + s;
    io.det_mask_pts = (tc::P2f *) (A.det_mask_pts + r), io.det_count = A.det_count + s, io.det_out = (const tc::P2f *) (A.det_out + r);
    return io;
}

// the stream's detection job as the dense ROI table detect.hip takes (tracking.cc:629-645; the host path builds the same list in icg_detect)
__device__ void build_rois(const TrkArena &A, const tc::Cfg &C, int s) {
    const int nblk   = C.block_cnts;
    const bool job   = A.det_slot[s] >= 0;
    det_roi *R       = A.rois + (size_t) s * nblk;
    const int32_t *q = A.det_quota + (size_t) s * tc::MAX_BLOCKS;
    for (int k = tc::lane(); k < nblk; k += tc::NL) { // a block per lane
        det_roi r;
        const int cols = k % C.block_cols, rows = k / C.block_cols;
        r.job   = s;
        r.block = k;
        r.rx    = cols * C.block_w;
        r.ry    = rows * C.block_h;
        r.rw    = C.block_w;
        r.rh    = C.block_h;
        if (k != nblk - 1) {
            r.rw -= 5;
            r.rh -= 5;
        }
        int quota = job ? q[k] : 0;
        if (quota > C.max_block_features) quota = C.max_block_features;
        r.quota     = quota;
        r.cand_base = k * C.block_w * C.block_h;
        R[k]        = r;
    }
    A.det_mask_begin[s] = s * tc::MAX_ROWS;
    if (!job) A.det_mask_count[s] = 0;
}

// block-order assembly of the refined corners with the block origin added (tracking.cc:669-685; the tail of icg_detect)
// A block per lane (block_cnts <= MAX_BLOCKS = 64 = the wave): where a block's corners start in the list is the number of corners the blocks
// before it hand over — an exclusive prefix sum over the lanes —, cut at max_per_job as the one-by-one loop cuts it.
__device__ void assemble_corners(const TrkArena &A, const tc::Cfg &C, int s) {
    static_assert(tc::MAX_BLOCKS <= 64, "assemble_corners: a block per lane");
    int cnt = 0;
    if (A.det_slot[s] >= 0) {
        const int nblk = C.block_cnts, max_pb = C.max_block_features;
        float2 *out    = A.det_out + (size_t) s * tc::MAX_ROWS;
        const int k    = tc::lane();
        det_roi R{};
        int n = 0;
        if (k < nblk) {
            R = A.rois[(size_t) s * nblk + k];
            n = R.quota > 0 ? A.corner_cnt[(size_t) s * nblk + k] : 0;
            if (n < 0) n = 0;
        }
        const int incl = (int) tc::wave_scan_incl((uint32_t) n);
        int begin      = incl - n;
        if (begin > C.max_per_job) begin = C.max_per_job;
        int take = n;
        if (begin + take > C.max_per_job) take = C.max_per_job - begin;
        const float2 *src = A.corners + ((size_t) s * nblk + (k < nblk ? k : 0)) * max_pb;
        for (int i = 0; i < max_pb; i++) // (every lane walks its own block: <= max_block_features steps for the wave)
            if (i < take) {
                const float2 c = src[i];
                out[begin + i] = make_float2((float) R.rx + c.x, (float) R.ry + c.y);
            }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        cnt             = total < C.max_per_job ? total : C.max_per_job;
    }
    A.det_count[s] = cnt;
    tc::sync(); // (the stage body that follows reads the list: every lane, entries other lanes stored)
}

__device__ void begin_frame(int s, tc::Stream *streams, const TrkArena &A, const TrkInput *in) {
    const TrkInput I = in[s];
    A.active[s]      = I.valid;
    A.work[4 * s] = A.work[4 * s + 1] = A.work[4 * s + 2] = A.work[4 * s + 3] = 0;
    if (!I.valid) {
        A.pre_slot[s] = -1;
        A.det_slot[s] = -1;
        A.lk_count[s] = A.rs_count[s] = A.tri_count[s] = A.tri_n_tcw[s] = 0;
        A.det_mask_count[s] = 0;
        return;
    }
    tc::Cfg dummy; // (stage_begin_frame needs no configuration)
    (void) dummy;
    tc::Io io = io_of(A, dummy, s);
    tc::Pose pose;
    for (int k = 0; k < 9; k++) pose.R[k] = I.pose[k];
    for (int k = 0; k < 3; k++) pose.t[k] = I.pose[9 + k];
    tc::stage_begin_frame(streams[s], io, I.stamp, pose, I.image);
}

// stage: 0 (new frame), 1, 2, 12 (1 then 2 in one launch: no stream queued a detection), 120 (0 then 12: additionally no histogram gate, so
// nothing of the preprocessing is needed before the prediction), 3, 4, 5, 6 (+ end of frame + results)
// TRK_WAVES streams per workgroup (a wave each, no workgroup barrier).  Measured (profiles/r04_device_tracker.md): 8 per workgroup — tried to
// keep the ~115 KB of stage code off most CUs' instruction caches — is SLOWER than one (12 x 64: 74.3 k vs 84.8 k frames/s; 2 confined CPUs
// 63.9 k vs 72.9 k): the stage bodies are latency chains, and eight of them sharing one CU's LDS and memory pipeline lengthen every chain.
#define TRK_WAVES 1
__global__ __launch_bounds__(64 * TRK_WAVES) void k_trk_stage(int stage, int n, tc::Stream *streams, TrkArena A, tc::Cfg C, const uint32_t *buckets_after,
                                                              icg_tracker_result *results, const TrkInput *in) {
    __shared__ tc::Scratch XS[TRK_WAVES];
    const int wave = (int) (threadIdx.x >> 6);
    tc::Scratch &X = XS[wave];
    const int s = blockIdx.x * TRK_WAVES + wave;
    if (s >= n) return; // all 64 lanes run the stage body (track_core.h "execution model"): redundantly where it is sequential
    tc::Stream &S = streams[s];
    TC_MARK_START();
    if (stage == 0 || stage == 120) {
        begin_frame(s, streams, A, in);
        tc::sync();
        TC_MARK(1);
        if (stage == 0) return;
        stage = 12;
    }
    const bool active = A.active[s] != 0;
    tc::Io io = io_of(A, C, s);
    bool log_ok  = false;
    int log_rows = 0;
    if (active) {
        switch (stage) {
        case 1:
            tc::stage_on_preprocess(S, C, io, X);
            build_rois(A, C, s);
            A.work[4 * s + 1] += A.det_slot[s] >= 0 ? 1 : 0;
            break;
        case 2:
            assemble_corners(A, C, s);
            tc::stage_on_detect_a(S, C, io, X);
            A.work[4 * s] = A.lk_count[s];
            break;
        case 12:
            tc::stage_on_preprocess(S, C, io, X);
            TC_MARK(10);
            if (!S.done && A.det_slot[s] >= 0) { // a detection was queued but this step runs without the detection-A launches
                S.overflow |= tc::OVF_INTERNAL;
                A.det_slot[s] = -1;
                S.det_job     = -1;
            }
            A.det_count[s] = 0;
            tc::stage_on_detect_a(S, C, io, X);
            TC_MARK(13);
            A.work[4 * s] = A.lk_count[s];
            break;
        case 3:
            tc::stage_on_lk(S, C, io, buckets_after, X);
            A.work[4 * s + 2] = A.rs_count[s] > 0 ? 1 : 0;
            break;
        case 4:
            tc::stage_on_ransac(S, C, io, X);
            TC_MARK(32);
            A.work[4 * s + 3] = A.tri_count[s];
            break;
        case 5:
            tc::stage_on_triangulate(S, C, io, buckets_after, X);
            TC_MARK(46);
            build_rois(A, C, s);
            TC_MARK(47);
            A.work[4 * s + 1] += A.det_slot[s] >= 0 ? 1 : 0;
            break;
        case 6:
            assemble_corners(A, C, s);
            TC_MARK(50);
            tc::stage_on_detect_b(S, C, io);
            TC_MARK(51);
            // the frame's tracking.txt line (TableTracker::endFrame writes it before the window keeper runs: same condition, same count)
            tc::sync();
            log_ok   = S.log_valid && S.result == tc::TRACK_TRACKING && S.mode == tc::M_TRACK && S.lost_reset != 2;
            log_rows = S.cur >= 0 ? S.frame[S.cur].n_rows : 0;
            tc::stage_end_frame(S, C);
            break;
        default: break;
        }
    } else if (stage == 1 || stage == 5) {
        build_rois(A, C, s); // (an idle stream's table entries must read "inactive")
    }
    tc::sync();
    TC_MARK(100 + (stage > 9 ? 2 : stage));
    if (stage == 6) {
        icg_tracker_result r;
        r.active          = active ? 1 : 0;
        r.state           = S.result;
        r.is_new_keyframe = S.isnewkeyframe;
        r.overflow        = S.overflow;
        r.n_features      = S.cur >= 0 ? S.frame[S.cur].n_rows : 0;
        r.n_candidates    = S.n_new;
        r.window_keyframes = S.n_map_kf;
        r.landmarks       = S.n_landmarks;
        r.frames = S.frames, r.keyframes = S.keyframes, r.tracked_sum = S.tracked_sum, r.digest = S.digest;
        r.frame_id = S.frame_id, r.keyframe_id = S.keyframe_id, r.mappoint_id = S.mappoint_id, r.last_input_fid = S.last_input_fid;
        r.need_detect_a = (S.isinitializing && (S.ref < 0 || S.n_ref == 0)) ? 1 : 0; // stage_on_preprocess :158-170
        r.n_log         = S.n_log;
        r.lk_points = A.work[4 * s], r.detect_jobs = A.work[4 * s + 1], r.ransac_sets = A.work[4 * s + 2], r.tri_points = A.work[4 * s + 3];
        r.log_valid = log_ok ? 1 : 0, r.log_features = log_rows;
        for (int k = 0; k < 5; k++) r.log_data[k] = S.log_data[k];
        results[s]      = r;
        TC_MARK(59);
    }
}

__global__ void k_trk_init(int n, tc::Stream *streams) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    tc::stream_init(streams[s], s * tc::MAX_SLOTS);
}
__global__ void k_trk_reset_log(tc::Stream *S) { S->n_log = 0; }

} // namespace

struct icg_tracker {
    icg_ctx *ctx      = nullptr;
    int n             = 0;
    tc::Cfg cfg{};
    icg_detect_grid grid{};
    tc::Stream *d_streams = nullptr;
    uint32_t *d_buckets   = nullptr;
    int32_t *d_vh         = nullptr;
    char *d_arena         = nullptr;
    TrkArena A{};
    TrkInput *h_in             = nullptr; // pinned, GPU-addressable
    icg_tracker_result *h_res  = nullptr; // pinned, written by k_trk_stage 6
    bool need_detect_a         = true;    // some stream starts its next frame with a detection
    hipEvent_t ev_done         = nullptr; // wait_mode 2
    int wait_mode              = 0;       // 0 adaptive (sleep + query), 1 the context's wait mode, 2 blocking event; ICG_TRACKER_WAIT=adaptive|ctx|block
    double step_us_ema         = 0;       // running estimate of a step's issue-to-completion time
};

extern "C" size_t icg_tracker_block_bytes(void) { return sizeof(tc::Stream); }

#if defined(TC_TIMING)
// profiling build: time between marks (10 ns ticks) summed over the streams since the library was loaded, printed when the process ends
static void tc_timing_dump() {
    unsigned long long h[256];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(tc::g_tc_marks), sizeof h) != hipSuccess) return;
    for (int k = 0; k < 128; k++)
        if (h[128 + k]) fprintf(stderr, "[tc timing] mark %3d hits %10llu total_us %12.1f us_per_hit %8.3f\n", k, h[128 + k], h[k] * 0.01, h[k] * 0.01 / h[128 + k]);
}
#endif
extern "C" void icg_tracker_destroy(icg_tracker *t) {
    if (!t) return;
    if (t->ctx) {
        (void) hipSetDevice(t->ctx->cfg.device);
        (void) hipStreamSynchronize(t->ctx->stream);
    }
    if (t->d_streams) (void) hipFree(t->d_streams);
    if (t->d_buckets) (void) hipFree(t->d_buckets);
    if (t->d_vh) (void) hipFree(t->d_vh);
    if (t->d_arena) (void) hipFree(t->d_arena);
    if (t->h_in) (void) hipHostFree(t->h_in);
    if (t->h_res) (void) hipHostFree(t->h_res);
    if (t->ev_done) (void) hipEventDestroy(t->ev_done);
    delete t;
}

extern "C" int icg_tracker_create(icg_ctx *ctx, int n_streams, const icg_tracker_config *cfg, const uint32_t *buckets_after, int n_buckets_after,
                                  icg_tracker **out) {
    if (!ctx || !cfg || !buckets_after || !out || n_streams <= 0) return ICG_ERR_INVALID;
    *out = nullptr;
#if defined(TC_TIMING)
    static std::once_flag tc_once;
    std::call_once(tc_once, [] { atexit(tc_timing_dump); });
#endif
    if (!ctx->has_cam) return icg_fail(ctx, ICG_ERR_INVALID, "icg_tracker_create: camera not set (icg_set_camera)");
    if (n_buckets_after < tc::MAX_ROWS + 2) return icg_fail(ctx, ICG_ERR_INVALID, "buckets_after must cover %d insertions", tc::MAX_ROWS + 1);
    if ((int) buckets_after[tc::MAX_ROWS + 1] > tc::MAX_BUCKETS) return icg_fail(ctx, ICG_ERR_CAPACITY, "bucket table larger than the block's");
    if (n_streams > ctx->cfg.max_batch) return icg_fail(ctx, ICG_ERR_CAPACITY, "%d streams > max_batch %d", n_streams, ctx->cfg.max_batch);
    if (ctx->cfg.n_slots < tc::MAX_SLOTS * n_streams) return icg_fail(ctx, ICG_ERR_CAPACITY, "the tracker needs %d frame slots per stream", tc::MAX_SLOTS);
    if (cfg->width != ctx->cfg.width || cfg->height != ctx->cfg.height) return icg_fail(ctx, ICG_ERR_INVALID, "tracker image size != context image size");
    if (cfg->block_cnts <= 0 || cfg->block_cnts > tc::MAX_BLOCKS || cfg->block_cnts != cfg->block_cols * cfg->block_rows ||
        cfg->max_per_job + 64 > tc::MAX_ROWS || cfg->window_size + 2 > tc::MAX_WINDOW || cfg->max_per_job != cfg->max_block_features * cfg->block_cnts)
        return icg_fail(ctx, ICG_ERR_CAPACITY, "tracker configuration exceeds the block's capacities");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_tracker *t = new icg_tracker;
    t->ctx         = ctx;
    t->n           = n_streams;
    memcpy(&t->cfg, cfg, sizeof(tc::Cfg));
    t->grid.block_cols = cfg->block_cols, t->grid.block_rows = cfg->block_rows, t->grid.block_w = cfg->block_w, t->grid.block_h = cfg->block_h;
    t->grid.min_dist = cfg->min_pixel_distance, t->grid.max_per_block = cfg->max_block_features;
    auto fail = [&](int rc) {
        icg_tracker_destroy(t);
        return rc;
    };
    const size_t n = (size_t) n_streams, R = tc::MAX_ROWS;
    if (icg_hip_check(ctx, hipMalloc((void **) &t->d_streams, sizeof(tc::Stream) * n), "hipMalloc tracker blocks")) return fail(ICG_ERR_NOMEM);
    if (icg_hip_check(ctx, hipMemsetAsync(t->d_streams, 0, sizeof(tc::Stream) * n, ctx->stream), "memset blocks")) return fail(ICG_ERR_HIP);
    if (icg_hip_check(ctx, hipMalloc((void **) &t->d_buckets, sizeof(uint32_t) * (size_t) n_buckets_after), "hipMalloc buckets")) return fail(ICG_ERR_NOMEM);
    if (icg_hip_check(ctx, hipMemcpy(t->d_buckets, buckets_after, sizeof(uint32_t) * (size_t) n_buckets_after, hipMemcpyHostToDevice), "copy buckets"))
        return fail(ICG_ERR_HIP);
    std::vector<int32_t> vh;
    if (icg_detect_circle_rows(cfg->min_pixel_distance, vh)) return fail(icg_fail(ctx, ICG_ERR_INVALID, "circle span table"));
    if (icg_hip_check(ctx, hipMalloc((void **) &t->d_vh, sizeof(int32_t) * vh.size()), "hipMalloc vh")) return fail(ICG_ERR_NOMEM);
    if (icg_hip_check(ctx, hipMemcpy(t->d_vh, vh.data(), sizeof(int32_t) * vh.size(), hipMemcpyHostToDevice), "copy vh")) return fail(ICG_ERR_HIP);
    // arenas: one allocation, carved (256-byte aligned pieces)
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off             = icg_align_up(off + bytes, 256);
        return at;
    };
    const size_t nblk = (size_t) cfg->block_cnts, mpb = (size_t) cfg->max_block_features;
    struct Piece {
        void **p;
        size_t at;
    };
    std::vector<Piece> pieces;
    auto want = [&](auto **p, size_t count) {
        typedef std::remove_pointer_t<std::remove_pointer_t<decltype(p)>> T;
        pieces.push_back({(void **) p, carve(sizeof(T) * count)});
    };
    TrkArena &A = t->A;
    want(&A.active, n), want(&A.pre_slot, n), want(&A.pre_hist, n);
    want(&A.lk_count, n), want(&A.lk_prev_slot, n * R), want(&A.lk_next_slot, n * R);
    want(&A.lk_prev, n * R), want(&A.lk_guess, n * R), want(&A.lk_out, n * R), want(&A.lk_undist, n * R), want(&A.lk_status, n * R);
    want(&A.rs_count, n), want(&A.rs_p1, n * R), want(&A.rs_p2, n * R), want(&A.rs_mask, n * R);
    want(&A.tri_count, n), want(&A.tri_n_tcw, n), want(&A.tri_T0, n * R), want(&A.tri_T1, n * R);
    want(&A.tri_Tcw, n * tc::MAX_TCW * 12), want(&A.tri_pc0, 3 * n * R), want(&A.tri_pc1, 3 * n * R), want(&A.tri_pw, 3 * n * R);
    want(&A.det_slot, n), want(&A.det_quota, n * tc::MAX_BLOCKS), want(&A.det_mask_begin, n), want(&A.det_mask_count, n), want(&A.det_count, n);
    want(&A.det_mask_pts, n * R), want(&A.det_out, n * R);
    want(&A.work, 4 * n);
    want(&A.rois, n * nblk), want(&A.corners, n * nblk * mpb), want(&A.corner_cnt, n * nblk), want(&A.picks, n * nblk * mpb), want(&A.pick_cnt, n * nblk);
    if (icg_hip_check(ctx, hipMalloc((void **) &t->d_arena, off), "hipMalloc tracker arenas")) return fail(ICG_ERR_NOMEM);
    if (icg_hip_check(ctx, hipMemsetAsync(t->d_arena, 0, off, ctx->stream), "memset arenas")) return fail(ICG_ERR_HIP);
    for (auto &pc : pieces) *pc.p = t->d_arena + pc.at;
    if (icg_hip_check(ctx, hipHostMalloc((void **) &t->h_in, sizeof(TrkInput) * n, hipHostMallocDefault), "hipHostMalloc inputs")) return fail(ICG_ERR_NOMEM);
    if (icg_hip_check(ctx, hipHostMalloc((void **) &t->h_res, sizeof(icg_tracker_result) * n, hipHostMallocDefault), "hipHostMalloc results"))
        return fail(ICG_ERR_NOMEM);
    {
        const char *wm = getenv("ICG_TRACKER_WAIT");
        t->wait_mode   = !wm ? 0 : wm[0] == 'b' ? 2 : wm[0] == 'c' ? 1 : 0;
        if (t->wait_mode == 2)
            if (icg_hip_check(ctx, hipEventCreateWithFlags(&t->ev_done, hipEventBlockingSync | hipEventDisableTiming), "hipEventCreate")) return fail(ICG_ERR_HIP);
    }
    memset(t->h_in, 0, sizeof(TrkInput) * n);
    memset(t->h_res, 0, sizeof(icg_tracker_result) * n);
    hipLaunchKernelGGL(k_trk_init, dim3((n_streams + 63) / 64), dim3(64), 0, ctx->stream, n_streams, t->d_streams);
    if (icg_hip_check(ctx, hipGetLastError(), "k_trk_init")) return fail(ICG_ERR_HIP);
    if (icg_hip_check(ctx, hipStreamSynchronize(ctx->stream), "tracker init")) return fail(ICG_ERR_HIP);
    *out = t;
    return ICG_OK;
}

static int launch_stage(icg_tracker *t, int stage, const char *name) {
    icg_ctx *ctx = t->ctx;
    icg_prof_scope ps(ctx, name);
    hipLaunchKernelGGL(k_trk_stage, dim3((t->n + TRK_WAVES - 1) / TRK_WAVES), dim3(64 * TRK_WAVES), 0, ctx->stream, stage, t->n, t->d_streams, t->A, t->cfg, (const uint32_t *) t->d_buckets, t->h_res,
                       (const TrkInput *) t->h_in);
    ICG_HIP(ctx, hipGetLastError());
    return ICG_OK;
}

extern "C" int icg_tracker_step(icg_tracker *t, const uint8_t *const *images, int stride, int channels, int images_on_device, const double *stamps,
                                const double *poses12, icg_tracker_result *results) {
    if (!t || !images || !stamps || !poses12 || !results) return ICG_ERR_INVALID;
    icg_ctx *ctx = t->ctx;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int n = t->n;
    for (int s = 0; s < n; s++) {
        TrkInput &I = t->h_in[s];
        I.valid     = images[s] ? 1 : 0;
        I.stamp     = stamps[s];
        memcpy(I.pose, poses12 + 12 * (size_t) s, sizeof I.pose);
        I.image = (unsigned long long) (uintptr_t) images[s];
    }
    int rc;
    const TrkArena &A = t->A;
    // without a detection at the start of the frame and without the histogram gate, the prediction needs nothing of the preprocessing:
    // new frame + state machine + prediction in ONE launch, the preprocessing behind it
    const bool fused_begin = !t->need_detect_a && !t->cfg.check_histogram;
    if ((rc = launch_stage(t, fused_begin ? 120 : 0, fused_begin ? "trk_stage_predict" : "trk_begin"))) return rc;
    if ((rc = icg_preprocess_launch_ind(ctx, n, A.pre_slot, images, stride, channels, images_on_device, t->cfg.check_histogram ? A.pre_hist : nullptr))) return rc;
    auto detect = [&]() {
        return icg_detect_launch_ind(ctx, n, &t->grid, A.rois, A.det_slot, A.det_mask_pts, A.det_mask_begin, A.det_mask_count, t->d_vh, A.picks, A.pick_cnt, A.corners, A.corner_cnt);
    };
    if (t->need_detect_a) {
        if ((rc = launch_stage(t, 1, "trk_stage_pre"))) return rc;
        if ((rc = detect())) return rc;
        if ((rc = launch_stage(t, 2, "trk_stage_predict"))) return rc;
    } else if (!fused_begin) {
        if ((rc = launch_stage(t, 12, "trk_stage_predict"))) return rc;
    }
    if ((rc = icg_lk_launch_segments(ctx, n, tc::MAX_ROWS, A.lk_count, A.lk_prev_slot, A.lk_next_slot, A.lk_prev, A.lk_guess, A.lk_out, A.lk_status, A.lk_undist)))
        return rc;
    if ((rc = launch_stage(t, 3, "trk_stage_lk"))) return rc;
    if ((rc = icg_fm_ransac_launch_sets(ctx, n, tc::MAX_ROWS, A.rs_count, A.rs_p1, A.rs_p2, t->cfg.reprojection_error_std, 0.99, A.rs_mask))) return rc;
    if ((rc = launch_stage(t, 4, "trk_stage_ransac"))) return rc;
    if ((rc = icg_triangulate_launch_segments(ctx, n, tc::MAX_ROWS, A.tri_count, A.tri_T0, A.tri_T1, tc::MAX_TCW, A.tri_Tcw, A.tri_pc0, A.tri_pc1, A.tri_pw)))
        return rc;
    if ((rc = launch_stage(t, 5, "trk_stage_tri"))) return rc;
    if ((rc = detect())) return rc;
    if ((rc = launch_stage(t, 6, "trk_stage_end"))) return rc;
    // ONE wait per step.  Default ("adaptive", below): sleep through most of the chain, then poll the stream.  ICG_TRACKER_WAIT=ctx selects
    // the context's wait mode (spin, or query + sleep every poll interval), ICG_TRACKER_WAIT=block an event with hipEventBlockingSync
    if (t->wait_mode == 2) { // "block": an event with hipEventBlockingSync (measured on ROCm 7.2: the runtime still spins — 9.5 cores busy with 12 groups)
        ICG_HIP(ctx, hipEventRecord(t->ev_done, ctx->stream));
        ICG_HIP(ctx, hipEventSynchronize(t->ev_done));
    } else if (t->wait_mode == 1) { // the context's wait mode (spin, or query + sleep every poll interval)
        if ((rc = icg_stream_wait(ctx))) return rc;
    } else {
        // default, "adaptive": the chain of a step takes milliseconds and about as long as the last one did — sleep through most of it
        // (85 % of the running estimate), then query the stream every 100 us.  A group's thread wakes ~10 times per step instead of ~100.
        const auto t_issue = std::chrono::steady_clock::now();
        if (t->step_us_ema > 400.0) {
            struct timespec ts;
            const long ns = (long) (850.0 * t->step_us_ema);
            ts.tv_sec = ns / 1000000000L, ts.tv_nsec = ns % 1000000000L;
            nanosleep(&ts, nullptr);
        }
        for (;;) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return icg_hip_check(ctx, q, "hipStreamQuery");
            struct timespec ts = {0, 100000L};
            nanosleep(&ts, nullptr);
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_issue).count();
        t->step_us_ema  = t->step_us_ema > 0 ? 0.8 * t->step_us_ema + 0.2 * us : us;
    }
    icg_prof_collect(ctx);
    bool need = false;
    int overflow = 0;
    for (int s = 0; s < n; s++) {
        results[s] = t->h_res[s];
        need |= results[s].need_detect_a != 0;
        overflow |= results[s].overflow;
    }
    t->need_detect_a = need;
    if (overflow) return icg_fail(ctx, ICG_ERR_CAPACITY, "tracker block capacity exceeded (flags 0x%x: rows 1, frames 2, map points 4, matrices 8, log 16, window 32, slots 64, internal 128)", overflow);
    return ICG_OK;
}

extern "C" int icg_tracker_download(icg_tracker *t, int stream, void *block) {
    if (!t || !block || stream < 0 || stream >= t->n) return ICG_ERR_INVALID;
    icg_ctx *ctx = t->ctx;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICG_HIP(ctx, hipMemcpy(block, t->d_streams + stream, sizeof(tc::Stream), hipMemcpyDeviceToHost));
    return ICG_OK;
}

extern "C" int icg_tracker_upload(icg_tracker *t, int stream, const void *block) {
    if (!t || !block || stream < 0 || stream >= t->n) return ICG_ERR_INVALID;
    icg_ctx *ctx = t->ctx;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICG_HIP(ctx, hipMemcpy(t->d_streams + stream, block, sizeof(tc::Stream), hipMemcpyHostToDevice));
    t->need_detect_a = true; // (whatever the uploaded state is: the next step takes the general path)
    return ICG_OK;
}

extern "C" int icg_tracker_reset_log(icg_tracker *t, int stream) {
    if (!t || stream < 0 || stream >= t->n) return ICG_ERR_INVALID;
    icg_ctx *ctx = t->ctx;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    hipLaunchKernelGGL(k_trk_reset_log, dim3(1), dim3(1), 0, ctx->stream, t->d_streams + stream);
    ICG_HIP(ctx, hipGetLastError());
    return ICG_OK;
}

extern "C" int icg_tracker_fetch_logs(icg_tracker *t, int n_req, const int32_t *streams, const int32_t *counts, void *out, int entry_stride) {
    if (!t || n_req < 0 || (n_req > 0 && (!streams || !counts || !out)) || entry_stride < 0) return ICG_ERR_INVALID;
    if (n_req == 0) return ICG_OK;
    icg_ctx *ctx = t->ctx;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    for (int k = 0; k < n_req; k++) {
        const int s = streams[k], n = counts[k];
        if (s < 0 || s >= t->n || n < 0 || n > tc::LOG_CAP || n > entry_stride) return icg_fail(ctx, ICG_ERR_INVALID, "icg_tracker_fetch_logs: bad request %d", k);
        if (n > 0)
            ICG_HIP(ctx, hipMemcpyAsync((char *) out + sizeof(tc::LmLog) * (size_t) k * entry_stride, (const char *) (t->d_streams + s) + offsetof(tc::Stream, log),
                                        sizeof(tc::LmLog) * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
        hipLaunchKernelGGL(k_trk_reset_log, dim3(1), dim3(1), 0, ctx->stream, t->d_streams + s);
    }
    ICG_HIP(ctx, hipGetLastError());
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICG_OK;
}
