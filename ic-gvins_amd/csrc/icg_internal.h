// Internal definitions shared by the HIP translation units of libicgvins_hip.so (gfx950 only).
#pragma once
#include <utility>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/icgvins_hip.h"

#define ICG_MAX_LEVELS 4
#define ICG_LK_WIN 21
#define ICG_LK_HALF 10
#define ICG_CLAHE_TILES 21

struct icg_level {
    int w, h, pitch;
    size_t off; // byte offset inside a slot
};

struct icg_prof_rec {
    int launches  = 0;
    double total_ms = 0;
};

// Deterministic assembly plan of a partition of the resident reprojection factors (reproj.hip).  The order in which every sum of the
// normal equations is formed is a function of the window's own factor list only — not of launch geometry, batch composition or timing:
//   runs    factors of a window grouped by their ordered (reference pose, observer pose) pair, list order kept inside a run; every run
//           is reduced by one wave into the 20 x 20 block [Ji Jj Je Jtd -r]^T [Ji Jj Je Jtd -r] (k_asm_runs);
//   lrec    factors of a window grouped by landmark, list order kept; the landmark rows (G_l, h_ll, b_l) are gathered from them.
struct icg_asm_plan {
    int n_runs = 0, Kmax = 1;
    std::vector<int32_t> run_off;   // W+1: window w owns runs [run_off[w], run_off[w+1])
    std::vector<int32_t> pose_off;  // W+1 into pose_glob
    std::vector<int32_t> pose_glob; // local pose number -> global pose index (ascending inside a window)
    char *d_buf = nullptr;          // one allocation for the five index arrays below
    size_t buf_cap = 0;
    int32_t *d_perm = nullptr;      // n: factor indices, run-major
    int32_t *d_runs = nullptr;      // n_runs x 4: first (into d_perm), count, local_i | local_j << 16, window
    int32_t *d_run_off = nullptr;   // W+1
    int32_t *d_pair_run = nullptr;  // W x Kmax^2: run of the ordered local pose pair, -1 = none
    int32_t *d_lrec = nullptr;      // n x 4: factor, local_i, local_j, 0 — landmark-major
    int32_t *d_lm_foff = nullptr;   // n_lm+1: landmark l owns d_lrec[lm_foff[l] .. lm_foff[l+1])
    double *d_part = nullptr;       // n_runs x 220: the runs' blocks (upper-triangular 2 x 2 tiles)
    size_t part_cap = 0;            // doubles
};
struct icg_partition {
    int W = 0;
    bool plan_valid = false;
    std::vector<int32_t> fac_off, lm_off; // W+1 each
    std::vector<int64_t> sys_off;         // W+1: start of window w's (H | b | inv) block inside d_sys, in doubles
    std::vector<double> damp;             // damping that went into each window's inv
    int sys_P = 0, sys_valid = 0;
    icg_asm_plan plan;
};

struct icg_ctx {
    icg_ctx_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    // completion wait of a call: spin on the stream (lowest latency) or sleep on a blocking event (frees the host core for
    // other contexts' threads; ICG_WAIT_MODE=block)
    hipEvent_t ev_wait = nullptr;
    bool wait_mode_env = false;
    long poll_sleep_ns = 0; // > 0: query the stream and sleep in between (ICG_WAIT_MODE=poll[:<us>])

    // frame slots: CLAHE image + LK pyramid, u8, per-level pitch
    int n_levels = 0;
    icg_level lv[ICG_MAX_LEVELS]{};
    size_t slot_bytes = 0;
    uint8_t *d_frames = nullptr;

    // preprocessing workspace (per batch lane)
    int raw_pitch      = 0;
    uint8_t *d_raw     = nullptr; // max_batch x raw_pitch x h  gray input
    uint8_t *d_bgr     = nullptr; // lazily: max_batch x w*3 x h
    uint8_t *d_lut     = nullptr; // max_batch x tiles^2 x 256
    double *d_histmean = nullptr; // max_batch

    // detection workspace (lazily allocated)
    uint32_t *d_roi_max   = nullptr;
    unsigned long long *d_cand = nullptr;
    int32_t *d_cand_cnt   = nullptr;
    int roi_state_cap     = 0;       // entries of d_roi_max / d_cand_cnt (zero between calls: k_select clears what it consumed)
    size_t cand_cap_per_roi = 0;

    // mirrored staging arena (pinned host <-> device), bump-allocated per call
    char *h_arena = nullptr;
    char *d_arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    size_t arena_inflight = 0; // end of the inputs staged by calls that returned without waiting (icg_call::finish_async): the next call stages behind them
    bool arena_overflow = false; // an allocation did not fit (sticky until the call's seal()/finish() reports it)

    // reprojection back-end resident state
    double *d_obs = nullptr;
    int32_t *d_fidx = nullptr; // 3 x n
    int n_factors_resident = 0, factors_cap = 0;
    double *d_rJ = nullptr; // n x 48 packed results (r2 + J46)
    int rJ_valid = 0, rJ_has_jac = 0;
    double *d_params = nullptr; // poses, ext, invdepth, td (device copy)
    size_t params_cap = 0;
    int last_n_poses = 0, last_n_lm = 0;
    double last_huber = 0.0;
    std::vector<int32_t> h_fidx; // host copy of d_fidx (3 x n): the assembly plans below are derived from it
    // pinned host memory a caller fills with a factor set before icg_reproj_commit_factors (icg_reproj_stage_factors): obs (15 x n doubles) | idx (3 x n)
    char *h_fstage = nullptr;
    size_t fstage_cap = 0;
    int fstage_n = -1;
    // f1: resident normal equations of the visual factors.  One window: H ((P+L)^2) | b | 1/(h_ll + damping) per landmark; a partition of
    // the factors into W windows (icg_reproj_set_windows) keeps one such block per window.  part_1 is the implicit partition "all resident
    // factors are one window" behind the single-window entry points — both run through the SAME kernels, so a window's sums are formed in
    // the same order alone and inside a batch.  d_sys holds the systems of whichever partition was assembled last.
    double *d_sys = nullptr;
    size_t sys_cap = 0; // doubles
    double sys_min_diag = 0.0, sys_max_diag = 0.0;
    icg_partition part_1, part_w;
    int32_t *d_fwin = nullptr;  // window of every factor (factors_cap), icg_reproj_eval_windows
    int32_t *d_lmwin = nullptr; // window of every landmark
    int lmwin_cap = 0;
    // f1, reduced systems solved on the device (icg_reproj_schur_windows_resident / _set_host_part_windows / _solve_backsub_windows):
    double *d_redS = nullptr;  // W x P x P, lower tiles written by k_schur_reduce_w
    size_t redS_cap = 0;       // doubles
    double *d_hostS = nullptr; // W x P(P+1)/2: packed lower triangle of every window's host-factor contribution (zero until set)
    size_t hostS_cap = 0;
    int hostS_P = 0, hostS_W = 0;


    icg_camera cam{};
    bool has_cam = false;

    // profiling
    bool prof_on = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct pending {
        std::string name;
        hipEvent_t a, b;
    };
    std::vector<pending> prof_pending;
    std::map<std::string, icg_prof_rec> prof;
};

int icg_fail(icg_ctx *ctx, int code, const char *fmt, ...);
int icg_hip_check(icg_ctx *ctx, hipError_t e, const char *what);
#define ICG_HIP(ctx, call)                                                                                             \
    do {                                                                                                               \
        int _rc = icg_hip_check((ctx), (call), #call);                                                                 \
        if (_rc) return _rc;                                                                                           \
    } while (0)

// arena ---------------------------------------------------------------------------------------------------
int icg_arena_reserve(icg_ctx *ctx, size_t bytes); // ensure capacity (may reallocate; only when arena_off == 0)
static inline size_t icg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// returns offset; host pointer = h_arena+off, device pointer = d_arena+off
size_t icg_arena_alloc(icg_ctx *ctx, size_t bytes);
template <typename T> static inline T *icg_h(icg_ctx *ctx, size_t off) { return reinterpret_cast<T *>(ctx->h_arena + off); }
template <typename T> static inline T *icg_d(icg_ctx *ctx, size_t off) { return reinterpret_cast<T *>(ctx->d_arena + off); }
int icg_arena_overflow_check(icg_ctx *ctx); // ICG_ERR_NOMEM (and the flag cleared) if an allocation since the last check did not fit
int icg_arena_h2d(icg_ctx *ctx, size_t begin, size_t end);
int icg_arena_d2h(icg_ctx *ctx, size_t begin, size_t end);
int icg_arena_drain(icg_ctx *ctx); // waits for calls that returned without waiting (arena_inflight) before the arena is reset or replaced

// profiling -----------------------------------------------------------------------------------------------
struct icg_prof_scope {
    icg_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    icg_prof_scope(icg_ctx *c, const char *n);
    ~icg_prof_scope();
};
void icg_prof_collect(icg_ctx *ctx); // call after stream sync

// slot helpers ---------------------------------------------------------------------------------------------
struct icg_pyr_desc { // passed by value to kernels
    uint8_t *base;    // d_frames
    unsigned long long slot_bytes;
    int n_levels;
    int w[ICG_MAX_LEVELS], h[ICG_MAX_LEVELS], pitch[ICG_MAX_LEVELS];
    unsigned int off[ICG_MAX_LEVELS];
};
icg_pyr_desc icg_make_pyr_desc(const icg_ctx *ctx);

// XCD-aware work mapping.  MI355X dispatches consecutive workgroups round-robin over its 8 XCDs, each with a private 4 MB
// L2.  Work items that share data (the LK points of one camera stream read the same two pyramids) are contiguous in the
// batch, so handing workgroup b the item  (b % 8) * ceil(n/8) + b / 8  gives every XCD one contiguous eighth of the batch
// (one stream's images per L2 instead of all of them in every L2).
// Launch 8*ceil(n/8) workgroups; items >= n exit.  A pure permutation: placement only, results unchanged.
#ifdef __HIPCC__
__device__ static inline int icg_xcd_chunked(int b, int n) { return (b & 7) * ((n + 7) >> 3) + (b >> 3); }
#endif
static inline int icg_xcd_grid(int n) { return 8 * ((n + 7) >> 3); }
// b / d for workgroup-index decoding without an integer division per thread: q = mulhi(b, ceil(2^32 / d)), exact for
// b < 2^32 / d (workgroup counts are far below that)
static inline unsigned int icg_div_magic(int d) { return (unsigned int) ((0x100000000ull + (unsigned int) d - 1u) / (unsigned int) d); }
#ifdef __HIPCC__
__device__ static inline int icg_div_by_magic(int b, unsigned int magic) { return (int) __umulhi((unsigned int) b, magic); }
#endif

#ifdef __HIPCC__
// sqrtf for x = 0 or x >= 2^-96 (no denormal argument or result): v_sqrt_f32 and the two residual tests of the compiler's own expansion of
// sqrtf — candidates r - 1 ulp, r, r + 1 ulp by the signs of the fused residuals x - c r — without the range scaling in front of it and behind
// it (5 of its 16 instructions).  Correctly rounded like the library's; the callers state why their argument is never a small non-zero value.
// (x = 0: r = 0, the pattern below it is a NaN whose test is false, the residual of the one above it is 0: the result is 0.)
__device__ static inline float icg_sqrt_unscaled(float x) {
    const float r  = __builtin_amdgcn_sqrtf(x);
    const float rm = __uint_as_float(__float_as_uint(r) - 1u), rp = __uint_as_float(__float_as_uint(r) + 1u);
    float q        = __builtin_fmaf(-rm, r, x) <= 0.f ? rm : r;
    q              = __builtin_fmaf(-rp, r, x) > 0.f ? rp : q;
    return q;
}
#endif

// single reflection, branch-free: valid for -n < i < 2n-1 (all stencil halos here overshoot by a few pixels at most)
__host__ __device__ static inline int icg_reflect1(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__host__ __device__ static inline int icg_reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

int icg_stream_wait_poll(icg_ctx *ctx);
static inline int icg_stream_wait(icg_ctx *ctx) {
    if (ctx->poll_sleep_ns > 0) return icg_stream_wait_poll(ctx);
    if (ctx->ev_wait) {
        int rc = icg_hip_check(ctx, hipEventRecord(ctx->ev_wait, ctx->stream), "hipEventRecord");
        if (rc) return rc;
        return icg_hip_check(ctx, hipEventSynchronize(ctx->ev_wait), "hipEventSynchronize");
    }
    return icg_hip_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

// Per-call staging helper.  The arena is pinned host memory that the GPU can address directly (hipHostMalloc), mirrored
// by a device arena at the same offsets:
//   in()/out()       mirrored: ONE H2D copy at seal(), ONE D2H copy at finish() — for data many workgroups re-read
//   in_zc()/out_zc() zero-copy: kernels read/write the pinned host memory over PCIe — for data touched once per
//                    lane/wave (point lists, status bytes); saves the copy API calls, which dominate at small batches
#define ICG_LAUNCH_GUARD(call)                \
    do {                                      \
        const int rc_guard_ = (call).outputs_done(); \
        if (rc_guard_) return rc_guard_;      \
    } while (0)

struct icg_call {
    icg_ctx *ctx;
    size_t mirror_lo = (size_t) -1, mirror_hi = 0;
    struct outrec {
        void *user;
        size_t off, bytes;
        bool zc;
    };
    std::vector<outrec> outs;
    std::vector<std::pair<size_t, size_t>> zc_regions; // [begin, end) of every zero-copy output: written by kernels through the host mapping
    explicit icg_call(icg_ctx *c) : ctx(c) { ctx->arena_off = ctx->arena_inflight; }
    int reserve(size_t bytes) {
        if (ctx->arena_inflight + bytes + 8192 <= ctx->arena_cap) return 0;
        if (int rc = icg_arena_drain(ctx)) return rc; // (growing replaces the arena: nothing may still be read from it)
        return icg_arena_reserve(ctx, bytes + 8192);
    }
    template <typename T> T *in(const T *src, size_t n) {
        size_t off = icg_arena_alloc(ctx, sizeof(T) * n);
        if (n) memcpy(ctx->h_arena + off, src, sizeof(T) * n);
        if (off < mirror_lo) mirror_lo = off;
        if (off + sizeof(T) * n > mirror_hi) mirror_hi = off + sizeof(T) * n;
        return reinterpret_cast<T *>(ctx->d_arena + off);
    }
    template <typename T> T *in_zc(const T *src, size_t n) {
        size_t off = icg_arena_alloc(ctx, sizeof(T) * n);
        if (n) memcpy(ctx->h_arena + off, src, sizeof(T) * n);
        return reinterpret_cast<T *>(ctx->h_arena + off);
    }
    // mirrored in-place block: uploaded at seal(), the same bytes copied back to `user` at finish() (pass user = nullptr to the
    // second argument to keep the result on the host side private: accumulators that the caller reads through another pointer)
    template <typename T> T *inout(const T *src, T *user, size_t n) {
        T *d = in(src, n);
        if (user) outs.push_back({(void *) user, (size_t) (reinterpret_cast<char *>(d) - ctx->d_arena), sizeof(T) * n, false});
        return d;
    }
    int overflowed() { return icg_arena_overflow_check(ctx); }
    // after the last in()/out()/out_zc() and BEFORE the first launch: an allocation that did not fit was redirected to offset 0, i.e. it
    // aliases the inputs staged there — no kernel may run on that (ICG_LAUNCH_GUARD returns ICG_ERR_NOMEM instead)
    int outputs_done() { return ctx->arena_overflow ? overflowed() : 0; }
    int seal() {
        if (ctx->arena_overflow) return overflowed();
        return mirror_hi > mirror_lo ? icg_arena_h2d(ctx, mirror_lo, mirror_hi) : 0;
    }
    template <typename T> T *out(T *user, size_t n) {
        size_t off = icg_arena_alloc(ctx, sizeof(T) * n);
        if (user) outs.push_back({(void *) user, off, sizeof(T) * n, false});
        return reinterpret_cast<T *>(ctx->d_arena + off);
    }
    template <typename T> T *out_zc(T *user, size_t n) {
        size_t off = icg_arena_alloc(ctx, sizeof(T) * n);
        if (user) outs.push_back({(void *) user, off, sizeof(T) * n, true});
        zc_regions.push_back({off, off + sizeof(T) * n});
        return reinterpret_cast<T *>(ctx->h_arena + off);
    }
    int finish() {
        int rc    = 0;
        if (ctx->arena_overflow) {
            (void) icg_stream_wait(ctx);
            return overflowed();
        }
        size_t lo = (size_t) -1, hi = 0;
        for (auto &o : outs)
            if (!o.zc) {
                if (o.off < lo) lo = o.off;
                if (o.off + o.bytes > hi) hi = o.off + o.bytes;
            }
        // one copy for the span of all mirrored outputs — unless a zero-copy region lies inside that span: the copy would overwrite what
        // the kernels wrote there through the host mapping with stale device-arena bytes; then every mirrored output is copied on its own
        bool spans_zc = false;
        for (auto &z : zc_regions) spans_zc |= z.first < hi && z.second > lo;
        if (hi > lo && !spans_zc) {
            rc = icg_arena_d2h(ctx, lo, hi);
        } else if (hi > lo) {
            for (auto &o : outs)
                if (!o.zc && !rc) rc = icg_arena_d2h(ctx, o.off, o.off + o.bytes);
        }
        if (rc) return rc;
        rc = icg_stream_wait(ctx);
        if (rc) return rc;
        icg_prof_collect(ctx);
        for (auto &o : outs) memcpy(o.user, ctx->h_arena + o.off, o.bytes);
        ctx->arena_off = ctx->arena_inflight = 0;
        return 0;
    }
    // A call without outputs (results stay resident: icg_reproj_eval_windows) returns as soon as its copies and kernels are enqueued — the
    // next call on the context is stream-ordered behind them and stages its inputs BEHIND this call's, which the H2D copy may still be
    // reading; the first call that waits (finish()) releases the arena.  Saves a host-device round trip per call (2 of 5 per LM step).
    int finish_async() {
        if (ctx->arena_overflow || !outs.empty() || !zc_regions.empty()) return finish();
        ctx->arena_inflight = icg_align_up(ctx->arena_off, 256);
        ctx->arena_off      = ctx->arena_inflight;
        return 0;
    }
};

// ---- device-resident tracker (tracker.hip): segmented / indirect launches of the primitives ----------------------------------------------
// Everything below is asynchronous on the context's stream and takes DEVICE pointers: the stage kernels of the tracker leave the work
// lists (and their lengths) in device memory, so no host round trip sizes a grid or builds a list between two stages.
struct det_roi {
    int job, block, rx, ry, rw, rh, quota, cand_base; // cand_base: offset into the job's candidate plane; quota <= 0: inactive entry
};
int icg_preprocess_launch_ind(icg_ctx *ctx, int n, const int32_t *d_slot_ind, const uint8_t *const *images, int stride, int channels, int src_on_device,
                              double *d_hist_mean);
int icg_lk_launch_segments(icg_ctx *ctx, int n_seg, int seg_cap, const int32_t *d_count, const int32_t *d_prev_slot, const int32_t *d_next_slot,
                           const float2 *d_prev, const float2 *d_guess, float2 *d_out, uint8_t *d_status, float2 *d_undist);
int icg_fm_ransac_launch_sets(icg_ctx *ctx, int n_sets, int seg_cap, const int32_t *d_count, const float2 *d_p1, const float2 *d_p2, double thresh,
                              double conf, uint8_t *d_mask);
int icg_triangulate_launch_segments(icg_ctx *ctx, int n_seg, int seg_cap, const int32_t *d_count, const int32_t *d_T0, const int32_t *d_T1, int tcw_cap,
                                    const double *d_Tcw, const double *d_pc0, const double *d_pc1, double *d_pw);
int icg_detect_circle_rows(int radius, std::vector<int32_t> &vh); // vh[a] = rows of a disc column at distance a (detect.hip); -1 on failure
int icg_detect_launch_ind(icg_ctx *ctx, int n_jobs, const icg_detect_grid *grid, const void *d_rois, const int32_t *d_slots, const float2 *d_mask_pts,
                          const int32_t *d_mask_begin, const int32_t *d_mask_cnt, const int32_t *d_vh, float2 *d_picks, int32_t *d_pick_cnt,
                          float2 *d_corners, int32_t *d_corner_cnt);
