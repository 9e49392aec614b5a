// Context, memory arenas and per-kernel HIP-event profiling for libicgvins_hip.so.
#include <cstdarg>

#include <sys/prctl.h>
#include <time.h>

#include "icg_internal.h"

int icg_fail(icg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int icg_hip_check(icg_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    return icg_fail(ctx, ICG_ERR_HIP, "HIP error %d (%s) at %s", (int) e, hipGetErrorString(e), what);
}

static thread_local std::string g_create_error;

extern "C" const char *icg_version(void) { return "icgvins-hip 0.1 (gfx950)"; }

extern "C" const char *icg_last_error(const icg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// Completion wait that does not occupy a host core: hipStreamSynchronize (and blocking events) busy-wait on this stack,
// which starves other contexts' threads when there are more contexts than cores.  Query a few times back to back (short
// kernels), then sleep between queries with a tight timer slack.
int icg_stream_wait_poll(icg_ctx *ctx) {
    static thread_local bool slack_set = false;
    if (!slack_set) {
        (void) prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
        slack_set = true;
    }
    struct timespec ts = {0, ctx->poll_sleep_ns};
    for (int it = 0;; it++) {
        hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) return ICG_OK;
        if (e != hipErrorNotReady) return icg_hip_check(ctx, e, "hipStreamQuery");
        if (it >= 4) nanosleep(&ts, nullptr);
    }
}

extern "C" int icg_ctx_set_wait_mode(icg_ctx *ctx, int mode, int sleep_us) {
    if (!ctx || (mode != ICG_WAIT_SPIN && mode != ICG_WAIT_POLL) || (mode == ICG_WAIT_POLL && sleep_us <= 0)) return ICG_ERR_INVALID;
    if (ctx->wait_mode_env) return ICG_OK; // ICG_WAIT_MODE wins
    ctx->poll_sleep_ns = mode == ICG_WAIT_POLL ? 1000L * sleep_us : 0;
    return ICG_OK;
}

icg_pyr_desc icg_make_pyr_desc(const icg_ctx *ctx) {
    icg_pyr_desc d{};
    d.base       = ctx->d_frames;
    d.slot_bytes = ctx->slot_bytes;
    d.n_levels   = ctx->n_levels;
    for (int l = 0; l < ICG_MAX_LEVELS; l++) {
        d.w[l]     = ctx->lv[l].w;
        d.h[l]     = ctx->lv[l].h;
        d.pitch[l] = ctx->lv[l].pitch;
        d.off[l]   = (unsigned int) ctx->lv[l].off;
    }
    return d;
}

extern "C" int icg_ctx_create(const icg_ctx_config *cfg, icg_ctx **out) {
    if (!cfg || !out) return ICG_ERR_INVALID;
    *out = nullptr;
    if (cfg->width < 32 || cfg->height < 32 || cfg->n_slots < 1 || cfg->max_batch < 1 || cfg->max_points < 1) {
        g_create_error = "icg_ctx_create: invalid configuration";
        return ICG_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
        g_create_error = "icg_ctx_create: no HIP device available (this library has no CPU fallback)";
        return ICG_ERR_NODEVICE;
    }
    icg_ctx *ctx = new icg_ctx();
    ctx->cfg     = *cfg;
    int rc       = 0;
    auto bail    = [&](int code) {
        g_create_error = ctx->err;
        icg_ctx_destroy(ctx);
        return code;
    };
    if ((rc = icg_hip_check(ctx, hipSetDevice(cfg->device), "hipSetDevice"))) return bail(rc);
    if ((rc = icg_hip_check(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate")))
        return bail(rc);

    {
        const char *wm = getenv("ICG_WAIT_MODE");
        if (wm && strncmp(wm, "poll", 4) == 0) ctx->poll_sleep_ns = (wm[4] == ':' ? atol(wm + 5) : 20) * 1000L;
        ctx->wait_mode_env = wm != nullptr;
        if (wm && strcmp(wm, "block") == 0 &&
            (rc = icg_hip_check(ctx, hipEventCreateWithFlags(&ctx->ev_wait, hipEventBlockingSync | hipEventDisableTiming), "hipEventCreate")))
            return bail(rc);
    }

    // pyramid geometry: SURVEY.md Appendix B.3 — level l is ((w+1)/2, (h+1)/2); stop when <= win.
    int w = cfg->width, h = cfg->height;
    size_t off    = 0;
    ctx->n_levels = 0;
    for (int l = 0; l < ICG_MAX_LEVELS; l++) {
        if (l > 0) {
            w = (w + 1) / 2;
            h = (h + 1) / 2;
            if (w <= ICG_LK_WIN || h <= ICG_LK_WIN) break;
        }
        ctx->lv[l].w     = w;
        ctx->lv[l].h     = h;
        ctx->lv[l].pitch = (int) icg_align_up((size_t) w, 128);
        ctx->lv[l].off   = off;
        off += icg_align_up((size_t) ctx->lv[l].pitch * h, 256);
        ctx->n_levels++;
    }
    ctx->slot_bytes = off;
    // what k_pyrdown_rows (image.hip) reads a source level with: aligned dwords, 12-byte windows that may end 8 bytes past a row
    for (int l = 0; l + 1 < ctx->n_levels; l++)
        if (ctx->lv[l].pitch % 4 != 0 || ctx->lv[l].off % 4 != 0 || ctx->lv[l].off + (size_t) ctx->lv[l].pitch * ctx->lv[l].h + 8 > ctx->slot_bytes ||
            ctx->lv[l].w < 8 || ctx->lv[l].h < 6 || ctx->lv[l].pitch < 16) {
            ctx->err = "icg_ctx_create: pyramid layout does not meet the row-stream kernel's invariants";
            return bail(ICG_ERR_INVALID);
        }
    ctx->raw_pitch  = ctx->lv[0].pitch;
    size_t tiles2   = (size_t) ICG_CLAHE_TILES * ICG_CLAHE_TILES;
    if ((rc = icg_hip_check(ctx, hipMalloc(&ctx->d_frames, ctx->slot_bytes * (size_t) cfg->n_slots), "hipMalloc frames")))
        return bail(rc);
    if ((rc = icg_hip_check(ctx, hipMalloc(&ctx->d_raw, (size_t) ctx->raw_pitch * cfg->height * cfg->max_batch),
                            "hipMalloc raw")))
        return bail(rc);
    if ((rc = icg_hip_check(ctx, hipMalloc(&ctx->d_lut, tiles2 * 256 * cfg->max_batch), "hipMalloc lut")))
        return bail(rc);
    if ((rc = icg_hip_check(ctx, hipMalloc(&ctx->d_histmean, sizeof(double) * cfg->max_batch), "hipMalloc hist")))
        return bail(rc);
    // staging arena: generous default, grows on demand
    size_t want = (size_t) cfg->max_points * 96 + (size_t) cfg->max_factors * (48 + 16) * 8 + (1u << 20);
    if ((rc = icg_arena_reserve(ctx, want))) return bail(rc);
    *out = ctx;
    return ICG_OK;
}

extern "C" void icg_ctx_destroy(icg_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->cfg.device);
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    for (auto e : ctx->ev_pool) (void) hipEventDestroy(e);
    if (ctx->ev_wait) (void) hipEventDestroy(ctx->ev_wait);
    void *dev[] = {ctx->d_frames, ctx->d_raw,     ctx->d_bgr,  ctx->d_lut,      ctx->d_histmean,
                   ctx->d_roi_max, ctx->d_cand, ctx->d_cand_cnt, ctx->d_arena,    ctx->d_obs,
                   ctx->d_fidx,   ctx->d_rJ,      ctx->d_params,   ctx->d_sys,
                   ctx->d_fwin,   ctx->d_lmwin};
    for (void *p : dev)
        if (p) (void) hipFree(p);
    if (ctx->d_redS) (void) hipFree(ctx->d_redS);
    if (ctx->d_hostS) (void) hipFree(ctx->d_hostS);
    for (icg_partition *pt : {&ctx->part_1, &ctx->part_w}) {
        if (pt->plan.d_buf) (void) hipFree(pt->plan.d_buf);
        if (pt->plan.d_part) (void) hipFree(pt->plan.d_part);
    }

    if (ctx->h_arena) (void) hipHostFree(ctx->h_arena);
    if (ctx->h_fstage) (void) hipHostFree(ctx->h_fstage);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int icg_ctx_sync(icg_ctx *ctx) {
    if (!ctx) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    icg_prof_collect(ctx);
    return ICG_OK;
}

extern "C" void *icg_ctx_stream(icg_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

extern "C" int icg_set_camera(icg_ctx *ctx, const icg_camera *cam) {
    if (!ctx || !cam) return ICG_ERR_INVALID;
    ctx->cam     = *cam;
    ctx->has_cam = true;
    return ICG_OK;
}

extern "C" int icg_pyramid_levels(const icg_ctx *ctx) { return ctx ? ctx->n_levels : 0; }

// ---- arena -------------------------------------------------------------------------------------------------
int icg_arena_reserve(icg_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->arena_cap) return 0;
    if (ctx->arena_off != 0) return icg_fail(ctx, ICG_ERR_NOMEM, "arena grow requested mid-call");
    size_t cap = icg_align_up(bytes + bytes / 2, 1 << 16);
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // only the staging arena pair is replaced here: the resident reduced systems (d_redS) and the packed host parts (d_hostS) are
    // independent allocations that live until icg_ctx_destroy
    if (ctx->h_arena) (void) hipHostFree(ctx->h_arena);
    if (ctx->d_arena) (void) hipFree(ctx->d_arena);
    ctx->h_arena = nullptr;
    ctx->d_arena = nullptr;
    ctx->arena_cap = 0;
    ICG_HIP(ctx, hipHostMalloc((void **) &ctx->h_arena, cap, hipHostMallocDefault));
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_arena, cap));
    ctx->arena_cap = cap;
    return 0;
}

int icg_arena_drain(icg_ctx *ctx) {
    if (ctx->arena_inflight == 0) return 0;
    int rc = icg_hip_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    ctx->arena_inflight = 0;
    ctx->arena_off      = 0;
    return rc;
}

size_t icg_arena_alloc(icg_ctx *ctx, size_t bytes) {
    size_t off = icg_align_up(ctx->arena_off, 256);
    if (off + bytes > ctx->arena_cap) {
        // the caller under-counted its reserve(): never hand out memory past the arena.  The allocation is redirected to
        // offset 0 (in bounds as long as one allocation fits) and the sticky flag fails the call at seal()/finish().
        ctx->arena_overflow = true;
        off                 = 0;
        if (bytes > ctx->arena_cap) abort(); // no in-bounds answer exists
    } else {
        ctx->arena_off = off + bytes;
    }
    return off;
}

int icg_arena_overflow_check(icg_ctx *ctx) {
    if (!ctx->arena_overflow) return 0;
    ctx->arena_overflow = false;
    ctx->arena_off      = 0;
    return icg_fail(ctx, ICG_ERR_NOMEM, "staging arena overflow: the entry point reserved too little");
}

int icg_arena_h2d(icg_ctx *ctx, size_t begin, size_t end) {
    if (end <= begin) return 0;
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_arena + begin, ctx->h_arena + begin, end - begin, hipMemcpyHostToDevice,
                                ctx->stream));
    return 0;
}
int icg_arena_d2h(icg_ctx *ctx, size_t begin, size_t end) {
    if (end <= begin) return 0;
    ICG_HIP(ctx, hipMemcpyAsync(ctx->h_arena + begin, ctx->d_arena + begin, end - begin, hipMemcpyDeviceToHost,
                                ctx->stream));
    return 0;
}

// ---- profiling ---------------------------------------------------------------------------------------------
static hipEvent_t prof_event(icg_ctx *ctx) {
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[ctx->ev_used++];
}

icg_prof_scope::icg_prof_scope(icg_ctx *c, const char *n) : ctx(c), name(n) {
    if (!ctx->prof_on) return;
    a = prof_event(ctx);
    b = prof_event(ctx);
    if (a) (void) hipEventRecord(a, ctx->stream);
}
icg_prof_scope::~icg_prof_scope() {
    if (!ctx->prof_on || !a || !b) return;
    (void) hipEventRecord(b, ctx->stream);
    ctx->prof_pending.push_back({name, a, b});
}

void icg_prof_collect(icg_ctx *ctx) {
    for (auto &p : ctx->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &r = ctx->prof[p.name];
            r.launches++;
            r.total_ms += ms;
        }
    }
    ctx->prof_pending.clear();
    ctx->ev_used = 0;
}

extern "C" int icg_prof_enable(icg_ctx *ctx, int enable) {
    if (!ctx) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    icg_prof_collect(ctx);
    ctx->prof_on = enable != 0;
    ctx->prof.clear();
    return ICG_OK;
}

extern "C" int icg_prof_get(icg_ctx *ctx, const char *kernel_name, int *launches, double *total_ms) {
    if (!ctx || !kernel_name) return ICG_ERR_INVALID;
    auto it = ctx->prof.find(kernel_name);
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.total_ms;
    return ICG_OK;
}

extern "C" int icg_prof_names(icg_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return ICG_ERR_INVALID;
    std::string s;
    for (auto &kv : ctx->prof) {
        s += kv.first;
        s += "\n";
    }
    snprintf(buf, (size_t) buflen, "%s", s.c_str());
    return ICG_OK;
}

// ---- raw device memory helpers -----------------------------------------------------------------------------
extern "C" int icg_dev_alloc(icg_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipMalloc(dptr, bytes));
    return ICG_OK;
}
extern "C" int icg_dev_free(icg_ctx *ctx, void *dptr) {
    if (!ctx) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipFree(dptr));
    return ICG_OK;
}
extern "C" int icg_dev_upload(icg_ctx *ctx, void *dptr, const void *host, size_t bytes) {
    if (!ctx) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice));
    return ICG_OK;
}
extern "C" int icg_dev_download(icg_ctx *ctx, void *host, const void *dptr, size_t bytes) {
    if (!ctx) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICG_HIP(ctx, hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost));
    return ICG_OK;
}
