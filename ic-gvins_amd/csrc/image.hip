// F1: Tracking::preprocessing on device — BGR->gray, CLAHE(3.0, 21x21) and the LK image pyramid.
//
// Reference call sites: tracking/tracking.cc:107-142 (preprocessing), :63 (createCLAHE(3.0, Size(21,21))),
// :385-393 (calcOpticalFlowPyrLK rebuilds the pyramid of both images on every call; here it is built once per
// frame and kept resident in the frame slot for the two frames it is used in).
// Arithmetic definitions: SURVEY.md Appendix B.1 (gray, exact), B.2 (CLAHE: integer histogram/clip/redistribute,
// float LUT scale and float bilinear LUT blend in OpenCV's association order), B.3 (pyrDown, exact integer).
//
// Kernels are HBM-bound u8 stencils/histograms (no MFMA shape here):
//   k_clahe_lut   one workgroup per (tile, frame): LDS histogram with reflect-101 padded reads, clip,
//                 redistribute, 256-bin block scan -> LUT (441 x 256 B per frame, stays in L2)
//   k_clahe_apply one workgroup per (interpolation strip, 256-px chunk, frame): the <=2x8 LUTs the chunk needs are
//                 staged in LDS, rows are read/written as coalesced uchar4 per lane
//   k_pyrdown     64x16 output tile per workgroup, (2*64+3)x(2*16+3) input tile staged in LDS, separable
//                 [1 4 6 4 1] in exact integers
// Algorithmic bytes per frame: CLAHE 3*W*H, pyramid 1.640625*W*H.
#include <algorithm>

#include "icg_internal.h"

// per-launch job list passed BY VALUE in the kernel arguments (no staging copy, no host sync needed)
#define PRE_MAX_JOBS 64
struct pre_jobs {
    const uint8_t *src[PRE_MAX_JOBS]; // gray source image per job (device memory)
    int32_t slot[PRE_MAX_JOBS];       // destination frame slot per job
    int stride;                       // source row stride in bytes
};

// ---------------------------------------------------------------------------------------------------------
__global__ void k_bgr2gray(const uint8_t *bgr, int w, int h, int sstride, size_t sbatch, uint8_t *gray, int gstride,
                           size_t gbatch) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    int b = blockIdx.z;
    if (x >= w) return;
    const uint8_t *p = bgr + (size_t) b * sbatch + (size_t) y * sstride + 3 * x;
    gray[(size_t) b * gbatch + (size_t) y * gstride + x] =
        (uint8_t) ((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
}

__global__ void k_hist256(pre_jobs jobs, int w, int h, unsigned int *hist /*n x 256*/) {
    __shared__ unsigned int sh[256];
    int b = blockIdx.y;
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *src = jobs.src[b];
    const int stride   = jobs.stride;
    int total          = w * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int y = i / w, x = i - y * w;
        atomicAdd(&sh[src[(size_t) y * stride + x]], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[(size_t) b * 256 + threadIdx.x], sh[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
struct clahe_geom {
    int w, h, tw, th, clip;
    float lut_scale, inv_tw, inv_th;
};

__global__ __launch_bounds__(256) void k_clahe_lut(pre_jobs jobs, clahe_geom g, uint8_t *lut /* n x tiles^2 x 256 */) {
    __shared__ int hist[256];
    __shared__ int scan[2][256];
    __shared__ int red[256];
    const int t    = threadIdx.x;
    const int tile = blockIdx.x;
    const int b    = blockIdx.y;
    const int ty = tile / ICG_CLAHE_TILES, tx = tile - ty * ICG_CLAHE_TILES;
    const uint8_t *src = jobs.src[b];
    const int stride   = jobs.stride;
    hist[t] = 0;
    __syncthreads();
    const int area = g.tw * g.th;
    for (int i = t; i < area; i += 256) {
        int yy = i / g.tw, xx = i - yy * g.tw;
        int sx = icg_reflect101(tx * g.tw + xx, g.w);
        int sy = icg_reflect101(ty * g.th + yy, g.h);
        atomicAdd(&hist[src[(size_t) sy * stride + sx]], 1);
    }
    __syncthreads();
    int hv     = hist[t];
    int excess = hv > g.clip ? hv - g.clip : 0;
    if (hv > g.clip) hv = g.clip;
    red[t] = excess;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    const int clipped  = red[0];
    const int batch_add = clipped / 256;
    const int residual = clipped - batch_add * 256;
    hv += batch_add;
    if (residual != 0) {
        int step = 256 / residual;
        if (step < 1) step = 1;
        if ((t % step) == 0 && (t / step) < residual) hv++;
    }
    // inclusive scan over 256 bins (Hillis-Steele, double buffered)
    int cur = 0;
    scan[0][t] = hv;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int v = scan[cur][t];
        if (t >= d) v += scan[cur][t - d];
        scan[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    int sum = scan[cur][t];
    float f = rintf((float) sum * g.lut_scale);
    int iv  = (int) f;
    iv      = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
    lut[((size_t) b * ICG_CLAHE_TILES * ICG_CLAHE_TILES + tile) * 256 + t] = (uint8_t) iv;
}

#define CLAHE_CHUNK 256 // pixels per workgroup row segment (64 lanes x uchar4)
#define CLAHE_MAXCOLS 24

__global__ __launch_bounds__(256) void k_clahe_apply(pre_jobs jobs, clahe_geom g, const uint8_t *lut, uint8_t *frames,
                                                     size_t slot_bytes, int dpitch) {
    __shared__ uint8_t slut[2][CLAHE_MAXCOLS][256];
    const int strip = blockIdx.x; // ty1_raw = strip-1
    const int chunk = blockIdx.y;
    const int b     = blockIdx.z;
    const int t     = threadIdx.x;
    const int T     = ICG_CLAHE_TILES;

    int ty1 = strip - 1, ty2 = strip;
    if (ty1 < 0) ty1 = 0;
    if (ty2 > T - 1) ty2 = T - 1;

    const int x_begin = chunk * CLAHE_CHUNK;
    int x_end         = x_begin + CLAHE_CHUNK;
    if (x_end > g.w) x_end = g.w;
    // tile-column range touched by this chunk
    int c_lo = (int) floorf(x_begin * g.inv_tw - 0.5f);
    int c_hi = (int) floorf((x_end - 1) * g.inv_tw - 0.5f) + 1;
    if (c_lo < 0) c_lo = 0;
    if (c_hi > T - 1) c_hi = T - 1;
    const int ncols = c_hi - c_lo + 1; // <= CLAHE_MAXCOLS by construction of the launch (checked on host)

    const uint8_t *blut = lut + (size_t) b * T * T * 256;
    for (int i = t; i < 2 * ncols * 256; i += 256) {
        int r   = i / (ncols * 256);
        int rem = i - r * ncols * 256;
        int c   = rem >> 8;
        int v   = rem & 255;
        slut[r][c][v] = blut[((size_t) (r == 0 ? ty1 : ty2) * T + (c_lo + c)) * 256 + v];
    }
    __syncthreads();

    const int wave = t >> 6, lane = t & 63;
    const int x0 = x_begin + lane * 4;
    int c1[4], c2[4];
    float xa[4], xa1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int x     = x0 + k;
        float txf = x * g.inv_tw - 0.5f;
        int tx1   = (int) floorf(txf);
        int tx2   = tx1 + 1;
        xa[k]     = txf - tx1;
        xa1[k]    = 1.0f - xa[k];
        if (tx1 < 0) tx1 = 0;
        if (tx2 > T - 1) tx2 = T - 1;
        c1[k] = tx1 - c_lo;
        c2[k] = tx2 - c_lo;
        if (c1[k] < 0) c1[k] = 0;
        if (c2[k] > ncols - 1) c2[k] = ncols - 1;
        if (c1[k] > ncols - 1) c1[k] = ncols - 1;
    }
    if (x0 >= g.w) return;

    // candidate rows of this strip: floor(y*inv_th - 0.5) == strip-1
    int y_lo = (strip - 1) * g.th + g.th / 2 - 2;
    int y_hi = strip * g.th + g.th / 2 + 3;
    if (y_lo < 0) y_lo = 0;
    if (y_hi > g.h) y_hi = g.h;
    const uint8_t *src = jobs.src[b];
    const int stride   = jobs.stride;
    uint8_t *dst       = frames + (size_t) jobs.slot[b] * slot_bytes;
    for (int y = y_lo + wave; y < y_hi; y += 4) {
        float tyf = y * g.inv_th - 0.5f;
        int tyr   = (int) floorf(tyf);
        if (tyr != strip - 1) continue;
        float ya = tyf - tyr, ya1 = 1.0f - ya;
        uchar4 p = *reinterpret_cast<const uchar4 *>(src + (size_t) y * stride + x0);
        unsigned char in[4] = {p.x, p.y, p.z, p.w};
        unsigned char out[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int v     = in[k];
            float l11 = slut[0][c1[k]][v], l12 = slut[0][c2[k]][v];
            float l21 = slut[1][c1[k]][v], l22 = slut[1][c2[k]][v];
            float res = (l11 * xa1[k] + l12 * xa[k]) * ya1 + (l21 * xa1[k] + l22 * xa[k]) * ya;
            int iv    = (int) rintf(res);
            out[k]    = (unsigned char) (iv < 0 ? 0 : (iv > 255 ? 255 : iv));
        }
        *reinterpret_cast<uchar4 *>(dst + (size_t) y * dpitch + x0) = make_uchar4(out[0], out[1], out[2], out[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------
#define PD_TW 64
#define PD_TH 16
#define PD_IW (2 * PD_TW + 3)
#define PD_IH (2 * PD_TH + 3)

__global__ __launch_bounds__(256) void k_pyrdown(uint8_t *frames, size_t slot_bytes, pre_jobs jobs,
                                                 unsigned int src_off, int sw, int sh, int spitch,
                                                 unsigned int dst_off, int dw, int dh, int dpitch) {
    __shared__ uint8_t in[PD_IH][PD_IW + 1];
    __shared__ int tmp[PD_IH][PD_TW];
    const int b      = blockIdx.z;
    uint8_t *slot    = frames + (size_t) jobs.slot[b] * slot_bytes;
    const uint8_t *s = slot + src_off;
    uint8_t *d       = slot + dst_off;
    const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH; // output tile origin
    const int ix0 = 2 * ox - 2, iy0 = 2 * oy - 2;
    const int t = threadIdx.x;
    for (int i = t; i < PD_IH * PD_IW; i += 256) {
        int r = i / PD_IW, c = i - r * PD_IW;
        int sx = icg_reflect101(ix0 + c, sw), sy = icg_reflect101(iy0 + r, sh);
        in[r][c] = s[(size_t) sy * spitch + sx];
    }
    __syncthreads();
    for (int i = t; i < PD_IH * PD_TW; i += 256) {
        int r = i / PD_TW, c = i - r * PD_TW;
        const uint8_t *p = &in[r][2 * c];
        tmp[r][c]        = p[0] + p[4] + 4 * (p[1] + p[3]) + 6 * p[2];
    }
    __syncthreads();
    for (int i = t; i < PD_TH * PD_TW; i += 256) {
        int r = i / PD_TW, c = i - r * PD_TW;
        int x = ox + c, y = oy + r;
        if (x < dw && y < dh) {
            int v = tmp[2 * r][c] + tmp[2 * r + 4][c] + 4 * (tmp[2 * r + 1][c] + tmp[2 * r + 3][c]) + 6 * tmp[2 * r + 2][c];
            d[(size_t) y * dpitch + x] = (uint8_t) ((v + 128) >> 8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
extern "C" int icg_frames_preprocess(icg_ctx *ctx, int n, const int32_t *slots, const uint8_t *const *images,
                                     int stride, int channels, int src_on_device, double *hist_mean) {
    if (!ctx || n < 0 || (n > 0 && (!slots || !images))) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (n > ctx->cfg.max_batch) return icg_fail(ctx, ICG_ERR_CAPACITY, "preprocess batch %d > max_batch %d", n, ctx->cfg.max_batch);
    if (channels != 1 && channels != 3) return icg_fail(ctx, ICG_ERR_INVALID, "channels must be 1 or 3");
    const int w = ctx->cfg.width, h = ctx->cfg.height;
    if (stride < w * channels) return icg_fail(ctx, ICG_ERR_INVALID, "stride too small");
    for (int k = 0; k < n; k++)
        if (slots[k] < 0 || slots[k] >= ctx->cfg.n_slots || !images[k]) return icg_fail(ctx, ICG_ERR_INVALID, "bad slot/image %d", k);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));

    // CLAHE geometry
    const int T = ICG_CLAHE_TILES;
    int ew = w, eh = h;
    if (w % T != 0 || h % T != 0) {
        ew = w + (T - w % T);
        eh = h + (T - h % T);
    }
    clahe_geom g;
    g.w = w;
    g.h = h;
    g.tw = ew / T;
    g.th = eh / T;
    int area    = g.tw * g.th;
    g.lut_scale = 255.0f / area;
    g.clip      = (int) (3.0 * area / 256);
    if (g.clip < 1) g.clip = 1;
    g.inv_tw = 1.0f / g.tw;
    g.inv_th = 1.0f / g.th;
    if (CLAHE_CHUNK / g.tw + 3 > CLAHE_MAXCOLS)
        return icg_fail(ctx, ICG_ERR_INVALID, "image too small for CLAHE chunking (tile width %d)", g.tw);

    const size_t raw_batch = (size_t) ctx->raw_pitch * h;
    hipMemcpyKind kind     = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    unsigned int *d_hist   = nullptr;
    if (hist_mean) {
        ctx->arena_off = 0;
        int rc         = icg_arena_reserve(ctx, sizeof(unsigned int) * 256 * (size_t) n + 4096);
        if (rc) return rc;
        d_hist = icg_d<unsigned int>(ctx, icg_arena_alloc(ctx, sizeof(unsigned int) * 256 * (size_t) n));
        ICG_HIP(ctx, hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * 256 * (size_t) n, ctx->stream));
    }

    for (int base = 0; base < n; base += PRE_MAX_JOBS) {
        const int m = std::min(PRE_MAX_JOBS, n - base);
        pre_jobs jobs;
        memset(&jobs, 0, sizeof jobs);
        for (int k = 0; k < m; k++) jobs.slot[k] = slots[base + k];
        if (channels == 3) {
            if (!ctx->d_bgr) ICG_HIP(ctx, hipMalloc((void **) &ctx->d_bgr, (size_t) w * 3 * h * ctx->cfg.max_batch));
            for (int k = 0; k < m; k++)
                ICG_HIP(ctx, hipMemcpy2DAsync(ctx->d_bgr + (size_t) (base + k) * w * 3 * h, (size_t) w * 3, images[base + k], stride,
                                              (size_t) w * 3, h, kind, ctx->stream));
            icg_prof_scope ps(ctx, "bgr2gray");
            hipLaunchKernelGGL(k_bgr2gray, dim3((w + 255) / 256, h, m), dim3(256), 0, ctx->stream,
                               ctx->d_bgr + (size_t) base * w * 3 * h, w, h, w * 3, (size_t) w * 3 * h,
                               ctx->d_raw + (size_t) base * raw_batch, ctx->raw_pitch, raw_batch);
            for (int k = 0; k < m; k++) jobs.src[k] = ctx->d_raw + (size_t) (base + k) * raw_batch;
            jobs.stride = ctx->raw_pitch;
        } else {
            // device-resident gray frames are consumed in place when uchar4 loads are aligned; otherwise (and for host
            // frames) they are first brought into the pitched staging planes
            bool direct = src_on_device && (stride % 4 == 0);
            for (int k = 0; k < m && direct; k++) direct = (((uintptr_t) images[base + k]) % 4) == 0;
            if (direct) {
                for (int k = 0; k < m; k++) jobs.src[k] = images[base + k];
                jobs.stride = stride;
            } else {
                for (int k = 0; k < m; k++) {
                    ICG_HIP(ctx, hipMemcpy2DAsync(ctx->d_raw + (size_t) (base + k) * raw_batch, ctx->raw_pitch, images[base + k],
                                                  stride, w, h, kind, ctx->stream));
                    jobs.src[k] = ctx->d_raw + (size_t) (base + k) * raw_batch;
                }
                jobs.stride = ctx->raw_pitch;
            }
        }
        if (hist_mean) {
            icg_prof_scope ps(ctx, "hist256");
            hipLaunchKernelGGL(k_hist256, dim3(64, m), dim3(256), 0, ctx->stream, jobs, w, h, d_hist + (size_t) base * 256);
        }
        uint8_t *lut = ctx->d_lut + (size_t) base * T * T * 256;
        {
            icg_prof_scope ps(ctx, "clahe_lut");
            hipLaunchKernelGGL(k_clahe_lut, dim3(T * T, m), dim3(256), 0, ctx->stream, jobs, g, lut);
        }
        {
            icg_prof_scope ps(ctx, "clahe_apply");
            hipLaunchKernelGGL(k_clahe_apply, dim3(T + 1, (w + CLAHE_CHUNK - 1) / CLAHE_CHUNK, m), dim3(256), 0, ctx->stream, jobs,
                               g, lut, ctx->d_frames, ctx->slot_bytes, ctx->lv[0].pitch);
        }
        for (int l = 1; l < ctx->n_levels; l++) {
            icg_prof_scope ps(ctx, "pyrdown");
            const icg_level &a = ctx->lv[l - 1], &bb = ctx->lv[l];
            hipLaunchKernelGGL(k_pyrdown, dim3((bb.w + PD_TW - 1) / PD_TW, (bb.h + PD_TH - 1) / PD_TH, m), dim3(256), 0,
                               ctx->stream, ctx->d_frames, ctx->slot_bytes, jobs, (unsigned int) a.off, a.w, a.h, a.pitch,
                               (unsigned int) bb.off, bb.w, bb.h, bb.pitch);
        }
    }
    ICG_HIP(ctx, hipGetLastError());
    if (!hist_mean) return ICG_OK; // asynchronous: later calls on this context are stream-ordered behind these kernels

    std::vector<unsigned int> hc((size_t) n * 256);
    ICG_HIP(ctx, hipMemcpyAsync(hc.data(), d_hist, sizeof(unsigned int) * 256 * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    icg_prof_collect(ctx);
    // tracking.cc:98-102: float histogram, (float)k product in float, /256.0 and accumulation in double
    for (int k = 0; k < n; k++) {
        double acc = 0;
        for (int i = 0; i < 256; i++) {
            float hf = (float) hc[(size_t) k * 256 + i];
            acc += hf * (float) i / 256.0;
        }
        hist_mean[k] = acc / ((double) w * h);
    }
    ctx->arena_off = 0;
    return ICG_OK;
}

extern "C" int icg_frame_download(icg_ctx *ctx, int slot, int level, uint8_t *dst, int dst_stride) {
    if (!ctx || !dst || slot < 0 || slot >= ctx->cfg.n_slots || level < 0 || level >= ctx->n_levels) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const icg_level &L = ctx->lv[level];
    if (dst_stride < L.w) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ICG_HIP(ctx, hipMemcpy2D(dst, dst_stride, ctx->d_frames + (size_t) slot * ctx->slot_bytes + L.off, L.pitch, L.w, L.h,
                             hipMemcpyDeviceToHost));
    return ICG_OK;
}
