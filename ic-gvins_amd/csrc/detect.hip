// F7: Tracking::featuresDetection on device — mask discs, gridded Shi-Tomasi (cv::goodFeaturesToTrack) and
// cv::cornerSubPix, batched over frames of many streams.
//
// Reference: tracking/tracking.cc:576-688 (mask :609-620, ROI per block :632-645, GFTT :647, subpix :651).
// Arithmetic definition: SURVEY.md Appendix B.7/B.8 with the formulation pinned in oracle/orc_detect.cc:
// exact-integer Sobel x (1/3060) in float, covariance products in float, 3x3 box sums accumulated in double in raster
// order (reflect-101 at the ROI edge), min-eigenvalue in float; threshold 0.01*max over unmasked ROI pixels; 3x3
// non-maximum test inside the ROI; candidates ordered by (response desc, raster address desc); greedy minimum-distance
// selection; sub-pixel refinement with sequential double accumulation (bit-identical to the CPU restatement).
//
// Kernels (HBM/L2-bound stencils, no MFMA shape):
//   k_mask_discs  one workgroup per existing feature: midpoint-circle span table -> zero spans in the u8 mask
//   k_min_eig     62x8 response tile per wave, a lane per column, separable exact box sums; per-ROI masked maximum by
//                 an order-preserving uint atomicMax
//   k_candidates  threshold + 3x3 NMS + mask, 4 rows per lane -> (key = response bits << 32 | raster index) appended per ROI
//   k_select      one workgroup per ROI: repeated block-wide arg-max over live candidates + min-distance kill
//                 (equivalent to sort + greedy grid test, needs no sort and no capacity cap)
//   k_subpix      one wavefront per corner: 13x13 bilinear patch and the 121 gradient terms in parallel through LDS,
//                 the five 121-term sums each sequentially in raster order (IEEE order == CPU order), one lane per sum
#include <cfloat>

#include "icg_internal.h"

struct det_roi {
    int job, block, rx, ry, rw, rh, quota, cand_base; // cand_base: offset into the job's candidate plane
};

__device__ __forceinline__ unsigned int f32_order_key(float f) {
    unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(unsigned int k) {
    unsigned int b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// ---------------------------------------------------------------------------------------------------------
// The mask plane is GENERATION-TAGGED: a pixel is masked in this call iff mask[p] == gen (gen cycles 1..255 per context;
// the plane is cleared only when gen wraps).  This removes the 0.9 MB-per-frame memset launch of the 255/0 formulation.
// One 64-lane workgroup per existing feature; each lane fills whole rows of the midpoint-circle span table with
// dword stores (head/tail bytes) instead of testing every pixel of the bounding square.
__global__ __launch_bounds__(64) void k_mask_discs(int n_pts, const float2 *pts, const int32_t *pt_job, int radius,
                                                   const int32_t *halfw /*radius+1*/, uint8_t *mask, int pitch, int w,
                                                   int h, size_t plane, unsigned int gen) {
    const int i = blockIdx.x;
    if (i >= n_pts) return;
    const float2 p = pts[i];
    const int cx = (int) rintf(p.x), cy = (int) rintf(p.y);
    uint8_t *m   = mask + (size_t) pt_job[i] * plane;
    const unsigned int g4 = gen * 0x01010101u;
    for (int r = threadIdx.x; r <= 2 * radius; r += 64) {
        const int dy = r - radius, y = cy + dy;
        if (y < 0 || y >= h) continue;
        const int hw = halfw[dy < 0 ? -dy : dy];
        if (hw < 0) continue;
        int x0 = cx - hw, x1 = cx + hw; // inclusive span
        if (x0 < 0) x0 = 0;
        if (x1 > w - 1) x1 = w - 1;
        uint8_t *row = m + (size_t) y * pitch;
        int x = x0;
        for (; x <= x1 && (x & 3); x++) row[x] = (uint8_t) gen;
        for (; x + 3 <= x1; x += 4) *reinterpret_cast<unsigned int *>(row + x) = g4; // pitch % 128 == 0: aligned
        for (; x <= x1; x++) row[x] = (uint8_t) gen;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_min_eig: one WAVE per 62 x 8 response tile, one lane per image column (64 lanes = 62 outputs + 2 halo columns), four
// waves (32 rows) per workgroup.
//   stage 1  each lane walks DOWN its column: 12 image rows -> the 10 Sobel pairs of its halo column with the running
//            differences d[r] = p[r][x+1]-p[r][x-1], s[r] = p[r][x-1]+2p[r][x]+p[r][x+1]  (gx = d[r-1]+2d[r]+d[r+1],
//            gy = s[r+1]-s[r-1]), the 3 float products per entry;  tiles touching the ROI or image border take a
//            per-entry path with explicit reflect-101
//   exchange the 3x10 products of a lane go to LDS once; the left/right neighbours' are read back
//   stage 2  SEPARABLE 3x3 box sums in double: the terms are floats in [s^2, (1020 s)^2] (s = 1/3060, |Sobel| <= 1020), so
//            every partial sum of nine of them is a multiple of 2^-47 below 1 and exactly representable — the raster-order
//            double accumulation of the CPU restatement and the separable one are bit-identical
//   stage 3  min-eigenvalue in float for the lane's 8 pixels, response store, masked per-ROI maximum (order-preserving
//            uint atomicMax, one per wave)
// ~1.7 issued instructions per output pixel instead of 6.7 for the one-pixel-per-lane formulation.
#define EIG_TW 62 // output columns per wave
#define EIG_TH 8  // output rows per wave
#define EIG_WAVES 4

__device__ __forceinline__ void eig_sobel_slow(const uint8_t *img, int pitch, int w, int h, const det_roi &R, int xcol, int yrow,
                                               int &gx, int &gy) {
    // ROI coordinate with reflect-101 at the ROI edge (cov is a fresh ROI-sized Mat in OpenCV); coordinates further out
    // belong to partial tiles and are clamped (their outputs are never stored)
    const int x = icg_reflect1(min(max(xcol, -1), R.rw), R.rw), y = icg_reflect1(min(max(yrow, -1), R.rh), R.rh);
    const int X = R.rx + x, Y = R.ry + y;
    // Sobel on REAL image pixels (peeks outside the ROI); reflect-101 only at true image borders
    const int xm = icg_reflect1(X - 1, w), xp = icg_reflect1(X + 1, w);
    const int ym = icg_reflect1(Y - 1, h), yp = icg_reflect1(Y + 1, h);
    const uint8_t *r0 = img + (size_t) ym * pitch, *r1 = img + (size_t) Y * pitch, *r2 = img + (size_t) yp * pitch;
    const int p00 = r0[xm], p01 = r0[X], p02 = r0[xp];
    const int p10 = r1[xm], p12 = r1[xp];
    const int p20 = r2[xm], p21 = r2[X], p22 = r2[xp];
    gx = (p02 - p00) + 2 * (p12 - p10) + (p22 - p20);
    gy = (p20 - p00) + 2 * (p21 - p01) + (p22 - p02);
}

__global__ __launch_bounds__(64 * EIG_WAVES) void k_min_eig(const det_roi *rois, const uint8_t *frames, size_t slot_bytes,
                                                            const int32_t *slots, int pitch, int w, int h,
                                                            const uint8_t *mask, size_t mask_plane, unsigned int gen, float *eig,
                                                            size_t eig_plane, unsigned int *roi_max, int gx, int gy, int n_blocks,
                                                            unsigned int m_roi, unsigned int m_gx) {
    __shared__ float cov[EIG_WAVES][3][EIG_TH + 2][64];
    // 1-D launch, ROI-major and XCD-chunked: a ROI's blocks (and later its k_candidates blocks) share one XCD's L2
    const int bl = icg_xcd_chunked(blockIdx.x, n_blocks);
    if (bl >= n_blocks) return; // whole workgroup
    const int roi = icg_div_by_magic(bl, m_roi), rem = bl - roi * (gx * gy);
    const int by = icg_div_by_magic(rem, m_gx), bx = rem - by * gx;
    const det_roi R = rois[roi];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tx0 = bx * EIG_TW, ty0 = (by * EIG_WAVES + wv) * EIG_TH;
    const bool live = tx0 < R.rw && ty0 < R.rh; // wave-uniform; dead waves still join the barrier
    const uint8_t *img = frames + (size_t) slots[R.job] * slot_bytes;
    const float s      = (float) (1.0 / 3060.0);
    float cx[EIG_TH + 2], cm[EIG_TH + 2], cy[EIG_TH + 2]; // dx*dx, dx*dy, dy*dy of the lane's halo column
    if (live) {
        // the lane's halo column xcol = tx0-1+lane, halo rows ty0-1 .. ty0+8
        const int xcol = tx0 - 1 + lane;
        const bool interior = tx0 >= 1 && tx0 + EIG_TW <= R.rw - 1 && ty0 >= 1 && ty0 + EIG_TH <= R.rh - 1 && R.rx + tx0 >= 2 &&
                              R.rx + tx0 + EIG_TW + 1 <= w - 1 && R.ry + ty0 >= 2 && R.ry + ty0 + EIG_TH + 1 <= h - 1; // wave-uniform
        int gxv[EIG_TH + 2], gyv[EIG_TH + 2];
        if (interior) {
            typedef unsigned int __attribute__((aligned(1))) u32u;
            const uint8_t *p = img + (size_t) (R.ry + ty0 - 2) * pitch + (R.rx + xcol - 1); // image row of halo row -1, column x-1
            int d[EIG_TH + 4], sm[EIG_TH + 4];
#pragma unroll
            for (int r = 0; r < EIG_TH + 4; r++) {
                const unsigned int v = *reinterpret_cast<const u32u *>(p + (size_t) r * pitch);
                const int a = v & 0xff, b = (v >> 8) & 0xff, c = (v >> 16) & 0xff;
                d[r]  = c - a;
                sm[r] = a + 2 * b + c;
            }
#pragma unroll
            for (int r = 0; r < EIG_TH + 2; r++) {
                gxv[r] = d[r] + 2 * d[r + 1] + d[r + 2];
                gyv[r] = sm[r + 2] - sm[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < EIG_TH + 2; r++) eig_sobel_slow(img, pitch, w, h, R, xcol, ty0 - 1 + r, gxv[r], gyv[r]);
        }
#pragma unroll
        for (int r = 0; r < EIG_TH + 2; r++) {
            const float dx = (float) gxv[r] * s, dy = (float) gyv[r] * s;
            cx[r] = dx * dx;
            cm[r] = dx * dy;
            cy[r] = dy * dy;
            cov[wv][0][r][lane] = cx[r];
            cov[wv][1][r][lane] = cm[r];
            cov[wv][2][r][lane] = cy[r];
        }
    }
    __syncthreads();
    if (!live) return;
    unsigned int key = 0;
    const int x = tx0 + lane - 1; // output column of this lane (lanes 1..62)
    if (lane >= 1 && lane <= EIG_TW && x < R.rw) {
        const int ln = lane - 1, lp = lane + 1;
        double ha[EIG_TH + 2], hb[EIG_TH + 2], hc[EIG_TH + 2]; // horizontal 3-sums per halo row
#pragma unroll
        for (int r = 0; r < EIG_TH + 2; r++) {
            ha[r] = ((double) cov[wv][0][r][ln] + (double) cx[r]) + (double) cov[wv][0][r][lp];
            hb[r] = ((double) cov[wv][1][r][ln] + (double) cm[r]) + (double) cov[wv][1][r][lp];
            hc[r] = ((double) cov[wv][2][r][ln] + (double) cy[r]) + (double) cov[wv][2][r][lp];
        }
        const int X = R.rx + x;
        float *erow         = eig + (size_t) R.job * eig_plane + (size_t) (R.ry + ty0) * w + X;
        const uint8_t *mrow = mask + (size_t) R.job * mask_plane + (size_t) (R.ry + ty0) * pitch + X;
#pragma unroll
        for (int k = 0; k < EIG_TH; k++) {
            if (ty0 + k < R.rh) {
                const double sa = (ha[k] + ha[k + 1]) + ha[k + 2], sb = (hb[k] + hb[k + 1]) + hb[k + 2];
                const double sc = (hc[k] + hc[k + 1]) + hc[k + 2];
                const float a = (float) sa * 0.5f, b = (float) sb, c = (float) sc * 0.5f;
                const float e = (a + c) - sqrtf((a - c) * (a - c) + b * b);
                erow[(size_t) k * w] = e;
                if (mrow[(size_t) k * pitch] != (uint8_t) gen) {
                    const unsigned int ke = f32_order_key(e);
                    key                   = ke > key ? ke : key;
                }
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned int o = __shfl_xor(key, m, 64);
        key                  = o > key ? o : key;
    }
    if (lane == 0 && key) atomicMax(&roi_max[roi], key);
}

// ---------------------------------------------------------------------------------------------------------
// 4 vertically adjacent pixels per lane (one wave = a 64 x 4 strip, one workgroup = 64 x 16): the 6x3 response values a
// lane needs are loaded once for its 4 pixels and every load is a fully coalesced row segment.  OpenCV's test
// "thresholded value equals the 3x3 maximum of the thresholded map" is evaluated as
// v > thresh && v >= every RAW neighbour  (identical: a neighbour at or below the threshold is below v, a neighbour
// above it keeps its value).
#define CAND_PY 4
__global__ __launch_bounds__(256) void k_candidates(const det_roi *rois, int pitch, int w, const uint8_t *mask,
                                                    size_t mask_plane, unsigned int gen, const float *eig, size_t eig_plane,
                                                    const unsigned int *roi_max, unsigned long long *cand,
                                                    size_t cand_plane, int32_t *cand_cnt, int gx, int gy, int n_blocks,
                                                    unsigned int m_roi, unsigned int m_gx) {
    const int bl = icg_xcd_chunked(blockIdx.x, n_blocks); // ROI-major, XCD-chunked like k_min_eig
    if (bl >= n_blocks) return;
    const int roi = icg_div_by_magic(bl, m_roi), rem = bl - roi * (gx * gy);
    const int by = icg_div_by_magic(rem, m_gx), bx = rem - by * gx;
    const det_roi R = rois[roi];
    const int lane = threadIdx.x & 63;
    const int x  = bx * 64 + lane;
    const int y0 = (by * 4 + (threadIdx.x >> 6)) * CAND_PY;
    if (y0 >= R.rh - 1) return; // wave-uniform
    const unsigned int mk = roi_max[roi];
    const double maxVal   = mk ? (double) f32_from_order_key(mk) : 0.0;
    const float thresh    = (float) (maxVal * 0.01);
    bool is_cand[CAND_PY];
    float v[CAND_PY];
#pragma unroll
    for (int k = 0; k < CAND_PY; k++) {
        is_cand[k] = false;
        v[k]       = 0.f;
    }
    if (x >= 1 && x < R.rw - 1) {
        const float *e = eig + (size_t) R.job * eig_plane + (size_t) R.ry * w + (R.rx + x);
        float a[CAND_PY + 2][3];
#pragma unroll
        for (int j = 0; j < CAND_PY + 2; j++) {
            // ROI row y0-1+j; rows outside [0, rh) are never a valid centre's neighbour (clamped load, unused)
            int yj = y0 - 1 + j;
            yj     = yj < 0 ? 0 : (yj > R.rh - 1 ? R.rh - 1 : yj);
            const float *row = e + (size_t) yj * w;
            a[j][0] = row[-1];
            a[j][1] = row[0];
            a[j][2] = row[1];
        }
        const uint8_t *mcol = mask + (size_t) R.job * mask_plane + (size_t) R.ry * pitch + (R.rx + x);
#pragma unroll
        for (int k = 0; k < CAND_PY; k++) {
            const int y = y0 + k;
            if (y < 1 || y >= R.rh - 1) continue;
            const float c = a[k + 1][1];
            if (!(c > thresh) || c == 0.f) continue; // THRESH_TOZERO leaves 0 below the threshold, and 0 is never a corner
            float mx = fmaxf(fmaxf(a[k][0], a[k][1]), a[k][2]);
            mx       = fmaxf(mx, fmaxf(a[k + 1][0], a[k + 1][2]));
            mx       = fmaxf(mx, fmaxf(fmaxf(a[k + 2][0], a[k + 2][1]), a[k + 2][2]));
            v[k]       = c;
            is_cand[k] = (c >= mx) && mcol[(size_t) y * pitch] != (uint8_t) gen;
        }
    }
    // wave-aggregated append: one atomic per wavefront and row instead of one per candidate
#pragma unroll
    for (int k = 0; k < CAND_PY; k++) {
        const unsigned long long m = __ballot(is_cand[k]);
        if (m == 0) continue;
        int base = 0;
        const int leader = __ffsll((long long) m) - 1;
        if (lane == leader) base = atomicAdd(&cand_cnt[roi], __popcll(m));
        base = __shfl(base, leader, 64);
        if (is_cand[k]) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            cand[(size_t) R.job * cand_plane + R.cand_base + slot] =
                ((unsigned long long) f32_order_key(v[k]) << 32) | (unsigned int) ((y0 + k) * R.rw + x);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
#define DET_MAX_PER_BLOCK 64

__global__ __launch_bounds__(256) void k_select(const det_roi *rois, unsigned long long *cand, size_t cand_plane,
                                                const int32_t *cand_cnt, int min_dist, float2 *corners /*roi x max_pb*/,
                                                int32_t *corner_cnt, int max_pb) {
    __shared__ unsigned long long wbest[4];
    __shared__ unsigned long long best;
    const det_roi R = rois[blockIdx.x];
    unsigned long long *C = cand + (size_t) R.job * cand_plane + R.cand_base;
    const int n = cand_cnt[blockIdx.x];
    const int t = threadIdx.x;
    const double md2 = (double) min_dist * (double) min_dist;
    int quota = R.quota < max_pb ? R.quota : max_pb;
    int acc   = 0;
    while (acc < quota) {
        unsigned long long k = 0;
        for (int i = t; i < n; i += 256) {
            unsigned long long c = C[i];
            k                    = c > k ? c : k;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            unsigned long long o = __shfl_xor(k, m, 64);
            k                    = o > k ? o : k;
        }
        if ((t & 63) == 0) wbest[t >> 6] = k;
        __syncthreads();
        if (t == 0) {
            unsigned long long b = wbest[0];
            for (int i = 1; i < 4; i++) b = wbest[i] > b ? wbest[i] : b;
            best = b;
        }
        __syncthreads();
        const unsigned long long b = best;
        if (b == 0) break; // no live candidate left
        const int idx = (int) (b & 0xffffffffu);
        const int by = idx / R.rw, bx = idx - by * R.rw;
        if (t == 0) corners[(size_t) blockIdx.x * max_pb + acc] = make_float2((float) bx, (float) by);
        acc++;
        if (min_dist >= 1) {
            for (int i = t; i < n; i += 256) {
                unsigned long long c = C[i];
                if (!c) continue;
                int ci = (int) (c & 0xffffffffu);
                int cy = ci / R.rw, cx = ci - cy * R.rw;
                float dx = (float) (cx - bx), dy = (float) (cy - by);
                if (c == b || (double) (dx * dx + dy * dy) < md2) C[i] = 0;
            }
        } else {
            for (int i = t; i < n; i += 256)
                if (C[i] == b) C[i] = 0;
        }
        __syncthreads();
    }
    if (t == 0) corner_cnt[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------------------
struct subpix_mask_t {
    float m[121];
};

__global__ __launch_bounds__(64) void k_subpix(const det_roi *rois, const uint8_t *frames, size_t slot_bytes,
                                               const int32_t *slots, int pitch, float2 *corners,
                                               const int32_t *corner_cnt, int max_pb, subpix_mask_t M) {
    __shared__ float patch[13][13];
    __shared__ double terms[121][5];
    __shared__ float cur[2];
    __shared__ int stop;
    const int roi = blockIdx.x / max_pb, ci = blockIdx.x - roi * max_pb;
    if (ci >= corner_cnt[roi]) return;
    const det_roi R    = rois[roi];
    const uint8_t *img = frames + (size_t) slots[R.job] * slot_bytes;
    const int lane     = threadIdx.x;
    const float2 cT    = corners[(size_t) roi * max_pb + ci];
    float cIx = cT.x, cIy = cT.y;
    int iter = 0;
    while (true) {
        const float ox = cIx - 6.f, oy = cIy - 6.f;
        const int iox = (int) floorf(ox), ioy = (int) floorf(oy);
        const float fa = ox - iox, fb = oy - ioy;
        const float w00 = (1.f - fa) * (1.f - fb), w01 = fa * (1.f - fb), w10 = (1.f - fa) * fb, w11 = fa * fb;
        for (int i = lane; i < 169; i += 64) {
            int r = i / 13, c = i - r * 13;
            int x0 = min(max(iox + c, 0), R.rw - 1), x1 = min(max(iox + c + 1, 0), R.rw - 1);
            int y0 = min(max(ioy + r, 0), R.rh - 1), y1 = min(max(ioy + r + 1, 0), R.rh - 1);
            const uint8_t *p0 = img + (size_t) (R.ry + y0) * pitch + R.rx;
            const uint8_t *p1 = img + (size_t) (R.ry + y1) * pitch + R.rx;
            float s00 = p0[x0], s01 = p0[x1], s10 = p1[x0], s11 = p1[x1];
            patch[r][c] = s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11;
        }
        __syncthreads();
        for (int i = lane; i < 121; i += 64) {
            int r = i / 11, c = i - r * 11;
            double m   = M.m[i];
            double tgx = patch[r + 1][c + 2] - patch[r + 1][c];
            double tgy = patch[r + 2][c + 1] - patch[r][c + 1];
            double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            double px = c - 5, py = r - 5;
            terms[i][0] = gxx;
            terms[i][1] = gxy;
            terms[i][2] = gyy;
            terms[i][3] = gxx * px + gxy * py;
            terms[i][4] = gxy * px + gyy * py;
        }
        __syncthreads();
        // the five 121-term sums keep the CPU's sequential raster order, but run side by side on lanes 0..4
        double acc = 0;
        if (lane < 5) {
#pragma unroll 11
            for (int i = 0; i < 121; i++) acc += terms[i][lane];
        }
        const double a = __shfl(acc, 0, 64), b = __shfl(acc, 1, 64), c = __shfl(acc, 2, 64);
        const double bb1 = __shfl(acc, 3, 64), bb2 = __shfl(acc, 4, 64);
        if (lane == 0) {
            int st     = 0;
            double det = a * c - b * b;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) {
                st = 1; // break before updating
            } else {
                double scale = 1.0 / det;
                float c2x    = (float) (cIx + c * scale * bb1 - b * scale * bb2);
                float c2y    = (float) (cIy - b * scale * bb1 + a * scale * bb2);
                double err   = (double) ((c2x - cIx) * (c2x - cIx) + (c2y - cIy) * (c2y - cIy));
                cur[0]       = c2x;
                cur[1]       = c2y;
                if (c2x < 0 || c2x >= R.rw || c2y < 0 || c2y >= R.rh)
                    st = 2;
                else if (!(iter + 1 < 20 && err > 0.01 * 0.01))
                    st = 2;
            }
            stop = st;
        }
        __syncthreads();
        const int st = stop;
        if (st != 1) {
            cIx = cur[0];
            cIy = cur[1];
        }
        ++iter;
        __syncthreads();
        if (st) break;
    }
    if (lane == 0) {
        if (fabsf(cIx - cT.x) > 5 || fabsf(cIy - cT.y) > 5) {
            cIx = cT.x;
            cIy = cT.y;
        }
        corners[(size_t) roi * max_pb + ci] = make_float2(cIx, cIy);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
static void circle_halfwidths(int radius, std::vector<int32_t> &hw) {
    // spans of OpenCV drawing.cpp Circle() (midpoint algorithm) for a filled circle, per |row offset|
    hw.assign((size_t) radius + 1, -1);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++;
        err += plus;
        plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

static int ensure_detect_ws(icg_ctx *ctx) {
    if (ctx->d_eig) return 0;
    const size_t w = ctx->cfg.width, h = ctx->cfg.height, nb = ctx->cfg.max_batch;
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_eig, sizeof(float) * w * h * nb));
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_mask, (size_t) ctx->lv[0].pitch * h * nb));
    ICG_HIP(ctx, hipMemsetAsync(ctx->d_mask, 0, (size_t) ctx->lv[0].pitch * h * nb, ctx->stream));
    ctx->mask_gen = 0;
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_cand, sizeof(unsigned long long) * w * h * nb));
    return 0;
}

extern "C" int icg_detect(icg_ctx *ctx, int n, const int32_t *slots, const icg_detect_grid *grid, const int32_t *mask_off,
                          const float *mask_pts, const int32_t *quota, int max_per_job, float *out_pts,
                          int32_t *out_count, int32_t *out_block) {
    if (!ctx || n < 0) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (!slots || !grid || !mask_off || !quota || !out_pts || !out_count || max_per_job <= 0) return ICG_ERR_INVALID;
    if (n > ctx->cfg.max_batch) return icg_fail(ctx, ICG_ERR_CAPACITY, "detect batch %d > max_batch %d", n, ctx->cfg.max_batch);
    const int w = ctx->cfg.width, h = ctx->cfg.height, pitch = ctx->lv[0].pitch;
    const int nblk = grid->block_cols * grid->block_rows;
    if (nblk <= 0 || grid->block_w <= 6 || grid->block_h <= 6 || grid->block_cols * grid->block_w > w ||
        grid->block_rows * grid->block_h > h || grid->min_dist < 0 || grid->max_per_block <= 0 ||
        grid->max_per_block > DET_MAX_PER_BLOCK)
        return icg_fail(ctx, ICG_ERR_INVALID, "bad detection grid");
    for (int k = 0; k < n; k++)
        if (slots[k] < 0 || slots[k] >= ctx->cfg.n_slots) return icg_fail(ctx, ICG_ERR_INVALID, "bad slot");
    const int n_mask = mask_off[n];
    if (n_mask < 0 || (n_mask > 0 && !mask_pts)) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_detect_ws(ctx);
    if (rc) return rc;

    // ROI list (tracking.cc:629-645)
    std::vector<det_roi> rois;
    for (int b = 0; b < n; b++)
        for (int k = 0; k < nblk; k++) {
            int q = quota[(size_t) b * nblk + k];
            if (q <= 0) continue;
            if (q > grid->max_per_block) q = grid->max_per_block;
            int cols = k % grid->block_cols, rows = k / grid->block_cols;
            det_roi R;
            R.job   = b;
            R.block = k;
            R.rx    = cols * grid->block_w;
            R.ry    = rows * grid->block_h;
            R.rw    = grid->block_w;
            R.rh    = grid->block_h;
            if (k != nblk - 1) {
                R.rw -= 5;
                R.rh -= 5;
            }
            R.quota     = q;
            R.cand_base = k * grid->block_w * grid->block_h;
            rois.push_back(R);
        }
    const int n_roi = (int) rois.size();
    for (int b = 0; b < n; b++) out_count[b] = 0;
    if (n_roi == 0) return ICG_OK;
    const int max_pb = grid->max_per_block;

    std::vector<int32_t> hw;
    circle_halfwidths(grid->min_dist, hw);
    std::vector<int32_t> pt_job((size_t) n_mask);
    for (int b = 0; b < n; b++)
        for (int i = mask_off[b]; i < mask_off[b + 1]; i++) pt_job[i] = b;

    icg_call c(ctx);
    size_t need = sizeof(det_roi) * n_roi + sizeof(int32_t) * (n + hw.size() + n_mask) + sizeof(float) * 2 * n_mask +
                  (sizeof(float2) * max_pb + 16) * (size_t) n_roi + 8192;
    if ((rc = c.reserve(need))) return rc;
    const det_roi *d_rois  = c.in(rois.data(), (size_t) n_roi);
    const int32_t *d_slots = c.in(slots, (size_t) n);
    const int32_t *d_hw    = c.in(hw.data(), hw.size());
    const float2 *d_mpts   = (const float2 *) c.in(mask_pts, 2 * (size_t) n_mask);
    const int32_t *d_ptjob = c.in(pt_job.data(), (size_t) n_mask);
    // [roi_max | cand_cnt] start at zero: staged with the inputs (rides on the single H2D copy, no memset launch)
    std::vector<unsigned int> zeros(2 * (size_t) n_roi, 0u);
    unsigned int *d_rmax = const_cast<unsigned int *>(c.in(zeros.data(), zeros.size()));
    int32_t *d_ccnt      = (int32_t *) (d_rmax + n_roi);
    if ((rc = c.seal())) return rc;
    std::vector<float> h_corners((size_t) n_roi * max_pb * 2);
    std::vector<int32_t> h_cnt((size_t) n_roi);
    // corners are re-read by k_subpix: keep them in device memory, fetch with the single D2H of finish()
    float2 *d_corners    = (float2 *) c.out(h_corners.data(), (size_t) n_roi * max_pb * 2);
    int32_t *d_cnt       = c.out(h_cnt.data(), (size_t) n_roi);

    const size_t mask_plane = (size_t) pitch * h, eig_plane = (size_t) w * h, cand_plane = (size_t) w * h;
    // mask generation (see k_mask_discs): a full clear only when the 8-bit tag wraps
    if (++ctx->mask_gen > 255) {
        ICG_HIP(ctx, hipMemsetAsync(ctx->d_mask, 0, mask_plane * ctx->cfg.max_batch, ctx->stream));
        ctx->mask_gen = 1;
    }
    const unsigned int gen = (unsigned int) ctx->mask_gen;
    if (n_mask > 0) {
        icg_prof_scope ps(ctx, "detect_mask");
        hipLaunchKernelGGL(k_mask_discs, dim3(n_mask), dim3(64), 0, ctx->stream, n_mask, d_mpts, d_ptjob, grid->min_dist,
                           d_hw, ctx->d_mask, pitch, w, h, mask_plane, gen);
    }
    {
        icg_prof_scope ps(ctx, "detect_min_eig");
        const int gx = (grid->block_w + EIG_TW - 1) / EIG_TW, gy = (grid->block_h + EIG_TH * EIG_WAVES - 1) / (EIG_TH * EIG_WAVES);
        hipLaunchKernelGGL(k_min_eig, dim3(icg_xcd_grid(gx * gy * n_roi)), dim3(64 * EIG_WAVES), 0, ctx->stream, d_rois, ctx->d_frames,
                           ctx->slot_bytes, d_slots, pitch, w, h, ctx->d_mask, mask_plane, gen, ctx->d_eig, eig_plane, d_rmax, gx, gy,
                           gx * gy * n_roi, icg_div_magic(gx * gy), icg_div_magic(gx));
    }
    {
        icg_prof_scope ps(ctx, "detect_candidates");
        const int gx = (grid->block_w + 63) / 64, gy = (grid->block_h + 4 * CAND_PY - 1) / (4 * CAND_PY);
        hipLaunchKernelGGL(k_candidates, dim3(icg_xcd_grid(gx * gy * n_roi)), dim3(256), 0, ctx->stream, d_rois, pitch, w, ctx->d_mask,
                           mask_plane, gen, ctx->d_eig, eig_plane, d_rmax, ctx->d_cand, cand_plane, d_ccnt, gx, gy, gx * gy * n_roi,
                           icg_div_magic(gx * gy), icg_div_magic(gx));
    }
    {
        icg_prof_scope ps(ctx, "detect_select");
        hipLaunchKernelGGL(k_select, dim3(n_roi), dim3(256), 0, ctx->stream, d_rois, ctx->d_cand, cand_plane, d_ccnt,
                           grid->min_dist, d_corners, d_cnt, max_pb);
    }
    {
        subpix_mask_t M;
        for (int i = 0; i < 11; i++) {
            float y  = (float) (i - 5) / 5;
            float vy = std::exp(-y * y);
            for (int j = 0; j < 11; j++) {
                float x         = (float) (j - 5) / 5;
                M.m[i * 11 + j] = (float) (vy * std::exp(-x * x));
            }
        }
        icg_prof_scope ps(ctx, "detect_subpix");
        hipLaunchKernelGGL(k_subpix, dim3(n_roi * max_pb), dim3(64), 0, ctx->stream, d_rois, ctx->d_frames, ctx->slot_bytes,
                           d_slots, pitch, d_corners, d_cnt, max_pb, M);
    }
    ICG_HIP(ctx, hipGetLastError());
    if ((rc = c.finish())) return rc;

    // block-order assembly with the block origin added (tracking.cc:669-685)
    for (int r = 0; r < n_roi; r++) {
        const det_roi &R = rois[r];
        for (int i = 0; i < h_cnt[r]; i++) {
            int &cnt = out_count[R.job];
            if (cnt >= max_per_job) break;
            float x = (float) R.rx + h_corners[((size_t) r * max_pb + i) * 2];
            float y = (float) R.ry + h_corners[((size_t) r * max_pb + i) * 2 + 1];
            out_pts[((size_t) R.job * max_per_job + cnt) * 2]     = x;
            out_pts[((size_t) R.job * max_per_job + cnt) * 2 + 1] = y;
            if (out_block) out_block[(size_t) R.job * max_per_job + cnt] = R.block;
            cnt++;
        }
    }
    return ICG_OK;
}
