// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the detection stage of the reference front-end:
//   Tracking::featuresDetection   ic_gvins/ic_gvins/tracking/tracking.cc:576-688  (mask discs :609-620, per-block ROI
//                                 :632-645, goodFeaturesToTrack :647, cornerSubPix :651, block-order append :669-685)
// OpenCV algorithm definitions (imgproc/src/featureselect.cpp, corner.cpp, cornersubpix.cpp, drawing.cpp Circle()):
// SURVEY.md Appendix B.7 / B.8.  Formulation fixed here (and mirrored by the HIP kernels):
//   * Sobel 3x3 in exact integers on the REAL image pixels around the ROI (reflect-101 only at true image borders),
//     one float multiply by 1/3060; covariance products in float; 3x3 un-normalised box sum accumulated in double
//     in raster order with reflect-101 at the ROI edge, rounded once to float (OpenCV: double running sums);
//   * cornerSubPix samples the 13x13 patch with plain float bilinear weights and replicate border at the ROI edge
//     (OpenCV's 8u->32f getRectSubPix uses an algebraically equal recurrence; differences are ~1e-5 px);
//     the 11x11 Gaussian mask is passed in by the caller (expf is libm-specific).
// PARITY UNPINNED (no upstream golden vectors, OpenCV absent offline).
#include "orc_parallel.h"
#include "oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}
static inline int cv_round_f(float v) { return (int) lrintf(v); }
} // namespace

extern "C" {

// cv::circle(img, center, radius, value, FILLED) for shift=0, LINE_8: OpenCV drawing.cpp Circle() midpoint algorithm.
void orc_draw_filled_circle(uint8_t *mask, int w, int h, int stride, int cx, int cy, int radius, uint8_t value) {
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    auto hline = [&](int y, int x0, int x1) {
        if (y < 0 || y >= h) return;
        if (x0 < 0) x0 = 0;
        if (x1 >= w) x1 = w - 1;
        for (int x = x0; x <= x1; x++) mask[(size_t) y * stride + x] = value;
    };
    while (dx >= dy) {
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        hline(y11, x11, x12);
        hline(y12, x11, x12);
        hline(y21, x21, x22);
        hline(y22, x21, x22);
        dy++;
        err += plus;
        plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

// half-width table of the filled disc: hw[d] for |row offset| = d in [0, radius]; -1 where no pixel.
void orc_circle_halfwidths(int radius, int *hw) {
    for (int i = 0; i <= radius; i++) hw[i] = -1;
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

// cornerMinEigenVal(blockSize 3, ksize 3) of the ROI [rx,rx+rw) x [ry,ry+rh) of a w x h image. eig: rw*rh floats.
void orc_min_eigen_map(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw, int rh, float *eig) {
    const float s = (float) (1.0 / 3060.0);
    std::vector<float> cxx((size_t) rw * rh), cxy((size_t) rw * rh), cyy((size_t) rw * rh);
    auto P = [&](int x, int y) -> int { return img[(size_t) reflect101(y, h) * stride + reflect101(x, w)]; };
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            int X = rx + x, Y = ry + y;
            int gx = (P(X + 1, Y - 1) - P(X - 1, Y - 1)) + 2 * (P(X + 1, Y) - P(X - 1, Y)) + (P(X + 1, Y + 1) - P(X - 1, Y + 1));
            int gy = (P(X - 1, Y + 1) - P(X - 1, Y - 1)) + 2 * (P(X, Y + 1) - P(X, Y - 1)) + (P(X + 1, Y + 1) - P(X + 1, Y - 1));
            float dx = (float) gx * s, dy = (float) gy * s;
            cxx[(size_t) y * rw + x] = dx * dx;
            cxy[(size_t) y * rw + x] = dx * dy;
            cyy[(size_t) y * rw + x] = dy * dy;
        }
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            double sa = 0, sb = 0, sc = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    size_t k = (size_t) reflect101(y + j, rh) * rw + reflect101(x + i, rw);
                    sa += cxx[k];
                    sb += cxy[k];
                    sc += cyy[k];
                }
            float a = (float) sa * 0.5f, b = (float) sb, c = (float) sc * 0.5f;
            eig[(size_t) y * rw + x] = (float) ((a + c) - sqrtf((a - c) * (a - c) + b * b));
        }
}

// cv::goodFeaturesToTrack(block_image, out, max_corners, quality, min_dist, block_mask) on a ROI; corners are
// ROI-local (x,y) floats; returns the count.
int orc_good_features(const uint8_t *img, int w, int h, int stride, const uint8_t *mask, int mstride, int rx, int ry,
                      int rw, int rh, int max_corners, double quality, double min_dist, float *corners) {
    if (rw <= 0 || rh <= 0 || max_corners <= 0) return 0;
    std::vector<float> eig((size_t) rw * rh);
    orc_min_eigen_map(img, w, h, stride, rx, ry, rw, rh, eig.data());
    double maxVal = 0;
    bool any      = false;
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++)
            if (!mask || mask[(size_t) (ry + y) * mstride + rx + x]) {
                double v = eig[(size_t) y * rw + x];
                if (!any || v > maxVal) {
                    maxVal = v;
                    any    = true;
                }
            }
    if (!any) maxVal = 0;
    float thresh = (float) (maxVal * quality);
    for (auto &v : eig)
        if (!(v > thresh)) v = 0.f;
    struct Cand {
        float v;
        int idx;
    };
    std::vector<Cand> cand;
    for (int y = 1; y < rh - 1; y++)
        for (int x = 1; x < rw - 1; x++) {
            float val = eig[(size_t) y * rw + x];
            if (val == 0) continue;
            float mx = val;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) mx = std::max(mx, eig[(size_t) (y + j) * rw + x + i]);
            if (val == mx && (!mask || mask[(size_t) (ry + y) * mstride + rx + x])) cand.push_back({val, y * rw + x});
        }
    std::sort(cand.begin(), cand.end(), [](const Cand &a, const Cand &b) { return a.v > b.v ? true : (a.v < b.v ? false : a.idx > b.idx); });
    int n = 0;
    if (min_dist >= 1) {
        const int cell = (int) lrint(min_dist);
        const int gw = (rw + cell - 1) / cell, gh = (rh + cell - 1) / cell;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t) gw * gh);
        double md2 = min_dist * min_dist;
        for (auto &c : cand) {
            int y = c.idx / rw, x = c.idx - y * rw;
            bool good = true;
            int xc = x / cell, yc = y / cell;
            int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (auto &m : grid[(size_t) yy * gw + xx]) {
                        float dx = x - m.first, dy = y - m.second;
                        if (dx * dx + dy * dy < md2) {
                            good = false;
                            break;
                        }
                    }
            if (good) {
                grid[(size_t) yc * gw + xc].push_back({(float) x, (float) y});
                corners[2 * n]     = (float) x;
                corners[2 * n + 1] = (float) y;
                n++;
                if (n == max_corners) break;
            }
        }
    } else {
        for (auto &c : cand) {
            int y = c.idx / rw, x = c.idx - y * rw;
            corners[2 * n]     = (float) x;
            corners[2 * n + 1] = (float) y;
            n++;
            if (n == max_corners) break;
        }
    }
    return n;
}

// the 11x11 weighting mask of cornerSubPix(win=(5,5)) — host libm expf, passed to both implementations
void orc_subpix_mask(float *mask121) {
    for (int i = 0; i < 11; i++) {
        float y  = (float) (i - 5) / 5;
        float vy = std::exp(-y * y);
        for (int j = 0; j < 11; j++) {
            float x          = (float) (j - 5) / 5;
            mask121[i * 11 + j] = (float) (vy * std::exp(-x * x));
        }
    }
}

// cv::cornerSubPix(block_image, pts, (5,5), (-1,-1), (COUNT+EPS, 20, 0.01)) on a ROI; corners ROI-local, in place.
void orc_corner_subpix(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw, int rh, int n,
                       float *corners) {
    (void) w;
    (void) h;
    float mask[121];
    orc_subpix_mask(mask);
    const int max_iters = 20;
    const double eps    = 0.01 * 0.01;
    float patch[13][13];
    for (int p = 0; p < n; p++) {
        float cTx = corners[2 * p], cTy = corners[2 * p + 1];
        float cIx = cTx, cIy = cTy;
        int iter   = 0;
        double err = 0;
        do {
            // getRectSubPix(src, 13x13, cI): origin = cI - 6, bilinear, replicate border inside the ROI
            float ox = cIx - 6.f, oy = cIy - 6.f;
            int iox = (int) floorf(ox), ioy = (int) floorf(oy);
            float fa = ox - iox, fb = oy - ioy;
            float w00 = (1.f - fa) * (1.f - fb), w01 = fa * (1.f - fb), w10 = (1.f - fa) * fb, w11 = fa * fb;
            for (int i = 0; i < 13; i++)
                for (int j = 0; j < 13; j++) {
                    int x0 = std::min(std::max(iox + j, 0), rw - 1), x1 = std::min(std::max(iox + j + 1, 0), rw - 1);
                    int y0 = std::min(std::max(ioy + i, 0), rh - 1), y1 = std::min(std::max(ioy + i + 1, 0), rh - 1);
                    float s00 = img[(size_t) (ry + y0) * stride + rx + x0], s01 = img[(size_t) (ry + y0) * stride + rx + x1];
                    float s10 = img[(size_t) (ry + y1) * stride + rx + x0], s11 = img[(size_t) (ry + y1) * stride + rx + x1];
                    patch[i][j] = s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11;
                }
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            for (int i = 0; i < 11; i++) {
                double py = i - 5;
                for (int j = 0; j < 11; j++) {
                    double m   = mask[i * 11 + j];
                    double tgx = patch[i + 1][j + 2] - patch[i + 1][j];
                    double tgy = patch[i + 2][j + 1] - patch[i][j + 1];
                    double gxx = tgx * tgx * m;
                    double gxy = tgx * tgy * m;
                    double gyy = tgy * tgy * m;
                    double px  = j - 5;
                    a += gxx;
                    b += gxy;
                    c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            float c2x    = (float) (cIx + c * scale * bb1 - b * scale * bb2);
            float c2y    = (float) (cIy - b * scale * bb1 + a * scale * bb2);
            err          = (c2x - cIx) * (c2x - cIx) + (c2y - cIy) * (c2y - cIy);
            cIx          = c2x;
            cIy          = c2y;
            if (cIx < 0 || cIx >= rw || cIy < 0 || cIy >= rh) break;
        } while (++iter < max_iters && err > eps);
        if (std::fabs(cIx - cTx) > 5 || std::fabs(cIy - cTy) > 5) {
            cIx = cTx;
            cIy = cTy;
        }
        corners[2 * p]     = cIx;
        corners[2 * p + 1] = cIy;
    }
}

// One featuresDetection device job as exposed by icg_detect (see include/icgvins_hip.h): mask discs at mask_pts,
// per-block ROI + quota, GFTT + subpix, block-order output with block origin added (tracking.cc:669-685).
// grid6 = {block_cols, block_rows, block_w, block_h, min_dist, max_per_block}
int orc_detect(const uint8_t *img, int w, int h, int stride, const int *grid6, int n_mask, const float *mask_pts,
               const int *quota, int max_out, float *out_pts, int *out_block) {
    const int bc = grid6[0], br = grid6[1], bw = grid6[2], bh = grid6[3], md = grid6[4];
    std::vector<uint8_t> mask((size_t) w * h, 255);
    for (int i = 0; i < n_mask; i++)
        orc_draw_filled_circle(mask.data(), w, h, w, cv_round_f(mask_pts[2 * i]), cv_round_f(mask_pts[2 * i + 1]), md, 0);
    const int nb = bc * br;
    // every block's corners in its own list (the reference's tbb::parallel_for over blocks, tracking.cc:656), appended in block order
    std::vector<std::vector<float>> found((size_t) nb);
    std::vector<int> origin_x((size_t) nb, 0), origin_y((size_t) nb, 0);
    orc_parallel_chunks(nb, [&](int k0, int k1) {
        for (int k = k0; k < k1; k++) {
            int q = quota[k];
            if (q <= 0) continue;
            int cols = k % bc, rows = k / bc;
            int col_sta = cols * bw, col_end = col_sta + bw, row_sta = rows * bh, row_end = row_sta + bh;
            if (k != nb - 1) {
                col_end -= 5;
                row_end -= 5;
            }
            int rw = col_end - col_sta, rh = row_end - row_sta;
            std::vector<float> corners((size_t) 2 * q, 0.f);
            int n = orc_good_features(img, w, h, stride, mask.data(), w, col_sta, row_sta, rw, rh, q, 0.01, (double) md, corners.data());
            if (n > 0) orc_corner_subpix(img, w, h, stride, col_sta, row_sta, rw, rh, n, corners.data());
            corners.resize((size_t) 2 * std::max(0, n));
            found[(size_t) k].swap(corners);
            origin_x[(size_t) k] = col_sta, origin_y[(size_t) k] = row_sta;
        }
    });
    int total = 0;
    for (int k = 0; k < nb; k++) {
        const std::vector<float> &corners = found[(size_t) k];
        for (size_t i = 0; 2 * i < corners.size() && total < max_out; i++) {
            out_pts[2 * total]     = (float) origin_x[(size_t) k] + corners[2 * i];
            out_pts[2 * total + 1] = (float) origin_y[(size_t) k] + corners[2 * i + 1];
            if (out_block) out_block[total] = k;
            total++;
        }
    }
    return total;
}

} // extern "C"
