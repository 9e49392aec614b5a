// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of cv::calcOpticalFlowPyrLK as parameterised by the
// reference at ic_gvins/ic_gvins/tracking/tracking.cc:385-393 and :487-496 (win 21x21, maxLevel 3,
// criteria COUNT+EPS (30, 0.01), OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4) and of the forward/backward
// cull at tracking.cc:396-403 / :499-506 with isOnBorder (:847-849) and ptsDistance (:841-845).
// Algorithm: OpenCV modules/video/src/lkpyramid.cpp (LKTrackerInvoker) as defined in SURVEY.md Appendix B.5;
// window sums are accumulated as exact int64 and converted once (the survey's bit-stable formulation).
// PARITY UNPINNED (no upstream golden vectors, OpenCV absent offline).
#include "orc_parallel.h"
#include "oracle.h"
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

struct Level {
    int w, h;
    std::vector<uint8_t> img;
    std::vector<int16_t> deriv; // only for "prev" pyramids
    inline int px(int x, int y) const { return img[(size_t) reflect101(y, h) * w + reflect101(x, w)]; }
    inline int dx(int x, int y) const {
        if (x < 0 || x >= w || y < 0 || y >= h) return 0;
        return deriv[((size_t) y * w + x) * 2];
    }
    inline int dy(int x, int y) const {
        if (x < 0 || x >= w || y < 0 || y >= h) return 0;
        return deriv[((size_t) y * w + x) * 2 + 1];
    }
};

struct Pyr {
    std::vector<Level> lv;
};

void build_pyr(const uint8_t *img, int w, int h, int stride, int max_level, int win, bool with_deriv, Pyr &p) {
    int nl = orc_pyramid_levels(w, h, max_level, win);
    p.lv.resize(nl);
    p.lv[0].w = w;
    p.lv[0].h = h;
    p.lv[0].img.resize((size_t) w * h);
    for (int y = 0; y < h; y++) memcpy(&p.lv[0].img[(size_t) y * w], img + (size_t) y * stride, w);
    for (int l = 1; l < nl; l++) {
        Level &a = p.lv[l - 1];
        Level &b = p.lv[l];
        b.w      = (a.w + 1) / 2;
        b.h      = (a.h + 1) / 2;
        b.img.resize((size_t) b.w * b.h);
        orc_pyrdown(a.img.data(), a.w, a.h, a.w, b.img.data(), b.w);
    }
    if (with_deriv)
        for (int l = 0; l < nl; l++) {
            Level &a = p.lv[l];
            a.deriv.resize((size_t) a.w * a.h * 2);
            orc_scharr(a.img.data(), a.w, a.h, a.w, a.deriv.data());
        }
}

static inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
static inline int cv_round_f(float v) { return (int) lrintf(v); }

constexpr int WIN = 21;
constexpr int HALF = 10;
constexpr int MAX_ITERS = 30;

void lk_point(const Pyr &P, const Pyr &N, float px, float py, float *nx, float *ny, uint8_t *status, float *err) {
    const float FLT_SCALE = 1.f / (1 << 20);
    const double eps2     = 0.01 * 0.01;
    const float minEigThreshold = 1e-4f;
    int nl                = (int) P.lv.size();
    int maxLevel          = nl - 1;
    *status               = 1;
    if (err) *err = 0;
    float nextx = *nx, nexty = *ny; // nextPts[i] storage
    short Iw[WIN * WIN], dIx[WIN * WIN], dIy[WIN * WIN];

    for (int level = maxLevel; level >= 0; level--) {
        const Level &I = P.lv[level];
        const Level &J = N.lv[level];
        float scale    = (float) (1. / (1 << level));
        float prevx = px * scale, prevy = py * scale;
        float nptx, npty;
        if (level == maxLevel) {
            nptx = nextx * scale;
            npty = nexty * scale;
        } else {
            nptx = nextx * 2.f;
            npty = nexty * 2.f;
        }
        nextx = nptx;
        nexty = npty;

        prevx -= (float) HALF;
        prevy -= (float) HALF;
        int ipx = (int) floorf(prevx), ipy = (int) floorf(prevy);
        if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
            if (level == 0) {
                *status = 0;
                if (err) *err = 0;
            }
            continue;
        }
        float a = prevx - ipx, b = prevy - ipy;
        int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << 14));
        int iw01 = cv_round_f(a * (1.f - b) * (1 << 14));
        int iw10 = cv_round_f((1.f - a) * b * (1 << 14));
        int iw11 = (1 << 14) - iw00 - iw01 - iw10;

        int64_t iA11 = 0, iA12 = 0, iA22 = 0;
        for (int y = 0; y < WIN; y++)
            for (int x = 0; x < WIN; x++) {
                int X = ipx + x, Y = ipy + y;
                int ival = descale(I.px(X, Y) * iw00 + I.px(X + 1, Y) * iw01 + I.px(X, Y + 1) * iw10 +
                                       I.px(X + 1, Y + 1) * iw11,
                                   14 - 5);
                int ixval = descale(I.dx(X, Y) * iw00 + I.dx(X + 1, Y) * iw01 + I.dx(X, Y + 1) * iw10 +
                                        I.dx(X + 1, Y + 1) * iw11,
                                    14);
                int iyval = descale(I.dy(X, Y) * iw00 + I.dy(X + 1, Y) * iw01 + I.dy(X, Y + 1) * iw10 +
                                        I.dy(X + 1, Y + 1) * iw11,
                                    14);
                Iw[y * WIN + x]  = (short) ival;
                dIx[y * WIN + x] = (short) ixval;
                dIy[y * WIN + x] = (short) iyval;
                iA11 += (int64_t) ixval * ixval;
                iA12 += (int64_t) ixval * iyval;
                iA22 += (int64_t) iyval * iyval;
            }
        float A11 = (float) iA11 * FLT_SCALE, A12 = (float) iA12 * FLT_SCALE, A22 = (float) iA22 * FLT_SCALE;
        float D      = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
        if (minEig < minEigThreshold || D < FLT_EPSILON) {
            if (level == 0) *status = 0;
            continue;
        }
        D = 1.f / D;
        nptx -= (float) HALF;
        npty -= (float) HALF;
        float pdx = 0, pdy = 0;
        for (int j = 0; j < MAX_ITERS; j++) {
            int inx = (int) floorf(nptx), iny = (int) floorf(npty);
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                if (level == 0) *status = 0;
                break;
            }
            a    = nptx - inx;
            b    = npty - iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << 14));
            iw01 = cv_round_f(a * (1.f - b) * (1 << 14));
            iw10 = cv_round_f((1.f - a) * b * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            int64_t ib1 = 0, ib2 = 0;
            for (int y = 0; y < WIN; y++)
                for (int x = 0; x < WIN; x++) {
                    int X = inx + x, Y = iny + y;
                    int diff = descale(J.px(X, Y) * iw00 + J.px(X + 1, Y) * iw01 + J.px(X, Y + 1) * iw10 +
                                           J.px(X + 1, Y + 1) * iw11,
                                       14 - 5) -
                               Iw[y * WIN + x];
                    ib1 += (int64_t) diff * dIx[y * WIN + x];
                    ib2 += (int64_t) diff * dIy[y * WIN + x];
                }
            float b1 = (float) ib1 * FLT_SCALE, b2 = (float) ib2 * FLT_SCALE;
            float dx = (float) ((A12 * b2 - A22 * b1) * D);
            float dy = (float) ((A12 * b1 - A11 * b2) * D);
            nptx += dx;
            npty += dy;
            nextx = nptx + (float) HALF;
            nexty = npty + (float) HALF;
            if ((double) dx * dx + (double) dy * dy <= eps2) break;
            if (j > 0 && std::fabs(dx + pdx) < 0.01 && std::fabs(dy + pdy) < 0.01) {
                nextx -= dx * 0.5f;
                nexty -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }
        if (*status && err && level == 0) {
            float ex = nextx - (float) HALF, ey = nexty - (float) HALF;
            int inx = (int) floorf(ex), iny = (int) floorf(ey);
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                *status = 0;
                continue;
            }
            float aa = ex - inx, bb = ey - iny;
            iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << 14));
            iw01 = cv_round_f(aa * (1.f - bb) * (1 << 14));
            iw10 = cv_round_f((1.f - aa) * bb * (1 << 14));
            iw11 = (1 << 14) - iw00 - iw01 - iw10;
            int64_t errval = 0;
            for (int y = 0; y < WIN; y++)
                for (int x = 0; x < WIN; x++) {
                    int X = inx + x, Y = iny + y;
                    int diff = descale(J.px(X, Y) * iw00 + J.px(X + 1, Y) * iw01 + J.px(X, Y + 1) * iw10 +
                                           J.px(X + 1, Y + 1) * iw11,
                                       14 - 5) -
                               Iw[y * WIN + x];
                    errval += std::abs(diff);
                }
            *err = (float) errval * 1.f / (32 * WIN * WIN);
        }
    }
    *nx = nextx;
    *ny = nexty;
}

} // namespace

extern "C" {

// One calcOpticalFlowPyrLK call: prev/next are single-channel 8-bit images; next_pts holds the initial guess
// on entry (OPTFLOW_USE_INITIAL_FLOW) and the result on exit.
void orc_lk_track(const uint8_t *prev, const uint8_t *next, int w, int h, int stride, int n, const float *prev_pts,
                  float *next_pts, uint8_t *status, float *err) {
    Pyr P, N;
    build_pyr(prev, w, h, stride, 3, WIN, true, P);
    build_pyr(next, w, h, stride, 3, WIN, false, N);
    orc_parallel_chunks(n, [&](int i0, int i1) { // (points are independent: OpenCV's own parallel_for_ over them)
        for (int i = i0; i < i1; i++) {
            float e = 0;
            lk_point(P, N, prev_pts[2 * i], prev_pts[2 * i + 1], &next_pts[2 * i], &next_pts[2 * i + 1], &status[i], &e);
            if (err) err[i] = e;
        }
    });
}

// Forward + backward + cull exactly as tracking.cc:385-403 (or :487-506):
//   fwd: prev->next from guess; bwd: next->prev with initial = prev_pts; keep iff both status, !isOnBorder(fwd),
//   ||bwd - prev|| < 0.5 (double).
void orc_lk_track_fb(const uint8_t *prev, const uint8_t *next, int w, int h, int stride, int n, const float *prev_pts,
                     const float *guess_pts, float *out_pts, uint8_t *status) {
    std::vector<float> rev(prev_pts, prev_pts + 2 * (size_t) n), err(n);
    std::vector<uint8_t> st_rev(n);
    memcpy(out_pts, guess_pts, sizeof(float) * 2 * (size_t) n);
    orc_lk_track(prev, next, w, h, stride, n, prev_pts, out_pts, status, err.data());
    orc_lk_track(next, prev, w, h, stride, n, out_pts, rev.data(), st_rev.data(), err.data());
    for (int k = 0; k < n; k++) {
        float x = out_pts[2 * k], y = out_pts[2 * k + 1];
        bool border = x < 5.0 || y < 5.0 || (x > (w - 5.0)) || (y > (h - 5.0));
        double dx = rev[2 * k] - prev_pts[2 * k], dy = rev[2 * k + 1] - prev_pts[2 * k + 1];
        double dist = std::sqrt(dx * dx + dy * dy);
        status[k]   = (status[k] && st_rev[k] && !border && dist < 0.5) ? 1 : 0;
    }
}

} // extern "C"
