// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the OpenCV image operations the reference's
// front-end calls (the arithmetic lives in OpenCV, un-vendored, version unpinned ">=3.2.0": README.md:58):
//   cvtColor BGR2GRAY          reference call site ic_gvins/ic_gvins/tracking/tracking.cc:112
//   CLAHE(3.0, 21x21)->apply   reference call site tracking.cc:63,139     (OpenCV imgproc/src/clahe.cpp)
//   buildOpticalFlowPyramid    inside calcOpticalFlowPyrLK, tracking.cc:385-393,487-496 (video/src/lkpyramid.cpp, imgproc pyrDown)
//   calcSharrDeriv             same
// Definitions follow SURVEY.md Appendix B.1-B.4.  PARITY UNPINNED: the reference ships no tests/golden vectors
// and OpenCV is not available offline; exact-integer steps are reproduced exactly by construction.
#include "oracle.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}
static inline int cv_round_f(float v) { return (int) lrintf(v); } // round-half-even under default rounding mode
static inline unsigned char sat_u8_from_float(float v) {
    int iv = cv_round_f(v);
    return (unsigned char) (iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

extern "C" {

// B.1 (exact)
void orc_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray, int gstride) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p    = bgr + (size_t) y * stride + 3 * x;
            gray[(size_t) y * gstride + x] = (uint8_t) ((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
        }
}

// tracking.cc:88-105  calculateHistigram: mean of k/256 weighted by a float histogram
double orc_histogram_mean(const uint8_t *img, int w, int h, int stride) {
    std::vector<float> hist(256, 0.f);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) hist[img[(size_t) y * stride + x]] += 1.0f;
    double acc = 0;
    for (int k = 0; k < 256; k++) acc += hist[k] * (float) k / 256.0;
    return acc / ((double) w * h);
}

// B.2 CLAHE. lut_out (optional) receives tiles*tiles*256 bytes.
void orc_clahe(const uint8_t *src, int w, int h, int stride, double clip_limit, int tiles, uint8_t *dst, int dstride,
               uint8_t *lut_out) {
    int ew = w, eh = h;
    if (w % tiles != 0 || h % tiles != 0) {
        ew = w + (tiles - w % tiles);
        eh = h + (tiles - h % tiles);
    }
    int tw = ew / tiles, th = eh / tiles;
    int tile_area  = tw * th;
    float lutScale = 255.0f / tile_area;
    int clip       = (int) (clip_limit * tile_area / 256);
    if (clip < 1) clip = 1;

    std::vector<uint8_t> lut((size_t) tiles * tiles * 256);
    for (int ty = 0; ty < tiles; ty++)
        for (int tx = 0; tx < tiles; tx++) {
            int hist[256];
            memset(hist, 0, sizeof hist);
            for (int yy = 0; yy < th; yy++) {
                int sy = reflect101(ty * th + yy, h);
                for (int xx = 0; xx < tw; xx++) {
                    int sx = reflect101(tx * tw + xx, w);
                    hist[src[(size_t) sy * stride + sx]]++;
                }
            }
            int clipped = 0;
            for (int i = 0; i < 256; i++)
                if (hist[i] > clip) {
                    clipped += hist[i] - clip;
                    hist[i] = clip;
                }
            int batch    = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; i++) hist[i] += batch;
            if (residual != 0) {
                int step = 256 / residual;
                if (step < 1) step = 1;
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
            int sum        = 0;
            uint8_t *tlut  = &lut[((size_t) ty * tiles + tx) * 256];
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                tlut[i] = sat_u8_from_float((float) sum * lutScale);
            }
        }
    if (lut_out) memcpy(lut_out, lut.data(), lut.size());

    float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    std::vector<uint8_t> out((size_t) w * h);
    for (int y = 0; y < h; y++) {
        float tyf = y * inv_th - 0.5f;
        int ty1   = (int) floorf(tyf);
        int ty2   = ty1 + 1;
        float ya  = tyf - ty1;
        float ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles - 1) ty2 = tiles - 1;
        for (int x = 0; x < w; x++) {
            float txf = x * inv_tw - 0.5f;
            int tx1   = (int) floorf(txf);
            int tx2   = tx1 + 1;
            float xa  = txf - tx1;
            float xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles - 1) tx2 = tiles - 1;
            int v     = src[(size_t) y * stride + x];
            float l11 = lut[((size_t) ty1 * tiles + tx1) * 256 + v];
            float l12 = lut[((size_t) ty1 * tiles + tx2) * 256 + v];
            float l21 = lut[((size_t) ty2 * tiles + tx1) * 256 + v];
            float l22 = lut[((size_t) ty2 * tiles + tx2) * 256 + v];
            float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            out[(size_t) y * w + x] = sat_u8_from_float(res);
        }
    }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t) y * dstride, &out[(size_t) y * w], w);
}

// B.3 pyrDown (exact). dst is ((w+1)/2) x ((h+1)/2).
void orc_pyrdown(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    std::vector<int> tmp((size_t) dw * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t) y * stride;
        for (int x = 0; x < dw; x++) {
            int a = s[reflect101(2 * x - 2, w)], b = s[reflect101(2 * x - 1, w)], c = s[reflect101(2 * x, w)],
                d = s[reflect101(2 * x + 1, w)], e = s[reflect101(2 * x + 2, w)];
            tmp[(size_t) y * dw + x] = a + e + 4 * (b + d) + 6 * c;
        }
    }
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int r0 = tmp[(size_t) reflect101(2 * y - 2, h) * dw + x], r1 = tmp[(size_t) reflect101(2 * y - 1, h) * dw + x],
                r2 = tmp[(size_t) reflect101(2 * y, h) * dw + x], r3 = tmp[(size_t) reflect101(2 * y + 1, h) * dw + x],
                r4 = tmp[(size_t) reflect101(2 * y + 2, h) * dw + x];
            dst[(size_t) y * dstride + x] = (uint8_t) ((r0 + r4 + 4 * (r1 + r3) + 6 * r2 + 128) >> 8);
        }
}

// number of LK pyramid levels actually built (maxLevel+1), B.3
int orc_pyramid_levels(int w, int h, int max_level, int win) {
    int levels = 1;
    for (int l = 1; l <= max_level; l++) {
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) break;
        levels++;
    }
    return levels;
}

// B.4 Scharr derivative (exact). deriv is interleaved (Ix,Iy) int16, w*h*2.
void orc_scharr(const uint8_t *src, int w, int h, int stride, int16_t *deriv) {
    std::vector<int> t0(w), t1(w);
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = src + (size_t) reflect101(y - 1, h) * stride;
        const uint8_t *r1 = src + (size_t) y * stride;
        const uint8_t *r2 = src + (size_t) reflect101(y + 1, h) * stride;
        for (int x = 0; x < w; x++) {
            t0[x] = 3 * (r0[x] + r2[x]) + 10 * r1[x];
            t1[x] = r2[x] - r0[x];
        }
        for (int x = 0; x < w; x++) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            deriv[((size_t) y * w + x) * 2 + 0] = (int16_t) (t0[xp] - t0[xm]);
            deriv[((size_t) y * w + x) * 2 + 1] = (int16_t) (3 * (t1[xm] + t1[xp]) + 10 * t1[x]);
        }
    }
}

} // extern "C"
