// Deterministic synthetic camera streams for tests and bench (SURVEY.md §8(d) generator): a procedurally textured,
// slanted plane seen by the radtan pinhole camera of config/gvins.yaml from a given pose.  Integer-hash texture, so
// every machine renders the same bytes.  Pure input synthesis — no tracker logic here.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {
inline uint64_t splitmix64(uint64_t &x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z          = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z          = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
inline uint32_t hash2(uint32_t x, uint32_t y, uint32_t seed) {
    uint64_t s = ((uint64_t) x << 32) ^ y ^ ((uint64_t) seed << 17) ^ 0x1C0671A5ull;
    return (uint32_t) (splitmix64(s) >> 32);
}
} // namespace

extern "C" {

// size x size wrap-around texture: 3 octaves of value noise + random bright/dark 3..7 px blobs
void icgs_make_texture(int size, uint32_t seed, uint8_t *out) {
    std::vector<float> acc((size_t) size * size, 20.f);
    const int oct[3]   = {64, 16, 5};
    const float amp[3] = {70.f, 45.f, 30.f};
    for (int o = 0; o < 3; o++) {
        const int cell = oct[o];
        const int g    = (size + cell - 1) / cell;
        for (int y = 0; y < size; y++) {
            int y0 = y / cell;
            float fy = (float) (y - y0 * cell) / cell;
            for (int x = 0; x < size; x++) {
                int x0 = x / cell;
                float fx = (float) (x - x0 * cell) / cell;
                auto v = [&](int gx, int gy) { return (hash2((uint32_t) (gx % g), (uint32_t) (gy % g), seed + o) & 0xffff) / 65535.f; };
                float a = v(x0, y0), b = v(x0 + 1, y0), c = v(x0, y0 + 1), d = v(x0 + 1, y0 + 1);
                acc[(size_t) y * size + x] += amp[o] * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy);
            }
        }
    }
    uint64_t st = 0xB10B5ull ^ seed;
    const int nblobs = (int) ((size_t) size * size / 450);
    for (int i = 0; i < nblobs; i++) {
        int bx = (int) (splitmix64(st) % (uint64_t) size), by = (int) (splitmix64(st) % (uint64_t) size);
        int bs = 3 + (int) (splitmix64(st) % 5);
        float bv = (splitmix64(st) & 1) ? 70.f : -70.f;
        for (int y = 0; y < bs; y++)
            for (int x = 0; x < bs; x++) acc[(size_t) ((by + y) % size) * size + (bx + x) % size] += bv;
    }
    for (size_t i = 0; i < acc.size(); i++) {
        float v = acc[i];
        out[i]  = (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : (int) (v + 0.5f)));
    }
}

// per-pixel undistorted normalised ray (x,y) for the radtan camera cam10 = fx,fy,cx,cy,skew,k1,k2,p1,p2,k3
void icgs_ray_table(const double *cam, int w, int h, float *xy) {
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], k1 = cam[5], k2 = cam[6], p1 = cam[7], p2 = cam[8], k3 = cam[9];
    for (int v = 0; v < h; v++)
        for (int u = 0; u < w; u++) {
            double x0 = (u - cx) / fx, y0 = (v - cy) / fy, x = x0, y = y0;
            for (int it = 0; it < 8; it++) {
                double r2 = x * x + y * y, icd = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
                double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
                x = (x0 - dX) * icd;
                y = (y0 - dY) * icd;
            }
            xy[((size_t) v * w + u) * 2]     = (float) x;
            xy[((size_t) v * w + u) * 2 + 1] = (float) y;
        }
}

// Render one frame: plane n.P = d (world), texture axes e1,e2 (unit, in-plane), texels_per_m; pose12 = R (camera->world,
// row-major) | t.  out: w x h u8 with the given stride.
void icgs_render(const uint8_t *tex, int tsize, double texels_per_m, const float *ray_xy, int w, int h, const double *pose12,
                 const double *plane4, const double *e1, const double *e2, uint8_t *out, int stride, int threads) {
    const double *R = pose12, *t = pose12 + 9;
    const double nt = plane4[0] * t[0] + plane4[1] * t[1] + plane4[2] * t[2];
    const bool pow2 = tsize > 0 && (tsize & (tsize - 1)) == 0;
    auto rows = [&](int y0, int y1) {
        for (int v = y0; v < y1; v++)
            for (int u = 0; u < w; u++) {
                double x = ray_xy[((size_t) v * w + u) * 2], y = ray_xy[((size_t) v * w + u) * 2 + 1];
                double rx = R[0] * x + R[1] * y + R[2], ry = R[3] * x + R[4] * y + R[5], rz = R[6] * x + R[7] * y + R[8];
                double den = plane4[0] * rx + plane4[1] * ry + plane4[2] * rz;
                uint8_t val = 0;
                if (den > 1e-9) {
                    double lam = (plane4[3] - nt) / den;
                    if (lam > 0) {
                        double px = t[0] + lam * rx, py = t[1] + lam * ry, pz = t[2] + lam * rz;
                        double a = (px * e1[0] + py * e1[1] + pz * e1[2]) * texels_per_m;
                        double b = (px * e2[0] + py * e2[1] + pz * e2[2]) * texels_per_m;
                        // floor without the libm call (|a|, |b| are texel coordinates, far below 2^53: the conversions are exact)
                        long ia = (long) a, ib = (long) b;
                        if ((double) ia > a) ia--;
                        if ((double) ib > b) ib--;
                        double fa = (double) ia, fb = (double) ib;
                        double wa = a - fa, wb = b - fb;
                        // wrap-around texel indices: two wraps per pixel instead of sixteen divisions (a mask when the size is a power of
                        // two); the same texels and the same arithmetic as T(i, j) = tex[wrap(j)][wrap(i)] at (ia, ib) .. (ia+1, ib+1)
                        const long i0 = pow2 ? (ia & (tsize - 1)) : ((ia % tsize) + tsize) % tsize, i1 = i0 + 1 == tsize ? 0 : i0 + 1;
                        const long j0 = pow2 ? (ib & (tsize - 1)) : ((ib % tsize) + tsize) % tsize, j1 = j0 + 1 == tsize ? 0 : j0 + 1;
                        const uint8_t *r0 = tex + (size_t) j0 * tsize, *r1 = tex + (size_t) j1 * tsize;
                        double s = ((double) r0[i0] * (1 - wa) + (double) r0[i1] * wa) * (1 - wb) + ((double) r1[i0] * (1 - wa) + (double) r1[i1] * wa) * wb;
                        val = (uint8_t) (s + 0.5);
                    }
                }
                out[(size_t) v * stride + u] = val;
            }
    };
    if (threads <= 1) {
        rows(0, h);
        return;
    }
    std::vector<std::thread> th;
    for (int k = 0; k < threads; k++) th.emplace_back(rows, h * k / threads, h * (k + 1) / threads);
    for (auto &x : th) x.join();
}

} // extern "C"
