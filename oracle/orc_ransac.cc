// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of
//   cv::findFundamentalMat(pts1, pts2, FM_RANSAC, 1.5, 0.99, status)   reference call site tracking/tracking.cc:547-555
// following OpenCV calib3d/src/fundam.cpp (FMEstimatorCallback::run7Point / computeError) and ptsetreg.cpp
// (RANSACPointSetRegistrator::run, getSubset, RANSACUpdateNumIters) as defined in SURVEY.md Appendix B.9.
// Formulation fixed here (mirrored by the HIP path so that decisions agree bit-for-bit):
//   * null space of the 7x9 system: one-sided (Hestenes) Jacobi on the 7 columns of A^T + basis completion, using only
//     + - * / sqrt (OpenCV: JacobiSVD on the same matrix; the two-dimensional null space is basis independent);
//   * cubic solved WITHOUT libm transcendentals (bisection on a Cauchy bracket + Newton polish + deflation; OpenCV's solveCubic
//     uses acos/cos/cbrt which differ between host and device libm), roots RETURNED IN cv::solveCubic's ORDER — three real roots:
//     (smallest, largest, middle) = -2*sqrt(Q)*cos(theta/3 + {0, 2pi/3, 4pi/3}) - a1/3 with theta/3 in [0, pi/3]; the quadratic
//     fallback: (q/a1, a3/q) with q the larger-magnitude one of (-a2 +- sqrt(disc))/2 — the order decides which of several
//     equally-scoring models wins the strict `good > maxGood` test of RANSACPointSetRegistrator::run;
//   * getSubset as in OpenCV 4.x ptsetreg.cpp: seven distinct indices, then FMEstimatorCallback::checkSubset (fundam.cpp) =
//     !haveCollinearPoints(ms1, 7) && !haveCollinearPoints(ms2, 7) (calib3d precomp.hpp: the LAST point of the subset against
//     every pair of the earlier ones, |cross| <= FLT_EPSILON * (|dx1|+|dy1|+|dx2|+|dy2|), double arithmetic on the float
//     coordinates); a rejected subset has consumed its RNG draws and the draw is repeated (at most 10000 attempts);
//   * scoring in double, compared as float against (float)(thresh^2), exactly as computeError/findInliers;
//   * the best/niters recurrence is replayed sequentially, so the result equals the sequential algorithm.
// PARITY UNPINNED (no upstream golden vectors, OpenCV absent offline).
#include "oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// One-sided Jacobi on an m x n matrix stored row-major in G (m rows, n cols); V (n x n) accumulates rotations.
// After convergence columns of G are orthogonal; their norms are the singular values.
void hestenes(double *G, int m, int n, double *V, int max_sweeps) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        bool changed = false;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < m; k++) {
                    double gp = G[k * n + p], gq = G[k * n + q];
                    alpha += gp * gp;
                    beta += gq * gq;
                    gamma += gp * gq;
                }
                if (gamma == 0.0) continue;
                if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                changed     = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t    = 1.0 / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                if (zeta < 0) t = -t;
                double c = 1.0 / std::sqrt(1.0 + t * t);
                double s = c * t;
                for (int k = 0; k < m; k++) {
                    double gp = G[k * n + p], gq = G[k * n + q];
                    G[k * n + p] = c * gp - s * gq;
                    G[k * n + q] = s * gp + c * gq;
                }
                for (int k = 0; k < n; k++) {
                    double vp = V[k * n + p], vq = V[k * n + q];
                    V[k * n + p] = c * vp - s * vq;
                    V[k * n + q] = s * vp + c * vq;
                }
            }
        if (!changed) break;
    }
}

// Real roots of c0 x^3 + c1 x^2 + c2 x + c3 = 0 without transcendentals, in cv::solveCubic's order (see the header); returns
// count (0..3).
int solve_cubic_real(const double c[4], double roots[3]) {
    double a = c[0], b = c[1], cc = c[2], d = c[3];
    double scale = std::fmax(std::fmax(std::fabs(a), std::fabs(b)), std::fmax(std::fabs(cc), std::fabs(d)));
    if (scale == 0) return 0;
    int n = 0;
    if (std::fabs(a) <= 1e-14 * scale) {
        // quadratic (or linear)
        if (std::fabs(b) <= 1e-14 * scale) {
            if (std::fabs(cc) <= 1e-14 * scale) return 0;
            roots[0] = -d / cc;
            return 1;
        }
        double disc = cc * cc - 4 * b * d;
        if (disc < 0) return 0;
        double sq = std::sqrt(disc);
        double q  = -0.5 * (cc + (cc >= 0 ? sq : -sq)); // the larger-magnitude one of q1 = (-a2+d)/2, q2 = -(a2+d)/2
        roots[0]  = q / b;                               // mathfuncs.cpp solveCubic: x0 = q/a1, x1 = a3/q
        roots[1]  = (q != 0) ? d / q : roots[0];
        return disc > 0 ? 2 : 1;
    }
    double p = b / a, q = cc / a, r = d / a; // x^3 + p x^2 + q x + r
    auto f  = [&](double x) { return ((x + p) * x + q) * x + r; };
    auto df = [&](double x) { return (3 * x + 2 * p) * x + q; };
    double B  = 1.0 + std::fmax(std::fabs(p), std::fmax(std::fabs(q), std::fabs(r)));
    double lo = -B, hi = B; // f(lo) < 0 < f(hi)
    for (int it = 0; it < 200; it++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        if (f(mid) < 0)
            lo = mid;
        else
            hi = mid;
    }
    double x1 = 0.5 * (lo + hi);
    for (int it = 0; it < 2; it++) {
        double dfx = df(x1);
        if (dfx != 0) {
            double xn = x1 - f(x1) / dfx;
            if (xn >= -B && xn <= B) x1 = xn;
        }
    }
    roots[n++] = x1;
    // deflate: x^2 + (p + x1) x + (q + (p + x1) x1)
    double b2 = p + x1, c2 = q + b2 * x1;
    double disc = b2 * b2 - 4 * c2;
    if (disc >= 0) {
        double sq = std::sqrt(disc);
        double qq = -0.5 * (b2 + (b2 >= 0 ? sq : -sq));
        double r1 = qq, r2 = (qq != 0) ? c2 / qq : qq;
        roots[n++] = r1;
        roots[n++] = r2;
    }
    std::sort(roots, roots + n);
    if (n == 3) std::swap(roots[1], roots[2]); // (smallest, largest, middle)
    return n;
}

// Null space of the 7x9 system A, given M = A^T (9x7, row-major).  As cv::SVDecomp does for m < n, the one-sided Jacobi
// runs on the 7 full-rank columns of A^T (fast convergence); the two missing right singular vectors of A are then
// obtained by completing the orthonormal basis of R^9 (twice-repeated Gram-Schmidt from the least-represented
// coordinate axes).  Only + - * / sqrt; fixed operation order.
void null_space_9x7(double *M, double *f1, double *f2) {
    const int m = 9, n = 7;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool changed = false;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < m; k++) {
                    double gp = M[k * n + p], gq = M[k * n + q];
                    alpha += gp * gp;
                    beta += gq * gq;
                    gamma += gp * gq;
                }
                if (gamma == 0.0) continue;
                if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                changed     = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t    = 1.0 / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                if (zeta < 0) t = -t;
                double c = 1.0 / std::sqrt(1.0 + t * t);
                double s = c * t;
                for (int k = 0; k < m; k++) {
                    double gp = M[k * n + p], gq = M[k * n + q];
                    M[k * n + p] = c * gp - s * gq;
                    M[k * n + q] = s * gp + c * gq;
                }
            }
        if (!changed) break;
    }
    // basis B: 9 vectors of length 9 (columns 0..6 = normalised columns of M, 7..8 = completion)
    double B[9][9];
    for (int c = 0; c < n; c++) {
        double s = 0;
        for (int k = 0; k < m; k++) s += M[k * n + c] * M[k * n + c];
        double w = std::sqrt(s);
        for (int k = 0; k < m; k++) B[c][k] = (w > 0) ? M[k * n + c] / w : 0.0;
    }
    for (int t = 0; t < 2; t++) {
        const int nb = n + t;
        int js = 0;
        double best = 0;
        for (int j = 0; j < m; j++) {
            double d = 0;
            for (int c = 0; c < nb; c++) d += B[c][j] * B[c][j];
            if (j == 0 || d < best) {
                best = d;
                js   = j;
            }
        }
        double v[9];
        for (int k = 0; k < m; k++) v[k] = (k == js) ? 1.0 : 0.0;
        for (int pass = 0; pass < 2; pass++)
            for (int c = 0; c < nb; c++) {
                double d = 0;
                for (int k = 0; k < m; k++) d += B[c][k] * v[k];
                for (int k = 0; k < m; k++) v[k] -= d * B[c][k];
            }
        double s = 0;
        for (int k = 0; k < m; k++) s += v[k] * v[k];
        double w = std::sqrt(s);
        for (int k = 0; k < m; k++) B[nb][k] = (w > 0) ? v[k] / w : 0.0;
    }
    for (int k = 0; k < m; k++) {
        f1[k] = B[7][k];
        f2[k] = B[8][k];
    }
}

struct Rng {
    uint64_t state;
    explicit Rng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t) (unsigned) state * 4164903690U + (unsigned) (state >> 32);
        return (unsigned) state;
    }
    int uniform(int a, int b) { return a == b ? a : (int) (next() % (unsigned) (b - a) + a); }
};

// calib3d precomp.hpp haveCollinearPoints(m, count): the last of `count` points against every pair of the earlier ones
bool have_collinear_points(const float *pts /*count x 2*/, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = pts[2 * j] - pts[2 * i];
        double dy1 = pts[2 * j + 1] - pts[2 * i + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = pts[2 * k] - pts[2 * i];
            double dy2 = pts[2 * k + 1] - pts[2 * i + 1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

// RANSACPointSetRegistrator::getSubset (ptsetreg.cpp, OpenCV 4.x) for modelPoints = 7 with FMEstimatorCallback::checkSubset.
// pts1 == nullptr: no subset check (the bare index stream).  Returns false after max_attempts rejected subsets.
bool get_subset(Rng &rng, int n, const float *pts1, const float *pts2, int idx[7], int max_attempts) {
    for (int attempt = 0; attempt < max_attempts; attempt++) {
        float a[14], b[14];
        for (int i = 0; i < 7; i++) {
            int v;
            for (v = rng.uniform(0, n); std::find(idx, idx + i, v) != idx + i; v = rng.uniform(0, n)) {
            }
            idx[i] = v;
            if (pts1) {
                a[2 * i] = pts1[2 * v], a[2 * i + 1] = pts1[2 * v + 1];
                b[2 * i] = pts2[2 * v], b[2 * i + 1] = pts2[2 * v + 1];
            }
        }
        if (!pts1 || (!have_collinear_points(a, 7) && !have_collinear_points(b, 7))) return true;
    }
    return false;
}

int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p  = std::max(p, 0.);
    p  = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num   = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num   = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int) lrint(num / denom);
}

} // namespace

extern "C" {

// 7-point algorithm on 7 correspondences (m1,m2: 7x2 doubles holding float-valued pixel coordinates).
// F: up to 3 row-major 3x3 matrices. Returns the number of models.
int orc_seven_point(const double *m1, const double *m2, double *F) {
    // M = A^T (9 x 7): row r, column c = coefficient r of equation c
    double M[9 * 7], f1[9], f2[9];
    for (int i = 0; i < 7; i++) {
        double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        M[0 * 7 + i] = x2 * x1;
        M[1 * 7 + i] = x2 * y1;
        M[2 * 7 + i] = x2;
        M[3 * 7 + i] = y2 * x1;
        M[4 * 7 + i] = y2 * y1;
        M[5 * 7 + i] = y2;
        M[6 * 7 + i] = x1;
        M[7 * 7 + i] = y1;
        M[8 * 7 + i] = 1;
    }
    null_space_9x7(M, f1, f2);
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];
    double c[4], t0, t1, t2;
    t0   = f2[4] * f2[8] - f2[5] * f2[7];
    t1   = f2[3] * f2[8] - f2[5] * f2[6];
    t2   = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0   = f1[4] * f1[8] - f1[5] * f1[7];
    t1   = f1[3] * f1[8] - f1[5] * f1[6];
    t2   = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    double roots[3];
    int n = solve_cubic_real(c, roots);
    for (int k = 0; k < n; k++) {
        double lambda = roots[k], mu = 1.;
        double s   = f1[8] * roots[k] + f2[8];
        double *Fk = F + 9 * k;
        if (std::fabs(s) > DBL_EPSILON) {
            mu = 1. / s;
            lambda *= mu;
            Fk[8] = 1.;
        } else
            Fk[8] = 0.;
        for (int i = 0; i < 8; i++) Fk[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

// inlier mask of one model (computeError + findInliers); returns the inlier count
int orc_fm_score(const double *F, int n, const float *pts1, const float *pts2, double thresh, uint8_t *mask) {
    float t = (float) (thresh * thresh);
    int nz  = 0;
    for (int i = 0; i < n; i++) {
        double x1 = pts1[2 * i], y1 = pts1[2 * i + 1], x2 = pts2[2 * i], y2 = pts2[2 * i + 1];
        double a = F[0] * x1 + F[1] * y1 + F[2];
        double b = F[3] * x1 + F[4] * y1 + F[5];
        double c = F[6] * x1 + F[7] * y1 + F[8];
        double s2 = 1. / (a * a + b * b);
        double d2 = x2 * a + y2 * b + c;
        a         = F[0] * x2 + F[3] * y2 + F[6];
        b         = F[1] * x2 + F[4] * y2 + F[7];
        c         = F[2] * x2 + F[5] * y2 + F[8];
        double s1 = 1. / (a * a + b * b);
        double d1 = x1 * a + y1 * b + c;
        float e   = (float) std::max(d1 * d1 * s1, d2 * d2 * s2);
        int f     = e <= t;
        if (mask) mask[i] = (uint8_t) f;
        nz += f;
    }
    return nz;
}

// the hypothesis index stream of RANSACPointSetRegistrator (getSubset, OpenCV 4.x): idx_out = n_hyp x 7.  pts1/pts2 (n_points x 2
// floats, may be NULL) enable FMEstimatorCallback::checkSubset.  Returns the number of hypotheses produced (< n_hyp if getSubset gave up).
int orc_ransac_subsets(int n_points, const float *pts1, const float *pts2, int n_hyp, int32_t *idx_out) {
    Rng rng((uint64_t) -1);
    for (int h = 0; h < n_hyp; h++)
        if (!get_subset(rng, n_points, pts1, pts2, idx_out + 7 * h, 10000)) return h;
    return n_hyp;
}

// the cubic solver alone (tests pin the root order against cv::solveCubic's closed forms)
int orc_solve_cubic(const double *coeffs4, double *roots3) { return solve_cubic_real(coeffs4, roots3); }

int orc_have_collinear_points(const float *pts, int count) { return have_collinear_points(pts, count) ? 1 : 0; }

// Full findFundamentalMat(FM_RANSAC). Returns 1 if a model was found. mask: n bytes (all zero on failure).
int orc_find_fundamental_ransac(int n, const float *pts1, const float *pts2, double thresh, double conf,
                                uint8_t *mask, double *F_out, int *iters_out) {
    const int modelPoints = 7;
    int niters            = 1000;
    int maxGood           = 0;
    memset(mask, 0, n);
    if (n < 15) return 0; // findFundamentalMat switches to LMedS below 15 points; the reference never calls it then
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    if (thresh <= 0) thresh = 3;
    Rng rng((uint64_t) -1);
    std::vector<uint8_t> cur(n);
    double bestF[9] = {0};
    int iter;
    for (iter = 0; iter < niters; iter++) {
        int idx[7];
        if (!get_subset(rng, n, pts1, pts2, idx, 10000)) { // ptsetreg.cpp run(): no valid subset -> fail on the first iteration, else stop
            if (iter == 0) {
                if (iters_out) *iters_out = 0;
                return 0;
            }
            break;
        }
        double m1[14], m2[14], F[27];
        for (int i = 0; i < 7; i++) {
            m1[2 * i]     = pts1[2 * idx[i]];
            m1[2 * i + 1] = pts1[2 * idx[i] + 1];
            m2[2 * i]     = pts2[2 * idx[i]];
            m2[2 * i + 1] = pts2[2 * idx[i] + 1];
        }
        int nmodels = orc_seven_point(m1, m2, F);
        for (int k = 0; k < nmodels; k++) {
            int good = orc_fm_score(F + 9 * k, n, pts1, pts2, thresh, cur.data());
            if (good > std::max(maxGood, modelPoints - 1)) {
                memcpy(mask, cur.data(), n);
                memcpy(bestF, F + 9 * k, sizeof bestF);
                maxGood = good;
                niters  = update_num_iters(conf, (double) (n - good) / n, modelPoints, niters);
            }
        }
    }
    if (iters_out) *iters_out = iter;
    if (F_out) memcpy(F_out, bestF, sizeof bestF);
    if (maxGood == 0) {
        memset(mask, 0, n);
        return 0;
    }
    return 1;
}

// Tracking::triangulatePoint (tracking.cc:800-811): smallest right singular vector of the 4x4 DLT matrix
// (Eigen jacobiSvd in the reference; one-sided Jacobi here), dehomogenised.
void orc_triangulate_point(const double *T0, const double *T1, const double *pc0, const double *pc1, double *pw) {
    double D[16], V[16];
    for (int j = 0; j < 4; j++) {
        D[0 * 4 + j] = pc0[0] * T0[2 * 4 + j] - T0[0 * 4 + j];
        D[1 * 4 + j] = pc0[1] * T0[2 * 4 + j] - T0[1 * 4 + j];
        D[2 * 4 + j] = pc1[0] * T1[2 * 4 + j] - T1[0 * 4 + j];
        D[3 * 4 + j] = pc1[1] * T1[2 * 4 + j] - T1[1 * 4 + j];
    }
    hestenes(D, 4, 4, V, 30);
    int best = 0;
    double bn = 0;
    for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += D[k * 4 + j] * D[k * 4 + j];
        if (j == 0 || s < bn) {
            bn   = s;
            best = j;
        }
    }
    double wv = V[3 * 4 + best];
    pw[0]     = V[0 * 4 + best] / wv;
    pw[1]     = V[1 * 4 + best] / wv;
    pw[2]     = V[2 * 4 + best] / wv;
}

} // extern "C"
