// F6: cv::findFundamentalMat(FM_RANSAC, 1.5 px, 0.99) outlier cull (tracking/tracking.cc:547-555) and
// F8: Tracking::triangulatePoint (tracking/tracking.cc:800-811), batched.
//
// Algorithm definition: SURVEY.md Appendix B.9/B.10 with the formulation pinned in oracle/orc_ransac.cc
// (one-sided Jacobi null space, transcendental-free cubic, double scoring with float compare, sequential
// best/niters replay).  Mapping:
//   * the hypothesis index stream depends only on cv::RNG's state, not on scores, so it is generated up front on the
//     host (same LCG as cv::RNG((uint64)-1)) for a chunk of hypotheses;
//   * k_fm_hypothesis: 32 hypotheses per workgroup.  Solve: a LANE per hypothesis — as cv::SVDecomp does for m < n, the one-sided
//     Jacobi runs on the 7 full-rank columns of A^T (9x7, REGISTER resident, fully unrolled) and the two null vectors come from
//     completing the orthonormal basis; only + - * / sqrt are used so results match the CPU restatement bit-for-bit.  Score: a WAVE
//     per model — 64 points per step, symmetric epipolar distance in double, inlier bits by wave ballot, count by popcount;
//   * the host replays the `best / niters` recurrence over the score array in hypothesis order, which makes the result
//     identical to the sequential algorithm; further chunks are generated only if niters demands them.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <map>
#include <mutex>

#include "icg_internal.h"

struct fm_set {
    int pt_begin, n_pts; // range in the concatenated point arrays
    int hyp_begin;       // first hypothesis slot of this set in this launch
    int n_hyp;
    int word_begin; // first u64 word of this set's inlier bit masks (per model: words_per_model)
    int words_per_model;
};

__device__ int dev_solve_cubic_real(const double c[4], double roots[3]) {
    double a = c[0], b = c[1], cc = c[2], d = c[3];
    double scale = fmax(fmax(fabs(a), fabs(b)), fmax(fabs(cc), fabs(d)));
    if (scale == 0) return 0;
    int n = 0;
    if (fabs(a) <= 1e-14 * scale) {
        if (fabs(b) <= 1e-14 * scale) {
            if (fabs(cc) <= 1e-14 * scale) return 0;
            roots[0] = -d / cc;
            return 1;
        }
        double disc = cc * cc - 4 * b * d;
        if (disc < 0) return 0;
        double sq = sqrt(disc);
        double q  = -0.5 * (cc + (cc >= 0 ? sq : -sq));
        roots[0]  = q / b; // cv::solveCubic's quadratic branch: x0 = q/a1, x1 = a3/q (q the larger-magnitude root of the resolvent)
        roots[1]  = (q != 0) ? d / q : roots[0];
        return disc > 0 ? 2 : 1;
    }
    const double p = b / a, q = cc / a, r = d / a;
    const double B = 1.0 + fmax(fabs(p), fmax(fabs(q), fabs(r)));
    double lo = -B, hi = B;
    for (int it = 0; it < 200; it++) {
        double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        double fm = ((mid + p) * mid + q) * mid + r;
        if (fm < 0)
            lo = mid;
        else
            hi = mid;
    }
    double x1 = 0.5 * (lo + hi);
    for (int it = 0; it < 2; it++) {
        double dfx = (3 * x1 + 2 * p) * x1 + q;
        if (dfx != 0) {
            double fx = ((x1 + p) * x1 + q) * x1 + r;
            double xn = x1 - fx / dfx;
            if (xn >= -B && xn <= B) x1 = xn;
        }
    }
    roots[n++] = x1;
    double b2 = p + x1, c2 = q + b2 * x1;
    double disc = b2 * b2 - 4 * c2;
    if (disc >= 0) {
        double sq = sqrt(disc);
        double qq = -0.5 * (b2 + (b2 >= 0 ? sq : -sq));
        double r1 = qq, r2 = (qq != 0) ? c2 / qq : qq;
        roots[n++] = r1;
        roots[n++] = r2;
    }
    // ascending insertion sort (n <= 3)
    for (int i = 1; i < n; i++) {
        double v = roots[i];
        int j    = i - 1;
        while (j >= 0 && roots[j] > v) {
            roots[j + 1] = roots[j];
            j--;
        }
        roots[j + 1] = v;
    }
    if (n == 3) { // cv::solveCubic's order of three real roots: (smallest, largest, middle)
        const double mid = roots[1];
        roots[1]         = roots[2];
        roots[2]         = mid;
    }
    return n;
}

// The seven-point solve of ONE hypothesis by the calling lane: up to three fundamental matrices into out27, their number into *n_out.
__device__ __forceinline__ void seven_point_solve(const fm_set &S, const int32_t *idx /*7, set-local*/, const float2 *pts1, const float2 *pts2,
                                                  double *out27, int *n_out) {
    // M = A^T (9 x 7) register resident; see oracle/orc_ransac.cc null_space_9x7 for the definition this mirrors.
    double M[9][7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float2 a = pts1[S.pt_begin + idx[i]], b = pts2[S.pt_begin + idx[i]];
        const double x1 = a.x, y1 = a.y, x2 = b.x, y2 = b.y;
        M[0][i] = x2 * x1;
        M[1][i] = x2 * y1;
        M[2][i] = x2;
        M[3][i] = y2 * x1;
        M[4][i] = y2 * y1;
        M[5][i] = y2;
        M[6][i] = x1;
        M[7][i] = y1;
        M[8][i] = 1;
    }
    for (int sweep = 0; sweep < 30; sweep++) {
        bool changed = false;
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = p + 1; q < 7; q++) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    double gp = M[k][p], gq = M[k][q];
                    alpha += gp * gp;
                    beta += gq * gq;
                    gamma += gp * gq;
                }
                if (gamma != 0.0 && !(fabs(gamma) <= 1e-15 * sqrt(alpha * beta))) {
                    changed     = true;
                    double zeta = (beta - alpha) / (2.0 * gamma);
                    double t    = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    if (zeta < 0) t = -t;
                    double c = 1.0 / sqrt(1.0 + t * t);
                    double s = c * t;
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        double gp = M[k][p], gq = M[k][q];
                        M[k][p] = c * gp - s * gq;
                        M[k][q] = s * gp + c * gq;
                    }
                }
            }
        if (!changed) break;
    }
    double B[9][9]; // B[c][k]: basis vector c, coordinate k
#pragma unroll
    for (int c = 0; c < 7; c++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += M[k][c] * M[k][c];
        double w = sqrt(s);
#pragma unroll
        for (int k = 0; k < 9; k++) B[c][k] = (w > 0) ? M[k][c] / w : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int nb = 7 + t;
        int js = 0;
        double best = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            double d = 0;
#pragma unroll
            for (int c = 0; c < nb; c++) d += B[c][j] * B[c][j];
            if (j == 0 || d < best) {
                best = d;
                js   = j;
            }
        }
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = (k == js) ? 1.0 : 0.0;
#pragma unroll
        for (int pass = 0; pass < 2; pass++)
#pragma unroll
            for (int c = 0; c < nb; c++) {
                double d = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) d += B[c][k] * v[k];
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] -= d * B[c][k];
            }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += v[k] * v[k];
        double w = sqrt(s);
#pragma unroll
        for (int k = 0; k < 9; k++) B[nb][k] = (w > 0) ? v[k] / w : 0.0;
    }
    double f1[9], f2[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        f1[k] = B[7][k];
        f2[k] = B[8][k];
    }
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
    const int n = dev_solve_cubic_real(c, roots);
    double *out = out27;
    for (int k = 0; k < n; k++) {
        double lambda = roots[k], mu = 1.;
        double s   = f1[8] * roots[k] + f2[8];
        double *Fk = out + 9 * k;
        if (fabs(s) > DBL_EPSILON) {
            mu = 1. / s;
            lambda *= mu;
            Fk[8] = 1.;
        } else
            Fk[8] = 0.;
        for (int i = 0; i < 8; i++) Fk[i] = f1[i] * lambda + f2[i] * mu;
    }
    *n_out = n;
}

// k_fm_hypothesis: a workgroup of four waves takes FM_HPW hypotheses.  Solve phase: LANE PER HYPOTHESIS in wave 0 — the seven-point
// solve is a strictly serial FP64 chain per hypothesis (Jacobi sweeps, basis completion, cubic) whose rotations cannot be spread over
// lanes without changing its rounding, but different hypotheses are independent, so 32 of them advance in the lanes of one wave at the
// cost of one (round 2: one wave per hypothesis with 63 idle lanes, 23 % of the queue time for < 10 % of the work).  The models pass
// through LDS; scoring phase: WAVE PER MODEL — the four waves take the (hypothesis, model) pairs in turn, 64 points per step, symmetric
// epipolar distance in double, inlier bits by wave ballot, count by popcount.  Still one launch per RANSAC round, every input read
// where the host staged it (pinned memory, zero-copy).  Per hypothesis the arithmetic is the one of round 2: bit-identical masks.
#define FM_HPW 32
__global__ __launch_bounds__(256, 1) void k_fm_hypothesis(int n_hyp_total, const fm_set *sets, const int32_t *hyp_set,
                                                          const int32_t *hyp_idx /*n_hyp x 7 (set-local)*/, const float2 *pts1,
                                                          const float2 *pts2, float thresh2, int32_t *good /*n_hyp x 3*/,
                                                          unsigned long long *bits) {
    __shared__ double Fm[FM_HPW][27];
    __shared__ int n_sh[FM_HPW];
    const int hyp0 = blockIdx.x * FM_HPW, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int n_here = min(FM_HPW, n_hyp_total - hyp0);
    if (n_here <= 0) return;
    if (wave == 0 && lane < n_here) {
        const int hyp = hyp0 + lane;
        int n         = 0;
        seven_point_solve(sets[hyp_set[hyp]], hyp_idx + 7 * (size_t) hyp, pts1, pts2, Fm[lane], &n);
        n_sh[lane] = n;
    }
    __syncthreads();
    for (int pair = wave; pair < 3 * n_here; pair += 4) {
        const int hl = pair / 3, model = pair - 3 * hl, hyp = hyp0 + hl;
        if (model >= n_sh[hl]) {
            if (lane == 0) good[hyp * 3 + model] = -1;
            continue;
        }
        const fm_set S  = sets[hyp_set[hyp]];
        const double *F = Fm[hl] + 9 * model;
        const double F0 = F[0], F1 = F[1], F2 = F[2], F3 = F[3], F4 = F[4], F5 = F[5], F6 = F[6], F7 = F[7], F8 = F[8];
        unsigned long long *w = bits + S.word_begin + ((size_t) (hyp - S.hyp_begin) * 3 + model) * S.words_per_model;
        int count = 0;
        for (int base = 0; base < S.n_pts; base += 64) {
            const int i = base + lane;
            bool in     = false;
            if (i < S.n_pts) {
                const float2 p1 = pts1[S.pt_begin + i], p2 = pts2[S.pt_begin + i];
                const double x1 = p1.x, y1 = p1.y, x2 = p2.x, y2 = p2.y;
                double a = F0 * x1 + F1 * y1 + F2;
                double b = F3 * x1 + F4 * y1 + F5;
                double c = F6 * x1 + F7 * y1 + F8;
                double s2 = 1. / (a * a + b * b);
                double d2 = x2 * a + y2 * b + c;
                a         = F0 * x2 + F3 * y2 + F6;
                b         = F1 * x2 + F4 * y2 + F7;
                c         = F2 * x2 + F5 * y2 + F8;
                double s1 = 1. / (a * a + b * b);
                double d1 = x1 * a + y1 * b + c;
                float e   = (float) fmax(d1 * d1 * s1, d2 * d2 * s2);
                in        = e <= thresh2;
            }
            const unsigned long long m = __ballot(in);
            if (lane == 0) w[base >> 6] = m;
            count += __popcll(m);
        }
        if (lane == 0) good[hyp * 3 + model] = count;
    }
}

// ---------------------------------------------------------------------------------------------------------
namespace {
struct cv_rng { // cv::RNG (core/operations.hpp): MWC generator, coefficient 4164903690
    uint64_t state;
    explicit cv_rng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t) (unsigned) state * 4164903690U + (unsigned) (state >> 32);
        return (unsigned) state;
    }
    int uniform(int a, int b) { return a == b ? a : (int) (next() % (unsigned) (b - a) + a); }
};

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) { // ptsetreg.cpp RANSACUpdateNumIters
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

// calib3d precomp.hpp haveCollinearPoints(m, 7): the last point of the subset against every pair of the earlier ones
bool have_collinear_points(const float *pts, const int *idx) {
    const int i = 6;
    for (int j = 0; j < i; j++) {
        double dx1 = pts[2 * idx[j]] - pts[2 * idx[i]];
        double dy1 = pts[2 * idx[j] + 1] - pts[2 * idx[i] + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = pts[2 * idx[k]] - pts[2 * idx[i]];
            double dy2 = pts[2 * idx[k] + 1] - pts[2 * idx[i] + 1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

// RANSACPointSetRegistrator::getSubset (ptsetreg.cpp, OpenCV 4.x) with FMEstimatorCallback::checkSubset (fundam.cpp): a rejected
// subset has consumed its RNG draws; false after 10000 rejected attempts
bool get_subset(cv_rng &rng, int n, const float *p1, const float *p2, int idx[7]) {
    for (int attempt = 0; attempt < 10000; attempt++) {
        for (int i = 0; i < 7; i++) {
            int v;
            for (v = rng.uniform(0, n); std::find(idx, idx + i, v) != idx + i; v = rng.uniform(0, n)) {
            }
            idx[i] = v;
        }
        if (!have_collinear_points(p1, idx) && !have_collinear_points(p2, idx)) return true;
    }
    return false;
}

struct set_state {
    int begin, n;
    cv_rng rng{(uint64_t) -1};
    int iter = 0, niters = 1000, max_good = 0;
    bool done = false;
    std::vector<uint8_t> best;
};
} // namespace

extern "C" int icg_fm_ransac(icg_ctx *ctx, int n_sets, const int32_t *offsets, const float *pts1, const float *pts2,
                             double thresh, double conf, uint8_t *mask) {
    if (!ctx || n_sets < 0) return ICG_ERR_INVALID;
    if (n_sets == 0) return ICG_OK;
    if (!offsets || !pts1 || !pts2 || !mask) return ICG_ERR_INVALID;
    const int total = offsets[n_sets];
    if (total > ctx->cfg.max_points) return icg_fail(ctx, ICG_ERR_CAPACITY, "%d points > max_points %d", total, ctx->cfg.max_points);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (thresh <= 0) thresh = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;

    std::vector<set_state> st((size_t) n_sets);
    int active = 0;
    for (int s = 0; s < n_sets; s++) {
        st[s].begin = offsets[s];
        st[s].n     = offsets[s + 1] - offsets[s];
        if (st[s].n < 0) return ICG_ERR_INVALID;
        if (st[s].n < 15) { // the reference only calls findFundamentalMat with >= 15 points: leave untouched
            for (int i = 0; i < st[s].n; i++) mask[st[s].begin + i] = 1;
            st[s].done = true;
        } else {
            st[s].best.assign((size_t) st[s].n, 0);
            active++;
        }
    }
    int chunk = 16;
    while (active > 0) {
        // hypotheses of this round: each active set gets min(chunk, niters - iter)
        std::vector<fm_set> sets;
        std::vector<int> set_of;       // launch set -> global set
        std::vector<int32_t> hyp_set;  // per hypothesis -> launch set
        std::vector<int32_t> hyp_idx;  // 7 per hypothesis
        int words = 0;
        for (int s = 0; s < n_sets; s++) {
            set_state &S = st[s];
            if (S.done) continue;
            int nh = std::min(chunk, S.niters - S.iter);
            fm_set f;
            f.pt_begin        = S.begin;
            f.n_pts           = S.n;
            f.hyp_begin       = (int) hyp_set.size();
            f.n_hyp           = nh;
            f.words_per_model = (S.n + 63) / 64;
            f.word_begin      = words;
            words += nh * 3 * f.words_per_model;
            for (int h = 0; h < nh; h++) {
                int idx[7];
                if (!get_subset(S.rng, S.n, pts1 + 2 * (size_t) S.begin, pts2 + 2 * (size_t) S.begin, idx)) {
                    // ptsetreg.cpp run(): no valid subset -> the iterations end here (nothing found if this was the first one)
                    S.niters = S.iter + h;
                    nh       = h;
                    break;
                }
                hyp_set.push_back((int32_t) sets.size());
                hyp_idx.insert(hyp_idx.end(), idx, idx + 7);
            }
            f.n_hyp = nh; // (words keeps the planned count: unused tail words are harmless)
            set_of.push_back(s);
            sets.push_back(f);
        }
        const int nh_total = (int) hyp_set.size();
        std::vector<int32_t> h_good((size_t) nh_total * 3);
        std::vector<unsigned long long> h_bits((size_t) words);
        if (nh_total > 0) { // (no hypothesis at all: every active set ran out of valid subsets)
        icg_call c(ctx);
        size_t need = sizeof(fm_set) * sets.size() + sizeof(int32_t) * 8 * (size_t) nh_total + sizeof(float) * 4 * (size_t) total +
                      (size_t) nh_total * (27 * 8 + 4 + 12) + (size_t) words * 8 + 16384;
        int rc = c.reserve(need);
        if (rc) return rc;
        // everything the launch reads stays where it is staged (pinned memory, read over PCIe): no upload launch
        const fm_set *d_sets  = c.in_zc(sets.data(), sets.size());
        const int32_t *d_hset = c.in_zc(hyp_set.data(), (size_t) nh_total);
        const int32_t *d_hidx = c.in_zc(hyp_idx.data(), 7 * (size_t) nh_total);
        const float2 *d_p1    = (const float2 *) c.in_zc(pts1, 2 * (size_t) total);
        const float2 *d_p2    = (const float2 *) c.in_zc(pts2, 2 * (size_t) total);
        if ((rc = c.seal())) return rc;
        int32_t *d_good   = c.out_zc(h_good.data(), (size_t) nh_total * 3);
        unsigned long long *d_bits = c.out_zc(h_bits.data(), (size_t) words);
        ICG_LAUNCH_GUARD(c);
        {
            icg_prof_scope ps(ctx, "fm_hypothesis");
            hipLaunchKernelGGL(k_fm_hypothesis, dim3((nh_total + FM_HPW - 1) / FM_HPW), dim3(256), 0, ctx->stream, nh_total, d_sets, d_hset, d_hidx, d_p1, d_p2,
                               (float) (thresh * thresh), d_good, d_bits);
        }
        ICG_HIP(ctx, hipGetLastError());
        if ((rc = c.finish())) return rc;
        }

        // sequential replay of RANSACPointSetRegistrator::run over the scores (ptsetreg.cpp)
        for (size_t ls = 0; ls < sets.size(); ls++) {
            set_state &S    = st[set_of[ls]];
            const fm_set &f = sets[ls];
            for (int h = 0; h < f.n_hyp && S.iter < S.niters; h++, S.iter++) {
                for (int m = 0; m < 3; m++) {
                    int good = h_good[(size_t) (f.hyp_begin + h) * 3 + m];
                    if (good < 0) break;
                    if (good > std::max(S.max_good, 7 - 1)) {
                        const unsigned long long *w = &h_bits[f.word_begin + ((size_t) h * 3 + m) * f.words_per_model];
                        for (int i = 0; i < S.n; i++) S.best[i] = (uint8_t) ((w[i >> 6] >> (i & 63)) & 1ull);
                        S.max_good = good;
                        S.niters   = ransac_update_num_iters(conf, (double) (S.n - good) / S.n, 7, S.niters);
                    }
                }
            }
            if (S.iter >= S.niters) {
                S.done = true;
                active--;
                if (S.max_good > 0)
                    memcpy(mask + S.begin, S.best.data(), (size_t) S.n);
                else
                    memset(mask + S.begin, 0, (size_t) S.n);
            }
        }
        if (chunk < 256) chunk *= 2;
    }
    return ICG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// F8: 4x4 DLT, smallest right singular vector by one-sided Jacobi (registers), one lane per point.
// one point: T0 / T1 its two 3x4 camera matrices, pc0 / pc1 / pw entry i of the arrays
__device__ __forceinline__ void triangulate_point(int i, const double *T0, const double *T1, const double *pc0, const double *pc1, double *pw) {
    double D[4][4], V[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        D[0][j] = pc0[3 * i] * T0[2 * 4 + j] - T0[0 * 4 + j];
        D[1][j] = pc0[3 * i + 1] * T0[2 * 4 + j] - T0[1 * 4 + j];
        D[2][j] = pc1[3 * i] * T1[2 * 4 + j] - T1[0 * 4 + j];
        D[3][j] = pc1[3 * i + 1] * T1[2 * 4 + j] - T1[1 * 4 + j];
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) V[a][b] = (a == b) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool changed = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    alpha += D[k][p] * D[k][p];
                    beta += D[k][q] * D[k][q];
                    gamma += D[k][p] * D[k][q];
                }
                if (gamma == 0.0) continue;
                if (fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
                changed     = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t    = 1.0 / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                if (zeta < 0) t = -t;
                double c = 1.0 / sqrt(1.0 + t * t);
                double s = c * t;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    double gp = D[k][p], gq = D[k][q];
                    D[k][p] = c * gp - s * gq;
                    D[k][q] = s * gp + c * gq;
                    double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - s * vq;
                    V[k][q] = s * vp + c * vq;
                }
            }
        if (!changed) break;
    }
    int best  = 0;
    double bn = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) s += D[k][j] * D[k][j];
        if (j == 0 || s < bn) {
            bn   = s;
            best = j;
        }
    }
    double v0 = V[0][0], v1 = V[1][0], v2 = V[2][0], v3 = V[3][0];
#pragma unroll
    for (int j = 1; j < 4; j++)
        if (best == j) {
            v0 = V[0][j];
            v1 = V[1][j];
            v2 = V[2][j];
            v3 = V[3][j];
        }
    pw[3 * i]     = v0 / v3;
    pw[3 * i + 1] = v1 / v3;
    pw[3 * i + 2] = v2 / v3;
}

__global__ void k_triangulate(int n, const int32_t *T0_idx, const int32_t *T1_idx, const double *Tcw12, const double *pc0,
                              const double *pc1, double *pw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    triangulate_point(i, Tcw12 + 12 * (size_t) T0_idx[i], Tcw12 + 12 * (size_t) T1_idx[i], pc0, pc1, pw);
}

// segmented form (device-resident tracker): stream s owns entries [s * seg_cap, s * seg_cap + count[s]) of the point arrays and the
// camera matrices [s * tcw_cap, ...) — its T0 / T1 indices are local to that table
__global__ void k_triangulate_seg(int n_seg, int seg_cap, const int32_t *count, const int32_t *T0_idx, const int32_t *T1_idx, int tcw_cap,
                                  const double *Tcw12, const double *pc0, const double *pc1, double *pw) {
    const int s = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg || k >= count[s]) return;
    const int i      = s * seg_cap + k;
    const double *T  = Tcw12 + 12 * (size_t) s * tcw_cap;
    triangulate_point(i, T + 12 * (size_t) T0_idx[i], T + 12 * (size_t) T1_idx[i], pc0, pc1, pw);
}

extern "C" int icg_triangulate(icg_ctx *ctx, int n, const int32_t *T0_idx, const int32_t *T1_idx, int n_T,
                               const double *Tcw12, const double *pc0, const double *pc1, double *pw) {
    if (!ctx || n < 0) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (!T0_idx || !T1_idx || !Tcw12 || !pc0 || !pc1 || !pw || n_T <= 0) return ICG_ERR_INVALID;
    if (n > ctx->cfg.max_points) return icg_fail(ctx, ICG_ERR_CAPACITY, "%d points > max_points %d", n, ctx->cfg.max_points);
    for (int i = 0; i < n; i++)
        if (T0_idx[i] < 0 || T0_idx[i] >= n_T || T1_idx[i] < 0 || T1_idx[i] >= n_T)
            return icg_fail(ctx, ICG_ERR_INVALID, "pose index out of range at point %d", i);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    int rc = c.reserve((size_t) n * (8 + 48 + 24) + (size_t) n_T * 96);
    if (rc) return rc;
    const int32_t *d_i0 = c.in_zc(T0_idx, (size_t) n);
    const int32_t *d_i1 = c.in_zc(T1_idx, (size_t) n);
    const double *d_T   = c.in_zc(Tcw12, 12 * (size_t) n_T);
    const double *d_p0  = c.in_zc(pc0, 3 * (size_t) n);
    const double *d_p1  = c.in_zc(pc1, 3 * (size_t) n);
    double *d_pw = c.out_zc(pw, 3 * (size_t) n);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "triangulate");
        hipLaunchKernelGGL(k_triangulate, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, d_i0, d_i1, d_T, d_p0, d_p1, d_pw);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

// ---------------------------------------------------------------------------------------------------------
// Device-resident tracker (tracker.hip): cv::findFundamentalMat(FM_RANSAC) of one point set per stream, the WHOLE run in one launch — the
// host loop of icg_fm_ransac (subset draws, one launch per round, sequential replay of the best / niters recurrence) moves into the
// workgroup that owns the set:
//   round:  wave 0 draws up to FM_HPW subsets from the set's cv::RNG (getSubset + checkSubset with a collinearity test per lane,
//           RNG-consuming redraws) -> LDS
//           wave 0, a lane per hypothesis: seven-point solve (the serial FP64 chain of k_fm_hypothesis) -> models in LDS
//           four waves, a wave per (hypothesis, model): inlier bits by ballot + count -> LDS
//           thread 0 replays the hypotheses in order: best mask, max_good, niters = RANSACUpdateNumIters(...)
// Any chunking of the hypothesis stream gives the sequential algorithm's result (scores do not depend on one another; hypotheses past
// the updated niters are discarded) — the argument of icg_fm_ransac.  RANSACUpdateNumIters needs log and pow: their values for every
// (set size, inlier count) come from a table the HOST computed with the same libm calls the host path makes (fm_denom_table), so the
// iteration counts are the host path's bit for bit; the comparisons, the division and lrint run on the device in IEEE double.
__device__ __forceinline__ unsigned dev_rng_next(unsigned long long &state) {
    state = (unsigned long long) (unsigned) state * 4164903690U + (unsigned) (state >> 32);
    return (unsigned) state;
}
// getSubset + checkSubset of ONE hypothesis by a whole wave (round 5).  The draws (cv::RNG, duplicate rejection) are uniform integer work that
// every lane repeats; checkSubset's collinearity test — the last point against the 15 pairs of the other six, in both point sets: 30
// independent FP64 tests of calib3d's haveCollinearPoints(m, 7), an OR — runs a test per lane (lanes 30..63 repeat tests of lanes 0..29: every lane stays active, because FP64
// and the other half-rate VALU instructions take ~5x as long on gfx950 when 16 or fewer lanes are active — under `if (t == 0)` the
// up-to-32 subsets of a round cost more than the seven-point solves and the scoring together; profiles/ubench/valu_cost_r05.txt).
// idx_lds: the hypothesis' seven indices in LDS (written here, read by the solve).  Same subsets, same RNG consumption as dev_get_subset's
// sequential form (k_fm_hypothesis' host twin draws on the host).
__device__ bool dev_get_subset_wave(unsigned long long &rng, int n, const float2 *p1, const float2 *p2, int *idx_lds, int lane) {
    const int pr = lane % 15;
    const int j  = pr < 1 ? 1 : pr < 3 ? 2 : pr < 6 ? 3 : pr < 10 ? 4 : 5; // pairs (j, k), k < j < 6, in the order of haveCollinearPoints
    const int k  = pr - j * (j - 1) / 2;
    const float2 *pts = ((lane / 15) & 1) ? p2 : p1;
    for (int attempt = 0; attempt < 10000; attempt++) {
        int idx[7];
        for (int i = 0; i < 7; i++) {
            int v;
            for (;;) {
                v = (int) (dev_rng_next(rng) % (unsigned) n); // RNG::uniform(0, n), n > 0
                bool dup = false;
                for (int q = 0; q < i; q++) dup |= idx[q] == v;
                if (!dup) break;
            }
            idx[i] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // (the previous attempt's readers are done: one wave, LDS queue in order)
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 7; q++) idx_lds[q] = idx[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const float2 a = pts[idx_lds[6]], b = pts[idx_lds[j]], c = pts[idx_lds[k]];
        const double dx1 = b.x - a.x, dy1 = b.y - a.y; // (float differences widened, as the CPU form: pts[..].x - pts[..].x in float, then double)
        const double dx2 = c.x - a.x, dy2 = c.y - a.y;
        const bool bad   = fabs(dx2 * dy1 - dy2 * dx1) <= (double) FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
        if (__ballot(bad) == 0ull) return true;
    }
    return false;
}

#define FMS_MAX_WORDS 16 // 64-bit inlier words per model: sets of up to 1024 points

__global__ __launch_bounds__(256, 1) void k_fm_ransac_sets(int n_sets, int seg_cap, const int32_t *count, const float2 *pts1, const float2 *pts2,
                                                           float thresh2, double log_num, const double *denom_tab, int tab_n, uint8_t *mask) {
    __shared__ double Fm[FM_HPW][27];
    __shared__ int n_sh[FM_HPW];
    __shared__ int idx_sh[FM_HPW][7];
    __shared__ int good_sh[FM_HPW * 3];
    __shared__ unsigned long long bits_sh[FM_HPW * 3][FMS_MAX_WORDS];
    __shared__ unsigned long long best_sh[FMS_MAX_WORDS];
    __shared__ int nh_sh, iter_sh, niters_sh, max_good_sh;
    const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if (s >= n_sets) return;
    const int n = count[s];
    if (n <= 0) return; // no set for this stream in this step
    const float2 *p1 = pts1 + (size_t) s * seg_cap, *p2 = pts2 + (size_t) s * seg_cap;
    uint8_t *m       = mask + (size_t) s * seg_cap;
    if (n < 15 || n > tab_n || n > 64 * FMS_MAX_WORDS) { // the reference only calls findFundamentalMat with >= 15 points: leave untouched
        for (int i = t; i < n; i += 256) m[i] = 1;
        return;
    }
    const int words = (n + 63) >> 6;
    fm_set S;
    S.pt_begin = 0, S.n_pts = n, S.hyp_begin = 0, S.n_hyp = 0, S.word_begin = 0, S.words_per_model = words;
    unsigned long long rng = 0xffffffffffffffffull; // cv::RNG((uint64) -1): the fixed seed of every findFundamentalMat call
    if (t == 0) iter_sh = 0, niters_sh = 1000, max_good_sh = 0;
    if (t < FMS_MAX_WORDS) best_sh[t] = 0;
    __syncthreads();
    for (;;) {
        if (wave == 0) { // all 64 lanes: see dev_get_subset_wave
            const int it0 = iter_sh;
            int nh        = min(FM_HPW, niters_sh - it0);
            for (int h = 0; h < nh; h++) {
                if (!dev_get_subset_wave(rng, n, p1, p2, idx_sh[h], lane)) {
                    // ptsetreg.cpp run(): no valid subset -> the iterations end here (nothing found if this was the first one)
                    if (lane == 0) niters_sh = it0 + h;
                    nh = h;
                    break;
                }
            }
            if (lane == 0) nh_sh = nh;
        }
        __syncthreads();
        const int nh = nh_sh;
        if (nh <= 0) break;
        if (wave == 0 && lane < nh) {
            int nm = 0;
            seven_point_solve(S, idx_sh[lane], p1, p2, Fm[lane], &nm);
            n_sh[lane] = nm;
        }
        __syncthreads();
        for (int pair = wave; pair < 3 * nh; pair += 4) {
            const int hl = pair / 3, model = pair - 3 * hl;
            if (model >= n_sh[hl]) {
                if (lane == 0) good_sh[pair] = -1;
                continue;
            }
            const double *F = Fm[hl] + 9 * model;
            const double F0 = F[0], F1 = F[1], F2 = F[2], F3 = F[3], F4 = F[4], F5 = F[5], F6 = F[6], F7 = F[7], F8 = F[8];
            int cnt = 0;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                bool in     = false;
                if (i < n) {
                    const float2 a0 = p1[i], b0 = p2[i];
                    const double x1 = a0.x, y1 = a0.y, x2 = b0.x, y2 = b0.y;
                    double a = F0 * x1 + F1 * y1 + F2;
                    double b = F3 * x1 + F4 * y1 + F5;
                    double c = F6 * x1 + F7 * y1 + F8;
                    double s2 = 1. / (a * a + b * b);
                    double d2 = x2 * a + y2 * b + c;
                    a         = F0 * x2 + F3 * y2 + F6;
                    b         = F1 * x2 + F4 * y2 + F7;
                    c         = F2 * x2 + F5 * y2 + F8;
                    double s1 = 1. / (a * a + b * b);
                    double d1 = x1 * a + y1 * b + c;
                    float e   = (float) fmax(d1 * d1 * s1, d2 * d2 * s2);
                    in        = e <= thresh2;
                }
                const unsigned long long mm = __ballot(in);
                if (lane == 0) bits_sh[pair][base >> 6] = mm;
                cnt += __popcll(mm);
            }
            if (lane == 0) good_sh[pair] = cnt;
        }
        __syncthreads();
        if (t == 0) { // sequential replay of RANSACPointSetRegistrator::run over the scores (ptsetreg.cpp)
            int iter = iter_sh, niters = niters_sh, max_good = max_good_sh;
            for (int h = 0; h < nh && iter < niters; h++, iter++) {
                for (int mdl = 0; mdl < 3; mdl++) {
                    const int good = good_sh[h * 3 + mdl];
                    if (good < 0) break;
                    if (good > max(max_good, 7 - 1)) {
                        for (int w = 0; w < words; w++) best_sh[w] = bits_sh[h * 3 + mdl][w];
                        max_good = good;
                        // RANSACUpdateNumIters(conf, (n - good) / n, 7, niters) with log(1 - (1 - ep)^7) from the host's table
                        const double denom = denom_tab[(size_t) n * (tab_n + 1) + good];
                        if (denom != denom) { // (NaN marks 1 - (1 - ep)^7 < DBL_MIN: "return 0")
                            niters = 0;
                        } else {
                            niters = (denom >= 0 || -log_num >= niters * (-denom)) ? niters : (int) lrint(log_num / denom);
                        }
                    }
                }
            }
            iter_sh = iter, niters_sh = niters, max_good_sh = max_good;
        }
        __syncthreads();
        // (the stop test is read into a register by every thread BEFORE thread 0 may write niters_sh at the top of the next round)
        const bool stop = iter_sh >= niters_sh;
        __syncthreads();
        if (stop) break;
    }
    const bool found = max_good_sh > 0;
    for (int i = t; i < n; i += 256) m[i] = found ? (uint8_t) ((best_sh[i >> 6] >> (i & 63)) & 1ull) : (uint8_t) 0;
}

// log(1 - (1 - ep)^7) for ep = (n - good) / n, n = 0..tab_n, good = 0..n, with the libm calls of ransac_update_num_iters (NaN where that
// function returns 0 before taking the logarithm): one table per process and device, shared by every tracker
static int fm_denom_table(icg_ctx *ctx, int tab_n, const double **d_tab) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, double *> tabs;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(ctx->cfg.device, tab_n);
    auto it  = tabs.find(key);
    if (it == tabs.end()) {
        std::vector<double> h((size_t) (tab_n + 1) * (tab_n + 1), 0.0);
        for (int n = 1; n <= tab_n; n++)
            for (int good = 0; good <= n; good++) {
                double ep = (double) (n - good) / n;
                ep        = std::min(std::max(ep, 0.), 1.);
                double denom = 1. - std::pow(1. - ep, 7);
                h[(size_t) n * (tab_n + 1) + good] = denom < DBL_MIN ? std::nan("") : std::log(denom);
            }
        double *d = nullptr;
        ICG_HIP(ctx, hipMalloc((void **) &d, h.size() * sizeof(double)));
        ICG_HIP(ctx, hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        it = tabs.emplace(key, d).first;
    }
    *d_tab = it->second;
    return 0;
}

int icg_fm_ransac_launch_sets(icg_ctx *ctx, int n_sets, int seg_cap, const int32_t *d_count, const float2 *d_p1, const float2 *d_p2, double thresh,
                              double conf, uint8_t *d_mask) {
    if (thresh <= 0) thresh = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    const int tab_n    = std::min(seg_cap, 64 * FMS_MAX_WORDS);
    const double *d_tab = nullptr;
    int rc = fm_denom_table(ctx, tab_n, &d_tab);
    if (rc) return rc;
    double p         = std::min(std::max(conf, 0.), 1.);
    const double num = std::log(std::max(1. - p, DBL_MIN)); // the numerator of RANSACUpdateNumIters
    icg_prof_scope ps(ctx, "fm_ransac_sets");
    hipLaunchKernelGGL(k_fm_ransac_sets, dim3(n_sets), dim3(256), 0, ctx->stream, n_sets, seg_cap, d_count, d_p1, d_p2, (float) (thresh * thresh), num, d_tab,
                       tab_n, d_mask);
    ICG_HIP(ctx, hipGetLastError());
    return ICG_OK;
}

int icg_triangulate_launch_segments(icg_ctx *ctx, int n_seg, int seg_cap, const int32_t *d_count, const int32_t *d_T0, const int32_t *d_T1, int tcw_cap,
                                    const double *d_Tcw, const double *d_pc0, const double *d_pc1, double *d_pw) {
    icg_prof_scope ps(ctx, "triangulate");
    hipLaunchKernelGGL(k_triangulate_seg, dim3((seg_cap + 63) / 64, n_seg), dim3(64), 0, ctx->stream, n_seg, seg_cap, d_count, d_T0, d_T1, tcw_cap, d_Tcw,
                       d_pc0, d_pc1, d_pw);
    ICG_HIP(ctx, hipGetLastError());
    return ICG_OK;
}

// The same result as icg_fm_ransac with ONE launch and ONE wait (no host round per RANSAC chunk): the whole run of every set inside the
// workgroup that owns it (k_fm_ransac_sets).  Sets of more than 64 * FMS_MAX_WORDS points fall back to icg_fm_ransac.
extern "C" int icg_fm_ransac_device(icg_ctx *ctx, int n_sets, const int32_t *offsets, const float *pts1, const float *pts2, double thresh, double conf,
                                    uint8_t *mask) {
    if (!ctx || n_sets < 0) return ICG_ERR_INVALID;
    if (n_sets == 0) return ICG_OK;
    if (!offsets || !pts1 || !pts2 || !mask) return ICG_ERR_INVALID;
    int seg_cap = 1;
    for (int s = 0; s < n_sets; s++) {
        const int n = offsets[s + 1] - offsets[s];
        if (n < 0) return ICG_ERR_INVALID;
        seg_cap = std::max(seg_cap, n);
    }
    if (seg_cap > 64 * FMS_MAX_WORDS) return icg_fm_ransac(ctx, n_sets, offsets, pts1, pts2, thresh, conf, mask);
    seg_cap = (seg_cap + 63) & ~63;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t cells = (size_t) n_sets * seg_cap;
    std::vector<float> a(2 * cells, 0.f), b(2 * cells, 0.f);
    std::vector<int32_t> cnt((size_t) n_sets);
    for (int s = 0; s < n_sets; s++) {
        const int n = offsets[s + 1] - offsets[s];
        cnt[(size_t) s] = n;
        memcpy(a.data() + 2 * (size_t) s * seg_cap, pts1 + 2 * (size_t) offsets[s], sizeof(float) * 2 * (size_t) n);
        memcpy(b.data() + 2 * (size_t) s * seg_cap, pts2 + 2 * (size_t) offsets[s], sizeof(float) * 2 * (size_t) n);
    }
    std::vector<uint8_t> m(cells, 0);
    icg_call c(ctx);
    int rc = c.reserve(cells * (16 + 1) + sizeof(int32_t) * (size_t) n_sets + 4096);
    if (rc) return rc;
    const float2 *d_p1 = (const float2 *) c.in(a.data(), 2 * cells);
    const float2 *d_p2 = (const float2 *) c.in(b.data(), 2 * cells);
    const int32_t *d_n = c.in(cnt.data(), (size_t) n_sets);
    if ((rc = c.seal())) return rc;
    uint8_t *d_m = c.out(m.data(), cells);
    ICG_LAUNCH_GUARD(c);
    if ((rc = icg_fm_ransac_launch_sets(ctx, n_sets, seg_cap, d_n, d_p1, d_p2, thresh, conf, d_m))) return rc;
    if ((rc = c.finish())) return rc;
    for (int s = 0; s < n_sets; s++) memcpy(mask + offsets[s], m.data() + (size_t) s * seg_cap, (size_t) cnt[(size_t) s]);
    return ICG_OK;
}
