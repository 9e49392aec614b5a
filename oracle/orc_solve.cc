// ORACLE — TEST INFRASTRUCTURE ONLY.  Dense CPU restatement of the landmark elimination behind one Gauss-Newton / LM step of the
// reference's window optimization (GVINS::gvinsOptimization, ic_gvins.cc:1130-1239: Ceres LEVENBERG_MARQUARDT + DENSE_SCHUR,
// reprojection factors added at :1763-1837 with one 1x1 inverse-depth block per landmark) — SURVEY.md §8 row f1.
// PARITY STATUS: "parity unpinned" — Ceres Solver is an absent third-party dependency (README.md:46, >= 2.0); what is restated is
// the published algorithm (Schur complement of the damped normal equations, LM diagonal clamp(diag(J^T J), 1e-6, 1e32)/radius of
// ceres/internal/levenberg_marquardt_strategy.cc), anchored by algebra instead of golden vectors: the tests check that the
// reduced solve + back-substitution equals a dense solve of the full damped system (numpy), which does not depend on this file.
#include "oracle.h"
#include <cmath>
#include <vector>

extern "C" {

// H: N x N row-major with N = P + L (camera columns first, landmark l at P + l), b: N.  inv (L) receives 1/(h_ll + d_l)
// (0 for a landmark whose row is empty).
void orc_schur_reduce(int P, int L, const double *H, const double *b, double damp, double min_diag, double max_diag, double *S, double *s,
                      double *diag_cc, double *inv) {
    const size_t N = (size_t) P + L;
    for (int l = 0; l < L; l++) {
        double h = H[(P + (size_t) l) * N + P + l];
        inv[l]   = h > 0.0 ? 1.0 / (h + std::fmin(std::fmax(h, min_diag), max_diag) * damp) : 0.0;
    }
    for (int i = 0; i < P; i++) {
        for (int j = 0; j < P; j++) {
            double acc = 0.0;
            for (int l = 0; l < L; l++) acc += H[(P + (size_t) l) * N + i] * inv[l] * H[(P + (size_t) l) * N + j];
            S[(size_t) i * P + j] = H[(size_t) i * N + j] - acc;
        }
        double accs = 0.0;
        for (int l = 0; l < L; l++) accs += H[(P + (size_t) l) * N + i] * inv[l] * b[P + l];
        s[i] = b[i] - accs;
        if (diag_cc) diag_cc[i] = H[(size_t) i * N + i];
    }
}

void orc_schur_backsub(int P, int L, const double *H, const double *b, const double *inv, double damp, double min_diag, double max_diag,
                       const double *delta_c, double *delta_l, double *lm_terms /* 2, may be NULL: sum b_l^2 inv_l, sum d_l delta_l^2 */) {
    const size_t N = (size_t) P + L;
    double t0 = 0.0, t1 = 0.0;
    for (int l = 0; l < L; l++) {
        double acc = 0.0;
        for (int i = 0; i < P; i++) acc += H[(P + (size_t) l) * N + i] * delta_c[i];
        delta_l[l] = (b[P + l] - acc) * inv[l];
        if (inv[l] > 0.0) {
            t0 += b[P + l] * b[P + l] * inv[l];
            t1 += std::fmin(std::fmax(H[(P + (size_t) l) * N + P + l], min_diag), max_diag) * damp * delta_l[l] * delta_l[l];
        }
    }
    if (lm_terms) lm_terms[0] = t0, lm_terms[1] = t1;
}

// 0.5 sum rho(|r|^2) from Huber-corrected residuals (see k_reproj_cost): |r_c|^2 = s (inlier) or a sqrt(s) > a^2 (outlier)
double orc_reproj_cost(int n, const double *r, const uint8_t *active, double huber) {
    double acc = 0.0;
    for (int f = 0; f < n; f++) {
        if (active && !active[f]) continue;
        double q = r[2 * (size_t) f] * r[2 * (size_t) f] + r[2 * (size_t) f + 1] * r[2 * (size_t) f + 1];
        if (huber > 0.0 && q > huber * huber) q = 2.0 * q - huber * huber;
        acc += 0.5 * q;
    }
    return acc;
}
}
