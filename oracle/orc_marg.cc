// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the marginalization prior of the reference:
//   MarginalizationInfo::constructEquation   factors/marginalization_info.h:195-230  (H0 += Ji^T Jj, b0 -= Ji^T e)
//   MarginalizationInfo::schurElimination    :170-192  (Hmm^-1 by eigen-decomposition with 1e-8 floor, Schur complement)
//   MarginalizationInfo::linearization       :153-167  (eig(Hp) -> J0 = S^1/2 V^T, e0 = -S^-1/2 V^T bp)
//   MarginalizationFactor::Evaluate          factors/marginalization_factor.h:47-101
// The reference uses Eigen::SelfAdjointEigenSolver; here a cyclic Jacobi eigen-solver (eigenvalues ascending, like
// Eigen).  J0/e0 depend on the eigenvector basis (sign / degenerate subspaces), so tests compare the basis-invariant
// quantities J0^T J0 = Hp and J0^T e0 = -bp and direct values only where the basis is shared.
// PINNED against the reference's own pipeline compiled unmodified (oracle/ref_build -> oracle/_ref/libref_marg.so,
// tests/golden/marg_ref_golden.npz: Hp, bp, marginalization-factor cost and gradient at a perturbed point).
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

extern "C" {

// Symmetric eigen-decomposition by cyclic Jacobi. A: n x n row-major (symmetric). evals ascending; evecs: n x n row-major,
// column k = eigenvector k. Returns the number of sweeps used.
int orc_sym_eigen(int n, const double *A, double *evals, double *evecs) {
    std::vector<double> a(A, A + (size_t) n * n), v((size_t) n * n, 0.0);
    for (int i = 0; i < n; i++) v[(size_t) i * n + i] = 1.0;
    int sweep = 0;
    for (; sweep < 100; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) (i == j ? diag : off) += a[(size_t) i * n + j] * a[(size_t) i * n + j];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = a[(size_t) p * n + q];
                if (apq == 0.0) continue;
                double app = a[(size_t) p * n + p], aqq = a[(size_t) q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t     = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                if (theta < 0) t = -t;
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) { // columns p,q
                    double akp = a[(size_t) k * n + p], akq = a[(size_t) k * n + q];
                    a[(size_t) k * n + p] = c * akp - s * akq;
                    a[(size_t) k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) { // rows p,q
                    double apk = a[(size_t) p * n + k], aqk = a[(size_t) q * n + k];
                    a[(size_t) p * n + k] = c * apk - s * aqk;
                    a[(size_t) q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = v[(size_t) k * n + p], vkq = v[(size_t) k * n + q];
                    v[(size_t) k * n + p] = c * vkp - s * vkq;
                    v[(size_t) k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t) x * n + x] < a[(size_t) y * n + y]; });
    for (int k = 0; k < n; k++) {
        evals[k] = a[(size_t) order[k] * n + order[k]];
        for (int i = 0; i < n; i++) evecs[(size_t) i * n + k] = v[(size_t) i * n + order[k]];
    }
    return sweep;
}

// constructEquation restricted to reprojection factors (2 residuals; blocks pose_i, pose_j, ext, invdepth, td with local
// sizes 6,6,6,1,1).  r: n x 2, J: n x 46 (already robust-corrected if a loss is used).  col_*: local column of each
// parameter block or -1 when the block is constant / absent.  H0 (local x local) and b0 are ACCUMULATED into.
void orc_reproj_accumulate_normal(int n, const double *r, const double *J, const int32_t *idx_i, const int32_t *idx_j,
                                  const int32_t *idx_lm, const int32_t *col_pose, int col_ext, const int32_t *col_lm, int col_td,
                                  int local_size, double *H0, double *b0) {
    for (int f = 0; f < n; f++) {
        const double *Jf = J + 46 * (size_t) f, *rf = r + 2 * (size_t) f;
        int col[5]       = {col_pose[idx_i[f]], col_pose[idx_j[f]], col_ext, col_lm[idx_lm[f]], col_td};
        const int sz[5]  = {6, 6, 6, 1, 1};
        const int off[5] = {0, 14, 28, 42, 44};
        const int ld[5]  = {7, 7, 7, 1, 1};
        for (int a = 0; a < 5; a++) {
            if (col[a] < 0) continue;
            for (int b = 0; b < 5; b++) {
                if (col[b] < 0) continue;
                for (int x = 0; x < sz[a]; x++)
                    for (int y = 0; y < sz[b]; y++) {
                        double v = Jf[off[a] + x] * Jf[off[b] + y] + Jf[off[a] + ld[a] + x] * Jf[off[b] + ld[b] + y];
                        H0[(size_t) (col[a] + x) * local_size + col[b] + y] += v;
                    }
            }
            for (int x = 0; x < sz[a]; x++) b0[col[a] + x] -= Jf[off[a] + x] * rf[0] + Jf[off[a] + ld[a] + x] * rf[1];
        }
    }
}

// schurElimination + linearization. H0: n_total x n_total (marginalized block first, size m), b0: n_total.
// Outputs (r = n_total - m): J0 r x r, e0 r, Hp r x r, bp r. Returns r.
int orc_marginalize(int n_total, int m, const double *H0, const double *b0, double eps, double *J0, double *e0, double *Hp_out,
                    double *bp_out) {
    const int r = n_total - m;
    auto H = [&](int i, int j) { return H0[(size_t) i * n_total + j]; };
    std::vector<double> Hmm((size_t) m * m), evals(m), evecs((size_t) m * m), Hinv((size_t) m * m, 0.0);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) Hmm[(size_t) i * m + j] = 0.5 * (H(i, j) + H(j, i));
    orc_sym_eigen(m, Hmm.data(), evals.data(), evecs.data());
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) {
                double inv = evals[k] > eps ? 1.0 / evals[k] : 0.0;
                s += evecs[(size_t) i * m + k] * inv * evecs[(size_t) j * m + k];
            }
            Hinv[(size_t) i * m + j] = s;
        }
    // T = Hrm * Hmm^-1  (r x m)
    std::vector<double> T((size_t) r * m);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += H(m + i, k) * Hinv[(size_t) k * m + j];
            T[(size_t) i * m + j] = s;
        }
    std::vector<double> Hp((size_t) r * r), bp(r);
    for (int i = 0; i < r; i++) {
        for (int j = 0; j < r; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += T[(size_t) i * m + k] * H(k, m + j);
            Hp[(size_t) i * r + j] = H(m + i, m + j) - s;
        }
        double s = 0;
        for (int k = 0; k < m; k++) s += T[(size_t) i * m + k] * b0[k];
        bp[i] = b0[m + i] - s;
    }
    if (Hp_out) memcpy(Hp_out, Hp.data(), sizeof(double) * (size_t) r * r);
    if (bp_out) memcpy(bp_out, bp.data(), sizeof(double) * (size_t) r);
    std::vector<double> ev(r), V((size_t) r * r);
    orc_sym_eigen(r, Hp.data(), ev.data(), V.data());
    for (int k = 0; k < r; k++) {
        double S = ev[k] > eps ? ev[k] : 0.0, Sinv = ev[k] > eps ? 1.0 / ev[k] : 0.0;
        double ss = std::sqrt(S), si = std::sqrt(Sinv);
        double vb = 0;
        for (int i = 0; i < r; i++) {
            J0[(size_t) k * r + i] = ss * V[(size_t) i * r + k];
            vb += V[(size_t) i * r + k] * -bp[i];
        }
        e0[k] = si * vb;
    }
    return r;
}

// MarginalizationFactor::Evaluate. block_size: global sizes (7 = pose, local 6); block_index: local start index of each
// retained block minus the marginalized size. x0/x: parameter values concatenated by global size.
// jac (optional): per block a (r x global_size) row-major matrix, blocks concatenated.
void orc_marg_factor_eval(int r, int n_blocks, const int *block_size, const int *block_index, const double *x0_concat,
                          const double *x_concat, const double *J0, const double *e0, double *residuals, double *jac_concat) {
    std::vector<double> dx(r, 0.0);
    int off = 0;
    for (int b = 0; b < n_blocks; b++) {
        int size = block_size[b], index = block_index[b];
        const double *x = x_concat + off, *x0 = x0_concat + off;
        if (size == 7) {
            // dq = q0^-1 * q  (Eigen quaternion product, xyzw storage, ctor order w,x,y,z)
            double n2 = x0[3] * x0[3] + x0[4] * x0[4] + x0[5] * x0[5] + x0[6] * x0[6];
            double ax = -x0[3] / n2, ay = -x0[4] / n2, az = -x0[5] / n2, aw = x0[6] / n2;
            double bx = x[3], by = x[4], bz = x[5], bw = x[6];
            double dqx = aw * bx + ax * bw + ay * bz - az * by;
            double dqy = aw * by + ay * bw + az * bx - ax * bz;
            double dqz = aw * bz + az * bw + ax * by - ay * bx;
            double dqw = aw * bw - ax * bx - ay * by - az * bz;
            for (int k = 0; k < 3; k++) dx[index + k] = x[k] - x0[k];
            double sgn    = dqw < 0 ? -2.0 : 2.0;
            dx[index + 3] = sgn * dqx;
            dx[index + 4] = sgn * dqy;
            dx[index + 5] = sgn * dqz;
        } else {
            for (int k = 0; k < size; k++) dx[index + k] = x[k] - x0[k];
        }
        off += size;
    }
    for (int i = 0; i < r; i++) {
        double s = 0;
        for (int k = 0; k < r; k++) s += J0[(size_t) i * r + k] * dx[k];
        residuals[i] = e0[i] + s;
    }
    if (!jac_concat) return;
    double *out = jac_concat;
    for (int b = 0; b < n_blocks; b++) {
        int size = block_size[b], index = block_index[b], local = size == 7 ? 6 : size;
        for (int i = 0; i < r; i++)
            for (int j = 0; j < size; j++) out[(size_t) i * size + j] = j < local ? J0[(size_t) i * r + index + j] : 0.0;
        out += (size_t) r * size;
    }
}

} // extern "C"
