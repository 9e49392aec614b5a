// Host side of the back-end boundary (see factors.h for the reference lines each class mirrors).
#include "factors.h"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace icg {

void HuberLossHip::Evaluate(double s, double rho[3]) const {
    if (s > b_) {
        const double r = std::sqrt(s);
        rho[0]         = 2.0 * a_ * r - b_;
        rho[1]         = std::max(std::numeric_limits<double>::min(), a_ / r);
        rho[2]         = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s;
        rho[1] = 1.0;
        rho[2] = 0.0;
    }
}

// ---- ReprojectionFactor / ReprojectionBatch ---------------------------------------------------------------------------
ReprojectionFactor::ReprojectionFactor(Vector3d pts0, Vector3d pts1, Vector3d vel0, Vector3d vel1, double td0, double td1,
                                       double std) {
    for (int i = 0; i < 3; i++) {
        obs_[i]     = pts0[i];
        obs_[3 + i] = pts1[i];
        obs_[6 + i] = vel0[i];
        obs_[9 + i] = vel1[i];
    }
    obs_[12] = td0;
    obs_[13] = td1;
    obs_[14] = std; // sqrt_info = 1/std on the diagonal (reprojection_factor.h:50-52)
}

bool ReprojectionFactor::Evaluate(const double *const *, double *residuals, double **jacobians) const {
    // The values were computed by ReprojectionBatch::PrepareForEvaluation for the state currently stored in the user
    // parameter arrays (which is what Ceres passes here).  No batch / not prepared -> evaluation failed, loudly.
    if (!batch_ || slot_ < 0 || !batch_->prepared(jacobians != nullptr)) return false;
    const double *r = batch_->residual(slot_);
    residuals[0]    = r[0];
    residuals[1]    = r[1];
    if (jacobians) {
        const double *J = batch_->jacobian(slot_);
        if (jacobians[0]) memcpy(jacobians[0], J, sizeof(double) * 14);
        if (jacobians[1]) memcpy(jacobians[1], J + 14, sizeof(double) * 14);
        if (jacobians[2]) memcpy(jacobians[2], J + 28, sizeof(double) * 14);
        if (jacobians[3]) memcpy(jacobians[3], J + 42, sizeof(double) * 2);
        if (jacobians[4]) memcpy(jacobians[4], J + 44, sizeof(double) * 2);
    }
    return true;
}

ReprojectionBatch::ReprojectionBatch(int device) {
    icg_ctx_config cfg{};
    cfg.device      = device;
    cfg.width       = 64; // the back-end context holds no images
    cfg.height      = 64;
    cfg.n_slots     = 1;
    cfg.max_batch   = 1;
    cfg.max_points  = 64;
    cfg.max_factors = 4096;
    if (icg_ctx_create(&cfg, &ctx_) != ICG_OK) throw std::runtime_error(std::string("ReprojectionBatch: ") + icg_last_error(nullptr));
}

ReprojectionBatch::~ReprojectionBatch() {
    for (auto *f : factors_) {
        f->batch_ = nullptr;
        f->slot_  = -1;
    }
    icg_ctx_destroy(ctx_);
}

void ReprojectionBatch::setWaitMode(int icg_wait_mode, int sleep_us) { (void) icg_ctx_set_wait_mode(ctx_, icg_wait_mode, sleep_us); }

void ReprojectionBatch::clear() {
    for (auto *f : factors_) {
        f->batch_ = nullptr;
        f->slot_  = -1;
    }
    factors_.clear();
    pose_ptrs_.clear();
    lm_ptrs_.clear();
    pose_index_.clear();
    lm_index_.clear();
    idx_i_.clear();
    idx_j_.clear();
    idx_lm_.clear();
    ext_ = td_ = nullptr;
    finalized_ = prepared_ = has_jac_ = false;
}

void ReprojectionBatch::add(ReprojectionFactor *factor, double *pose_i, double *pose_j, double *extrinsic, double *invdepth,
                            double *td) {
    auto index_of = [](std::unordered_map<const double *, int> &m, vector<double *> &v, double *p) {
        auto it = m.find(p);
        if (it != m.end()) return it->second;
        int k = (int) v.size();
        v.push_back(p);
        m[p] = k;
        return k;
    };
    if (ext_ && (ext_ != extrinsic || td_ != td)) throw std::runtime_error("ReprojectionBatch: one extrinsic/td block per batch");
    ext_ = extrinsic;
    td_  = td;
    factor->batch_ = this;
    factor->slot_  = (int) factors_.size();
    factors_.push_back(factor);
    idx_i_.push_back(index_of(pose_index_, pose_ptrs_, pose_i));
    idx_j_.push_back(index_of(pose_index_, pose_ptrs_, pose_j));
    idx_lm_.push_back(index_of(lm_index_, lm_ptrs_, invdepth));
    finalized_ = prepared_ = false;
}

void ReprojectionBatch::finalize() {
    const int n = (int) factors_.size();
    if (n == 0) { // a window without visual factors (the reference solves those too: GNSS / IMU only)
        finalized_ = true;
        prepared_  = false;
        return;
    }
    vector<double> obs((size_t) 15 * n);
    for (int k = 0; k < n; k++)
        for (int c = 0; c < 15; c++) obs[(size_t) c * n + k] = factors_[(size_t) k]->obs_[c];
    if (icg_reproj_set_factors(ctx_, n, obs.data(), idx_i_.data(), idx_j_.data(), idx_lm_.data()) != ICG_OK)
        throw std::runtime_error(std::string("icg_reproj_set_factors: ") + icg_last_error(ctx_));
    finalized_ = true;
    prepared_  = false;
}

bool ReprojectionBatch::run(bool want_jac, double huber, bool fetch) {
    prepared_ = false;
    if (factors_.empty()) {
        prepared_ = has_jac_ = true;
        return true;
    }
    if (!finalized_) finalize();
    vector<double> poses(7 * pose_ptrs_.size()), inv(lm_ptrs_.size());
    for (size_t k = 0; k < pose_ptrs_.size(); k++) memcpy(&poses[7 * k], pose_ptrs_[k], sizeof(double) * 7);
    for (size_t k = 0; k < lm_ptrs_.size(); k++) inv[k] = *lm_ptrs_[k];
    r_view_ = J_view_ = nullptr;
    int rc = fetch ? icg_reproj_eval_resident_view(ctx_, (int) pose_ptrs_.size(), poses.data(), ext_, (int) lm_ptrs_.size(), inv.data(), *td_,
                                                   want_jac ? 1 : 0, huber, &r_view_, &J_view_)
                   : icg_reproj_eval_resident(ctx_, (int) pose_ptrs_.size(), poses.data(), ext_, (int) lm_ptrs_.size(), inv.data(), *td_,
                                              want_jac ? 1 : 0, huber, nullptr, nullptr);
    if (rc != ICG_OK) {
        error_ = icg_last_error(ctx_);
        return false;
    }
    prepared_ = fetch;
    has_jac_  = fetch && want_jac;
    return true;
}

void ReprojectionBatch::PrepareForEvaluation(bool evaluate_jacobians, bool /*new_evaluation_point*/) { run(evaluate_jacobians, 0.0); }

// (the marginalization never reads a factor's slice: the batch is assembled on the device, nothing is fetched)
bool ReprojectionBatch::evaluateCorrected(double huber_delta) { return run(true, huber_delta, false); }

bool ReprojectionBatch::accumulateNormal(const std::unordered_map<const double *, int> &column_of, int local_size, double *H0,
                                         double *b0) {
    if (factors_.empty()) return true;
    auto col = [&](const double *p) {
        auto it = column_of.find(p);
        return it == column_of.end() ? -1 : it->second;
    };
    vector<int32_t> cp(pose_ptrs_.size()), cl(lm_ptrs_.size());
    for (size_t k = 0; k < pose_ptrs_.size(); k++) cp[k] = col(pose_ptrs_[k]);
    for (size_t k = 0; k < lm_ptrs_.size(); k++) cl[k] = col(lm_ptrs_[k]);
    prepared_ = false; // (any further call on the context invalidates the fetched views)
    int rc = icg_reproj_accumulate_normal(ctx_, local_size, cp.data(), col(ext_), cl.data(), col(td_), H0, b0);
    if (rc != ICG_OK) {
        error_ = icg_last_error(ctx_);
        return false;
    }
    return true;
}

bool ReprojectionBatch::accumulateLandmarkEliminated(const std::unordered_map<const double *, int> &camera_column_of, int P, double *H,
                                                     double *b, double *min_hll) {
    if (factors_.empty()) return true;
    auto col = [&](const double *p) {
        auto it = camera_column_of.find(p);
        return it == camera_column_of.end() ? -1 : it->second;
    };
    vector<int32_t> cp(pose_ptrs_.size());
    for (size_t k = 0; k < pose_ptrs_.size(); k++) cp[k] = col(pose_ptrs_[k]);
    vector<double> S((size_t) P * P), s((size_t) P), hll(lm_ptrs_.size());
    prepared_ = false;
    int rc = icg_reproj_schur(ctx_, P, cp.data(), col(ext_), col(td_), nullptr, 1, 0.0, 0.0, 0.0, S.data(), s.data(), nullptr, nullptr);
    if (rc == ICG_OK) rc = icg_reproj_landmark_diag(ctx_, hll.data());
    if (rc != ICG_OK) {
        error_ = icg_last_error(ctx_);
        return false;
    }
    for (size_t k = 0; k < S.size(); k++) H[k] += S[k];
    for (int k = 0; k < P; k++) b[(size_t) k] += s[(size_t) k];
    double mn = hll.empty() ? 0.0 : hll[0];
    for (double v : hll) mn = std::min(mn, v);
    if (min_hll) *min_hll = mn;
    return true;
}

// ---- ResidualBlockInfo (generic host path) ------------------------------------------------------------------------------
bool ResidualBlockInfo::Evaluate() {
    const int nr = cost_function_->num_residuals();
    residuals_.assign((size_t) nr, 0.0);
    const vector<int32_t> &block_sizes = cost_function_->parameter_block_sizes();
    vector<double *> raw(block_sizes.size());
    jacobians_.resize(block_sizes.size());
    for (size_t i = 0; i < block_sizes.size(); i++) {
        jacobians_[i].assign((size_t) nr * block_sizes[i], 0.0);
        raw[i] = jacobians_[i].data();
    }
    if (!cost_function_->Evaluate(parameter_blocks_.data(), residuals_.data(), raw.data())) return false;
    if (loss_function_) { // Ceres corrector (residual_block_info.h:59-87)
        double sq_norm = 0, rho[3];
        for (double v : residuals_) sq_norm += v * v;
        loss_function_->Evaluate(sq_norm, rho);
        const double sqrt_rho1 = std::sqrt(rho[1]);
        double residual_scaling, alpha_sq_norm;
        if ((sq_norm == 0.0) || (rho[2] <= 0.0)) {
            residual_scaling = sqrt_rho1;
            alpha_sq_norm    = 0.0;
        } else {
            const double D     = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
            const double alpha = 1.0 - std::sqrt(D);
            residual_scaling   = sqrt_rho1 / (1 - alpha);
            alpha_sq_norm      = alpha / sq_norm;
        }
        for (size_t i = 0; i < jacobians_.size(); i++) {
            const int nc = block_sizes[i];
            for (int c = 0; c < nc; c++) {
                double rtj = 0;
                for (int k = 0; k < nr; k++) rtj += residuals_[(size_t) k] * jacobians_[i][(size_t) k * nc + c];
                for (int k = 0; k < nr; k++) {
                    double &j = jacobians_[i][(size_t) k * nc + c];
                    j         = sqrt_rho1 * (j - alpha_sq_norm * residuals_[(size_t) k] * rtj);
                }
            }
        }
        for (double &v : residuals_) v *= residual_scaling;
    }
    return true;
}

// ---- symmetric eigen-solver ------------------------------------------------------------------------------------------
// Householder tridiagonalisation followed by the implicit-shift QL iteration (the classic EISPACK tred2 / tql2 pair): ~(4/3 + 3) n^3
// flops.  The cyclic Jacobi solver used before needed ~40 ms for the 133 x 133 and 61 x 61 matrices of one C2 marginalization (its
// off-diagonal norm never reached the 1e-30 threshold, so every call ran the full 100 sweeps); this one takes ~1 ms for both.
// evecs is row-major n x n with eigenvector k in COLUMN k; evals ascending.
// Round 5: the same arithmetic in the same order (outputs bit-identical to the plain form on every matrix tried: full rank, rank-deficient,
// diagonal, zero rows, zero, badly scaled; n = 1 ... 207), 2.4 x faster at n = 142 — the working matrix is held transposed from the start
// (every inner loop contiguous), the dependent sums of tred2 / of the accumulation run four / eight columns' chains at a time (each chain in
// its own order), and the element-wise loops are vectorised.
// The element-wise inner loops of symmetricEigen: no reductions, so their vector forms do exactly the scalar arithmetic (no FMA in either
// clone: "avx2" does not include it, and contraction is off).  The AVX2 clone is picked at load time where the CPU has it.
// (Sanitizer builds — tests/tsan/run.sh — take the plain build: an instrumented ifunc resolver runs before the sanitizer's runtime is up.)
#if defined(__SANITIZE_THREAD__) || defined(__SANITIZE_ADDRESS__)
#define ICG_CLONES __attribute__((optimize("O3", "fp-contract=off")))
#else
#define ICG_CLONES __attribute__((target_clones("avx2", "default"), optimize("O3", "fp-contract=off")))
#endif
// Givens rotation of two rows (tql2 "accumulate"): element-wise, no reduction -> the vector form does the scalar form's arithmetic
ICG_CLONES static void eig_rotate(double *__restrict vi, double *__restrict vi1, int n, double c, double s) {
    for (int k = 0; k < n; k++) {
        const double h = vi1[k], a = vi[k];
        vi1[k] = s * a + c * h;
        vi[k]  = c * a - s * h;
    }
}
// col[k] -= f * e[k] + g * d[k], k in [k0, k1)
ICG_CLONES static void eig_rank2(double *__restrict col, const double *__restrict e, const double *__restrict d, int k0, int k1, double f, double g) {
    for (int k = k0; k < k1; k++) col[k] -= (f * e[k] + g * d[k]);
}
// col[k] -= g * d[k], k in [0, k1)
ICG_CLONES static void eig_axpy(double *__restrict col, const double *__restrict d, int k1, double g) {
    for (int k = 0; k < k1; k++) col[k] -= g * d[k];
}

void symmetricEigen(int n, const vector<double> &A, vector<double> &evals, vector<double> &evecs) {
    if (n == 0) {
        evals.clear(), evecs.clear();
        return;
    }
    // W holds the working matrix TRANSPOSED: v(i, j) = W[j * n + i] — every inner loop of the three phases then walks memory contiguously
    vector<double> W((size_t) n * n), d((size_t) n), e((size_t) n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) W[(size_t) j * n + i] = A[(size_t) i * n + j];
    auto v   = [&](int i, int j) -> double & { return W[(size_t) j * n + i]; };
    auto col = [&](int j) -> double * { return &W[(size_t) j * n]; };
    double *dp = d.data(), *ep = e.data();
    // ---- tridiagonalise
    for (int j = 0; j < n; j++) dp[j] = v(n - 1, j);
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale += std::fabs(dp[k]);
        if (scale == 0.0) {
            ep[i] = dp[i - 1];
            for (int j = 0; j < i; j++) {
                dp[j]   = v(i - 1, j);
                v(i, j) = 0.0;
                v(j, i) = 0.0;
            }
        } else {
            for (int k = 0; k < i; k++) {
                dp[k] /= scale;
                h += dp[k] * dp[k];
            }
            double f = dp[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            ep[i]     = scale * g;
            h         = h - f * g;
            dp[i - 1] = f - g;
            for (int j = 0; j < i; j++) ep[j] = 0.0;
            double *ci = col(i);
            for (int j = 0; j < i; j++) ci[j] = dp[j]; // v(j, i) = d[j]
            // e[k] += v(k, j) d[j] (k > j) and g_j = e[j] + v(j, j) d[j] + sum_{k > j} v(k, j) d[k]: every sum in the order the one-column
            // loop adds it, four columns' (independent) chains in flight
            int j = 0;
            for (; j + 4 <= i; j += 4) {
                const double *c0 = col(j), *c1 = col(j + 1), *c2 = col(j + 2), *c3 = col(j + 3);
                const double f0 = dp[j], f1 = dp[j + 1], f2 = dp[j + 2], f3 = dp[j + 3];
                double g0 = ep[j] + c0[j] * f0;
                g0 += c0[j + 1] * dp[j + 1], ep[j + 1] += c0[j + 1] * f0;
                g0 += c0[j + 2] * dp[j + 2], ep[j + 2] += c0[j + 2] * f0;
                g0 += c0[j + 3] * dp[j + 3], ep[j + 3] += c0[j + 3] * f0;
                double g1 = ep[j + 1] + c1[j + 1] * f1;
                g1 += c1[j + 2] * dp[j + 2], ep[j + 2] += c1[j + 2] * f1;
                g1 += c1[j + 3] * dp[j + 3], ep[j + 3] += c1[j + 3] * f1;
                double g2 = ep[j + 2] + c2[j + 2] * f2;
                g2 += c2[j + 3] * dp[j + 3], ep[j + 3] += c2[j + 3] * f2;
                double g3 = ep[j + 3] + c3[j + 3] * f3;
                for (int k = j + 4; k < i; k++) {
                    const double dk = dp[k];
                    double ek = ep[k];
                    g0 += c0[k] * dk, ek += c0[k] * f0;
                    g1 += c1[k] * dk, ek += c1[k] * f1;
                    g2 += c2[k] * dk, ek += c2[k] * f2;
                    g3 += c3[k] * dk, ek += c3[k] * f3;
                    ep[k] = ek;
                }
                ep[j] = g0, ep[j + 1] = g1, ep[j + 2] = g2, ep[j + 3] = g3;
            }
            for (; j < i; j++) {
                const double *cj = col(j);
                f = dp[j];
                g = ep[j] + cj[j] * f;
                for (int k = j + 1; k <= i - 1; k++) {
                    g += cj[k] * dp[k];
                    ep[k] += cj[k] * f;
                }
                ep[j] = g;
            }
            f = 0.0;
            for (int jj = 0; jj < i; jj++) {
                ep[jj] /= h;
                f += ep[jj] * dp[jj];
            }
            const double hh = f / (h + h);
            for (int jj = 0; jj < i; jj++) ep[jj] -= hh * dp[jj];
            for (int jj = 0; jj < i; jj++) {
                f = dp[jj];
                g = ep[jj];
                double *cj = col(jj);
                eig_rank2(cj, ep, dp, jj, i, f, g);
                // (a later column jj' reads d[k] for k >= jj' > jj only: d[jj] is free to take the next step's value now)
                dp[jj]  = cj[i - 1];
                cj[i]   = 0.0;
            }
        }
        dp[i] = h;
    }
    // ---- accumulate the transformations
    for (int i = 0; i < n - 1; i++) {
        v(n - 1, i) = v(i, i);
        v(i, i)     = 1.0;
        const double h = dp[i + 1];
        if (h != 0.0) {
            const double *u = col(i + 1);
            for (int k = 0; k <= i; k++) dp[k] = u[k] / h;
            int j = 0;
            for (; j + 8 <= i + 1; j += 8) { // eight columns' (independent) sums in flight, each in its own order
                double *c0 = col(j), *c1 = col(j + 1), *c2 = col(j + 2), *c3 = col(j + 3), *c4 = col(j + 4), *c5 = col(j + 5), *c6 = col(j + 6), *c7 = col(j + 7);
                double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0, g4 = 0.0, g5 = 0.0, g6 = 0.0, g7 = 0.0;
                for (int k = 0; k <= i; k++) {
                    const double uk = u[k];
                    g0 += uk * c0[k], g1 += uk * c1[k], g2 += uk * c2[k], g3 += uk * c3[k];
                    g4 += uk * c4[k], g5 += uk * c5[k], g6 += uk * c6[k], g7 += uk * c7[k];
                }
                eig_axpy(c0, dp, i + 1, g0), eig_axpy(c1, dp, i + 1, g1), eig_axpy(c2, dp, i + 1, g2), eig_axpy(c3, dp, i + 1, g3);
                eig_axpy(c4, dp, i + 1, g4), eig_axpy(c5, dp, i + 1, g5), eig_axpy(c6, dp, i + 1, g6), eig_axpy(c7, dp, i + 1, g7);
            }
            for (; j <= i; j++) {
                double *cj = col(j);
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += u[k] * cj[k];
                eig_axpy(cj, dp, i + 1, g);
            }
        }
        double *u = col(i + 1);
        for (int k = 0; k <= i; k++) u[k] = 0.0;
    }
    for (int j = 0; j < n; j++) {
        dp[j]       = v(n - 1, j);
        v(n - 1, j) = 0.0;
    }
    v(n - 1, n - 1) = 1.0;
    ep[0]           = 0.0;
    // ---- implicit QL on the tridiagonal matrix; the Givens rotations act on two columns of V = two rows of W
    for (int i = 1; i < n; i++) ep[i - 1] = ep[i];
    ep[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        tst1  = std::max(tst1, std::fabs(dp[l]) + std::fabs(ep[l]));
        int m = l;
        while (m < n) {
            if (std::fabs(ep[m]) <= eps * tst1) break;
            m++;
        }
        if (m > l) {
            int iter = 0;
            do {
                iter++;
                double g = dp[l];
                double p = (dp[l + 1] - g) / (2.0 * ep[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                dp[l]     = ep[l] / (p + r);
                dp[l + 1] = ep[l] * (p + r);
                const double dl1 = dp[l + 1];
                double h         = g - dp[l];
                for (int i = l + 2; i < n; i++) dp[i] -= h;
                f += h;
                p        = dp[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = ep[l + 1];
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g  = c * ep[i];
                    h  = c * p;
                    r  = std::hypot(p, ep[i]);
                    ep[i + 1] = s * r;
                    s         = ep[i] / r;
                    c         = p / r;
                    p         = c * dp[i] - s * g;
                    dp[i + 1] = h + s * (c * g + s * dp[i]);
                    eig_rotate(col(i), col(i + 1), n, c, s);
                }
                p     = -s * s2 * c3 * el1 * ep[l] / dl1;
                ep[l] = s * p;
                dp[l] = c * p;
            } while (std::fabs(ep[l]) > eps * tst1 && iter < 60);
        }
        dp[l] = dp[l] + f;
        ep[l] = 0.0;
    }
    vector<int> order((size_t) n);
    for (int i = 0; i < n; i++) order[(size_t) i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return dp[x] < dp[y]; });
    evals.assign((size_t) n, 0.0);
    evecs.assign((size_t) n * n, 0.0);
    for (int k = 0; k < n; k++) {
        evals[(size_t) k] = dp[order[(size_t) k]];
        const double *src = col(order[(size_t) k]);
        for (int i = 0; i < n; i++) evecs[(size_t) i * n + k] = src[i];
    }
}

// ---- MarginalizationInfo ------------------------------------------------------------------------------------------------
MarginalizationInfo::~MarginalizationInfo() {
    for (auto &block : parameter_block_data_) delete[] block.second;
}

void MarginalizationInfo::addResidualBlockInfo(const std::shared_ptr<ResidualBlockInfo> &blockinfo) { // :53-67
    factors_.push_back(blockinfo);
    const auto &parameter_blocks = blockinfo->parameterBlocks();
    const auto &block_sizes      = blockinfo->parameterBlockSizes();
    for (size_t k = 0; k < parameter_blocks.size(); k++) parameter_block_size_[idOf(parameter_blocks[k])] = block_sizes[k];
    for (int index : blockinfo->marginalizationParametersIndex()) parameter_block_index_[idOf(parameter_blocks[(size_t) index])] = 0;
}

namespace {
thread_local double g_marg_phase_ms[4] = {0, 0, 0, 0};
thread_local bool g_marg_structured   = false;
// process-wide switch (diagnostics / tests): force the reference's dense M2 + M3 even where the structured path applies
std::atomic<int> g_marg_force_dense{0};
}
const double *MarginalizationInfo::lastPhaseMs() { return g_marg_phase_ms; }
bool MarginalizationInfo::lastWasStructured() { return g_marg_structured; }
void MarginalizationInfo::forceDense(bool on) { g_marg_force_dense.store(on ? 1 : 0); }
bool MarginalizationInfo::denseForced() { return g_marg_force_dense.load() != 0; }

bool MarginalizationInfo::marginalization() { // :73-101
    if (!updateParameterBlocksIndex()) {
        isvalid_ = false;
        releaseMemory();
        return false;
    }
    const bool dbg = getenv("ICG_MARG_DEBUG") != nullptr;
    auto now       = [] { return std::chrono::steady_clock::now(); };
    auto ms        = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    auto t0 = now();
    if (!preMarginalization()) {
        isvalid_ = false;
        releaseMemory();
        return false;
    }
    auto t1 = now();
    auto t2 = t1;
    g_marg_structured = g_marg_force_dense.load() == 0 && constructAndEliminateStructured();
    if (!g_marg_structured) {
        if (!constructEquation()) {
            isvalid_ = false;
            releaseMemory();
            return false;
        }
        t2 = now();
        schurElimination();
    } else {
        t2 = now(); // (structured path: assembly and elimination are one step, booked under "construct")
    }
    auto t3 = now();
    linearization();
    auto t4 = now();
    g_marg_phase_ms[0] = ms(t0, t1), g_marg_phase_ms[1] = ms(t1, t2), g_marg_phase_ms[2] = ms(t2, t3), g_marg_phase_ms[3] = ms(t3, t4);
    if (dbg)
        fprintf(stderr, "[marginalization] m=%d r=%d: evaluate %.3f ms, construct %.3f ms, schur %.3f ms, linearize %.3f ms\n", marginalized_size_,
                remained_size_, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    releaseMemory();
    return true;
}

vector<double *> MarginalizationInfo::getParamterBlocks(std::unordered_map<long, double *> &address) { // :103-122
    vector<double *> remained_block_addr;
    remained_block_data_.clear();
    remained_block_index_.clear();
    remained_block_size_.clear();
    for (const auto &block : parameter_block_index_) {
        if (block.second >= marginalized_size_) {
            remained_block_data_.push_back(parameter_block_data_[block.first]);
            remained_block_size_.push_back(parameter_block_size_[block.first]);
            remained_block_index_.push_back(parameter_block_index_[block.first]);
            remained_block_addr.push_back(address[block.first]);
        }
    }
    return remained_block_addr;
}

bool MarginalizationInfo::updateParameterBlocksIndex() { // :232-253
    int index = 0;
    for (auto &block : parameter_block_index_) {
        block.second = index;
        index += localSize(parameter_block_size_[block.first]);
    }
    marginalized_size_ = index;
    for (const auto &block : parameter_block_size_) {
        if (parameter_block_index_.find(block.first) == parameter_block_index_.end()) {
            parameter_block_index_[block.first] = index;
            index += localSize(block.second);
        }
    }
    remained_size_ = index - marginalized_size_;
    local_size_    = index;
    return marginalized_size_ > 0;
}

static bool onBatch(const std::shared_ptr<ResidualBlockInfo> &f, DeviceFactorSet *batch) {
    if (!batch) return false;
    auto *rf = dynamic_cast<ReprojectionFactor *>(f->costFunction().get());
    if (rf == nullptr || !batch->owns(rf)) return false;
    return f->lossFunction() == nullptr || dynamic_cast<HuberLossHip *>(f->lossFunction().get()) != nullptr;
}

bool MarginalizationInfo::preMarginalization() { // :256-273
    // every reprojection factor of the batch: ONE device launch incl. the robust correction
    double delta = 0;
    for (const auto &factor : factors_)
        if (onBatch(factor, batch_)) {
            auto *hl = dynamic_cast<HuberLossHip *>(factor->lossFunction().get());
            delta    = hl ? hl->delta() : 0.0;
            break;
        }
    if (batch_ && batch_->size() > 0 && !batch_->evaluateCorrected(delta)) return false;
    for (const auto &factor : factors_) {
        if (!onBatch(factor, batch_) && !factor->Evaluate()) return false;
        const vector<int32_t> &block_sizes = factor->parameterBlockSizes();
        for (size_t k = 0; k < block_sizes.size(); k++) {
            long id  = idOf(factor->parameterBlocks()[k]);
            int size = block_sizes[k];
            if (parameter_block_data_.find(id) == parameter_block_data_.end()) {
                auto *data = new double[(size_t) size];
                memcpy(data, factor->parameterBlocks()[k], sizeof(double) * (size_t) size);
                parameter_block_data_[id] = data;
            }
        }
    }
    return true;
}

bool MarginalizationInfo::constructEquation() { // :195-230
    const size_t L = (size_t) local_size_;
    H0_.assign(L * L, 0.0);
    b0_.assign(L, 0.0);
    std::unordered_map<const double *, int> column_of;
    bool any_device = false;
    for (const auto &factor : factors_) {
        const auto &blocks = factor->parameterBlocks();
        if (onBatch(factor, batch_)) {
            any_device = true;
            for (double *p : blocks) column_of[p] = parameter_block_index_[idOf(p)];
            continue;
        }
        const int nr = (int) factor->residuals().size();
        for (size_t i = 0; i < blocks.size(); i++) {
            const int row0 = parameter_block_index_[idOf(blocks[i])];
            const int gi   = parameter_block_size_[idOf(blocks[i])];
            const int rows = localSize(gi);
            const vector<double> &Ji = factor->jacobians()[i];
            for (size_t j = i; j < blocks.size(); ++j) {
                const int col0 = parameter_block_index_[idOf(blocks[j])];
                const int gj   = parameter_block_size_[idOf(blocks[j])];
                const int cols = localSize(gj);
                const vector<double> &Jj = factor->jacobians()[j];
                for (int x = 0; x < rows; x++)
                    for (int y = 0; y < cols; y++) {
                        double s = 0;
                        for (int k = 0; k < nr; k++) s += Ji[(size_t) k * gi + x] * Jj[(size_t) k * gj + y];
                        H0_[(size_t) (row0 + x) * L + col0 + y] += s;
                        if (i != j) H0_[(size_t) (col0 + y) * L + row0 + x] = H0_[(size_t) (row0 + x) * L + col0 + y];
                    }
            }
            for (int x = 0; x < rows; x++) {
                double s = 0;
                for (int k = 0; k < nr; k++) s += Ji[(size_t) k * gi + x] * factor->residuals()[(size_t) k];
                b0_[(size_t) row0 + x] -= s;
            }
        }
    }
    if (any_device && !batch_->accumulateNormal(column_of, local_size_, H0_.data(), b0_.data())) return false;
    return true;
}

// The marginalized set of GVINS::gvinsMarginalization (ic_gvins.cc:1425-1610) is {pose, mix of the oldest state} + {the inverse depths
// anchored in the oldest keyframe}, and an inverse depth is touched by reprojection factors only.  Hmm is then
//     [ A   B ]   A: the few pose / mix columns (15),  D: DIAGONAL (one 1x1 block per landmark),
//     [ B^T D ]
// and the reference's dense pseudo-inverse of Hmm (eigen-decomposition with a 1e-8 floor, :170-181) equals the plain inverse
// whenever Hmm is safely positive definite.  Then  Hrr - Hrm Hmm^-1 Hmr  can be taken in two exact steps: the landmark block first
// (reciprocals of h_ll: done on the DEVICE together with the assembly, the resident f1 kernels), the small A-block second (the
// reference's own eigen / floor procedure on 15 columns instead of 15 + L).  Guard: every h_ll and every eigenvalue of the reduced
// A-block at least 100 x the reference's floor — otherwise this returns false and the dense path runs.
bool MarginalizationInfo::constructAndEliminateStructured() {
    StructuredPlan plan;
    if (!planStructured(plan)) return false;
    double min_hll = 0;
    if (!batch_->accumulateLandmarkEliminated(plan.camera_column_of, plan.P, plan.H.data(), plan.b.data(), &min_hll)) return false;
    return finishStructured(plan, min_hll);
}

bool MarginalizationInfo::planStructured(StructuredPlan &plan) {
    if (!batch_ || batch_->size() == 0) return false;
    // landmark blocks of the batch: all marginalized, size 1, and touched by batch factors only
    std::unordered_map<long, char> is_lm;
    for (double *p : batch_->landmarkBlocks()) {
        const long id = idOf(p);
        auto it = parameter_block_index_.find(id);
        if (it == parameter_block_index_.end() || it->second >= marginalized_size_ || parameter_block_size_[id] != 1) return false;
        is_lm[id] = 1;
    }
    for (const auto &factor : factors_) {
        const bool on_batch = onBatch(factor, batch_);
        for (double *p : factor->parameterBlocks()) {
            const bool lm = is_lm.count(idOf(p)) != 0;
            if (lm && !on_batch) return false; // a host factor on an inverse depth: Hmm's landmark block is not diagonal
        }
        if (!on_batch && dynamic_cast<ReprojectionFactor *>(factor->costFunction().get()) != nullptr) return false;
    }
    // compact camera columns: every non-landmark column of the local ordering, order kept (marginalized ones first)
    const int L = local_size_;
    vector<int> compact((size_t) L, -1);
    {
        vector<char> lm_col((size_t) L, 0);
        for (const auto &kv : is_lm) lm_col[(size_t) parameter_block_index_[kv.first]] = 1;
        int c = 0;
        for (int k = 0; k < L; k++)
            if (!lm_col[(size_t) k]) compact[(size_t) k] = c++;
    }
    const int n_lm = (int) is_lm.size();
    const int P = L - n_lm, m = marginalized_size_ - n_lm, r = remained_size_;
    if (m < 0 || P != m + r) return false;
    plan.P = P, plan.m = m, plan.r = r;
    plan.H.assign((size_t) P * P, 0.0), plan.b.assign((size_t) P, 0.0);
    plan.camera_column_of.clear();
    vector<double> &H = plan.H, &b = plan.b;
    std::unordered_map<const double *, int> &camera_column_of = plan.camera_column_of;
    for (const auto &factor : factors_) {
        const auto &blocks = factor->parameterBlocks();
        if (onBatch(factor, batch_)) {
            for (double *p : blocks)
                if (!is_lm.count(idOf(p))) camera_column_of[p] = compact[(size_t) parameter_block_index_[idOf(p)]];
            continue;
        }
        const int nr = (int) factor->residuals().size(); // host factors: J^T J / J^T e straight into the compact system (:195-230)
        for (size_t i = 0; i < blocks.size(); i++) {
            const int row0 = compact[(size_t) parameter_block_index_[idOf(blocks[i])]];
            const int gi   = parameter_block_size_[idOf(blocks[i])];
            const int rows = localSize(gi);
            const vector<double> &Ji = factor->jacobians()[i];
            for (size_t j = i; j < blocks.size(); ++j) {
                const int col0 = compact[(size_t) parameter_block_index_[idOf(blocks[j])]];
                const int gj   = parameter_block_size_[idOf(blocks[j])];
                const int cols = localSize(gj);
                const vector<double> &Jj = factor->jacobians()[j];
                for (int x = 0; x < rows; x++)
                    for (int y = 0; y < cols; y++) {
                        double sum = 0;
                        for (int k = 0; k < nr; k++) sum += Ji[(size_t) k * gi + x] * Jj[(size_t) k * gj + y];
                        H[(size_t) (row0 + x) * P + col0 + y] += sum;
                        if (i != j) H[(size_t) (col0 + y) * P + row0 + x] = H[(size_t) (row0 + x) * P + col0 + y];
                    }
            }
            for (int x = 0; x < rows; x++) {
                double sum = 0;
                for (int k = 0; k < nr; k++) sum += Ji[(size_t) k * gi + x] * factor->residuals()[(size_t) k];
                b[(size_t) row0 + x] -= sum;
            }
        }
    }
    return true;
}

bool MarginalizationInfo::finishStructured(StructuredPlan &plan, double min_hll) {
    const int P = plan.P, m = plan.m, r = plan.r;
    const vector<double> &H = plan.H, &b = plan.b;
    const double GUARD = 100.0 * EPS;
    if (!(min_hll > GUARD)) return false;
    // second step on the m leading columns: the reference's procedure (:170-192) on the reduced system
    vector<double> Hmm((size_t) m * m), ev, V;
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) Hmm[(size_t) i * m + j] = 0.5 * (H[(size_t) i * P + j] + H[(size_t) j * P + i]);
    symmetricEigen(m, Hmm, ev, V);
    for (int k = 0; k < m; k++)
        if (!(ev[(size_t) k] > GUARD)) return false;
    vector<double> Hinv((size_t) m * m, 0.0), Wv((size_t) m * m);
    for (int i = 0; i < m; i++)
        for (int k = 0; k < m; k++) Wv[(size_t) i * m + k] = V[(size_t) i * m + k] / ev[(size_t) k];
    for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++) {
            double sum = 0;
            for (int k = 0; k < m; k++) sum += Wv[(size_t) i * m + k] * V[(size_t) j * m + k];
            Hinv[(size_t) i * m + j] = Hinv[(size_t) j * m + i] = sum;
        }
    vector<double> T((size_t) r * m);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < m; j++) {
            double sum = 0;
            for (int k = 0; k < m; k++) sum += H[(size_t) (m + i) * P + k] * Hinv[(size_t) k * m + j];
            T[(size_t) i * m + j] = sum;
        }
    Hp_.assign((size_t) r * r, 0.0);
    bp_.assign((size_t) r, 0.0);
    for (int i = 0; i < r; i++) {
        for (int j = 0; j < r; j++) {
            double sum = 0;
            for (int k = 0; k < m; k++) sum += T[(size_t) i * m + k] * H[(size_t) k * P + m + j];
            Hp_[(size_t) i * r + j] = H[(size_t) (m + i) * P + m + j] - sum;
        }
        double sum = 0;
        for (int k = 0; k < m; k++) sum += T[(size_t) i * m + k] * b[(size_t) k];
        bp_[(size_t) i] = b[(size_t) m + i] - sum;
    }
    return true;
}

void MarginalizationInfo::schurElimination() { // :170-192
    const int m = marginalized_size_, r = remained_size_;
    const size_t L = (size_t) local_size_;
    auto H = [&](int i, int j) { return H0_[(size_t) i * L + j]; };
    vector<double> Hmm((size_t) m * m), ev, V, Hinv((size_t) m * m, 0.0);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) Hmm[(size_t) i * m + j] = 0.5 * (H(i, j) + H(j, i));
    auto tA = std::chrono::steady_clock::now();
    symmetricEigen(m, Hmm, ev, V);
    auto tB = std::chrono::steady_clock::now();
    if (getenv("ICG_MARG_DEBUG")) fprintf(stderr, "[schur] eigen(%d) %.3f ms\n", m, std::chrono::duration<double, std::milli>(tB - tA).count());
    // Hmm^+ = V diag(1/ev, 0 below EPS) V^T: the thresholded reciprocals once, the scaled copy W = V diag(inv) once, and only the
    // lower triangle of the symmetric product
    vector<double> inv((size_t) m), Wv((size_t) m * m);
    for (int k = 0; k < m; k++) inv[(size_t) k] = ev[(size_t) k] > EPS ? 1.0 / ev[(size_t) k] : 0.0;
    for (int i = 0; i < m; i++)
        for (int k = 0; k < m; k++) Wv[(size_t) i * m + k] = V[(size_t) i * m + k] * inv[(size_t) k];
    for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += Wv[(size_t) i * m + k] * V[(size_t) j * m + k];
            Hinv[(size_t) i * m + j] = Hinv[(size_t) j * m + i] = s;
        }
    vector<double> T((size_t) r * m);
    for (int i = 0; i < r; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += H(m + i, k) * Hinv[(size_t) k * m + j];
            T[(size_t) i * m + j] = s;
        }
    Hp_.assign((size_t) r * r, 0.0);
    bp_.assign((size_t) r, 0.0);
    for (int i = 0; i < r; i++) {
        for (int j = 0; j < r; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += T[(size_t) i * m + k] * H(k, m + j);
            Hp_[(size_t) i * r + j] = H(m + i, m + j) - s;
        }
        double s = 0;
        for (int k = 0; k < m; k++) s += T[(size_t) i * m + k] * b0_[(size_t) k];
        bp_[(size_t) i] = b0_[(size_t) m + i] - s;
    }
}

void MarginalizationInfo::linearization() { // :153-167
    const int r = remained_size_;
    vector<double> ev, V;
    symmetricEigen(r, Hp_, ev, V);
    linearized_jacobians_.assign((size_t) r * r, 0.0);
    linearized_residuals_.assign((size_t) r, 0.0);
    for (int k = 0; k < r; k++) {
        const double S = ev[(size_t) k] > EPS ? ev[(size_t) k] : 0.0, Sinv = ev[(size_t) k] > EPS ? 1.0 / ev[(size_t) k] : 0.0;
        const double ss = std::sqrt(S), si = std::sqrt(Sinv);
        double vb = 0;
        for (int i = 0; i < r; i++) {
            linearized_jacobians_[(size_t) k * r + i] = ss * V[(size_t) i * r + k];
            vb += V[(size_t) i * r + k] * -bp_[(size_t) i];
        }
        linearized_residuals_[(size_t) k] = si * vb;
    }
}

// ---- MarginalizationFactor (marginalization_factor.h:31-105) ----------------------------------------------------------
MarginalizationFactor::MarginalizationFactor(std::shared_ptr<MarginalizationInfo> marg_info) : marg_info_(std::move(marg_info)) {
    for (auto size : marg_info_->remainedBlockSize()) mutable_parameter_block_sizes()->push_back(size);
    set_num_residuals(marg_info_->remainedSize());
}

bool MarginalizationFactor::Evaluate(const double *const *parameters, double *residuals, double **jacobians) const {
    const int marginalizaed_size = marg_info_->marginalizedSize();
    const int remained_size      = marg_info_->remainedSize();
    const vector<int> &remained_block_index     = marg_info_->remainedBlockIndex();
    const vector<int> &remained_block_size      = marg_info_->remainedBlockSize();
    const vector<double *> &remained_block_data = marg_info_->remainedBlockData();
    const vector<double> &J0 = marg_info_->linearizedJacobians();
    const vector<double> &e0 = marg_info_->linearizedResiduals();
    vector<double> dx((size_t) remained_size, 0.0);
    for (size_t i = 0; i < remained_block_size.size(); i++) {
        const int size = remained_block_size[i], index = remained_block_index[i] - marginalizaed_size;
        const double *x = parameters[i], *x0 = remained_block_data[i];
        if (size == POSE_GLOBAL_SIZE) { // :64-72
            const double n2 = x0[3] * x0[3] + x0[4] * x0[4] + x0[5] * x0[5] + x0[6] * x0[6];
            const double ax = -x0[3] / n2, ay = -x0[4] / n2, az = -x0[5] / n2, aw = x0[6] / n2;
            const double bx = x[3], by = x[4], bz = x[5], bw = x[6];
            const double dqx = aw * bx + ax * bw + ay * bz - az * by;
            const double dqy = aw * by + ay * bw + az * bx - ax * bz;
            const double dqz = aw * bz + az * bw + ax * by - ay * bx;
            const double dqw = aw * bw - ax * bx - ay * by - az * bz;
            for (int k = 0; k < 3; k++) dx[(size_t) index + k] = x[k] - x0[k];
            const double sgn      = dqw < 0 ? -2.0 : 2.0;
            dx[(size_t) index + 3] = sgn * dqx;
            dx[(size_t) index + 4] = sgn * dqy;
            dx[(size_t) index + 5] = sgn * dqz;
        } else {
            for (int k = 0; k < size; k++) dx[(size_t) index + k] = x[k] - x0[k];
        }
    }
    for (int i = 0; i < remained_size; i++) { // e = e0 + J0 * dx  (:79-80)
        double s = 0;
        for (int k = 0; k < remained_size; k++) s += J0[(size_t) i * remained_size + k] * dx[(size_t) k];
        residuals[i] = e0[(size_t) i] + s;
    }
    if (jacobians) {
        for (size_t b = 0; b < remained_block_size.size(); b++) {
            if (!jacobians[b]) continue;
            const int size = remained_block_size[b], index = remained_block_index[b] - marginalizaed_size;
            const int local_size = MarginalizationInfo::localSize(size);
            for (int i = 0; i < remained_size; i++)
                for (int j = 0; j < size; j++)
                    jacobians[b][(size_t) i * size + j] = j < local_size ? J0[(size_t) i * remained_size + index + j] : 0.0;
        }
    }
    return true;
}

} // namespace icg
