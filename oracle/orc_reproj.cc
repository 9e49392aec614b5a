// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). CPU restatement of
//   ReprojectionFactor::Evaluate        reference ic_gvins/ic_gvins/factors/reprojection_factor.h:55-147
//   ResidualBlockInfo robust correction reference ic_gvins/ic_gvins/factors/residual_block_info.h:59-87
// Parity status: no golden vectors exist upstream.  PINNED against the reference's own header compiled
// unmodified (oracle/ref_build -> oracle/_ref/libref_reproj.so, tests/golden/reproj_ref_golden.npz, 1e-12) and by the
// finite-difference / algebraic known-answer tests in tests/test_oracle_reproj.py; the robust corrector additionally
// through the reference's ResidualBlockInfo in the marginalization golden (tests/golden/marg_ref_golden.npz).
#include "oracle.h"
#include "orc_math.h"

using namespace orc;

namespace {

struct Mat23 {
    double m[2][3];
};

static inline void mul_23_33(const Mat23 &a, const M3 &b, double out[2][3]) {
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) out[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
}

// One factor. obs = {pts0[3], pts1[3], vel0[3], vel1[3], td0, td1, std}
// out48 = r[2], J_pose_i[2x7], J_pose_j[2x7], J_ext[2x7], J_invdepth[2], J_td[2]  (row-major, 7th column zero)
void reproj_one(const double *o, const double *pi, const double *pj, const double *pe, double id0, double td,
                int want_jac, double *r2, double *J46) {
    V3 pts0{o[0], o[1], o[2]}, pts1{o[3], o[4], o[5]}, vel0{o[6], o[7], o[8]}, vel1{o[9], o[10], o[11]};
    double td0 = o[12], td1 = o[13], sinfo = 1.0 / o[14];

    V3 p0{pi[0], pi[1], pi[2]};
    Q4 q0 = quat_wxyz(pi[6], pi[3], pi[4], pi[5]);
    V3 p1{pj[0], pj[1], pj[2]};
    Q4 q1 = quat_wxyz(pj[6], pj[3], pj[4], pj[5]);
    V3 tic{pe[0], pe[1], pe[2]};
    Q4 qic = quat_wxyz(pe[6], pe[3], pe[4], pe[5]);

    V3 pts_0_td = pts0 - (td - td0) * vel0; // :73
    V3 pts_1_td = pts1 - (td - td1) * vel1; // :74

    V3 pts_c_0 = pts_0_td / id0;                 // :76
    V3 pts_b_0 = qrot(qic, pts_c_0) + tic;       // :77
    V3 pts_n   = qrot(q0, pts_b_0) + p0;         // :78
    V3 pts_b_1 = qrot(qinv(q1), pts_n - p1);     // :79
    V3 pts_1   = qrot(qinv(qic), pts_b_1 - tic); // :80

    double d1 = pts_1.z;

    r2[0] = sinfo * (pts_1.x / d1 - pts_1_td.x); // :86-87
    r2[1] = sinfo * (pts_1.y / d1 - pts_1_td.y);

    if (!want_jac) return;

    M3 cb0n = qmat(q0);       // :90
    M3 cnb1 = m3_T(qmat(q1)); // :91
    M3 cbc  = m3_T(qmat(qic)); // :92
    Mat23 reduce;
    reduce.m[0][0] = sinfo * (1.0 / d1);
    reduce.m[0][1] = sinfo * 0.0;
    reduce.m[0][2] = sinfo * (-pts_1.x / (d1 * d1));
    reduce.m[1][0] = sinfo * 0.0;
    reduce.m[1][1] = sinfo * (1.0 / d1);
    reduce.m[1][2] = sinfo * (-pts_1.y / (d1 * d1));

    double *Ji = J46, *Jj = J46 + 14, *Je = J46 + 28, *Jr = J46 + 42, *Jt = J46 + 44;
    double t[2][3];

    M3 cbc_cnb1 = m3_mul(cbc, cnb1);
    // pose i  :98-107
    {
        M3 left  = cbc_cnb1;
        M3 right = m3_mul(m3_mul(m3_mul(m3_neg(cbc), cnb1), cb0n), skew(pts_b_0));
        mul_23_33(reduce, left, t);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) Ji[i * 7 + j] = t[i][j];
        mul_23_33(reduce, right, t);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) Ji[i * 7 + 3 + j] = t[i][j];
        Ji[6] = Ji[13] = 0;
    }
    // pose j  :109-118
    {
        M3 left  = m3_mul(m3_neg(cbc), cnb1);
        M3 right = m3_mul(cbc, skew(pts_b_1));
        mul_23_33(reduce, left, t);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) Jj[i * 7 + j] = t[i][j];
        mul_23_33(reduce, right, t);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) Jj[i * 7 + 3 + j] = t[i][j];
        Jj[6] = Jj[13] = 0;
    }
    M3 tmp_r = m3_mul(m3_mul(m3_mul(cbc, cnb1), cb0n), m3_T(cbc)); // :125
    // extrinsic  :120-133
    {
        M3 left  = m3_mul(cbc, m3_sub(m3_mul(cnb1, cb0n), m3_identity()));
        V3 inner = m3_vec(cbc, m3_vec(cnb1, m3_vec(cb0n, tic) + p0 - p1) - tic);
        M3 right = m3_add(m3_add(m3_mul(m3_neg(tmp_r), skew(pts_c_0)), skew(m3_vec(tmp_r, pts_c_0))), skew(inner));
        mul_23_33(reduce, left, t);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) Je[i * 7 + j] = t[i][j];
        mul_23_33(reduce, right, t);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) Je[i * 7 + 3 + j] = t[i][j];
        Je[6] = Je[13] = 0;
    }
    // inverse depth :135-138   -reduce * cbc*cnb1*cb0n*cbc^T * pts_0_td / id0^2
    {
        double nr[2][3];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) nr[i][j] = -reduce.m[i][j];
        Mat23 nred;
        memcpy(nred.m, nr, sizeof nr);
        mul_23_33(nred, tmp_r, t);
        double idsq = id0 * id0;
        for (int i = 0; i < 2; i++)
            Jr[i] = (t[i][0] * pts_0_td.x + t[i][1] * pts_0_td.y + t[i][2] * pts_0_td.z) / idsq;
        // td :140-143
        for (int i = 0; i < 2; i++)
            Jt[i] = (t[i][0] * vel0.x + t[i][1] * vel0.y + t[i][2] * vel0.z) / id0 + sinfo * (i == 0 ? vel1.x : vel1.y);
    }
}

} // namespace

extern "C" {

void orc_reproj_eval_batch(int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j,
                           const int32_t *idx_lm, const double *poses, const double *ext, const double *invdepth,
                           double td, int want_jac, double *out_r, double *out_J) {
    for (int k = 0; k < n; k++) {
        double o[15];
        for (int c = 0; c < 15; c++) o[c] = obs_soa[(size_t) c * n + k];
        double J[46];
        reproj_one(o, poses + 7 * (size_t) idx_i[k], poses + 7 * (size_t) idx_j[k], ext, invdepth[idx_lm[k]], td,
                   want_jac, out_r + 2 * (size_t) k, J);
        if (want_jac && out_J) memcpy(out_J + 46 * (size_t) k, J, sizeof J);
    }
}

// Single factor with explicit parameter blocks, i.e. the ceres::CostFunction::Evaluate call shape.
void orc_reproj_eval_one(const double *obs15, const double *pose_i, const double *pose_j, const double *ext,
                         double invdepth, double td, int want_jac, double *r2, double *J46) {
    reproj_one(obs15, pose_i, pose_j, ext, invdepth, td, want_jac, r2, J46);
}

// Huber loss as ceres::HuberLoss(a): rho[0..2] for s = squared norm.
static void huber(double a, double s, double rho[3]) {
    double b = a * a;
    if (s > b) {
        double r = std::sqrt(s);
        rho[0]   = 2.0 * a * r - b;
        rho[1]   = std::fmax(2.2250738585072014e-308, a / r);
        rho[2]   = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s;
        rho[1] = 1.0;
        rho[2] = 0.0;
    }
}

// ResidualBlockInfo::Evaluate robust correction (residual_block_info.h:59-87) for a 2-residual factor with
// Jacobian laid out as 46 doubles (3 x [2x7] + 2 + 2).  huber_delta <= 0 -> no loss.
void orc_huber_correct_2x46(int n, double huber_delta, double *r, double *J) {
    if (huber_delta <= 0) return;
    for (int k = 0; k < n; k++) {
        double *rk = r + 2 * (size_t) k;
        double *Jk = J ? J + 46 * (size_t) k : nullptr;
        double sq_norm = rk[0] * rk[0] + rk[1] * rk[1];
        double rho[3];
        huber(huber_delta, sq_norm, rho);
        double sqrt_rho1 = std::sqrt(rho[1]);
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
        if (Jk) {
            // J = sqrt_rho1 * (J - alpha_sq_norm * r * (r^T J)) column-wise
            const int off[5]  = {0, 14, 28, 42, 44};
            const int cols[5] = {7, 7, 7, 1, 1};
            for (int b = 0; b < 5; b++) {
                double *B = Jk + off[b];
                int nc    = cols[b];
                for (int c = 0; c < nc; c++) {
                    double j0 = B[c], j1 = B[nc + c];
                    double rtj = rk[0] * j0 + rk[1] * j1;
                    B[c]       = sqrt_rho1 * (j0 - alpha_sq_norm * rk[0] * rtj);
                    B[nc + c]  = sqrt_rho1 * (j1 - alpha_sq_norm * rk[1] * rtj);
                }
            }
        }
        rk[0] *= residual_scaling;
        rk[1] *= residual_scaling;
    }
}

} // extern "C"
