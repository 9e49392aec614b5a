// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the IMU preintegration of the reference:
//   PreintegrationBase::integration / compensationBias   preintegration/preintegration_base.cc:39-70, 86-92
//   PreintegrationNormal  integrationProcess :183-192, updateJacobianAndCovariance :198-232, resetState :234-243,
//                         setNoiseMatrix :245-253, evaluate :38-70, residualJacobian* :72-142   (preintegration_normal.cc)
//   PreintegrationEarth   integrationProcess :205-260, updateJacobianAndCovariance :266-303, resetState :305-323,
//                         evaluate :37-90, residualJacobian* :92-164                             (preintegration_earth.cc)
// Odo / EarthOdo variants are dead in the reference build (isuseodo=false, ic_gvins.cc:100) and out of scope.
// `iewn` is an explicit input (SURVEY.md hazard H9: the reference reads an unset `station`, i.e. effectively lat 0).
// Fully specified in-tree.  PINNED against the reference's own sources compiled unmodified (oracle/ref_build ->
// oracle/_ref/libref_preint.so; tests/golden/preint_ref_golden.npz: state/Jacobian/covariance 1e-12, whitened residual and
// Jacobians 1e-9, both variants) and by the analytic tests in tests/test_oracle_preint.py (constant-rate closed forms,
// finite-difference Jacobians).
#include "oracle.h"
#include "orc_math.h"
#include <vector>

using namespace orc;

namespace {

struct State {
    V3 p;
    Q4 q;
    V3 v, bg, ba;
};
State load_state(const double *s) {
    State st;
    st.p  = v3(s[0], s[1], s[2]);
    st.q  = Q4{s[3], s[4], s[5], s[6]};
    st.v  = v3(s[7], s[8], s[9]);
    st.bg = v3(s[10], s[11], s[12]);
    st.ba = v3(s[13], s[14], s[15]);
    return st;
}
void store_state(const State &st, double *s) {
    s[0] = st.p.x, s[1] = st.p.y, s[2] = st.p.z;
    s[3] = st.q.x, s[4] = st.q.y, s[5] = st.q.z, s[6] = st.q.w;
    s[7] = st.v.x, s[8] = st.v.y, s[9] = st.v.z;
    s[10] = st.bg.x, s[11] = st.bg.y, s[12] = st.bg.z;
    s[13] = st.ba.x, s[14] = st.ba.y, s[15] = st.ba.z;
}

struct Imu {
    double time, dt;
    V3 dtheta, dvel;
};
Imu load_imu(const double *p) { return Imu{p[0], p[1], v3(p[2], p[3], p[4]), v3(p[5], p[6], p[7])}; }

typedef double M15[15][15];

void mat_mul15(const M15 a, const M15 b, M15 out) {
    M15 t;
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            double s = 0;
            for (int k = 0; k < 15; k++) s += a[i][k] * b[k][j];
            t[i][j] = s;
        }
    memcpy(out, t, sizeof(M15));
}
void set_block(M15 m, int r, int c, const M3 &b) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) m[r + i][c + j] = b.m[i][j];
}
M3 get_block(const M15 m, int r, int c) {
    M3 b;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) b.m[i][j] = m[r + i][c + j];
    return b;
}

// jacobian_ = phi*jacobian_ ; covariance_ = phi P phi^T + 0.5 dt (phi M + M phi^T), M = gt noise gt^T
void propagate(const M15 phi, const double gt[15][12], const double noise[12], double dt, M15 jac, M15 cov) {
    mat_mul15(phi, jac, jac);
    M15 M, phiT, t1, t2;
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            double s = 0;
            for (int k = 0; k < 12; k++) s += gt[i][k] * noise[k] * gt[j][k];
            M[i][j]    = s;
            phiT[i][j] = phi[j][i];
        }
    mat_mul15(phi, M, t1);  // phi * M
    mat_mul15(M, phiT, t2); // M * phi^T
    M15 pc;
    mat_mul15(phi, cov, pc);
    mat_mul15(pc, phiT, pc);
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) cov[i][j] = pc[i][j] + 0.5 * dt * (t1[i][j] + t2[i][j]);
}

// 4x4 left/right quaternion product matrices, bottom-right 3x3 (rotation.h:103-119)
M3 qleft_br(Q4 q) { return m3_add(m3_scale(m3_identity(), q.w), skew(v3(q.x, q.y, q.z))); }
M3 qright_br(Q4 q) { return m3_sub(m3_scale(m3_identity(), q.w), skew(v3(q.x, q.y, q.z))); }
// bottom-right 3x3 of quaternionleft(a) * quaternionright(b)
M3 qleft_qright_br(Q4 a, Q4 b) {
    double L[4][4], R[4][4];
    auto fill = [](double M[4][4], Q4 q, double sgn) {
        M[0][0] = q.w;
        M[0][1] = -q.x, M[0][2] = -q.y, M[0][3] = -q.z;
        M[1][0] = q.x, M[2][0] = q.y, M[3][0] = q.z;
        M3 s = skew(v3(q.x, q.y, q.z));
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M[1 + i][1 + j] = (i == j ? q.w : 0.0) + sgn * s.m[i][j];
    };
    fill(L, a, 1.0);
    fill(R, b, -1.0);
    M3 out;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += L[1 + i][k] * R[k][1 + j];
            out.m[i][j] = s;
        }
    return out;
}

// dense helpers for evaluate: inverse by Gauss-Jordan with partial pivoting, lower Cholesky
bool invert15(const M15 a, M15 inv) {
    double w[15][30];
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            w[i][j]      = a[i][j];
            w[i][15 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 15; c++) {
        int piv = c;
        for (int r = c + 1; r < 15; r++)
            if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 30; j++) std::swap(w[c][j], w[piv][j]);
        double d = w[c][c];
        for (int j = 0; j < 30; j++) w[c][j] /= d;
        for (int r = 0; r < 15; r++)
            if (r != c) {
                double f = w[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 30; j++) w[r][j] -= f * w[c][j];
            }
    }
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) inv[i][j] = w[i][15 + j];
    return true;
}
void cholesky_lower15(const M15 a, M15 L) {
    memset(L, 0, sizeof(M15));
    for (int j = 0; j < 15; j++) {
        double s = a[j][j];
        for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        L[j][j] = std::sqrt(s);
        for (int i = j + 1; i < 15; i++) {
            double t = a[i][j];
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t / L[j][j];
        }
    }
}

} // namespace

extern "C" {

void orc_preint_integrate(int variant, int n_imu, const double *imu, const double *state0, const double *params,
                          double *cur_state, double *delta_state, double *jac_out, double *cov_out, double *delta_time_out,
                          double *pn_out) {
    const double gyr_arw = params[0], acc_vrw = params[1], gbstd = params[2], abstd = params[3], corr_time = params[4];
    const V3 gravity = v3(0, 0, params[5]);
    const V3 iewn    = v3(params[6], params[7], params[8]);
    State cur        = load_state(state0);
    // resetState (normal :234-243, earth :305-323)
    State delta;
    delta.p = delta.v = v3(0, 0, 0);
    delta.q           = quat_identity();
    delta.bg          = cur.bg;
    delta.ba          = cur.ba;
    const Q4 q0       = cur.q;
    double delta_time = 0;
    M15 jac, cov;
    memset(jac, 0, sizeof jac);
    memset(cov, 0, sizeof cov);
    for (int i = 0; i < 15; i++) jac[i][i] = 1.0;
    double noise[12];
    for (int i = 0; i < 3; i++) {
        noise[i]     = gyr_arw * gyr_arw;
        noise[3 + i] = acc_vrw * acc_vrw;
        noise[6 + i] = 2 * gbstd * gbstd / corr_time;
        noise[9 + i] = 2 * abstd * abstd / corr_time;
    }

    for (int index = 1; index < n_imu; index++) {
        Imu pre = load_imu(imu + 8 * (index - 1)), curi = load_imu(imu + 8 * index);
        // compensationBias (base :86-92)
        pre.dtheta  = pre.dtheta - pre.dt * delta.bg;
        pre.dvel    = pre.dvel - pre.dt * delta.ba;
        curi.dtheta = curi.dtheta - curi.dt * delta.bg;
        curi.dvel   = curi.dvel - curi.dt * delta.ba;
        const double dt = curi.dt;
        delta_time += dt;

        V3 dvfb = curi.dvel + 0.5 * cross(curi.dtheta, curi.dvel) +
                  (1.0 / 12.0) * (cross(pre.dtheta, curi.dvel) + cross(pre.dvel, curi.dtheta));
        V3 dtheta = curi.dtheta + (1.0 / 12.0) * cross(pre.dtheta, curi.dtheta);

        M15 phi;
        memset(phi, 0, sizeof phi);
        double gt[15][12];
        memset(gt, 0, sizeof gt);

        if (variant == 0) {
            // PreintegrationBase::integration :39-70
            V3 dvel = m3_vec(qmat(cur.q), dvfb) + gravity * dt;
            cur.p   = cur.p + dt * cur.v + 0.5 * dt * dvel;
            cur.v   = cur.v + dvel;
            cur.q   = qnormalized(qmul(cur.q, rotvec2quat(dtheta)));
            dvel    = m3_vec(qmat(delta.q), dvfb);
            delta.p = delta.p + dt * delta.v + 0.5 * dt * dvel;
            delta.v = delta.v + dvel;
            delta.q = qnormalized(qmul(delta.q, rotvec2quat(dtheta)));
            // updateJacobianAndCovariance normal :198-232
            M3 Rq = qmat(delta.q);
            set_block(phi, 0, 0, m3_identity());
            set_block(phi, 0, 3, m3_scale(m3_identity(), dt));
            set_block(phi, 3, 3, m3_identity());
            set_block(phi, 3, 6, m3_mul(m3_neg(Rq), skew(curi.dvel)));
            set_block(phi, 3, 12, m3_scale(m3_neg(Rq), dt));
            set_block(phi, 6, 6, m3_sub(m3_identity(), skew(curi.dtheta)));
            set_block(phi, 6, 9, m3_scale(m3_neg(m3_identity()), dt));
            set_block(phi, 9, 9, m3_scale(m3_identity(), 1 - dt / corr_time));
            set_block(phi, 12, 12, m3_scale(m3_identity(), 1 - dt / corr_time));
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) gt[3 + i][3 + j] = Rq.m[i][j];
                gt[6 + i][0 + i]  = 1.0;
                gt[9 + i][6 + i]  = 1.0;
                gt[12 + i][9 + i] = 1.0;
            }
        } else {
            // PreintegrationEarth::integrationProcess :205-260
            V3 dv_cor_g = (gravity - 2.0 * cross(iewn, cur.v)) * dt;
            V3 dnn      = -iewn * dt;
            Q4 qnn      = rotvec2quat(dnn);
            M3 half     = m3_scale(m3_add(m3_identity(), qmat(qnn)), 0.5);
            V3 dvel     = m3_vec(m3_mul(half, qmat(cur.q)), dvfb) + dv_cor_g;
            cur.p       = cur.p + dt * cur.v + 0.5 * dt * dvel;
            cur.v       = cur.v + dvel;
            if (pn_out) {
                pn_out[4 * (index - 1)]     = dt;
                pn_out[4 * (index - 1) + 1] = cur.p.x;
                pn_out[4 * (index - 1) + 2] = cur.p.y;
                pn_out[4 * (index - 1) + 3] = cur.p.z;
            }
            cur.q   = qnormalized(qmul(qmul(qnn, cur.q), rotvec2quat(dtheta)));
            dnn     = -(delta_time - 0.5 * dt) * iewn;
            dvel    = m3_vec(qmat(qmul(qmul(qmul(qinv(q0), rotvec2quat(dnn)), q0), delta.q)), dvfb);
            delta.p = delta.p + dt * delta.v + 0.5 * dt * dvel;
            delta.v = delta.v + dvel;
            delta.q = qnormalized(qmul(delta.q, rotvec2quat(dtheta)));
            // updateJacobianAndCovariance earth :266-303
            V3 dnn2 = -iewn * delta_time;
            M3 cbb0 = m3_neg(qmat(qmul(qmul(qmul(qinv(q0), rotvec2quat(dnn2)), q0), delta.q)));
            set_block(phi, 0, 0, m3_identity());
            set_block(phi, 0, 3, m3_scale(m3_identity(), dt));
            set_block(phi, 3, 3, m3_identity());
            set_block(phi, 3, 6, m3_mul(cbb0, skew(curi.dvel)));
            set_block(phi, 3, 12, m3_scale(cbb0, dt));
            set_block(phi, 6, 6, m3_sub(m3_identity(), skew(curi.dtheta)));
            set_block(phi, 6, 9, m3_scale(m3_neg(m3_identity()), dt));
            set_block(phi, 9, 9, m3_scale(m3_identity(), 1 - dt / corr_time));
            set_block(phi, 12, 12, m3_scale(m3_identity(), 1 - dt / corr_time));
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) gt[3 + i][3 + j] = cbb0.m[i][j];
                gt[6 + i][0 + i]  = -1.0;
                gt[9 + i][6 + i]  = 1.0;
                gt[12 + i][9 + i] = 1.0;
            }
        }
        propagate(phi, gt, noise, dt, jac, cov);
    }
    store_state(cur, cur_state);
    store_state(delta, delta_state);
    memcpy(jac_out, jac, sizeof jac);
    memcpy(cov_out, cov, sizeof cov);
    *delta_time_out = delta_time;
}

// PreintegrationFactor::Evaluate (preintegration_factor.h:45-69) = evaluate + 4 Jacobian blocks.
// jacobians: 15x7 | 15x9 | 15x7 | 15x9 row-major, concatenated (pose0, mix0, pose1, mix1).
void orc_preint_evaluate(int variant, const double *delta_state, const double *jac_in, const double *cov_in, double delta_time,
                         const double *gravity3, const double *iewn3, int n_pn, const double *pn, const double *q0_xyzw,
                         const double *pose0, const double *mix0, const double *pose1, const double *mix1, double *residuals,
                         double *jacobians) {
    (void) q0_xyzw;
    State d = load_state(delta_state);
    M15 jac, cov, inv, L;
    memcpy(jac, jac_in, sizeof jac);
    memcpy(cov, cov_in, sizeof cov);
    invert15(cov, inv);
    for (int i = 0; i < 15; i++) // symmetrise like a self-adjoint view would
        for (int j = 0; j < i; j++) inv[j][i] = inv[i][j];
    cholesky_lower15(inv, L);
    // sqrt_information = L^T
    M15 S;
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) S[i][j] = L[j][i];

    V3 p0 = v3(pose0[0], pose0[1], pose0[2]), p1 = v3(pose1[0], pose1[1], pose1[2]);
    Q4 q0 = quat_wxyz(pose0[6], pose0[3], pose0[4], pose0[5]), q1 = quat_wxyz(pose1[6], pose1[3], pose1[4], pose1[5]);
    V3 v0 = v3(mix0[0], mix0[1], mix0[2]), bg0 = v3(mix0[3], mix0[4], mix0[5]), ba0 = v3(mix0[6], mix0[7], mix0[8]);
    V3 v1 = v3(mix1[0], mix1[1], mix1[2]), bg1 = v3(mix1[3], mix1[4], mix1[5]), ba1 = v3(mix1[6], mix1[7], mix1[8]);
    V3 gravity = v3(gravity3[0], gravity3[1], gravity3[2]);
    V3 iewn    = v3(iewn3[0], iewn3[1], iewn3[2]);

    M3 dp_dbg = get_block(jac, 0, 9), dp_dba = get_block(jac, 0, 12), dv_dbg = get_block(jac, 3, 9), dv_dba = get_block(jac, 3, 12),
       dq_dbg = get_block(jac, 6, 9);
    V3 dbg = bg0 - d.bg, dba = ba0 - d.ba;
    V3 corrected_p = d.p + m3_vec(dp_dba, dba) + m3_vec(dp_dbg, dbg);
    V3 corrected_v = d.v + m3_vec(dv_dba, dba) + m3_vec(dv_dbg, dbg);
    Q4 corrected_q = qmul(d.q, rotvec2quat(m3_vec(dq_dbg, dbg)));
    const double T = delta_time;

    double r[15];
    double J0[15][7], J1[15][9], J2[15][7], J3[15][9];
    memset(J0, 0, sizeof J0);
    memset(J1, 0, sizeof J1);
    memset(J2, 0, sizeof J2);
    memset(J3, 0, sizeof J3);
    auto put = [](auto &J, int r0, int c0, const M3 &b) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) J[r0 + i][c0 + j] = b.m[i][j];
    };
    M3 cnb0 = qmat(qinv(q0));
    if (variant == 0) {
        V3 dpn = p1 - p0 - v0 * T - 0.5 * gravity * T * T;
        V3 dvn = v1 - v0 - gravity * T;
        V3 rp  = qrot(qinv(q0), dpn) - corrected_p;
        V3 rv  = qrot(qinv(q0), dvn) - corrected_v;
        Q4 qe  = qmul(qmul(qinv(corrected_q), qinv(q0)), q1);
        r[0] = rp.x, r[1] = rp.y, r[2] = rp.z, r[3] = rv.x, r[4] = rv.y, r[5] = rv.z;
        r[6] = 2 * qe.x, r[7] = 2 * qe.y, r[8] = 2 * qe.z;
        // pose0 :72-91
        put(J0, 0, 0, m3_neg(cnb0));
        put(J0, 0, 3, skew(qrot(qinv(q0), dpn)));
        put(J0, 3, 3, skew(qrot(qinv(q0), dvn)));
        put(J0, 6, 3, m3_neg(qleft_qright_br(qmul(qinv(q1), q0), corrected_q)));
        // pose1 :93-105
        put(J2, 0, 0, cnb0);
        put(J2, 6, 3, qleft_br(qe));
        // mix0 :107-131
        put(J1, 0, 0, m3_scale(m3_neg(cnb0), T));
        put(J1, 0, 3, m3_neg(dp_dbg));
        put(J1, 0, 6, m3_neg(dp_dba));
        put(J1, 3, 0, m3_neg(cnb0));
        put(J1, 3, 3, m3_neg(dv_dbg));
        put(J1, 3, 6, m3_neg(dv_dba));
        put(J1, 6, 3, m3_mul(m3_neg(qleft_br(qmul(qmul(qinv(q1), q0), d.q))), dq_dbg));
        put(J1, 9, 3, m3_neg(m3_identity()));
        put(J1, 12, 6, m3_neg(m3_identity()));
        // mix1 :133-142
        put(J3, 3, 0, cnb0);
        put(J3, 9, 3, m3_identity());
        put(J3, 12, 6, m3_identity());
    } else {
        M3 iewn_skew = skew(iewn);
        V3 p_cor     = v3(0, 0, 0);
        for (int k = 0; k < n_pn; k++) p_cor = p_cor + (v3(pn[4 * k + 1], pn[4 * k + 2], pn[4 * k + 3]) - p0) * pn[4 * k];
        p_cor     = m3_vec(m3_scale(iewn_skew, 2.0), p_cor);
        V3 v_cor  = m3_vec(m3_scale(iewn_skew, 2.0), p1 - p0);
        Q4 qnn    = rotvec2quat(-iewn * T);
        V3 dpn    = p1 - p0 - v0 * T - 0.5 * gravity * T * T + p_cor;
        V3 dvn    = v1 - v0 - gravity * T + v_cor;
        Q4 qb0b1  = qmul(qmul(qinv(q1), qnn), q0);
        V3 rp     = m3_vec(cnb0, dpn) - corrected_p;
        V3 rv     = m3_vec(cnb0, dvn) - corrected_v;
        Q4 qe     = qmul(qb0b1, corrected_q);
        r[0] = rp.x, r[1] = rp.y, r[2] = rp.z, r[3] = rv.x, r[4] = rv.y, r[5] = rv.z;
        r[6] = 2 * qe.x, r[7] = 2 * qe.y, r[8] = 2 * qe.z;
        // pose0 :92-110
        put(J0, 0, 0, m3_sub(m3_neg(cnb0), m3_scale(m3_mul(m3_scale(cnb0, 2.0), iewn_skew), T)));
        put(J0, 0, 3, skew(m3_vec(cnb0, dpn)));
        put(J0, 3, 0, m3_mul(m3_scale(cnb0, -2.0), iewn_skew));
        put(J0, 3, 3, skew(m3_vec(cnb0, dvn)));
        put(J0, 6, 3, qleft_qright_br(qb0b1, corrected_q));
        // pose1 :112-125
        put(J2, 0, 0, cnb0);
        put(J2, 3, 0, m3_mul(m3_scale(cnb0, 2.0), iewn_skew));
        put(J2, 6, 3, m3_neg(qright_br(qe)));
        // mix0 :127-152
        put(J1, 0, 0, m3_scale(m3_neg(cnb0), T));
        put(J1, 0, 3, m3_neg(dp_dbg));
        put(J1, 0, 6, m3_neg(dp_dba));
        put(J1, 3, 0, m3_neg(cnb0));
        put(J1, 3, 3, m3_neg(dv_dbg));
        put(J1, 3, 6, m3_neg(dv_dba));
        put(J1, 6, 3, m3_mul(qleft_br(qmul(qb0b1, d.q)), dq_dbg));
        put(J1, 9, 3, m3_neg(m3_identity()));
        put(J1, 12, 6, m3_neg(m3_identity()));
        // mix1 :154-164
        put(J3, 3, 0, cnb0);
        put(J3, 9, 3, m3_identity());
        put(J3, 12, 6, m3_identity());
    }
    V3 rbg = bg1 - bg0, rba = ba1 - ba0;
    r[9] = rbg.x, r[10] = rbg.y, r[11] = rbg.z, r[12] = rba.x, r[13] = rba.y, r[14] = rba.z;

    for (int i = 0; i < 15; i++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += S[i][k] * r[k];
        residuals[i] = s;
    }
    if (!jacobians) return;
    auto emit = [&](auto &J, int cols, double *out) {
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < cols; j++) {
                double s = 0;
                for (int k = 0; k < 15; k++) s += S[i][k] * J[k][j];
                out[i * cols + j] = s;
            }
    };
    emit(J0, 7, jacobians);
    emit(J1, 9, jacobians + 105);
    emit(J2, 7, jacobians + 240);
    emit(J3, 9, jacobians + 345);
}

} // extern "C"
