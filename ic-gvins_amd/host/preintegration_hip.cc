// Preintegration on the HIP C ABI: P1 (the per-sample inner loop) runs batched on the device through icg_preint_batch;
// P2 (residual + Jacobians, preintegration_normal.cc:38-142 / preintegration_earth.cc:37-164, wrapped by
// preintegration_factor.h:45-69) is a handful of 3x3 / 15x15 operations per factor (<= 15 factors per solve) and stays
// on the host, in the Ceres callback thread that asks for it.
#include <cmath>
#include <cstring>

#include "earth.h"
#include "factors.h"

namespace icg {

namespace {
struct V3 {
    double x, y, z;
};
struct Q4 {
    double x, y, z, w;
};
struct M3 {
    double m[3][3];
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Q4 qinv(Q4 q) {
    double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return {-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
}
inline Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline V3 qrot(Q4 q, V3 v) {
    V3 qv{q.x, q.y, q.z};
    V3 uv = cross(qv, v);
    uv    = uv + uv;
    return v + q.w * uv + cross(qv, uv);
}
inline M3 qmat(Q4 q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return M3{{{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}}};
}
inline Q4 rotvec2quat(V3 rv) {
    double angle = std::sqrt(rv.x * rv.x + rv.y * rv.y + rv.z * rv.z);
    V3 axis      = rv;
    if (angle > 0) axis = rv * (1.0 / angle);
    double s = std::sin(0.5 * angle), c = std::cos(0.5 * angle);
    return {s * axis.x, s * axis.y, s * axis.z, c};
}
inline M3 skew(V3 v) { return M3{{{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}}}; }
inline M3 eye() { return M3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
inline M3 scale(const M3 &a, double s) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] * s;
    return r;
}
inline M3 add(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
inline M3 mul(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
inline V3 mv(const M3 &a, V3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 qleft_br(Q4 q) { return add(scale(eye(), q.w), skew({q.x, q.y, q.z})); }
inline M3 qright_br(Q4 q) { return add(scale(eye(), q.w), scale(skew({q.x, q.y, q.z}), -1.0)); }
M3 qleft_qright_br(Q4 a, Q4 b) { // bottom-right 3x3 of quaternionleft(a) * quaternionright(b), rotation.h:103-119
    double L[4][4], R[4][4];
    auto fill = [](double M[4][4], Q4 q, double sgn) {
        M[0][0] = q.w;
        M[0][1] = -q.x, M[0][2] = -q.y, M[0][3] = -q.z;
        M[1][0] = q.x, M[2][0] = q.y, M[3][0] = q.z;
        M3 s = skew({q.x, q.y, q.z});
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

void stateToArray(const IntegrationState &s, double *a) {
    a[0] = s.p[0], a[1] = s.p[1], a[2] = s.p[2];
    a[3] = s.q.x, a[4] = s.q.y, a[5] = s.q.z, a[6] = s.q.w;
    for (int i = 0; i < 3; i++) {
        a[7 + i]  = s.v[i];
        a[10 + i] = s.bg[i];
        a[13 + i] = s.ba[i];
    }
}
void stateFromArray(const double *a, IntegrationState &s) {
    s.p = Vector3d(a[0], a[1], a[2]);
    s.q = Quaterniond{a[3], a[4], a[5], a[6]};
    s.v  = Vector3d(a[7], a[8], a[9]);
    s.bg = Vector3d(a[10], a[11], a[12]);
    s.ba = Vector3d(a[13], a[14], a[15]);
}
} // namespace

static void refreshEarthRate(const IntegrationParameters &P, const IntegrationState &state, Vector3d &iewn);

Preintegration::Preintegration(std::shared_ptr<IntegrationParameters> parameters, const IMU &imu0, const IntegrationState &state,
                               Variant v)
    : parameters_(std::move(parameters)), variant_(v), start_state_(state), current_state_(state) {
    imu_buffer_.push_back(imu0);
    delta_state_.bg = state.bg;
    delta_state_.ba = state.ba;
    jacobian_.assign(225, 0.0);
    covariance_.assign(225, 0.0);
    for (int i = 0; i < 15; i++) jacobian_[(size_t) i * 16] = 1.0;
    refreshEarthRate(*parameters_, state, iewn_);
}

// preintegration_earth.cc:319-321 (resetState, run by the constructor and by every reintegration)
static void refreshEarthRate(const IntegrationParameters &P, const IntegrationState &state, Vector3d &iewn) {
    iewn = P.has_station ? Earth::iewn(P.station, state.p) : P.iewn;
}

void Preintegration::reintegration(const IntegrationState &state) {
    start_state_ = state;
    dirty_       = true;
    if (variant_ == EARTH) refreshEarthRate(*parameters_, state, iewn_);
}

bool Preintegration::integrateBatch(icg_ctx *ctx, const vector<Preintegration *> &list, std::string *err) {
    // one launch per (variant, parameter values): icg_preint_batch takes one parameter vector per call, and every interval must be
    // integrated with ITS OWN Earth rate (the one evaluate() uses).  Intervals of one estimator share their values in practice.
    auto paramsOf = [](const Preintegration *p, double *out9) {
        const IntegrationParameters &P = *p->parameters_;
        const double v[9] = {P.gyr_arw, P.acc_vrw, P.gyr_bias_std, P.acc_bias_std, P.corr_time, P.gravity, p->iewn_[0], p->iewn_[1], p->iewn_[2]};
        memcpy(out9, v, sizeof v);
    };
    vector<Preintegration *> pending;
    for (auto *p : list)
        if (p->dirty_) pending.push_back(p);
    while (!pending.empty()) {
        double params[9];
        paramsOf(pending[0], params);
        const int variant = (int) pending[0]->variant_;
        vector<Preintegration *> todo, rest;
        for (auto *p : pending) {
            double q[9];
            paramsOf(p, q);
            ((int) p->variant_ == variant && memcmp(q, params, sizeof q) == 0 ? todo : rest).push_back(p);
        }
        pending.swap(rest);
        vector<int32_t> offsets{0};
        vector<double> imu, state0;
        for (auto *p : todo) {
            for (const IMU &s : p->imu_buffer_) {
                const double row[8] = {s.time, s.dt, s.dtheta[0], s.dtheta[1], s.dtheta[2], s.dvel[0], s.dvel[1], s.dvel[2]};
                imu.insert(imu.end(), row, row + 8);
            }
            offsets.push_back((int32_t) (imu.size() / 8));
            double a[16];
            stateToArray(p->start_state_, a);
            state0.insert(state0.end(), a, a + 16);
        }
        const size_t n = todo.size();
        vector<double> cur(16 * n), del(16 * n), jac(225 * n), cov(225 * n), dt(n), pn(imu.size() / 2);
        int rc = icg_preint_batch(ctx, variant, (int) n, offsets.data(), imu.data(), state0.data(), params, cur.data(), del.data(),
                                  jac.data(), cov.data(), dt.data(), pn.data());
        if (rc != ICG_OK) {
            if (err) *err = icg_last_error(ctx);
            return false;
        }
        for (size_t k = 0; k < n; k++) {
            Preintegration *p = todo[k];
            stateFromArray(&cur[16 * k], p->current_state_);
            stateFromArray(&del[16 * k], p->delta_state_);
            p->current_state_.time = p->imu_buffer_.back().time;
            p->jacobian_.assign(jac.begin() + 225 * (long) k, jac.begin() + 225 * (long) (k + 1));
            p->covariance_.assign(cov.begin() + 225 * (long) k, cov.begin() + 225 * (long) (k + 1));
            p->delta_time_ = dt[k];
            const int b = offsets[k], cnt = offsets[k + 1] - offsets[k];
            p->pn_.assign(pn.begin() + 4 * (long) b, pn.begin() + 4 * (long) (b + cnt - 1));
            p->updateSqrtInformation();
            p->dirty_ = false;
        }
    }
    return true;
}

// sqrt_information = LLT(cov^-1).matrixL().transpose()  (normal :39-40, earth :39-40).  The reference forms it inside every evaluate();
// it only depends on the covariance, so it is formed once when an integration result arrives (same arithmetic, same value).
void Preintegration::updateSqrtInformation() {
    typedef double M15[15][15];
    M15 cov, inv, L;
    memcpy(cov, covariance_.data(), sizeof cov);
    sqrt_information_ok_ = false;
    sqrt_information_.assign(225, 0.0);
    double w[15][30];
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            w[i][j]      = cov[i][j];
            w[i][15 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 15; c++) {
        int piv = c;
        for (int r = c + 1; r < 15; r++)
            if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return;
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
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < i; j++) inv[j][i] = inv[i][j];
    memset(L, 0, sizeof L);
    for (int j = 0; j < 15; j++) {
        double s = inv[j][j];
        for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        L[j][j] = std::sqrt(s);
        for (int i = j + 1; i < 15; i++) {
            double t = inv[i][j];
            for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
            L[i][j] = t / L[j][j];
        }
    }
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) sqrt_information_[(size_t) i * 15 + j] = L[j][i];
    sqrt_information_ok_ = true;
}

bool Preintegration::evaluate(const double *const *parameters, double *residuals, double **jacobians) const {
    if (dirty_) return false; // not integrated for the current buffer/start state: evaluation failed, loudly
    if (!sqrt_information_ok_) return false; // singular covariance
    typedef double M15[15][15];
    M15 jac, S;
    memcpy(jac, jacobian_.data(), sizeof jac);
    memcpy(S, sqrt_information_.data(), sizeof S);
    // constructState (normal :162-180)
    const double *pose0 = parameters[0], *mix0 = parameters[1], *pose1 = parameters[2], *mix1 = parameters[3];
    V3 p0{pose0[0], pose0[1], pose0[2]}, p1{pose1[0], pose1[1], pose1[2]};
    Q4 q0{pose0[3], pose0[4], pose0[5], pose0[6]}, q1{pose1[3], pose1[4], pose1[5], pose1[6]};
    V3 v0{mix0[0], mix0[1], mix0[2]}, bg0{mix0[3], mix0[4], mix0[5]}, ba0{mix0[6], mix0[7], mix0[8]};
    V3 v1{mix1[0], mix1[1], mix1[2]}, bg1{mix1[3], mix1[4], mix1[5]}, ba1{mix1[6], mix1[7], mix1[8]};
    V3 gravity{0, 0, parameters_->gravity};
    V3 iewn{iewn_[0], iewn_[1], iewn_[2]};
    const IntegrationState &d = delta_state_;
    V3 dp{d.p[0], d.p[1], d.p[2]}, dv{d.v[0], d.v[1], d.v[2]}, dbg0{d.bg[0], d.bg[1], d.bg[2]}, dba0{d.ba[0], d.ba[1], d.ba[2]};
    Q4 dq{d.q.x, d.q.y, d.q.z, d.q.w};
    auto blk = [&](int r, int c) {
        M3 b;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) b.m[i][j] = jac[r + i][c + j];
        return b;
    };
    M3 dp_dbg = blk(0, 9), dp_dba = blk(0, 12), dv_dbg = blk(3, 9), dv_dba = blk(3, 12), dq_dbg = blk(6, 9);
    V3 dbg = bg0 - dbg0, dba = ba0 - dba0;
    V3 corrected_p = dp + mv(dp_dba, dba) + mv(dp_dbg, dbg);
    V3 corrected_v = dv + mv(dv_dba, dba) + mv(dv_dbg, dbg);
    Q4 corrected_q = qmul(dq, rotvec2quat(mv(dq_dbg, dbg)));
    const double T = delta_time_;

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
    if (variant_ == NORMAL) {
        V3 dpn = p1 - p0 - v0 * T - 0.5 * gravity * T * T;
        V3 dvn = v1 - v0 - gravity * T;
        V3 rp = qrot(qinv(q0), dpn) - corrected_p, rv = qrot(qinv(q0), dvn) - corrected_v;
        Q4 qe = qmul(qmul(qinv(corrected_q), qinv(q0)), q1);
        r[0] = rp.x, r[1] = rp.y, r[2] = rp.z, r[3] = rv.x, r[4] = rv.y, r[5] = rv.z, r[6] = 2 * qe.x, r[7] = 2 * qe.y, r[8] = 2 * qe.z;
        put(J0, 0, 0, scale(cnb0, -1.0));
        put(J0, 0, 3, skew(qrot(qinv(q0), dpn)));
        put(J0, 3, 3, skew(qrot(qinv(q0), dvn)));
        put(J0, 6, 3, scale(qleft_qright_br(qmul(qinv(q1), q0), corrected_q), -1.0));
        put(J2, 0, 0, cnb0);
        put(J2, 6, 3, qleft_br(qe));
        put(J1, 0, 0, scale(cnb0, -T));
        put(J1, 0, 3, scale(dp_dbg, -1.0));
        put(J1, 0, 6, scale(dp_dba, -1.0));
        put(J1, 3, 0, scale(cnb0, -1.0));
        put(J1, 3, 3, scale(dv_dbg, -1.0));
        put(J1, 3, 6, scale(dv_dba, -1.0));
        put(J1, 6, 3, mul(scale(qleft_br(qmul(qmul(qinv(q1), q0), dq)), -1.0), dq_dbg));
    } else {
        M3 iewn_skew = skew(iewn);
        V3 p_cor{0, 0, 0};
        for (size_t k = 0; k + 3 < pn_.size(); k += 4) p_cor = p_cor + (V3{pn_[k + 1], pn_[k + 2], pn_[k + 3]} - p0) * pn_[k];
        p_cor    = mv(scale(iewn_skew, 2.0), p_cor);
        V3 v_cor = mv(scale(iewn_skew, 2.0), p1 - p0);
        Q4 qnn   = rotvec2quat(-iewn * T);
        V3 dpn   = p1 - p0 - v0 * T - 0.5 * gravity * T * T + p_cor;
        V3 dvn   = v1 - v0 - gravity * T + v_cor;
        Q4 qb0b1 = qmul(qmul(qinv(q1), qnn), q0);
        V3 rp = mv(cnb0, dpn) - corrected_p, rv = mv(cnb0, dvn) - corrected_v;
        Q4 qe = qmul(qb0b1, corrected_q);
        r[0] = rp.x, r[1] = rp.y, r[2] = rp.z, r[3] = rv.x, r[4] = rv.y, r[5] = rv.z, r[6] = 2 * qe.x, r[7] = 2 * qe.y, r[8] = 2 * qe.z;
        put(J0, 0, 0, add(scale(cnb0, -1.0), scale(mul(scale(cnb0, 2.0), iewn_skew), -T)));
        put(J0, 0, 3, skew(mv(cnb0, dpn)));
        put(J0, 3, 0, mul(scale(cnb0, -2.0), iewn_skew));
        put(J0, 3, 3, skew(mv(cnb0, dvn)));
        put(J0, 6, 3, qleft_qright_br(qb0b1, corrected_q));
        put(J2, 0, 0, cnb0);
        put(J2, 3, 0, mul(scale(cnb0, 2.0), iewn_skew));
        put(J2, 6, 3, scale(qright_br(qe), -1.0));
        put(J1, 0, 0, scale(cnb0, -T));
        put(J1, 0, 3, scale(dp_dbg, -1.0));
        put(J1, 0, 6, scale(dp_dba, -1.0));
        put(J1, 3, 0, scale(cnb0, -1.0));
        put(J1, 3, 3, scale(dv_dbg, -1.0));
        put(J1, 3, 6, scale(dv_dba, -1.0));
        put(J1, 6, 3, mul(qleft_br(qmul(qb0b1, dq)), dq_dbg));
    }
    put(J1, 9, 3, scale(eye(), -1.0));
    put(J1, 12, 6, scale(eye(), -1.0));
    put(J3, 3, 0, cnb0);
    put(J3, 9, 3, eye());
    put(J3, 12, 6, eye());
    V3 rbg = bg1 - bg0, rba = ba1 - ba0;
    r[9] = rbg.x, r[10] = rbg.y, r[11] = rbg.z, r[12] = rba.x, r[13] = rba.y, r[14] = rba.z;
    for (int i = 0; i < 15; i++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += S[i][k] * r[k];
        residuals[i] = s;
    }
    if (jacobians) {
        auto emit = [&](auto &J, int cols, double *out) {
            if (!out) return;
            for (int i = 0; i < 15; i++)
                for (int j = 0; j < cols; j++) {
                    double s = 0;
                    for (int k = 0; k < 15; k++) s += S[i][k] * J[k][j];
                    out[i * cols + j] = s;
                }
        };
        emit(J0, 7, jacobians[0]);
        emit(J1, 9, jacobians[1]);
        emit(J2, 7, jacobians[2]);
        emit(J3, 9, jacobians[3]);
    }
    return true;
}

PreintegrationFactor::PreintegrationFactor(std::shared_ptr<Preintegration> preintegration)
    : preintegration_(std::move(preintegration)) {
    *mutable_parameter_block_sizes() = std::vector<int32_t>{7, 9, 7, 9}; // numBlocksParameters (normal :148-150)
    set_num_residuals(15);
}

} // namespace icg
