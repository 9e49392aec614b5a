// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own IMU preintegration (PreintegrationNormal / PreintegrationEarth:
// /root/reference/ic_gvins/ic_gvins/preintegration/preintegration_{base,normal,earth}.{h,cc}, preintegration_factor.h, common/earth.h,
// compiled unmodified from where they lie) behind C entry points.  Linear algebra comes from the Eigen-interface shim in
// shim/ (NOT real Eigen — stated in DESIGN.md).
#include "common/earth.h"
#include "preintegration/preintegration_earth.h"
#include "preintegration/preintegration_factor.h"
#include "preintegration/preintegration_normal.h"

#include "preintegration/preintegration_base.cc"   // the reference sources themselves (single translation unit)
#include "preintegration/preintegration_earth.cc"
#include "preintegration/preintegration_normal.cc"

namespace {
template <typename Base> struct ProbeT : public Base { // protected state made readable
    using Base::Base;
    const Eigen::MatrixXd &jac() const { return this->jacobian_; }
    const Eigen::MatrixXd &cov() const { return this->covariance_; }
};
typedef ProbeT<PreintegrationNormal> Probe;
typedef ProbeT<PreintegrationEarth> ProbeEarth;

IntegrationState make_state(const double *s16) { // p3 q4(xyzw) v3 bg3 ba3
    IntegrationState st;
    st.time = 0;
    st.p    = Vector3d(s16[0], s16[1], s16[2]);
    st.q    = Quaterniond(s16[6], s16[3], s16[4], s16[5]);
    st.v    = Vector3d(s16[7], s16[8], s16[9]);
    st.bg   = Vector3d(s16[10], s16[11], s16[12]);
    st.ba   = Vector3d(s16[13], s16[14], s16[15]);
    return st;
}
void put_state(const IntegrationState &st, double *s16) {
    for (int k = 0; k < 3; k++) {
        s16[k]      = st.p[k];
        s16[7 + k]  = st.v[k];
        s16[10 + k] = st.bg[k];
        s16[13 + k] = st.ba[k];
    }
    s16[3] = st.q.x(), s16[4] = st.q.y(), s16[5] = st.q.z(), s16[6] = st.q.w();
}
template <typename P>
std::shared_ptr<P> build_t(int n_imu, const double *imu, const double *state0, const double *params, const double *station) {
    auto par          = std::make_shared<IntegrationParameters>();
    if (station) par->station = Vector3d(station[0], station[1], station[2]);
    par->gyr_arw      = params[0];
    par->acc_vrw      = params[1];
    par->gyr_bias_std = params[2];
    par->acc_bias_std = params[3];
    par->corr_time    = params[4];
    par->gravity      = params[5];
    auto mk = [&](int k) {
        IMU m;
        m.time   = imu[8 * k];
        m.dt     = imu[8 * k + 1];
        m.dtheta = Vector3d(imu[8 * k + 2], imu[8 * k + 3], imu[8 * k + 4]);
        m.dvel   = Vector3d(imu[8 * k + 5], imu[8 * k + 6], imu[8 * k + 7]);
        m.odovel = 0;
        return m;
    };
    auto pre = std::make_shared<P>(par, mk(0), make_state(state0));
    for (int k = 1; k < n_imu; k++) pre->addNewImu(mk(k));
    return pre;
}
std::shared_ptr<Probe> build(int n_imu, const double *imu, const double *state0, const double *params) {
    return build_t<Probe>(n_imu, imu, state0, params, nullptr);
}
template <typename P> void dump(P &pre, double *cur_state, double *delta_state, double *jac, double *cov, double *delta_time) {
    put_state(pre.currentState(), cur_state);
    put_state(pre.deltaState(), delta_state);
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            jac[i * 15 + j] = pre.jac().get(i, j);
            cov[i * 15 + j] = pre.cov().get(i, j);
        }
    *delta_time = pre.deltaTime();
}
} // namespace

extern "C" {
// same argument layout as orc_preint_integrate (oracle/oracle.h); jac/cov row-major 15x15
int ref_preint_integrate(int n_imu, const double *imu, const double *state0, const double *params, double *cur_state,
                         double *delta_state, double *jac, double *cov, double *delta_time) {
    auto pre = build(n_imu, imu, state0, params);
    put_state(pre->currentState(), cur_state);
    put_state(pre->deltaState(), delta_state);
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            jac[i * 15 + j] = pre->jac().get(i, j);
            cov[i * 15 + j] = pre->cov().get(i, j);
        }
    *delta_time = pre->deltaTime();
    return 0;
}
// Earth variant (preintegration_earth.cc): station = origin (lat, lon, h) of the local frame; iewn3 returns
// Earth::iewn(station, p0), the rotation rate the reference derives at resetState (preintegration_earth.cc:320)
int ref_preint_integrate_earth(int n_imu, const double *imu, const double *state0, const double *params, const double *station,
                               double *cur_state, double *delta_state, double *jac, double *cov, double *delta_time,
                               double *iewn3) {
    auto pre = build_t<ProbeEarth>(n_imu, imu, state0, params, station);
    dump(*pre, cur_state, delta_state, jac, cov, delta_time);
    Vector3d w = Earth::iewn(Vector3d(station[0], station[1], station[2]), Vector3d(state0[0], state0[1], state0[2]));
    iewn3[0] = w[0], iewn3[1] = w[1], iewn3[2] = w[2];
    return 0;
}
int ref_preint_factor_earth(int n_imu, const double *imu, const double *state0, const double *params, const double *station,
                            const double *pose0, const double *mix0, const double *pose1, const double *mix1, double *residuals,
                            double *jacobians) {
    PreintegrationFactor f(build_t<ProbeEarth>(n_imu, imu, state0, params, station));
    const double *p[4] = {pose0, mix0, pose1, mix1};
    double *J[4]       = {jacobians, jacobians + 105, jacobians + 240, jacobians + 345};
    return f.Evaluate(p, residuals, jacobians ? J : nullptr) ? 0 : 1;
}
// PreintegrationFactor::Evaluate on the interval built from (imu, state0): residual 15, Jacobians 15x7, 15x9, 15x7, 15x9
int ref_preint_factor(int n_imu, const double *imu, const double *state0, const double *params, const double *pose0,
                      const double *mix0, const double *pose1, const double *mix1, double *residuals, double *jacobians) {
    PreintegrationFactor f(build(n_imu, imu, state0, params));
    const double *p[4] = {pose0, mix0, pose1, mix1};
    double *J[4]       = {jacobians, jacobians + 105, jacobians + 240, jacobians + 345};
    return f.Evaluate(p, residuals, jacobians ? J : nullptr) ? 0 : 1;
}
}
