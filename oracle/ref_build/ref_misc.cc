// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own INS helpers (MISC::insMechanization, getCameraPoseFromInsWindow,
// statePoseInterpolation, getImuSeriesFromTo, redoInsMechanization: /root/reference/ic_gvins/ic_gvins/misc.{h,cc}, compiled
// unmodified from where they lie) behind C entry points.  Linear algebra comes from the Eigen-interface shim in shim/ (NOT real
// Eigen — stated in DESIGN.md).  SURVEY.md §8 row f4.
#include "misc.h"

#include "fileio/filesaver.cc" // the reference sources themselves (single translation unit)
#include "misc.cc"

namespace {
typedef std::deque<std::pair<IMU, IntegrationState>> InsWindow;

IMU make_imu(const double *m) { // time, dt, dtheta3, dvel3
    IMU imu;
    imu.time   = m[0];
    imu.dt     = m[1];
    imu.dtheta = Vector3d(m[2], m[3], m[4]);
    imu.dvel   = Vector3d(m[5], m[6], m[7]);
    imu.odovel = 0;
    return imu;
}
void put_imu(const IMU &imu, double *m) {
    m[0] = imu.time, m[1] = imu.dt;
    for (int k = 0; k < 3; k++) m[2 + k] = imu.dtheta[k], m[5 + k] = imu.dvel[k];
}
// state23: time, p3, q4 (x y z w), v3, bg3, ba3, sg3, sa3
IntegrationState make_state(const double *s) {
    IntegrationState st;
    st.time = s[0];
    st.p    = Vector3d(s[1], s[2], s[3]);
    st.q    = Quaterniond(s[7], s[4], s[5], s[6]);
    st.v    = Vector3d(s[8], s[9], s[10]);
    st.bg   = Vector3d(s[11], s[12], s[13]);
    st.ba   = Vector3d(s[14], s[15], s[16]);
    st.sg   = Vector3d(s[17], s[18], s[19]);
    st.sa   = Vector3d(s[20], s[21], s[22]);
    return st;
}
void put_state(const IntegrationState &st, double *s) {
    s[0] = st.time;
    for (int k = 0; k < 3; k++) {
        s[1 + k]  = st.p[k];
        s[8 + k]  = st.v[k];
        s[11 + k] = st.bg[k];
        s[14 + k] = st.ba[k];
        s[17 + k] = st.sg[k];
        s[20 + k] = st.sa[k];
    }
    s[4] = st.q.x(), s[5] = st.q.y(), s[6] = st.q.z(), s[7] = st.q.w();
}
// cfg8: gravity3, iewn3, iswithearth, iswithscale
IntegrationConfiguration make_config(const double *c) {
    IntegrationConfiguration cfg;
    cfg.isuseodo    = false;
    cfg.origin      = Vector3d(0, 0, 0);
    cfg.gravity     = Vector3d(c[0], c[1], c[2]);
    cfg.iewn        = Vector3d(c[3], c[4], c[5]);
    cfg.iswithearth = c[6] != 0;
    cfg.iswithscale = c[7] != 0;
    return cfg;
}
Pose make_pose(const double *p12) { // R row-major 9, t 3
    Pose p;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) p.R(i, j) = p12[3 * i + j];
        p.t[i] = p12[9 + i];
    }
    return p;
}
void put_pose(const Pose &p, double *p12) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) p12[3 * i + j] = p.R(i, j);
        p12[9 + i] = p.t[i];
    }
}
InsWindow make_window(int n, const double *imu, const double *states) {
    InsWindow w;
    for (int k = 0; k < n; k++) {
        IntegrationState st;
        if (states) st = make_state(states + 23 * (size_t) k);
        w.emplace_back(make_imu(imu + 8 * (size_t) k), st);
    }
    return w;
}
} // namespace

extern "C" {
// MISC::insMechanization (misc.cc:151-206) applied to imu[1..n-1] in sequence starting from state23 (in/out); traj (may be
// NULL) receives the state after every sample, (n-1) x 23
int ref_ins_mechanize(const double *cfg8, int n_imu, const double *imu, double *state23, double *traj) {
    IntegrationConfiguration cfg = make_config(cfg8);
    IntegrationState st          = make_state(state23);
    for (int k = 1; k < n_imu; k++) {
        MISC::insMechanization(cfg, make_imu(imu + 8 * (size_t) (k - 1)), make_imu(imu + 8 * (size_t) k), st);
        if (traj) put_state(st, traj + 23 * (size_t) (k - 1));
    }
    put_state(st, state23);
    return 0;
}
// MISC::getCameraPoseFromInsWindow (misc.cc:67-83); window: IMU times (imu n x 8) + states (n x 23).  Returns found (0/1)
int ref_ins_camera_pose(int n_win, const double *imu, const double *states, const double *pose_b_c12, double time, double *pose12) {
    Pose pose;
    bool ok = MISC::getCameraPoseFromInsWindow(make_window(n_win, imu, states), make_pose(pose_b_c12), time, pose);
    put_pose(pose, pose12);
    return ok ? 1 : 0;
}
size_t ref_ins_window_index(int n_win, const double *imu, double time) {
    return MISC::getInsWindowIndex(make_window(n_win, imu, nullptr), time);
}
// MISC::getImuSeriesFromTo (misc.cc:307-361): -1 when it fails, else the number of samples written (cap x 8)
int ref_imu_series(int n_win, const double *imu, double start, double end, int cap, double *series) {
    vector<IMU> out;
    if (!MISC::getImuSeriesFromTo(make_window(n_win, imu, nullptr), start, end, out)) return -1;
    if ((int) out.size() > cap) return -2;
    for (size_t k = 0; k < out.size(); k++) put_imu(out[k], series + 8 * k);
    return (int) out.size();
}
// MISC::writeNavResult (misc.cc:417-499) through the reference's own FileSaver: writes <dir>/nav.txt, err.txt, traj.txt (text).  The
// function keeps a static call counter and only writes every 10th call: `calls` consecutive calls are made with the same state,
// so ceil(calls / 10) rows appear (call it with multiples of 10 to keep the counter aligned between invocations).  origin = config.origin (lat, lon [rad], h).  sodo is state.sodo.
int ref_write_nav_result(const double *cfg8, const double *origin3, const double *state23, double sodo, const char *dir, int calls) {
    IntegrationConfiguration cfg = make_config(cfg8);
    cfg.origin                   = Vector3d(origin3[0], origin3[1], origin3[2]);
    IntegrationState st          = make_state(state23);
    st.sodo                      = sodo;
    std::string d(dir);
    auto nav  = FileSaver::create(d + "/nav.txt", 11);
    auto errf = FileSaver::create(d + "/err.txt", 7);
    auto traj = FileSaver::create(d + "/traj.txt", 8);
    if (!nav->isOpen() || !errf->isOpen() || !traj->isOpen()) return -1;
    for (int k = 0; k < calls; k++) MISC::writeNavResult(cfg, st, nav, errf, traj);
    return 0;
}

// MISC::redoInsMechanization (misc.cc:208-261): window (imu n x 8, states n x 23) updated in place; returns the new window length
// (expired entries are dropped from the front: imu/states are compacted to the front of the arrays)
int ref_redo_ins(const double *cfg8, const double *updated_state23, int reserved, int n_win, double *imu, double *states) {
    InsWindow w = make_window(n_win, imu, states);
    MISC::redoInsMechanization(make_config(cfg8), make_state(updated_state23), (size_t) reserved, w);
    for (size_t k = 0; k < w.size(); k++) {
        put_imu(w[k].first, imu + 8 * k);
        put_state(w[k].second, states + 23 * k);
    }
    return (int) w.size();
}
}
