// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the INS helpers in front of the tracker:
//   MISC::getInsWindowIndex          misc.cc:30-65      MISC::getCameraPoseFromInsWindow  misc.cc:67-83
//   MISC::statePoseInterpolation     misc.cc:85-100     MISC::stateToCameraPose           misc.cc:102-108
//   MISC::insMechanization           misc.cc:151-206    MISC::redoInsMechanization        misc.cc:208-261
//   MISC::isNeedInterpolation        misc.cc:263-286    MISC::imuInterpolation            misc.cc:288-305
//   MISC::getImuSeriesFromTo         misc.cc:307-361    Rotation::quaternion2vector       common/rotation.h:78-81
// PINNED against the reference's own misc.cc compiled unmodified (oracle/ref_build -> oracle/_ref/libref_misc.so;
// tests/golden/ins_ref_golden.npz): index/series decisions exact, states and poses to 1e-12.
// Layouts: imu rows of 8 (time, dt, dtheta3, dvel3); state rows of 23 (time, p3, q4 xyzw, v3, bg3, ba3, sg3, sa3);
// cfg8 = gravity3, iewn3, iswithearth, iswithscale; poses 12 = R row-major 9, t 3.
#include "oracle.h"
#include "orc_math.h"
#include <vector>

using namespace orc;

namespace {
const double MINIMUM_TIME_INTERVAL = 0.0001; // misc.h:72

struct Imu {
    double time, dt;
    V3 dtheta, dvel;
};
Imu load_imu(const double *p) { return Imu{p[0], p[1], v3(p[2], p[3], p[4]), v3(p[5], p[6], p[7])}; }
void store_imu(const Imu &m, double *p) {
    p[0] = m.time, p[1] = m.dt;
    p[2] = m.dtheta.x, p[3] = m.dtheta.y, p[4] = m.dtheta.z;
    p[5] = m.dvel.x, p[6] = m.dvel.y, p[7] = m.dvel.z;
}
struct State {
    double time;
    V3 p;
    Q4 q;
    V3 v, bg, ba, sg, sa;
};
State load_state(const double *s) {
    return State{s[0], v3(s[1], s[2], s[3]), Q4{s[4], s[5], s[6], s[7]}, v3(s[8], s[9], s[10]), v3(s[11], s[12], s[13]),
                 v3(s[14], s[15], s[16]), v3(s[17], s[18], s[19]), v3(s[20], s[21], s[22])};
}
void store_state(const State &st, double *s) {
    s[0] = st.time;
    s[1] = st.p.x, s[2] = st.p.y, s[3] = st.p.z;
    s[4] = st.q.x, s[5] = st.q.y, s[6] = st.q.z, s[7] = st.q.w;
    s[8] = st.v.x, s[9] = st.v.y, s[10] = st.v.z;
    s[11] = st.bg.x, s[12] = st.bg.y, s[13] = st.bg.z;
    s[14] = st.ba.x, s[15] = st.ba.y, s[16] = st.ba.z;
    s[17] = st.sg.x, s[18] = st.sg.y, s[19] = st.sg.z;
    s[20] = st.sa.x, s[21] = st.sa.y, s[22] = st.sa.z;
}
V3 scale3(V3 a, V3 one_minus) { return v3(a.x * one_minus.x, a.y * one_minus.y, a.z * one_minus.z); }

// misc.cc:151-206
void mechanize(const double *cfg8, const Imu &pre, const Imu &cur, State &st) {
    const V3 gravity = v3(cfg8[0], cfg8[1], cfg8[2]), iewn = v3(cfg8[3], cfg8[4], cfg8[5]);
    const bool withearth = cfg8[6] != 0, withscale = cfg8[7] != 0;
    V3 cur_dtheta = cur.dtheta - cur.dt * st.bg, cur_dvel = cur.dvel - cur.dt * st.ba; // :155-159
    V3 pre_dtheta = pre.dtheta - pre.dt * st.bg, pre_dvel = pre.dvel - pre.dt * st.ba;
    if (withscale) { // :161-168
        V3 og = v3(1.0 - st.sg.x, 1.0 - st.sg.y, 1.0 - st.sg.z), oa = v3(1.0 - st.sa.x, 1.0 - st.sa.y, 1.0 - st.sa.z);
        cur_dtheta = scale3(cur_dtheta, og), cur_dvel = scale3(cur_dvel, oa);
        pre_dtheta = scale3(pre_dtheta, og), pre_dvel = scale3(pre_dvel, oa);
    }
    const double dt = cur.dt;
    st.time         = cur.time;
    V3 dvfb   = (cur_dvel + 0.5 * cross(cur_dtheta, cur_dvel)) +
              1.0 / 12.0 * (cross(pre_dtheta, cur_dvel) + cross(pre_dvel, cur_dtheta)); // :174-175
    V3 dtheta = cur_dtheta + 1.0 / 12.0 * cross(pre_dtheta, cur_dtheta);              // :176
    V3 dvel;
    if (withearth) { // :181-193
        V3 dv_cor_g = (gravity - 2.0 * cross(iewn, st.v)) * dt;
        V3 dnn      = (-iewn) * dt;
        Q4 qnn      = rotvec2quat(dnn);
        M3 half     = m3_scale(m3_add(m3_identity(), qmat(qnn)), 0.5);
        dvel        = m3_vec(m3_mul(half, qmat(st.q)), dvfb) + dv_cor_g;
        st.q        = qnormalized(qmul(qmul(qnn, st.q), rotvec2quat(dtheta)));
    } else { // :194-200
        dvel = m3_vec(qmat(st.q), dvfb) + gravity * dt;
        st.q = qnormalized(qmul(st.q, rotvec2quat(dtheta)));
    }
    st.p = st.p + (dt * st.v + (0.5 * dt) * dvel); // :203
    st.v = st.v + dvel;                            // :205
}

// Rotation::quaternion2vector (rotation.h:78-81) = Eigen AngleAxis(q): angle * axis
V3 quat2rotvec(Q4 q) {
    V3 vec   = v3(q.x, q.y, q.z);
    double n = norm(vec);
    if (n != 0.0) {
        double angle = 2.0 * std::atan2(n, std::fabs(q.w));
        if (q.w < 0) n = -n;
        return angle * (vec / n);
    }
    return 0.0 * v3(1, 0, 0);
}

// misc.cc:30-65 over the IMU times of a window (imu rows of 8)
size_t window_index(int n, const double *imu, double time) {
    auto t = [&](size_t k) { return imu[8 * k]; };
    if (n <= 0 || t(0) > time || t((size_t) n - 1) <= time) return 0;
    size_t index = 0, sta = 0, end = (size_t) n;
    int counts = 0;
    while (true) {
        size_t mid    = (sta + end) / 2;
        double first  = t(mid - 1), second = t(mid);
        if (first <= time && time < second) {
            index = mid;
            break;
        } else if (first > time) {
            end = mid;
        } else if (second <= time) {
            sta = mid;
        }
        if (counts++ > 15) break;
    }
    return index;
}

int need_interpolation(const Imu &imu0, const Imu &imu1, double mid) { // misc.cc:263-286
    if (imu0.time < mid && imu1.time > mid) {
        double dt = mid - imu0.time;
        if (dt < MINIMUM_TIME_INTERVAL) return -1;
        dt = imu1.time - mid;
        if (dt < MINIMUM_TIME_INTERVAL) return 1;
        return 2;
    }
    return 0;
}
void imu_interpolation(const Imu &imu01, Imu &imu00, Imu &imu11, double mid) { // misc.cc:288-305 (imu11 may alias imu01)
    double scale = (imu01.time - mid) / imu01.dt;
    Imu buff     = imu01;
    imu00.time   = mid;
    imu00.dt     = buff.dt - (buff.time - mid);
    imu00.dtheta = buff.dtheta * (1 - scale);
    imu00.dvel   = buff.dvel * (1 - scale);
    imu11.time   = buff.time;
    imu11.dt     = buff.time - mid;
    imu11.dtheta = buff.dtheta * scale;
    imu11.dvel   = buff.dvel * scale;
}
} // namespace

extern "C" {

void orc_ins_mechanize(const double *cfg8, int n_imu, const double *imu, double *state23, double *traj) {
    State st = load_state(state23);
    for (int k = 1; k < n_imu; k++) {
        mechanize(cfg8, load_imu(imu + 8 * (size_t) (k - 1)), load_imu(imu + 8 * (size_t) k), st);
        if (traj) store_state(st, traj + 23 * (size_t) (k - 1));
    }
    store_state(st, state23);
}

int64_t orc_ins_window_index(int n_win, const double *imu, double time) { return (int64_t) window_index(n_win, imu, time); }

int orc_ins_camera_pose(int n_win, const double *imu, const double *states, const double *pose_b_c12, double time,
                        double *pose12) {
    size_t index = window_index(n_win, imu, time);
    V3 p;
    Q4 q;
    if (index > 0) { // misc.cc:73-78 + statePoseInterpolation :85-100
        State s0 = load_state(states + 23 * (index - 1)), s1 = load_state(states + 23 * index);
        V3 dp    = s1.p - s0.p;
        Q4 dq    = qmul(qinv(s1.q), s0.q);
        V3 rvec  = quat2rotvec(dq);
        double scale = (time - s0.time) / (s1.time - s0.time);
        rvec     = rvec * scale;
        dq       = rotvec2quat(rvec);
        p        = s0.p + dp * scale;
        q        = qnormalized(qmul(s0.q, qinv(dq)));
    } else { // :79-82
        State s = load_state(states + 23 * (size_t) (n_win - 1));
        p = s.p, q = s.q;
    }
    M3 R  = qmat(q); // stateToCameraPose :102-108
    M3 Rb;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rb.m[i][j] = pose_b_c12[3 * i + j];
    V3 t  = p + m3_vec(R, v3(pose_b_c12[9], pose_b_c12[10], pose_b_c12[11]));
    M3 Rc = m3_mul(R, Rb);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) pose12[3 * i + j] = Rc.m[i][j];
    pose12[9] = t.x, pose12[10] = t.y, pose12[11] = t.z;
    return index > 0 ? 1 : 0;
}

int orc_imu_series(int n_win, const double *imu, double start, double end, int cap, double *series) {
    size_t is = window_index(n_win, imu, start), ie = window_index(n_win, imu, end);
    if (is == 0 && ie == 0) return -1;
    // the reference indexes window[is-1] with is == 0 when only one end is found (UB); flagged instead of reproduced
    if (is == 0 || ie == 0) return -3;
    std::vector<Imu> out;
    Imu imu0 = load_imu(imu + 8 * (is - 1)), imu1 = load_imu(imu + 8 * is), mid;
    int isneed = need_interpolation(imu0, imu1, start);
    if (isneed == -1) {
        out.push_back(imu0);
        out.push_back(imu1);
    } else if (isneed == 1) {
        out.push_back(imu1);
    } else if (isneed == 2) {
        imu_interpolation(imu1, mid, imu1, start);
        out.push_back(mid);
        out.push_back(imu1);
    }
    for (size_t k = is + 1; k + 1 < ie; k++) out.push_back(load_imu(imu + 8 * k)); // k < ie - 1, ie >= 1
    imu0   = load_imu(imu + 8 * (ie - 1));
    imu1   = load_imu(imu + 8 * ie);
    isneed = need_interpolation(imu0, imu1, end);
    if (isneed == -1) {
        out.push_back(imu0);
    } else if (isneed == 1) {
        out.push_back(imu0);
        out.push_back(imu1);
    } else if (isneed == 2) {
        out.push_back(imu0);
        imu_interpolation(imu1, mid, imu1, end);
        out.push_back(mid);
    }
    if (out.empty()) return -4; // series.back() on an empty vector in the reference (UB)
    out.back().time = end;
    if ((int) out.size() > cap) return -2;
    for (size_t k = 0; k < out.size(); k++) store_imu(out[k], series + 8 * k);
    return (int) out.size();
}

int orc_redo_ins(const double *cfg8, const double *updated_state23, int reserved, int n_win, double *imu, double *states) {
    State st     = load_state(updated_state23);
    size_t index = window_index(n_win, imu, st.time);
    if (index == 0) return n_win; // :214-217
    Imu imu0 = load_imu(imu + 8 * (index - 1)), imu1 = load_imu(imu + 8 * index);
    int isneed = need_interpolation(imu0, imu1, st.time);
    if (isneed == -1) {
        mechanize(cfg8, imu0, imu1, st);
        store_state(st, states + 23 * index);
    } else if (isneed == 1) {
        st.time = imu1.time;
        store_state(st, states + 23 * index);
    } else if (isneed == 2) {
        imu_interpolation(imu1, imu0, imu1, st.time);
        mechanize(cfg8, imu0, imu1, st);
        store_state(st, states + 23 * index);
    }
    for (size_t k = index + 1; k < (size_t) n_win; k++) { // :245-251
        imu0 = imu1;
        imu1 = load_imu(imu + 8 * k);
        mechanize(cfg8, imu0, imu1, st);
        store_state(st, states + 23 * k);
    }
    if (index < (size_t) reserved) return n_win; // :254-260
    size_t counts = index - (size_t) reserved;
    if (counts) {
        memmove(imu, imu + 8 * counts, sizeof(double) * 8 * ((size_t) n_win - counts));
        memmove(states, states + 23 * counts, sizeof(double) * 23 * ((size_t) n_win - counts));
    }
    return n_win - (int) counts;
}
}

// ---- result rows of MISC::writeNavResult (misc.cc:417-499): the three text lines the reference dumps through FileSaver ----------
//   nav  (11): 0, time, lat [deg], lon [deg], h, v3, roll, pitch, heading [deg]     (Earth::local2global earth.h:194-208,
//                                                                                      Rotation::matrix2euler rotation.h:44-66)
//   err  (7 | 13 values + sodo): time, bg [deg/h], ba [mGal] (, sg, sa [ppm]), sodo
//   traj (8): time, p3, q4 (x y z w)
// PINNED against the reference's own misc.cc + filesaver.cc (oracle/_ref/libref_misc.so): the text of the rows is compared.
namespace {
const double WGS84_RA = 6378137.0000000000, WGS84_E1 = 0.0066943799901413156; // earth.h:36-39
const double R2D = 180.0 / M_PI;                                                // angle.h:30
double earth_RN(double lat) { // earth.h:66-69
    double sinlat = std::sin(lat);
    return WGS84_RA / std::sqrt(1.0 - WGS84_E1 * sinlat * sinlat);
}
M3 earth_cne(V3 blh) { // earth.h:71-93
    double sinlat = std::sin(blh.x), sinlon = std::sin(blh.y), coslat = std::cos(blh.x), coslon = std::cos(blh.y);
    M3 d;
    d.m[0][0] = -sinlat * coslon, d.m[0][1] = -sinlon, d.m[0][2] = -coslat * coslon;
    d.m[1][0] = -sinlat * sinlon, d.m[1][1] = coslon, d.m[1][2] = -coslat * sinlon;
    d.m[2][0] = coslat, d.m[2][1] = 0, d.m[2][2] = -sinlat;
    return d;
}
V3 earth_blh2ecef(V3 blh) { // earth.h:117-130
    double coslat = std::cos(blh.x), sinlat = std::sin(blh.x), coslon = std::cos(blh.y), sinlon = std::sin(blh.y);
    double rn = earth_RN(blh.x), rnh = rn + blh.z;
    return v3(rnh * coslat * coslon, rnh * coslat * sinlon, (rnh - rn * WGS84_E1) * sinlat);
}
V3 earth_ecef2blh(V3 ecef) { // earth.h:132-150
    double p = std::sqrt(ecef.x * ecef.x + ecef.y * ecef.y);
    double rn, lat, lon, h = 0, h2;
    lat = std::atan(ecef.z / (p * (1.0 - WGS84_E1)));
    lon = 2.0 * std::atan2(ecef.y, ecef.x + p);
    do {
        h2  = h;
        rn  = earth_RN(lat);
        h   = p / std::cos(lat) - rn;
        lat = std::atan(ecef.z / (p * (1.0 - WGS84_E1 * rn / (rn + h))));
    } while (std::fabs(h - h2) > 1.0e-4);
    return v3(lat, lon, h);
}
V3 matrix2euler(const M3 &dcm) { // rotation.h:44-66
    V3 e;
    e.y = std::atan(-dcm.m[2][0] / std::sqrt(dcm.m[2][1] * dcm.m[2][1] + dcm.m[2][2] * dcm.m[2][2]));
    if (dcm.m[2][0] <= -0.999) {
        e.x = std::atan2(dcm.m[2][1], dcm.m[2][2]);
        e.z = std::atan2((dcm.m[1][2] - dcm.m[0][1]), (dcm.m[0][2] + dcm.m[1][1]));
    } else if (dcm.m[2][0] >= 0.999) {
        e.x = std::atan2(dcm.m[2][1], dcm.m[2][2]);
        e.z = M_PI + std::atan2((dcm.m[1][2] + dcm.m[0][1]), (dcm.m[0][2] - dcm.m[1][1]));
    } else {
        e.x = std::atan2(dcm.m[2][1], dcm.m[2][2]);
        e.z = std::atan2(dcm.m[1][0], dcm.m[0][0]);
    }
    if (e.z < 0) e.z = M_PI * 2 + e.z;
    return e;
}
} // namespace

extern "C" {
// nav[11], errrow[14] (n_err values used), traj[8]; returns n_err (8 without scale factors, 14 with)
int orc_nav_result_rows(const double *origin3, int iswithscale, const double *state23, double sodo, double *nav, double *errrow, double *traj) {
    State st   = load_state(state23);
    V3 origin  = v3(origin3[0], origin3[1], origin3[2]);
    V3 ecef0   = earth_blh2ecef(origin); // Earth::local2global(origin, Pose{R, p})
    M3 cn0e    = earth_cne(origin);
    V3 ecef1   = ecef0 + m3_vec(cn0e, st.p);
    V3 blh1    = earth_ecef2blh(ecef1);
    M3 cn1e    = earth_cne(blh1);
    M3 Rg      = m3_mul(m3_mul(m3_T(cn1e), cn0e), qmat(st.q));
    V3 pos     = blh1;
    pos.x *= R2D, pos.y *= R2D; // pos.segment(0, 2) *= R2D
    V3 att = matrix2euler(Rg) * R2D;
    V3 bg  = st.bg * R2D * 3600;
    V3 ba  = st.ba * 1e5;
    const double n[11] = {0, st.time, pos.x, pos.y, pos.z, st.v.x, st.v.y, st.v.z, att.x, att.y, att.z};
    memcpy(nav, n, sizeof n);
    int k = 0;
    errrow[k++] = st.time;
    errrow[k++] = bg.x, errrow[k++] = bg.y, errrow[k++] = bg.z;
    errrow[k++] = ba.x, errrow[k++] = ba.y, errrow[k++] = ba.z;
    if (iswithscale) {
        V3 sg = st.sg * 1e6, sa = st.sa * 1e6;
        errrow[k++] = sg.x, errrow[k++] = sg.y, errrow[k++] = sg.z;
        errrow[k++] = sa.x, errrow[k++] = sa.y, errrow[k++] = sa.z;
    }
    errrow[k++] = sodo;
    const double t[8] = {st.time, st.p.x, st.p.y, st.p.z, st.q.x, st.q.y, st.q.z, st.q.w};
    memcpy(traj, t, sizeof t);
    return k;
}
}
