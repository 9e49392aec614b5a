// MISC (misc.cc) on the MI355X host layer: window bookkeeping on the host, propagation on the device.  See misc_hip.h.
#include "misc_hip.h"

#include <cmath>

#include "../../include/icgvins_hip.h"
#include "earth.h"

namespace icg {

namespace {
void imuToRow(const IMU &m, double *r) {
    r[0] = m.time, r[1] = m.dt;
    for (int k = 0; k < 3; k++) r[2 + k] = m.dtheta[k], r[5 + k] = m.dvel[k];
}
void stateToRow(const IntegrationState &s, double *r) { // time, p3, q4 xyzw, v3, bg3, ba3, sg3, sa3
    r[0] = s.time;
    for (int k = 0; k < 3; k++) {
        r[1 + k]  = s.p[k];
        r[8 + k]  = s.v[k];
        r[11 + k] = s.bg[k];
        r[14 + k] = s.ba[k];
        r[17 + k] = s.sg[k];
        r[20 + k] = s.sa[k];
    }
    r[4] = s.q.x, r[5] = s.q.y, r[6] = s.q.z, r[7] = s.q.w;
}
void rowToState(const double *r, IntegrationState &s) {
    s.time = r[0];
    for (int k = 0; k < 3; k++) {
        s.p[k]  = r[1 + k];
        s.v[k]  = r[8 + k];
        s.bg[k] = r[11 + k];
        s.ba[k] = r[14 + k];
        s.sg[k] = r[17 + k];
        s.sa[k] = r[20 + k];
    }
    s.q = Quaterniond{r[4], r[5], r[6], r[7]};
}
void configToRow(const IntegrationConfiguration &c, double *r) {
    for (int k = 0; k < 3; k++) r[k] = c.gravity[k], r[3 + k] = c.iewn[k];
    r[6] = c.iswithearth ? 1.0 : 0.0;
    r[7] = c.iswithscale ? 1.0 : 0.0;
}
bool fail(icg_ctx *ctx, std::string *err) {
    if (err) *err = icg_last_error(ctx);
    return false;
}
} // namespace

FileSaver::FileSaver(const std::string &filename, int columns, int filetype) : columns_(columns), filetype_(filetype) {
    fp_ = fopen(filename.c_str(), filetype == TEXT ? "w" : "wb");
}
FileSaver::~FileSaver() {
    if (fp_) fclose(fp_);
}
void FileSaver::dump(const std::vector<double> &data) { // filesaver.cc:51-66
    if (!fp_) return;
    if (filetype_ == BINARY) {
        fwrite(data.data(), sizeof(double), data.size(), fp_);
        return;
    }
    for (double v : data) fprintf(fp_, "%-15.9lf ", v);
    fprintf(fp_, "\n");
}
void FileSaver::flush() {
    if (fp_) fflush(fp_);
}

void MISC::writeNavResult(const IntegrationConfiguration &config, const IntegrationState &state, const FileSaver::Ptr &navfile,
                          const FileSaver::Ptr &errfile, const FileSaver::Ptr &trajfile, int *counter) { // misc.cc:417-499
    static int process_counts = 0;
    int &counts               = counter ? *counter : process_counts;
    if (counts++ % 10) return;
    std::vector<double> result;
    const double time = state.time;
    // Earth::local2global(config.origin, Pose{R, p})
    Vector3d ecef0 = Earth::blh2ecef(config.origin);
    Matrix3d cn0e  = Earth::cne(config.origin);
    Vector3d ecef1 = ecef0 + cn0e * state.p;
    Vector3d pos   = Earth::ecef2blh(ecef1);
    Matrix3d cn1e  = Earth::cne(pos);
    Matrix3d Rg    = (cn1e.transpose() * cn0e) * Rotation::quaternion2matrix(state.q);
    pos[0] *= R2D, pos[1] *= R2D;
    Vector3d att = Rotation::matrix2euler(Rg) * R2D;
    Vector3d bg  = (state.bg * R2D) * 3600;
    Vector3d ba  = state.ba * 1e5;
    result = {0, time, pos[0], pos[1], pos[2], state.v[0], state.v[1], state.v[2], att[0], att[1], att[2]};
    navfile->dump(result);
    navfile->flush();
    result = {time, bg[0], bg[1], bg[2], ba[0], ba[1], ba[2]};
    if (config.iswithscale) {
        Vector3d sg = state.sg * 1e6, sa = state.sa * 1e6;
        result.insert(result.end(), {sg[0], sg[1], sg[2], sa[0], sa[1], sa[2]});
    }
    result.push_back(state.sodo);
    errfile->dump(result);
    errfile->flush();
    result = {time, state.p[0], state.p[1], state.p[2], state.q.x, state.q.y, state.q.z, state.q.w};
    trajfile->dump(result);
}

size_t MISC::getInsWindowIndex(const InsWindow &window, double time) { // misc.cc:30-65
    if (window.empty() || (window.front().first.time > time) || (window.back().first.time <= time)) return 0;
    size_t index = 0, sta = 0, end = window.size();
    int counts = 0;
    while (true) {
        size_t mid    = (sta + end) / 2;
        double first  = window[mid - 1].first.time;
        double second = window[mid].first.time;
        if ((first <= time) && (time < second)) {
            index = mid;
            break;
        } else if (first > time) {
            end = mid;
        } else if (second <= time) {
            sta = mid;
        }
        if (counts++ > 15) break; // 2^16 entries: the reference logs an error and returns 0
    }
    return index;
}

int MISC::isNeedInterpolation(const IMU &imu0, const IMU &imu1, double mid) { // misc.cc:263-286
    if (imu0.time < mid && imu1.time > mid) {
        double dt = mid - imu0.time;
        if (dt < MINIMUM_TIME_INTERVAL) return -1;
        dt = imu1.time - mid;
        if (dt < MINIMUM_TIME_INTERVAL) return 1;
        return 2;
    }
    return 0;
}

void MISC::imuInterpolation(const IMU &imu01, IMU &imu00, IMU &imu11, double mid) { // misc.cc:288-305 (imu11 may alias imu01)
    double scale = (imu01.time - mid) / imu01.dt;
    IMU buff     = imu01;
    imu00.time   = mid;
    imu00.dt     = buff.dt - (buff.time - mid);
    imu00.dtheta = buff.dtheta * (1 - scale);
    imu00.dvel   = buff.dvel * (1 - scale);
    imu00.odovel = buff.odovel * (1 - scale);
    imu11.time   = buff.time;
    imu11.dt     = buff.time - mid;
    imu11.dtheta = buff.dtheta * scale;
    imu11.dvel   = buff.dvel * scale;
    imu11.odovel = buff.odovel * scale;
}

bool MISC::getImuSeriesFromTo(const InsWindow &ins_windows, double start, double end, std::vector<IMU> &series) { // misc.cc:307-361
    size_t is = getInsWindowIndex(ins_windows, start);
    size_t ie = getInsWindowIndex(ins_windows, end);
    // the reference only rejects (0, 0) and then indexes window[is-1] / window[ie-1]: one missing end is undefined behaviour
    // there, a failure here
    if (is == 0 || ie == 0) return false;
    IMU imu0, imu1, imu;
    series.clear();
    imu0 = ins_windows[is - 1].first;
    imu1 = ins_windows[is].first;
    int isneed = isNeedInterpolation(imu0, imu1, start);
    if (isneed == -1) {
        series.push_back(imu0);
        series.push_back(imu1);
    } else if (isneed == 1) {
        series.push_back(imu1);
    } else if (isneed == 2) {
        imuInterpolation(imu1, imu, imu1, start);
        series.push_back(imu);
        series.push_back(imu1);
    }
    for (size_t k = is + 1; k + 1 < ie; k++) series.push_back(ins_windows[k].first);
    imu0   = ins_windows[ie - 1].first;
    imu1   = ins_windows[ie].first;
    isneed = isNeedInterpolation(imu0, imu1, end);
    if (isneed == -1) {
        series.push_back(imu0);
    } else if (isneed == 1) {
        series.push_back(imu0);
        series.push_back(imu1);
    } else if (isneed == 2) {
        series.push_back(imu0);
        imuInterpolation(imu1, imu, imu1, end);
        series.push_back(imu);
    }
    if (series.empty()) return false; // series.back() on an empty vector in the reference
    series.back().time = end;
    return true;
}

bool MISC::insMechanizationBatch(icg_ctx *ctx, const IntegrationConfiguration &config, const std::vector<const std::vector<IMU> *> &series,
                                 const std::vector<IntegrationState *> &states, std::vector<std::vector<IntegrationState>> *trajectories,
                                 std::string *err) {
    const size_t n = series.size();
    if (states.size() != n) {
        if (err) *err = "insMechanizationBatch: series/states size mismatch";
        return false;
    }
    std::vector<int32_t> offsets{0};
    std::vector<double> imu, st(23 * n);
    for (size_t s = 0; s < n; s++) {
        for (const IMU &m : *series[s]) {
            double row[8];
            imuToRow(m, row);
            imu.insert(imu.end(), row, row + 8);
        }
        offsets.push_back((int32_t) (imu.size() / 8));
        stateToRow(*states[s], &st[23 * s]);
    }
    double cfg[8];
    configToRow(config, cfg);
    std::vector<double> traj;
    if (trajectories) traj.resize(23 * (imu.size() / 8));
    if (imu.empty()) imu.resize(8);
    if (icg_ins_mechanize_batch(ctx, (int) n, offsets.data(), imu.data(), cfg, st.data(), trajectories ? traj.data() : nullptr) != ICG_OK)
        return fail(ctx, err);
    for (size_t s = 0; s < n; s++) rowToState(&st[23 * s], *states[s]);
    if (trajectories) {
        trajectories->assign(n, {});
        for (size_t s = 0; s < n; s++)
            for (int row = offsets[s] + 1; row < offsets[s + 1]; row++) {
                IntegrationState t = *states[s]; // carries the fields the mechanization does not touch
                rowToState(&traj[23 * (size_t) row], t);
                (*trajectories)[s].push_back(t);
            }
    }
    return true;
}

bool MISC::getCameraPoseFromInsWindowBatch(icg_ctx *ctx, const std::vector<const InsWindow *> &windows, const Pose &pose_b_c,
                                           const std::vector<double> &times, std::vector<Pose> &poses, std::vector<uint8_t> &found,
                                           std::string *err) {
    const size_t n = windows.size();
    if (times.size() != n) {
        if (err) *err = "getCameraPoseFromInsWindowBatch: windows/times size mismatch";
        return false;
    }
    std::vector<double> brackets(16 * n, 0.0), out(12 * n);
    std::vector<int32_t> interp(n, 0);
    found.assign(n, 0);
    auto put = [](const IntegrationState &s, double *r) {
        r[0] = s.time;
        for (int k = 0; k < 3; k++) r[1 + k] = s.p[k];
        r[4] = s.q.x, r[5] = s.q.y, r[6] = s.q.z, r[7] = s.q.w;
    };
    for (size_t s = 0; s < n; s++) {
        const InsWindow &w = *windows[s];
        if (w.empty()) {
            if (err) *err = "getCameraPoseFromInsWindowBatch: empty INS window";
            return false;
        }
        size_t index = getInsWindowIndex(w, times[s]); // misc.cc:70
        if (index > 0) {
            put(w[index - 1].second, &brackets[16 * s]);
            put(w[index].second, &brackets[16 * s + 8]);
            interp[s] = 1, found[s] = 1;
        } else {
            put(w.back().second, &brackets[16 * s]); // :79-82
        }
    }
    double pbc[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) pbc[3 * i + j] = pose_b_c.R(i, j);
        pbc[9 + i] = pose_b_c.t[i];
    }
    if (icg_ins_camera_pose_batch(ctx, (int) n, brackets.data(), interp.data(), pbc, times.data(), out.data()) != ICG_OK) return fail(ctx, err);
    poses.resize(n);
    for (size_t s = 0; s < n; s++)
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) poses[s].R(i, j) = out[12 * s + 3 * i + j];
            poses[s].t[i] = out[12 * s + 9 + i];
        }
    return true;
}

bool MISC::redoInsMechanizationBatch(icg_ctx *ctx, const IntegrationConfiguration &config, const std::vector<IntegrationState> &updated_states,
                                     size_t reserved_ins_num, const std::vector<InsWindow *> &ins_windows, std::string *err) {
    const size_t n = ins_windows.size();
    if (updated_states.size() != n) {
        if (err) *err = "redoInsMechanizationBatch: size mismatch";
        return false;
    }
    // per stream: the IMU series to re-propagate (first entry = imu_pre) and where its states go back into the window
    std::vector<std::vector<IMU>> series(n);
    std::vector<IntegrationState> start(n);
    std::vector<size_t> first_row(n, 0), index_of(n, 0);
    std::vector<int> mode(n, 0);
    for (size_t s = 0; s < n; s++) {
        InsWindow &w  = *ins_windows[s];
        start[s]      = updated_states[s];
        size_t index  = getInsWindowIndex(w, start[s].time);
        index_of[s]   = index;
        if (index == 0) continue; // :214-217: nothing is touched
        IMU imu0 = w[index - 1].first, imu1 = w[index].first;
        int isneed = isNeedInterpolation(imu0, imu1, start[s].time);
        mode[s]    = isneed;
        if (isneed == -1) { // the updated state sits on the previous node
            series[s].push_back(imu0);
            series[s].push_back(imu1);
            first_row[s] = index;
        } else if (isneed == 1) { // ... on the current node: the state is taken over as is (:233-236)
            start[s].time   = imu1.time;
            w[index].second = start[s];
            series[s].push_back(imu1);
            first_row[s] = index + 1;
        } else if (isneed == 2) {
            imuInterpolation(imu1, imu0, imu1, start[s].time);
            series[s].push_back(imu0);
            series[s].push_back(imu1);
            first_row[s] = index;
        } else { // time exactly on a node boundary pattern the reference does not handle (isneed == 0): it only propagates k > index
            series[s].push_back(imu1);
            first_row[s] = index + 1;
        }
        for (size_t k = index + 1; k < w.size(); k++) series[s].push_back(w[k].first);
    }
    std::vector<const std::vector<IMU> *> sp(n);
    std::vector<IntegrationState *> stp(n);
    for (size_t s = 0; s < n; s++) sp[s] = &series[s], stp[s] = &start[s];
    std::vector<std::vector<IntegrationState>> traj;
    if (!insMechanizationBatch(ctx, config, sp, stp, &traj, err)) return false;
    for (size_t s = 0; s < n; s++) {
        if (index_of[s] == 0) continue;
        InsWindow &w = *ins_windows[s];
        for (size_t k = 0; k < traj[s].size(); k++) w[first_row[s] + k].second = traj[s][k];
        if (index_of[s] < reserved_ins_num) continue; // :254-260
        size_t counts = index_of[s] - reserved_ins_num;
        for (size_t k = 0; k < counts; k++) w.pop_front();
    }
    return true;
}

} // namespace icg
