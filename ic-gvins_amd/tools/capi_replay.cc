// C entry points of the replay harness and of the small navigation factors (ctypes plumbing for tests / tools; see replay.h, gvins_hip.h,
// nav_factors.h, earth.h).
#include <cstring>
#include <stdexcept>

#include "replay.h"

using namespace icg;

static void set_err(char *err, int errlen, const char *msg) {
    if (err && errlen > 0) {
        strncpy(err, msg, (size_t) errlen - 1);
        err[errlen - 1] = 0;
    }
}

extern "C" {

// summary16: imu, gnss, gnss dropped, frames, frames tracked, keyframes, optimizations, marginalizations, INS launches, lost,
// reprojection factors, chi2-removed factors, final state, wall seconds, data seconds, time nodes (unused = 0)
int icgh_replay_run(const char *configfile, const char *outputpath, const char *imufile, const char *gnssfile, const char *imagelist, int imu_is_rate,
                    double start_time, double end_time, double *summary16, char *err, int errlen) {
    try {
        ReplayOptions o;
        o.configfile = configfile ? configfile : "";
        o.outputpath = outputpath ? outputpath : "";
        o.imufile    = imufile ? imufile : "";
        o.gnssfile   = gnssfile ? gnssfile : "";
        o.imagelist  = imagelist ? imagelist : "";
        o.imu_is_rate = imu_is_rate != 0;
        o.start_time = start_time, o.end_time = end_time;
        ReplaySummary s;
        std::string e;
        // one camera stream: the window solve is a latency chain, so the host factors of a linearization run beside the device calls
        struct Overlap {
            Overlap() { WindowSolver::setHostFactorOverlap(true); }
            ~Overlap() { WindowSolver::setHostFactorOverlap(false); }
        } overlap;
        if (!Replay::run(o, s, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        const double v[16] = {(double) s.imu, (double) s.gnss, (double) s.gnss_dropped, (double) s.frames, (double) s.counters.frames_tracked,
                              (double) s.counters.keyframes, (double) s.counters.optimizations, (double) s.counters.marginalizations,
                              (double) s.counters.ins_launches, (double) s.counters.lost, (double) s.counters.reprojection_factors,
                              (double) s.counters.chi2_removed, (double) s.final_state, s.wall_seconds, s.data_seconds, 0.0};
        if (summary16) memcpy(summary16, v, sizeof v);
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}

// ---- the small factors and helpers, for the golden comparison with the reference's headers -------------------------------------
// kind: 0 GnssFactor (aux: blh3, std3, lever3; x: pose7)   1 ImuErrorFactor (x: mix9)   2 ImuPosePriorFactor (aux: pose7, std6; x: pose7)
//       3 ImuMixPriorFactor (aux: mix9, std9; x: mix9).  J row-major num_residuals x block size.
int icgh_nav_factor(int kind, const double *aux, const double *x, double *residuals, double *jacobian) {
    const double *params[1] = {x};
    double *J[1]            = {jacobian};
    try {
        if (kind == 0) {
            GNSS g;
            g.blh = Vector3d(aux[0], aux[1], aux[2]);
            g.std = Vector3d(aux[3], aux[4], aux[5]);
            return GnssFactor(g, Vector3d(aux[6], aux[7], aux[8])).Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
        } else if (kind == 1) {
            return ImuErrorFactor().Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
        } else if (kind == 2) {
            return ImuPosePriorFactor(aux, aux + 7).Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
        } else if (kind == 3) {
            return ImuMixPriorFactor(aux, aux + 9).Evaluate(params, residuals, jacobian ? J : nullptr) ? 0 : 1;
        }
    } catch (const std::exception &) {
    }
    return -1;
}

// what: 0 gravity(blh) -> out[0]   1 global2local(origin=a, global=b) -> out3   2 local2global(origin=a, local=b) -> out3
//       3 iewn(origin=a, local=b) -> out3   4 euler2quaternion(a) -> out4 (x y z w)   5 matrix2euler(quaternion2matrix(q = a[0..3] xyzw)) -> out3
//       6 unix2gps(a[0]) -> out[0] = week, out[1] = second of week
int icgh_nav_helper(int what, const double *a, const double *b, double *out) {
    Vector3d A(a[0], a[1], a[2]), B = b ? Vector3d(b[0], b[1], b[2]) : Vector3d();
    Vector3d r;
    switch (what) {
    case 0: out[0] = Earth::gravity(A); return 0;
    case 1: r = Earth::global2local(A, B); break;
    case 2: r = Earth::local2global(A, B); break;
    case 3: r = Earth::iewn(A, B); break;
    case 4: {
        Quaterniond q = Rotation::euler2quaternion(A);
        out[0] = q.x, out[1] = q.y, out[2] = q.z, out[3] = q.w;
        return 0;
    }
    case 5: r = Rotation::matrix2euler(Rotation::quaternion2matrix(Quaterniond{a[0], a[1], a[2], a[3]})); break;
    case 6: {
        int week;
        double sow;
        GpsTime::unix2gps(a[0], week, sow);
        out[0] = week, out[1] = sow;
        return 0;
    }
    default: return -1;
    }
    out[0] = r[0], out[1] = r[1], out[2] = r[2];
    return 0;
}

// MISC::detectZeroVelocity on n rows of (dtheta3, dvel3); average6 out; returns 1 / 0
int icgh_detect_zero_velocity(int n, const double *rows6, double imudatarate, double *average6) {
    std::vector<IMU> buf((size_t) n);
    for (int k = 0; k < n; k++) {
        buf[(size_t) k].dtheta = Vector3d(rows6[6 * k], rows6[6 * k + 1], rows6[6 * k + 2]);
        buf[(size_t) k].dvel   = Vector3d(rows6[6 * k + 3], rows6[6 * k + 4], rows6[6 * k + 5]);
    }
    std::vector<double> avg;
    bool z = GVINS::detectZeroVelocity(buf, imudatarate, avg);
    for (int k = 0; k < 6; k++) average6[k] = avg[(size_t) k];
    return z ? 1 : 0;
}

} // extern "C"

extern "C" {
// Replay::loadPnm: dims3 = rows, cols, channels; the pixels (BGR order for colour) are copied when `out` holds at least rows*cols*channels bytes
int icgh_replay_load_pnm(const char *path, int32_t *dims3, uint8_t *out, int out_len, char *err, int errlen) {
    Mat image;
    std::string e;
    if (!Replay::loadPnm(path ? path : "", image, &e)) {
        set_err(err, errlen, e.c_str());
        return -2;
    }
    dims3[0] = image.rows, dims3[1] = image.cols, dims3[2] = image.chans;
    const size_t n = (size_t) image.rows * image.cols * image.chans;
    if (out && (size_t) out_len >= n) memcpy(out, image.data, n);
    return 0;
}
}

extern "C" {
// n independent replays of the same input files side by side (one estimator + host thread each), results under outputs[k].
// summaries: n x 16 as icgh_replay_run; *batch_wall_seconds = the whole batch.
int icgh_replay_run_many(int n, const char *configfile, const char *const *outputs, const char *imufile, const char *gnssfile, const char *imagelist,
                         int imu_is_rate, int wait_poll_us, double *summaries, double *batch_wall_seconds, char *err, int errlen) {
    try {
        std::vector<ReplayOptions> opts((size_t) n);
        for (int k = 0; k < n; k++) {
            ReplayOptions &o = opts[(size_t) k];
            o.configfile = configfile ? configfile : "";
            o.outputpath = outputs[k];
            o.imufile    = imufile ? imufile : "";
            o.gnssfile   = gnssfile ? gnssfile : "";
            o.imagelist  = imagelist ? imagelist : "";
            o.imu_is_rate  = imu_is_rate != 0;
            o.wait_poll_us = wait_poll_us;
        }
        std::vector<ReplaySummary> S;
        std::string e;
        double wall = 0;
        if (!Replay::runMany(opts, S, &wall, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        for (int k = 0; k < n; k++) {
            const ReplaySummary &s = S[(size_t) k];
            const double v[16] = {(double) s.imu, (double) s.gnss, (double) s.gnss_dropped, (double) s.frames, (double) s.counters.frames_tracked,
                                  (double) s.counters.keyframes, (double) s.counters.optimizations, (double) s.counters.marginalizations,
                                  (double) s.counters.ins_launches, (double) s.counters.lost, (double) s.counters.reprojection_factors,
                                  (double) s.counters.chi2_removed, (double) s.final_state, s.wall_seconds, s.data_seconds, 0.0};
            memcpy(summaries + 16 * (size_t) k, v, sizeof v);
        }
        if (batch_wall_seconds) *batch_wall_seconds = wall;
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}
}

extern "C" {
// as icgh_replay_run_many, but in `groups` lock-step groups (one thread each) with the window solves of a tick shared through the group's WindowSolverBatch.
// shared3: window solves, batched solve rounds, largest batch.
int icgh_replay_run_lockstep(int n, const char *configfile, const char *const *outputs, const char *imufile, const char *gnssfile, const char *imagelist,
                             int imu_is_rate, int groups, double *summaries, double *batch_wall_seconds, int64_t *shared3, char *err, int errlen) {
    try {
        std::vector<ReplayOptions> opts((size_t) n);
        for (int k = 0; k < n; k++) {
            ReplayOptions &o = opts[(size_t) k];
            o.configfile = configfile ? configfile : "";
            o.outputpath = outputs[k];
            o.imufile    = imufile ? imufile : "";
            o.gnssfile   = gnssfile ? gnssfile : "";
            o.imagelist  = imagelist ? imagelist : "";
            o.imu_is_rate = imu_is_rate != 0;
        }
        std::vector<ReplaySummary> S;
        std::string e;
        double wall = 0;
        long shared[3] = {0, 0, 0};
        if (!Replay::runLockstepGroups(opts, groups, S, &wall, shared, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        for (int k = 0; k < n; k++) {
            const ReplaySummary &s = S[(size_t) k];
            const double v[16] = {(double) s.imu, (double) s.gnss, (double) s.gnss_dropped, (double) s.frames, (double) s.counters.frames_tracked,
                                  (double) s.counters.keyframes, (double) s.counters.optimizations, (double) s.counters.marginalizations,
                                  (double) s.counters.ins_launches, (double) s.counters.lost, (double) s.counters.reprojection_factors,
                                  (double) s.counters.chi2_removed, (double) s.final_state, s.wall_seconds, s.data_seconds, 0.0};
            memcpy(summaries + 16 * (size_t) k, v, sizeof v);
        }
        if (batch_wall_seconds) *batch_wall_seconds = wall;
        if (shared3) shared3[0] = shared[0], shared3[1] = shared[1], shared3[2] = shared[2];
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}
}

extern "C" {
// batches / windows that went through a MarginalizationBatch in the lock-step runs since the last call (ICG_LOCKSTEP_MARG_BATCH=1)
void icgh_replay_lockstep_marg_counts(int64_t *out2) {
    long v[2];
    icg::Replay::takeLockstepMarginalizationCounts(v);
    out2[0] = v[0], out2[1] = v[1];
}

// lock-step groups over streams with their OWN input files (configs / imus / gnss / images: n entries each; a NULL gnss / images entry = none)
int icgh_replay_run_lockstep_files(int n, const char *const *configs, const char *const *outputs, const char *const *imus, const char *const *gnss,
                                   const char *const *images, int groups, double *summaries, double *batch_wall_seconds, int64_t *shared3, char *err,
                                   int errlen) {
    try {
        std::vector<ReplayOptions> opts((size_t) n);
        for (int k = 0; k < n; k++) {
            ReplayOptions &o = opts[(size_t) k];
            o.configfile = configs[k], o.outputpath = outputs[k], o.imufile = imus[k];
            o.gnssfile   = gnss && gnss[k] ? gnss[k] : "";
            o.imagelist  = images && images[k] ? images[k] : "";
        }
        std::vector<ReplaySummary> S;
        std::string e;
        double wall    = 0;
        long shared[3] = {0, 0, 0};
        if (!Replay::runLockstepGroups(opts, groups, S, &wall, shared, &e)) {
            set_err(err, errlen, e.c_str());
            return -2;
        }
        for (int k = 0; k < n; k++) {
            const ReplaySummary &s = S[(size_t) k];
            const double v[16] = {(double) s.imu, (double) s.gnss, (double) s.gnss_dropped, (double) s.frames, (double) s.counters.frames_tracked,
                                  (double) s.counters.keyframes, (double) s.counters.optimizations, (double) s.counters.marginalizations,
                                  (double) s.counters.ins_launches, (double) s.counters.lost, (double) s.counters.reprojection_factors,
                                  (double) s.counters.chi2_removed, (double) s.final_state, s.wall_seconds, s.data_seconds, 0.0};
            memcpy(summaries + 16 * (size_t) k, v, sizeof v);
        }
        if (batch_wall_seconds) *batch_wall_seconds = wall;
        if (shared3) shared3[0] = shared[0], shared3[1] = shared[1], shared3[2] = shared[2];
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return -1;
    }
}
}
