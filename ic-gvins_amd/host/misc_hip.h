// Host-side mirror of the reference's INS helper class (misc.h:37-77), SURVEY.md §8 row f4.
//
// The window bookkeeping (bracket search, near-node tests, IMU sample splitting) is compare/assign work on a host-resident
// deque and is done here exactly as the reference does it; the floating-point propagation — MISC::insMechanization over a
// series and the pose interpolation behind getCameraPoseFromInsWindow — runs on the device, batched over all streams with one
// launch (icg_ins_mechanize_batch / icg_ins_camera_pose_batch).  No CPU fallback: the *Batch entry points fail when the ABI
// call fails.
#pragma once
#include <cstdio>
#include <deque>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "factors.h"
#include "types.h"

namespace icg {

struct IntegrationConfiguration { // preintegration/integration_state.h:91-99
    bool isuseodo{false}, iswithscale{false}, iswithearth{false};
    Vector3d origin, gravity, iewn;
};

typedef std::deque<std::pair<IMU, IntegrationState>> InsWindow;

// fileio/filesaver.h:35-66: text mode = one "%-15.9lf " per value and a newline per dump(); binary mode = the doubles as they are
class FileSaver {
public:
    typedef std::shared_ptr<FileSaver> Ptr;
    enum { TEXT = 0, BINARY = 1 }; // fileio/filebase.h:35-38
    FileSaver(const std::string &filename, int columns, int filetype = TEXT);
    ~FileSaver();
    static Ptr create(const std::string &filename, int columns, int filetype = TEXT) { return std::make_shared<FileSaver>(filename, columns, filetype); }
    bool isOpen() const { return fp_ != nullptr; }
    void dump(const std::vector<double> &data);
    void flush();

private:
    FILE *fp_{nullptr};
    int columns_, filetype_;
};

class MISC {
public:
    static constexpr double MINIMUM_TIME_INTERVAL = 0.0001; // misc.h:72

    // ---- host, exact (misc.cc:30-65, 263-361) ----
    static size_t getInsWindowIndex(const InsWindow &window, double time);
    static int isNeedInterpolation(const IMU &imu0, const IMU &imu1, double mid);
    static void imuInterpolation(const IMU &imu01, IMU &imu00, IMU &imu11, double mid);
    static bool getImuSeriesFromTo(const InsWindow &ins_windows, double start, double end, std::vector<IMU> &series);

    // misc.cc:417-499: every 10th call (process-wide counter, as in the reference) appends one line to the navigation file
    // (0, time, lat/lon [deg], h, v, roll/pitch/heading [deg]), the IMU error file (time, bg [deg/h], ba [mGal], (sg, sa [ppm],) sodo)
    // and the trajectory file (time, p, q xyzw).  Host only: three text lines per 10 IMU epochs.
    // `counter`: the every-10th-call counter; null = the process-wide one of the reference (a function-local static there), a pointer =
    // the caller's own (one per estimator when several run in one process)
    static void writeNavResult(const IntegrationConfiguration &config, const IntegrationState &state, const FileSaver::Ptr &navfile,
                               const FileSaver::Ptr &errfile, const FileSaver::Ptr &trajfile, int *counter = nullptr);

    // ---- device, batched over streams ----
    // insMechanization (misc.cc:151-206) over series[s] (series[s][0] = imu_pre of the first step) starting from *states[s],
    // updated in place; trajectories (optional) receives the state after every sample of every series
    static bool insMechanizationBatch(icg_ctx *ctx, const IntegrationConfiguration &config, const std::vector<const std::vector<IMU> *> &series,
                                      const std::vector<IntegrationState *> &states, std::vector<std::vector<IntegrationState>> *trajectories,
                                      std::string *err = nullptr);
    // getCameraPoseFromInsWindow (misc.cc:67-83) for one (window, time) pair per stream; found[s] is its return value
    static bool getCameraPoseFromInsWindowBatch(icg_ctx *ctx, const std::vector<const InsWindow *> &windows, const Pose &pose_b_c,
                                                const std::vector<double> &times, std::vector<Pose> &poses, std::vector<uint8_t> &found,
                                                std::string *err = nullptr);
    // redoInsMechanization (misc.cc:208-261) for every stream: re-propagates each window from its updated state with ONE
    // mechanization launch, then drops the expired front of each window
    static bool redoInsMechanizationBatch(icg_ctx *ctx, const IntegrationConfiguration &config, const std::vector<IntegrationState> &updated_states,
                                          size_t reserved_ins_num, const std::vector<InsWindow *> &ins_windows, std::string *err = nullptr);
};

} // namespace icg
