// Replay harness — SURVEY.md §8 row f2: what the reference's ROS shell does (ROS/fusion_ros.cc:56-234: load the configuration, create
// the output directory, build GVINS, turn sensor messages into IMU / GNSS / Frame objects and feed them in time order), with files in
// place of topics:
//   IMU    text, one epoch per line.  "increment" (i2Nav text format): t dtheta_x dtheta_y dtheta_z dvel_x dvel_y dvel_z  [rad, m/s];
//          "rate" (the fields of sensor_msgs/Imu): t wx wy wz ax ay az [rad/s, m/s^2], multiplied by dt as imuCallback does (:136-141).
//          Front-right-down body axes (README.md:123).
//   GNSS   text: t lat[deg] lon[deg] h[m] std_n std_e std_d [m]  (the fields gnssCallback takes from sensor_msgs/NavSatFix, :170-178);
//          fixes with a zero std or any std >= gnssthreshold are dropped, fixes after gnssoutagetime too when isusegnssoutage (:181-198).
//   images a list file "t filename" per line; binary PGM (P5, MONO8) or PPM (P6, RGB, handed over as BGR8 after swapping) next to the list.
// Stamps above 1e9 are Unix seconds and converted with GpsTime::unix2gps (:128-131); smaller values are taken as GPS seconds of week.
// Events are delivered in stamp order (IMU before GNSS before image on ties), as `rosbag play` delivers a recorded bag.
#pragma once
#include <string>
#include <vector>

#include "gvins_hip.h"

namespace icg {

struct ReplayOptions {
    std::string configfile, outputpath;     // outputpath empty: the configuration's `outputpath`
    std::string imufile, gnssfile, imagelist;
    bool imu_is_rate{false};
    double start_time{0}, end_time{0};      // 0 = unbounded (after conversion to GPS seconds of week)
    int wait_poll_us{0};                    // > 0: the estimator's contexts poll + sleep instead of spinning (many replays per host)
};

struct ReplaySummary {
    long imu{0}, gnss{0}, gnss_dropped{0}, frames{0};
    GVINS::Counters counters;
    int final_state{GVINS::GVINS_ERROR};
    double wall_seconds{0}, data_seconds{0};
    std::string outputpath;
};

class Replay {
public:
    struct ImageEntry {
        double time;
        std::string path;
    };
    static bool readImuText(const std::string &path, bool is_rate, std::vector<IMU> &out, std::string *err = nullptr);
    static bool readGnssText(const std::string &path, std::vector<GNSS> &out, std::string *err = nullptr);
    static bool readImageList(const std::string &path, std::vector<ImageEntry> &out, std::string *err = nullptr);
    static bool loadPnm(const std::string &path, Mat &image, std::string *err = nullptr);
    static double toGpsSecondOfWeek(double stamp);
    // the whole run; false + err on I/O or estimator failure
    static bool run(const ReplayOptions &options, ReplaySummary &summary, std::string *err = nullptr);
    // independent replays side by side, one host thread and one estimator (own device contexts, own id space) each: the camera streams of
    // one GPU.  Per-stream results do not depend on what runs next to them.  wall_seconds (optional) = the whole batch.
    static bool runMany(const std::vector<ReplayOptions> &options, std::vector<ReplaySummary> &summaries, double *wall_seconds = nullptr,
                        std::string *err = nullptr);
    // the same replays in lock-step on ONE host thread: every stream advances by one IMU epoch per tick, and the window solves that become due
    // in a tick are solved TOGETHER through one WindowSolverBatch (one evaluation / assembly / elimination / back-substitution launch per LM
    // step for all of them, each window with its own trust region and stopping).  Per-stream results equal the stream replayed alone
    // (bit for bit on the CPU backend).  shared_solves (optional) = [window solves, batched launches' worth of solves, largest batch].
    static bool runLockstep(const std::vector<ReplayOptions> &options, std::vector<ReplaySummary> &summaries, double *wall_seconds = nullptr,
                            long *shared_solves = nullptr, std::string *err = nullptr, int solver_host_threads = 0);
    // `groups` lock-step groups side by side (one host thread + one WindowSolverBatch each, the streams dealt out in contiguous blocks):
    // the host work per stream that is not shared (tracking stages, INS, culling, marginalization) spreads over the groups' threads
    // batches / windows that went through a MarginalizationBatch in lock-step runs since the last call (ICG_LOCKSTEP_MARG_BATCH=1)
    static void takeLockstepMarginalizationCounts(long out[2]);
    static bool runLockstepGroups(const std::vector<ReplayOptions> &options, int groups, std::vector<ReplaySummary> &summaries, double *wall_seconds = nullptr,
                                  long *shared_solves = nullptr, std::string *err = nullptr);
};

} // namespace icg
