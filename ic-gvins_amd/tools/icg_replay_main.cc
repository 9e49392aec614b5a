// icg_replay — command-line front of the replay harness (ic-gvins_amd/host/replay.h): what `roslaunch ic_gvins ic_gvins.launch
// configfile:=...` + `rosbag play` do for the reference (README.md:100-109, ROS/fusion_ros.cc), with files in place of a ROS bag.
//   icg_replay --config gvins.yaml --imu imu.txt [--gnss gnss.txt] [--images cam0/images.txt] [--output DIR] [--imu-rate] [--start T] [--end T]
//              [--streams N [--lockstep-groups G]]
// --streams N replays the input N times side by side (results in DIR/stream<k>), one estimator per host thread, or — with --lockstep-groups G —
// as G lock-step groups whose window solves share one batched solver each: the throughput forms for the camera streams of one GPU.
// Results (gvins.nav, trajectory.csv, tracking.txt, statistics.txt, extrinsic.txt, mappoint.txt, IMU_ERR.bin, a copy of the configuration)
// go to --output or to the configuration's `outputpath`.  Needs an MI355X: the library behind it has no CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "replay.h"

static void usage() {
    fprintf(stderr, "usage: icg_replay --config gvins.yaml --imu imu.txt [--gnss gnss.txt] [--images images.txt] [--output DIR] [--imu-rate]\n"
                    "                  [--start GPS_SECOND] [--end GPS_SECOND] [--streams N [--lockstep-groups G]]\n"
                    "  imu.txt     t dtheta_x dtheta_y dtheta_z dvel_x dvel_y dvel_z   (increments; with --imu-rate: angular rate / specific force)\n"
                    "  gnss.txt    t lat[deg] lon[deg] h[m] std_n std_e std_d\n"
                    "  images.txt  t filename   (binary PGM / PPM next to the list)\n");
}

int main(int argc, char **argv) {
    icg::ReplayOptions o;
    int streams = 1, groups = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto value    = [&](std::string &dst) {
            if (i + 1 >= argc) {
                usage();
                exit(2);
            }
            dst = argv[++i];
        };
        std::string num;
        if (a == "--config") value(o.configfile);
        else if (a == "--imu") value(o.imufile);
        else if (a == "--gnss") value(o.gnssfile);
        else if (a == "--images") value(o.imagelist);
        else if (a == "--output") value(o.outputpath);
        else if (a == "--imu-rate") o.imu_is_rate = true;
        else if (a == "--start") value(num), o.start_time = atof(num.c_str());
        else if (a == "--end") value(num), o.end_time = atof(num.c_str());
        else if (a == "--streams") value(num), streams = atoi(num.c_str());
        else if (a == "--lockstep-groups") value(num), groups = atoi(num.c_str());
        else {
            usage();
            return 2;
        }
    }
    if (o.configfile.empty() || o.imufile.empty()) {
        usage();
        return 2;
    }
    std::string err;
    if (streams > 1) {
        if (o.outputpath.empty()) {
            fprintf(stderr, "icg_replay: --streams needs --output\n");
            return 2;
        }
        std::vector<icg::ReplayOptions> many((size_t) streams, o);
        for (int k = 0; k < streams; k++) {
            many[(size_t) k].outputpath   = o.outputpath + "/stream" + std::to_string(k);
            many[(size_t) k].wait_poll_us = 50;
        }
        std::vector<icg::ReplaySummary> S;
        double wall = 0;
        long shared[3] = {0, 0, 0};
        mkdir(o.outputpath.c_str(), 0755);
        bool ok = groups > 0 ? icg::Replay::runLockstepGroups(many, groups, S, &wall, shared, &err) : icg::Replay::runMany(many, S, &wall, &err);
        if (!ok) {
            fprintf(stderr, "icg_replay: %s\n", err.c_str());
            return 1;
        }
        double data = 0;
        long frames = 0, solves = 0;
        for (const auto &x : S) data += x.data_seconds, frames += x.counters.frames_tracked, solves += x.counters.optimizations;
        printf("replayed %d streams (%.2f s of data in total) in %.2f s: x%.1f real time summed, %.0f frames/s, %.0f window solves/s%s\n", streams, data, wall,
               wall > 0 ? data / wall : 0.0, wall > 0 ? frames / wall : 0.0, wall > 0 ? solves / wall : 0.0,
               groups > 0 ? (" (" + std::to_string(shared[1]) + " batched solve rounds, largest batch " + std::to_string(shared[2]) + ")").c_str() : "");
        return 0;
    }
    icg::ReplaySummary s;
    icg::WindowSolver::setHostFactorOverlap(true); // one stream: the host factors of a linearization run beside the device calls
    if (!icg::Replay::run(o, s, &err)) {
        fprintf(stderr, "icg_replay: %s\n", err.c_str());
        return 1;
    }
    printf("replayed %.2f s of data in %.2f s (x%.1f real time): %ld IMU, %ld GNSS (%ld dropped), %ld images; %ld frames tracked, %ld keyframes, %ld window "
           "solves, %ld marginalizations, %ld tracking losses; final state %d; results in %s\n",
           s.data_seconds, s.wall_seconds, s.wall_seconds > 0 ? s.data_seconds / s.wall_seconds : 0.0, s.imu, s.gnss, s.gnss_dropped, s.frames,
           s.counters.frames_tracked, s.counters.keyframes, s.counters.optimizations, s.counters.marginalizations, s.counters.lost, s.final_state,
           s.outputpath.c_str());
    return 0;
}
