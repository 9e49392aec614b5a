// ORACLE / TEST INFRASTRUCTURE ONLY — runs the REFERENCE's own estimator: GVINS (/root/reference/ic_gvins/ic_gvins/ic_gvins.{h,cc}) with
// misc.cc, the four preintegration variants, the factors and the tracker sources, all compiled unmodified from where they lie, behind one
// C entry point that plays a recorded sequence into it the way the ROS shell does (ROS/fusion_ros.cc:123-234).  What the build replaces is
// stated per shim: Eigen / yaml-cpp / glog / absl / tbb interfaces (shim/), the OpenCV image primitives (ref_tracking.cc: forwarded to the
// CPU restatement, so they stay unpinned) and Ceres — shim/ceres/problem_shim.h restates the published Levenberg-Marquardt loop on dense
// normal equations, independently of the product's solver.  The estimator's three threads run as in the reference, so the result depends
// on thread timing; the driver paces the input slower than real time so that tracking and optimization finish between events, which is
// the schedule the product's deterministic event loop follows.  SURVEY.md §8 row f2.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "ref_tracking.cc" // the tracker sources + the OpenCV entry points on the oracle primitives (and its own C entry points)

#include "preintegration/preintegration_base.cc"
#include "preintegration/preintegration_earth.cc"
#include "preintegration/preintegration_earth_odo.cc"
#include "preintegration/preintegration_normal.cc"
#include "preintegration/preintegration_odo.cc"

#include "misc.cc"

#include "ic_gvins.cc"

namespace {
class ReplayDrawer : public Drawer { // tracking/drawer.h: the RViz drawer of the ROS shell is not part of the estimator
public:
    void run() override {}
    void setFinished() override {}
    void addNewFixedMappoint(Vector3d) override {}
    void updateMap(const Eigen::Matrix4d &) override {}
    void updateFrame(Frame::Ptr) override {}
    void updateTrackedMapPoints(vector<cv::Point2f>, vector<cv::Point2f>, vector<MapPointType>) override {}
    void updateTrackedRefPoints(vector<cv::Point2f>, vector<cv::Point2f>) override {}
};
} // namespace

extern "C" {
// imu rows: t, dtheta3, dvel3 (increments; dt from consecutive stamps as imuCallback computes it); gnss rows: t, lat [rad], lon [rad], h, std3;
// images: n_img gray frames of w x h, row-major, one after the other, with their stamps.  slowdown: wall seconds per data second.
// Returns the final GVINS state; the result files are in outputpath.
int ref_gvins_run(const char *configfile, const char *outputpath, int n_imu, const double *imu_rows, int n_gnss, const double *gnss_rows, int n_img,
                  const double *img_stamps, const uint8_t *images, int w, int h, double slowdown) {
    Drawer::Ptr drawer = std::make_shared<ReplayDrawer>();
    auto gvins         = std::make_shared<GVINS>(configfile, outputpath, drawer);
    if (!gvins->isRunning()) return -100;
    int ii = 1, gi = 0, fi = 0; // the first IMU message only initialises dt (fusion_ros.cc:146-148)
    const double t_first = imu_rows[0];
    auto wall0           = std::chrono::steady_clock::now();
    auto wait_until      = [&](double t) {
        auto due = wall0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>((t - t_first) * slowdown));
        std::this_thread::sleep_until(due);
    };
    while (ii < n_imu || gi < n_gnss || fi < n_img) {
        const double ti = ii < n_imu ? imu_rows[7 * ii] : 1e300, tg = gi < n_gnss ? gnss_rows[7 * gi] : 1e300, tf = fi < n_img ? img_stamps[fi] : 1e300;
        if (ti <= tg && ti <= tf) {
            wait_until(ti);
            const double *r = imu_rows + 7 * (size_t) ii;
            IMU imu;
            imu.time   = r[0];
            imu.dt     = r[0] - imu_rows[7 * (size_t) (ii - 1)];
            imu.dtheta = Vector3d(r[1], r[2], r[3]);
            imu.dvel   = Vector3d(r[4], r[5], r[6]);
            imu.odovel = 0;
            while (!gvins->addNewImu(imu)) usleep(50); // try_lock failed: the shell retries with the next message (fusion_ros.cc:151-161)
            ii++;
        } else if (tg <= tf) {
            wait_until(tg);
            const double *r = gnss_rows + 7 * (size_t) gi;
            GNSS g;
            g.time       = r[0];
            g.blh        = Vector3d(r[1], r[2], r[3]);
            g.std        = Vector3d(r[4], r[5], r[6]);
            g.isyawvalid = false;
            g.yaw        = 0;
            gvins->addNewGnss(g);
            gi++;
        } else {
            wait_until(tf);
            Mat image(h, w, CV_8UC1);
            memcpy(image.data, images + (size_t) fi * w * h, (size_t) w * h);
            auto frame = Frame::createFrame(tf, image);
            while (!gvins->addNewFrame(frame)) usleep(50);
            fi++;
        }
    }
    usleep((useconds_t) (300000 * std::max(1.0, slowdown))); // let the last optimization finish
    int state = (int) gvins->gvinsState();
    gvins->setFinished();
    return state;
}
}
