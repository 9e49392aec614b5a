// Replay harness: files -> IMU / GNSS / Frame events -> icg::GVINS.  See replay.h.
#include "replay.h"

#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <array>
#include <atomic>
#include <thread>

#include "marg_batch.h"
#include "yaml_lite.h"

namespace icg {

namespace {
bool setErr(std::string *err, const std::string &what) {
    if (err) *err = what;
    return false;
}
std::string dirOf(const std::string &path) {
    size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}
bool readRows(const std::string &path, size_t min_columns, std::vector<std::vector<double>> &rows, std::string *err) {
    std::ifstream f(path);
    if (!f) return setErr(err, "cannot open " + path);
    std::string line;
    while (std::getline(f, line)) {
        for (char &c : line)
            if (c == ',') c = ' ';
        size_t a = line.find_first_not_of(" \t\r");
        if (a == std::string::npos || line[a] == '#') continue;
        std::istringstream ss(line);
        std::vector<double> row;
        double v;
        while (ss >> v) row.push_back(v);
        if (row.size() < min_columns) return setErr(err, path + ": a line has fewer than " + std::to_string(min_columns) + " columns");
        rows.push_back(std::move(row));
    }
    return true;
}
} // namespace

double Replay::toGpsSecondOfWeek(double stamp) {
    if (stamp < 1.0e9) return stamp;
    int week;
    double sow;
    GpsTime::unix2gps(stamp, week, sow);
    return sow;
}

bool Replay::readImuText(const std::string &path, bool is_rate, std::vector<IMU> &out, std::string *err) {
    std::vector<std::vector<double>> rows;
    if (!readRows(path, 7, rows, err)) return false;
    out.clear();
    IMU imu_pre;
    imu_pre.time = 0;
    for (const auto &r : rows) { // imuCallback (fusion_ros.cc:123-162)
        IMU imu;
        imu.time  = toGpsSecondOfWeek(r[0]);
        imu.dt    = imu.time - imu_pre.time;
        double s  = is_rate ? imu.dt : 1.0;
        imu.dtheta = Vector3d(r[1] * s, r[2] * s, r[3] * s);
        imu.dvel   = Vector3d(r[4] * s, r[5] * s, r[6] * s);
        imu.odovel = 0;
        bool ready = imu_pre.time != 0; // the first message only initialises dt (:146-148)
        imu_pre    = imu;
        if (ready) out.push_back(imu);
    }
    return true;
}

bool Replay::readGnssText(const std::string &path, std::vector<GNSS> &out, std::string *err) {
    std::vector<std::vector<double>> rows;
    if (!readRows(path, 7, rows, err)) return false;
    out.clear();
    for (const auto &r : rows) { // gnssCallback (fusion_ros.cc:164-199)
        GNSS g;
        g.time       = toGpsSecondOfWeek(r[0]);
        g.blh        = Vector3d(r[1] * D2R, r[2] * D2R, r[3]);
        g.std        = Vector3d(r[4], r[5], r[6]);
        g.isyawvalid = false;
        out.push_back(g);
    }
    return true;
}

bool Replay::readImageList(const std::string &path, std::vector<ImageEntry> &out, std::string *err) {
    std::ifstream f(path);
    if (!f) return setErr(err, "cannot open " + path);
    const std::string dir = dirOf(path);
    std::string line;
    out.clear();
    while (std::getline(f, line)) {
        for (char &c : line)
            if (c == ',') c = ' ';
        std::istringstream ss(line);
        double t;
        std::string name;
        if (!(ss >> t >> name)) continue;
        out.push_back({toGpsSecondOfWeek(t), name.front() == '/' ? name : dir + "/" + name});
    }
    return true;
}

bool Replay::loadPnm(const std::string &path, Mat &image, std::string *err) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) return setErr(err, "cannot open " + path);
    auto token = [&](std::string &tok) { // header tokens, '#' comments skipped
        tok.clear();
        int c;
        while ((c = fgetc(fp)) != EOF) {
            if (c == '#') {
                while ((c = fgetc(fp)) != EOF && c != '\n') {
                }
                continue;
            }
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                if (!tok.empty()) return true;
                continue;
            }
            tok.push_back((char) c);
        }
        return !tok.empty();
    };
    std::string magic, sw, sh, smax;
    bool ok = token(magic) && token(sw) && token(sh) && token(smax);
    int w = ok ? atoi(sw.c_str()) : 0, h = ok ? atoi(sh.c_str()) : 0, maxv = ok ? atoi(smax.c_str()) : 0;
    int chans = magic == "P5" ? 1 : (magic == "P6" ? 3 : 0);
    if (!ok || chans == 0 || w <= 0 || h <= 0 || maxv != 255) {
        fclose(fp);
        return setErr(err, path + ": not an 8-bit binary PGM/PPM");
    }
    image     = Mat(h, w, chans);
    size_t n  = (size_t) w * h * chans;
    size_t rd = fread(image.data, 1, n, fp);
    fclose(fp);
    if (rd != n) return setErr(err, path + ": truncated pixel data");
    if (chans == 3) // PPM stores RGB, the reference's colour input is BGR8 (fusion_ros.cc:209-211, tracking.cc:109-111)
        for (size_t k = 0; k < n; k += 3) std::swap(image.data[k], image.data[k + 2]);
    return true;
}

bool Replay::run(const ReplayOptions &options, ReplaySummary &summary, std::string *err) {
    YamlLite config;
    if (!YamlLite::load(options.configfile, config, err)) return false;
    std::string outputpath = options.outputpath.empty() ? (config.has("outputpath") ? config.str("outputpath") : std::string()) : options.outputpath;
    if (outputpath.empty()) return setErr(err, "no output path");
    struct stat st;
    if (stat(outputpath.c_str(), &st) != 0) mkdir(outputpath.c_str(), 0755); // fusion_ros.cc:76-83
    if (stat(outputpath.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return setErr(err, "Failed to open outputpath " + outputpath);
    summary.outputpath = outputpath;
    const bool isusegnssoutage = config.has("isusegnssoutage") && config.boolean("isusegnssoutage");
    const double gnssoutagetime = config.has("gnssoutagetime") ? config.real("gnssoutagetime") : 0.0;
    const double gnssthreshold  = config.has("gnssthreshold") ? config.real("gnssthreshold") : 1.0e9;

    std::vector<IMU> imus;
    std::vector<GNSS> gnss;
    std::vector<ImageEntry> images;
    if (!readImuText(options.imufile, options.imu_is_rate, imus, err)) return false;
    if (!options.gnssfile.empty() && !readGnssText(options.gnssfile, gnss, err)) return false;
    if (!options.imagelist.empty() && !readImageList(options.imagelist, images, err)) return false;

    GVINS gvins(options.configfile, outputpath, nullptr);
    if (!gvins.isRunning()) return setErr(err, "GVINS failed to start: " + gvins.error());
    if (options.wait_poll_us > 0) gvins.setWaitMode(ICG_WAIT_POLL, options.wait_poll_us);

    auto in_range = [&](double t) { return (options.start_time == 0 || t >= options.start_time) && (options.end_time == 0 || t <= options.end_time); };
    auto t0       = std::chrono::steady_clock::now();
    size_t ii = 0, gi = 0, fi = 0;
    double first = 0, last = 0;
    try {
        while (ii < imus.size() || gi < gnss.size() || fi < images.size()) {
            const double ti = ii < imus.size() ? imus[ii].time : 1e300, tg = gi < gnss.size() ? gnss[gi].time : 1e300,
                         tf = fi < images.size() ? images[fi].time : 1e300;
            if (ti <= tg && ti <= tf) {
                const IMU &imu = imus[ii++];
                if (!in_range(imu.time)) continue;
                if (first == 0) first = imu.time;
                last = imu.time;
                gvins.addNewImu(imu);
                summary.imu++;
            } else if (tg <= tf) {
                const GNSS &g = gnss[gi++];
                if (!in_range(g.time)) continue;
                bool bad = (g.std[0] == 0) || (g.std[1] == 0) || (g.std[2] == 0) ||
                           !((g.std[0] < gnssthreshold) && (g.std[1] < gnssthreshold) && (g.std[2] < gnssthreshold));
                if (bad || (isusegnssoutage && (g.time >= gnssoutagetime))) {
                    summary.gnss_dropped++;
                    continue;
                }
                gvins.addNewGnss(g);
                summary.gnss++;
            } else {
                const ImageEntry &e = images[fi++];
                if (!in_range(e.time)) continue;
                Mat image;
                if (!loadPnm(e.path, image, err)) return false;
                gvins.addNewFrame(Frame::createFrame(e.time, image, gvins.ids()));
                summary.frames++;
            }
        }
        gvins.setFinished();
    } catch (const std::exception &e) {
        return setErr(err, e.what());
    }
    summary.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    summary.data_seconds = last - first;
    summary.counters     = gvins.counters();
    summary.final_state  = (int) gvins.gvinsState();
    return true;
}

bool Replay::runMany(const std::vector<ReplayOptions> &options, std::vector<ReplaySummary> &summaries, double *wall_seconds, std::string *err) {
    const size_t n = options.size();
    summaries.assign(n, ReplaySummary());
    std::vector<std::string> errors(n);
    std::vector<char> ok(n, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    for (size_t k = 0; k < n; k++) threads.emplace_back([&, k]() { ok[k] = run(options[k], summaries[k], &errors[k]) ? 1 : 0; });
    for (auto &t : threads) t.join();
    if (wall_seconds) *wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t k = 0; k < n; k++)
        if (!ok[k]) return setErr(err, "replay " + std::to_string(k) + ": " + errors[k]);
    return true;
}

namespace {
// marginalizations that went through a MarginalizationBatch since the last Replay::takeLockstepMarginalizationCounts() (all groups)
std::atomic<long> g_lockstep_marg_batches{0}, g_lockstep_marg_windows{0};
struct LockstepStream {
    std::unique_ptr<GVINS> gvins;
    std::vector<IMU> imus;
    std::vector<GNSS> gnss;
    std::vector<Replay::ImageEntry> images;
    size_t ii = 0, gi = 0, fi = 0;
    bool isusegnssoutage = false;
    double gnssoutagetime = 0, gnssthreshold = 1e9, first = 0, last = 0;
    ReplaySummary *summary = nullptr;
    std::unique_ptr<BatchWindowProblem> problem;
    int n_visual = 0;
    bool alone   = false;
};
} // namespace

void Replay::takeLockstepMarginalizationCounts(long out[2]) {
    out[0] = g_lockstep_marg_batches.exchange(0);
    out[1] = g_lockstep_marg_windows.exchange(0);
}

bool Replay::runLockstep(const std::vector<ReplayOptions> &options, std::vector<ReplaySummary> &summaries, double *wall_seconds, long *shared_solves,
                         std::string *err, int solver_host_threads) {
    const size_t n = options.size();
    summaries.assign(n, ReplaySummary());
    std::vector<LockstepStream> S(n);
    for (size_t k = 0; k < n; k++) {
        const ReplayOptions &o = options[k];
        YamlLite config;
        if (!YamlLite::load(o.configfile, config, err)) return false;
        std::string outputpath = o.outputpath.empty() ? (config.has("outputpath") ? config.str("outputpath") : std::string()) : o.outputpath;
        if (outputpath.empty()) return setErr(err, "no output path");
        struct stat st;
        if (stat(outputpath.c_str(), &st) != 0) mkdir(outputpath.c_str(), 0755);
        if (stat(outputpath.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return setErr(err, "Failed to open outputpath " + outputpath);
        summaries[k].outputpath = outputpath;
        S[k].summary            = &summaries[k];
        S[k].isusegnssoutage    = config.has("isusegnssoutage") && config.boolean("isusegnssoutage");
        S[k].gnssoutagetime     = config.has("gnssoutagetime") ? config.real("gnssoutagetime") : 0.0;
        S[k].gnssthreshold      = config.has("gnssthreshold") ? config.real("gnssthreshold") : 1.0e9;
        if (!readImuText(o.imufile, o.imu_is_rate, S[k].imus, err)) return false;
        if (!o.gnssfile.empty() && !readGnssText(o.gnssfile, S[k].gnss, err)) return false;
        if (!o.imagelist.empty() && !readImageList(o.imagelist, S[k].images, err)) return false;
        S[k].gvins.reset(new GVINS(o.configfile, outputpath, nullptr));
        if (!S[k].gvins->isRunning()) return setErr(err, "GVINS failed to start: " + S[k].gvins->error());
        S[k].gvins->setDeferredWindowSolves(true);
    }
    long n_solves = 0, n_batches = 0, largest = 0, n_marg_batches = 0, n_marg_windows = 0;
    auto t0 = std::chrono::steady_clock::now();
    try {
        WindowSolverBatch batch(0, 1.0, solver_host_threads); // huber delta 1 of the reprojection factors (ic_gvins.cc:1773)
        // The marginalizations of a tick share their device launches too (MarginalizationBatch, host/marg_batch.h; Huber delta 0: the
        // reference builds the prior from uncorrected reprojection factors, ic_gvins.cc:1600-1606).  On by default since round 5 (the class
        // went through the replays on the MI355X: tests/test_gpu_zz_marg_batch.py, profiles/r05_first_call); ICG_LOCKSTEP_MARG_BATCH=0
        // gives every estimator its own MarginalizationInfo::marginalization() back.
        std::unique_ptr<MarginalizationBatch> marg_batch;
        {
            const char *e = getenv("ICG_LOCKSTEP_MARG_BATCH");
            if (!e || atoi(e) > 0) marg_batch.reset(new MarginalizationBatch(0, 0.0, solver_host_threads));
        }
        std::vector<LockstepStream *> due;
        bool any = true;
        while (any) {
            any = false;
            due.clear();
            for (LockstepStream &L : S) { // one IMU epoch of every stream, with the GNSS fixes / images that precede it
                const ReplayOptions &o = options[(size_t) (&L - &S[0])];
                auto in_range          = [&](double t) { return (o.start_time == 0 || t >= o.start_time) && (o.end_time == 0 || t <= o.end_time); };
                bool imu_done          = false;
                while (!imu_done && (L.ii < L.imus.size() || L.gi < L.gnss.size() || L.fi < L.images.size())) {
                    any             = true;
                    const double ti = L.ii < L.imus.size() ? L.imus[L.ii].time : 1e300, tg = L.gi < L.gnss.size() ? L.gnss[L.gi].time : 1e300,
                                 tf = L.fi < L.images.size() ? L.images[L.fi].time : 1e300;
                    if (ti <= tg && ti <= tf) {
                        const IMU &imu = L.imus[L.ii++];
                        if (!in_range(imu.time)) continue;
                        if (L.first == 0) L.first = imu.time;
                        L.last = imu.time;
                        L.gvins->addNewImu(imu);
                        L.summary->imu++;
                        imu_done = true;
                    } else if (tg <= tf) {
                        const GNSS &g = L.gnss[L.gi++];
                        if (!in_range(g.time)) continue;
                        bool bad = (g.std[0] == 0) || (g.std[1] == 0) || (g.std[2] == 0) ||
                                   !((g.std[0] < L.gnssthreshold) && (g.std[1] < L.gnssthreshold) && (g.std[2] < L.gnssthreshold));
                        if (bad || (L.isusegnssoutage && (g.time >= L.gnssoutagetime))) {
                            L.summary->gnss_dropped++;
                            continue;
                        }
                        L.gvins->addNewGnss(g);
                        L.summary->gnss++;
                    } else {
                        const ImageEntry &e = L.images[L.fi++];
                        if (!in_range(e.time)) continue;
                        Mat image;
                        if (!loadPnm(e.path, image, err)) return false;
                        L.gvins->addNewFrame(Frame::createFrame(e.time, image, L.gvins->ids()));
                        L.summary->frames++;
                    }
                }
                if (L.gvins->windowSolvePending()) due.push_back(&L);
            }
            if (due.empty()) continue;
            // ---- the window solves of this tick, together --------------------------------------------------------------------------------
            batch.clear();
            std::vector<LockstepStream *> batched;
            for (LockstepStream *L : due) {
                L->n_visual = L->gvins->beginWindowSolve();
                L->alone    = L->n_visual == 0 || L->gvins->windowCameraColumnsBound() > WindowSolverBatch::kMaxCameraColumns;
                if (L->alone) continue; // (no visual factors yet, or a window too wide for the batched assembly's LDS tile: 15 keyframes)
                L->problem.reset(new BatchWindowProblem(batch, batch.addWindow()));
                L->gvins->populateWindow(*L->problem, L->n_visual);
                batched.push_back(L);
            }
            n_solves += (long) due.size();
            if (!batched.empty()) {
                n_batches++;
                largest = std::max(largest, (long) batched.size());
                WindowSolver::Options opt;
                std::vector<WindowSolver::Summary> first, second;
                auto ta                = std::chrono::steady_clock::now();
                opt.max_num_iterations = batched[0]->gvins->firstNumIterations();
                if (!batch.solve(opt, &first)) return setErr(err, "batched window solve: " + batch.error());
                const double first_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
                for (LockstepStream *L : batched) L->gvins->betweenWindowSolves(*L->problem);
                std::vector<int> removed = batch.removeReprojectionFactorsByChi2(5.991);
                auto tb                  = std::chrono::steady_clock::now();
                opt.max_num_iterations   = batched[0]->gvins->secondNumIterations();
                if (!batch.solve(opt, &second)) return setErr(err, "batched window solve: " + batch.error());
                const double second_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
                for (LockstepStream *L : batched) {
                    const size_t w = (size_t) L->problem->window();
                    L->gvins->finishWindowSolve(first[w], second[w], first_ms, second_ms, removed[w]);
                }
            }
            std::vector<LockstepStream *> solved;
            for (LockstepStream *L : due) {
                if (L->alone)
                    L->gvins->solveWindowAlone(L->n_visual); // on the estimator's own WindowSolver (the window is already prepared)
                else if (marg_batch)
                    solved.push_back(L);
                else
                    L->gvins->afterWindowSolve();
                L->problem.reset();
            }
            if (!solved.empty()) {
                // ---- window maintenance of this tick; the marginalizations (one per estimator that is over its keyframe count), together -------
                for (LockstepStream *L : solved) L->gvins->afterWindowSolveBegin();
                for (;;) {
                    std::vector<LockstepStream *> marg;
                    for (LockstepStream *L : solved)
                        if (L->gvins->marginalizationDue()) marg.push_back(L);
                    if (marg.empty()) break;
                    marg_batch->clear();
                    std::vector<GVINS::MarginalizationJob> jobs(marg.size());
                    std::vector<int> window(marg.size(), -1);
                    std::vector<std::array<double *, 6>> fac; // (one estimator's factors are collected first: only a job WITH device factors gets a window)
                    for (size_t k = 0; k < marg.size(); k++) {
                        fac.clear();
                        std::vector<ReprojectionFactor *> ptr;
                        marg[k]->gvins->beginMarginalization(jobs[k], [&](ReprojectionFactor *f, double *pi, double *pj, double *ext, double *inv, double *td) {
                            ptr.push_back(f);
                            fac.push_back({pi, pj, ext, inv, td, nullptr});
                        });
                        if (ptr.empty()) continue;
                        window[k] = marg_batch->addWindow(jobs[k].info);
                        for (size_t i = 0; i < ptr.size(); i++) marg_batch->addReprojectionFactor(window[k], ptr[i], fac[i][0], fac[i][1], fac[i][2], fac[i][3], fac[i][4]);
                    }
                    std::vector<char> ok;
                    if (marg_batch->numWindows() > 0) {
                        if (!marg_batch->marginalize(&ok)) return setErr(err, "batched marginalization: " + marg_batch->error());
                        n_marg_batches++;
                        n_marg_windows += marg_batch->numWindows();
                    }
                    for (size_t k = 0; k < marg.size(); k++) {
                        const bool valid = window[k] >= 0 ? ok[(size_t) window[k]] != 0 : jobs[k].info->marginalization(); // (host factors only: on its own)
                        marg[k]->gvins->finishMarginalization(jobs[k], valid);
                    }
                    marg_batch->clear();
                }
                for (LockstepStream *L : solved) L->gvins->afterWindowSolveEnd();
            }
        }
        for (LockstepStream &L : S) L.gvins->setFinished();
    } catch (const std::exception &e) {
        return setErr(err, e.what());
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (wall_seconds) *wall_seconds = wall;
    if (shared_solves) shared_solves[0] = n_solves, shared_solves[1] = n_batches, shared_solves[2] = largest;
    g_lockstep_marg_batches.fetch_add(n_marg_batches), g_lockstep_marg_windows.fetch_add(n_marg_windows);
    for (LockstepStream &L : S) {
        L.summary->wall_seconds = wall;
        L.summary->data_seconds = L.last - L.first;
        L.summary->counters     = L.gvins->counters();
        L.summary->final_state  = (int) L.gvins->gvinsState();
    }
    return true;
}

bool Replay::runLockstepGroups(const std::vector<ReplayOptions> &options, int groups, std::vector<ReplaySummary> &summaries, double *wall_seconds,
                               long *shared_solves, std::string *err) {
    const size_t n = options.size();
    const size_t G = (size_t) std::max(1, std::min(groups, (int) n));
    if (G == 1) return runLockstep(options, summaries, wall_seconds, shared_solves, err);
    summaries.assign(n, ReplaySummary());
    std::vector<std::vector<ReplayOptions>> part(G);
    std::vector<std::vector<ReplaySummary>> out(G);
    std::vector<std::string> errors(G);
    std::vector<char> ok(G, 0);
    std::vector<std::array<long, 3>> shared(G);
    std::vector<size_t> begin(G + 1, 0);
    for (size_t g = 0; g < G; g++) begin[g + 1] = begin[g] + n / G + (g < n % G ? 1 : 0);
    for (size_t g = 0; g < G; g++) part[g].assign(options.begin() + (long) begin[g], options.begin() + (long) begin[g + 1]);
    auto t0 = std::chrono::steady_clock::now();
    const int solver_threads = std::max(1, 16 / (int) G);
    std::vector<std::thread> threads;
    for (size_t g = 0; g < G; g++)
        threads.emplace_back([&, g]() { ok[g] = runLockstep(part[g], out[g], nullptr, shared[g].data(), &errors[g], solver_threads) ? 1 : 0; });
    for (auto &t : threads) t.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (wall_seconds) *wall_seconds = wall;
    long total[3] = {0, 0, 0};
    for (size_t g = 0; g < G; g++) {
        if (!ok[g]) return setErr(err, "lock-step group " + std::to_string(g) + ": " + errors[g]);
        for (size_t k = 0; k < out[g].size(); k++) summaries[begin[g] + k] = out[g][k];
        total[0] += shared[g][0], total[1] += shared[g][1], total[2] = std::max(total[2], shared[g][2]);
    }
    if (shared_solves) shared_solves[0] = total[0], shared_solves[1] = total[1], shared_solves[2] = total[2];
    return true;
}

} // namespace icg
