// MarginalizationBatch: see marg_batch.h.  Reference: factors/marginalization_info.h:73-101 (marginalization), :153-273 (the steps).
#include "marg_batch.h"
#include "solver_batch_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace icg {

template <typename F> void MarginalizationBatch::forEachWindow(size_t n, F &&fn) {
    if (host_threads_ <= 1 || n < 4) {
        for (size_t w = 0; w < n; w++) fn(w);
        return;
    }
    if (!pool_) pool_.reset(new HostPool(host_threads_));
    const std::function<void(int)> f = [&](int w) { fn((size_t) w); };
    pool_->parallelFor((int) n, f);
}

static icg_ctx *backendContext(int device, const char *who) {
    icg_ctx_config cfg{};
    cfg.device = device, cfg.width = 64, cfg.height = 64, cfg.n_slots = 1, cfg.max_batch = 1, cfg.max_points = 64; // (holds no images)
    icg_ctx *ctx = nullptr;
    if (icg_ctx_create(&cfg, &ctx) != ICG_OK) throw std::runtime_error(std::string(who) + ": " + icg_last_error(nullptr));
    return ctx;
}

MarginalizationBatch::MarginalizationBatch(int device, double huber_delta, int host_threads) : device_(device), huber_(huber_delta) {
    host_threads_ = host_threads > 0 ? host_threads : (int) std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("ICG_SOLVER_THREADS")) host_threads_ = std::max(1, atoi(e)); // diagnostics (shared with WindowSolverBatch)
    ctx_ = backendContext(device, "MarginalizationBatch");
}

MarginalizationBatch::~MarginalizationBatch() {
    clear();
    if (dense_ctx_) icg_ctx_destroy(dense_ctx_);
    icg_ctx_destroy(ctx_);
}

void MarginalizationBatch::clear() {
    for (auto &W : windows_)
        if (W->info && W->info->batch_ == W.get()) W->info->batch_ = nullptr; // (an info may outlive the batch: it keeps only its results)
    windows_.clear();
    retired_.clear(); // (the factor records of the windows marginalized since the last clear())
    laid_out_  = false;
    n_factors_ = n_poses_ = n_lm_ = 0;
}

int MarginalizationBatch::addWindow(const std::shared_ptr<MarginalizationInfo> &info) {
    if (!info) throw std::runtime_error("MarginalizationBatch: null MarginalizationInfo");
    std::unique_ptr<Slice> W(new Slice);
    W->owner = this;
    W->info  = info;
    info->setDeviceFactors(W.get());
    windows_.push_back(std::move(W));
    laid_out_ = false;
    return (int) windows_.size() - 1;
}

void MarginalizationBatch::addReprojectionFactor(int w, const ReprojectionFactor *factor, double *pose_i, double *pose_j, double *extrinsic,
                                                 double *invdepth, double *td) {
    Slice &W = *windows_.at((size_t) w);
    if (!factor || !pose_i || !pose_j || !extrinsic || !invdepth || !td) throw std::runtime_error("MarginalizationBatch: null block");
    if (W.ext && (W.ext != extrinsic || W.td != td)) throw std::runtime_error("MarginalizationBatch: one extrinsic / td block per window");
    W.ext = extrinsic, W.td = td;
    auto index_of = [](std::unordered_map<const double *, int> &m, std::vector<double *> &v, double *p) {
        auto it = m.find(p);
        if (it != m.end()) return it->second;
        const int k = (int) v.size();
        v.push_back(p);
        m[p] = k;
        return k;
    };
    W.members.insert(factor);
    W.obs.insert(W.obs.end(), factor->observation(), factor->observation() + 15);
    W.idx_i.push_back(index_of(W.pose_index, W.poses, pose_i));
    W.idx_j.push_back(index_of(W.pose_index, W.poses, pose_j));
    W.idx_lm.push_back(index_of(W.lm_index, W.landmarks, invdepth));
    laid_out_ = false;
}

// the factor set of all windows, sorted by window, with the partition the *_windows calls work on
bool MarginalizationBatch::layout() {
    n_factors_ = n_poses_ = n_lm_ = 0;
    std::vector<int32_t> fac_off{0}, lm_off{0};
    std::unordered_map<const double *, size_t> pose_owner;
    for (size_t w = 0; w < windows_.size(); w++) {
        Slice &W    = *windows_[w];
        W.fac_begin = n_factors_, W.pose_begin = n_poses_, W.lm_begin = n_lm_;
        for (double *p : W.poses) {
            auto it = pose_owner.find(p);
            if (it != pose_owner.end() && it->second != w) {
                error_ = "a pose block is used by the reprojection factors of two windows";
                return false;
            }
            pose_owner[p] = w;
        }
        n_factors_ += W.size(), n_poses_ += (int) W.poses.size(), n_lm_ += (int) W.landmarks.size();
        fac_off.push_back(n_factors_), lm_off.push_back(n_lm_);
    }
    laid_out_ = true;
    if (n_factors_ == 0) return true; // (only host factors anywhere: nothing for the device)
    // the windows' factors are written by the pool's threads straight into the context's pinned staging block (34 MB of observations at
    // 256 C2 windows: through a pageable vector and hipMemcpy they were most of this function's 13 ms)
    double *obs = nullptr;
    int32_t *idx3 = nullptr;
    if (icg_reproj_stage_factors(ctx_, n_factors_, &obs, &idx3) != ICG_OK) {
        error_    = icg_last_error(ctx_);
        laid_out_ = false;
        return false;
    }
    int32_t *ii = idx3, *jj = idx3 + n_factors_, *ll = idx3 + 2 * (size_t) n_factors_;
    forEachWindow(windows_.size(), [&](size_t w) {
        const Slice &W = *windows_[w];
        for (int c = 0; c < 15; c++) { // (component-major: one contiguous destination run per component and window)
            double *dst = obs + (size_t) c * n_factors_ + (size_t) W.fac_begin;
            for (int k = 0; k < W.size(); k++) dst[k] = W.obs[(size_t) 15 * k + c];
        }
        for (int k = 0; k < W.size(); k++) {
            const size_t f = (size_t) W.fac_begin + (size_t) k;
            ii[f] = W.pose_begin + W.idx_i[(size_t) k], jj[f] = W.pose_begin + W.idx_j[(size_t) k], ll[f] = W.lm_begin + W.idx_lm[(size_t) k];
        }
    });
    if (icg_reproj_commit_factors(ctx_) != ICG_OK ||
        icg_reproj_set_windows(ctx_, (int) windows_.size(), fac_off.data(), lm_off.data()) != ICG_OK) {
        error_    = icg_last_error(ctx_);
        laid_out_ = false;
        return false;
    }
    return true;
}

bool MarginalizationBatch::Slice::evaluateCorrected(double huber_delta) {
    if (!evaluated) {
        err = "a window of a MarginalizationBatch is marginalized by MarginalizationBatch::marginalize(), not on its own";
        return false;
    }
    if (huber_delta != owner->huber_) {
        err = "the Huber delta of a window's reprojection factors differs from the batch's";
        return false;
    }
    return true; // (evaluated with every other window's factors, one launch: marginalize() phase 1)
}

bool MarginalizationBatch::Slice::accumulateLandmarkEliminated(const std::unordered_map<const double *, int> &, int, double *, double *, double *) {
    err = "a window of a MarginalizationBatch is marginalized by MarginalizationBatch::marginalize(), not on its own";
    return false;
}

bool MarginalizationBatch::Slice::accumulateNormal(const std::unordered_map<const double *, int> &column_of, int local_size, double *H0, double *b0) {
    return owner->denseNormalOfWindow(*this, column_of, local_size, H0, b0);
}

// The dense M2 of one window (marginalization_info.h:195-230): its factors alone on a one-window context, evaluated there once more (the
// batched evaluation lives in the partitioned context, whose dense assembly would mix the windows' shared columns).
bool MarginalizationBatch::denseNormalOfWindow(Slice &W, const std::unordered_map<const double *, int> &column_of, int local_size, double *H0,
                                               double *b0) {
    const int n = W.size();
    if (n == 0) return true;
    std::lock_guard<std::mutex> lock(dense_mutex_);
    try {
        if (!dense_ctx_) dense_ctx_ = backendContext(device_, "MarginalizationBatch (dense path)");
    } catch (const std::exception &e) {
        W.err = e.what();
        return false;
    }
    std::vector<double> obs((size_t) 15 * n);
    for (int k = 0; k < n; k++)
        for (int c = 0; c < 15; c++) obs[(size_t) c * n + k] = W.obs[(size_t) 15 * k + c];
    std::vector<double> poses(7 * W.poses.size()), inv(W.landmarks.size());
    for (size_t k = 0; k < W.poses.size(); k++) memcpy(&poses[7 * k], W.poses[k], sizeof(double) * 7);
    for (size_t k = 0; k < W.landmarks.size(); k++) inv[k] = *W.landmarks[k];
    auto col = [&](const double *p) {
        auto it = column_of.find(p);
        return it == column_of.end() ? -1 : it->second;
    };
    std::vector<int32_t> cp(W.poses.size()), cl(W.landmarks.size());
    for (size_t k = 0; k < W.poses.size(); k++) cp[k] = col(W.poses[k]);
    for (size_t k = 0; k < W.landmarks.size(); k++) cl[k] = col(W.landmarks[k]);
    int rc = icg_reproj_set_factors(dense_ctx_, n, obs.data(), W.idx_i.data(), W.idx_j.data(), W.idx_lm.data());
    if (rc == ICG_OK)
        rc = icg_reproj_eval_resident(dense_ctx_, (int) W.poses.size(), poses.data(), W.ext, (int) W.landmarks.size(), inv.data(), *W.td, 1, huber_,
                                      nullptr, nullptr);
    if (rc == ICG_OK) rc = icg_reproj_accumulate_normal(dense_ctx_, local_size, cp.data(), col(W.ext), cl.data(), col(W.td), H0, b0);
    if (rc != ICG_OK) {
        W.err = icg_last_error(dense_ctx_);
        return false;
    }
    return true;
}

bool MarginalizationBatch::marginalize(std::vector<char> *ok) {
    const size_t NW = windows_.size();
    if (ok) ok->assign(NW, 0);
    n_structured_ = n_dense_ = 0;
    phase_ms_[0] = phase_ms_[1] = phase_ms_[2] = phase_ms_[3] = 0;
    error_.clear();
    window_error_.clear();
    if (NW == 0) return true;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms  = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_layout = now();
    if (!laid_out_ && !layout()) return false;
    const double layout_ms = ms(t_layout, now());
    auto fail = [&](const std::string &what) {
        error_ = what;
        for (auto &W : windows_) W->evaluated = false;
        return false;
    };

    // ---- 1: every window's reprojection factors, one launch (residual_block_info.h:44-88 with the corrector :59-87 on the device) -------
    auto t0 = now();
    if (n_factors_ > 0) {
        std::vector<double> poses(7 * (size_t) n_poses_), ext(7 * NW, 0.0), inv((size_t) n_lm_), td(NW, 0.0);
        forEachWindow(NW, [&](size_t w) {
            const Slice &W = *windows_[w];
            for (size_t k = 0; k < W.poses.size(); k++) memcpy(&poses[7 * ((size_t) W.pose_begin + k)], W.poses[k], sizeof(double) * 7);
            for (size_t k = 0; k < W.landmarks.size(); k++) inv[(size_t) W.lm_begin + k] = *W.landmarks[k];
            if (W.ext)
                memcpy(&ext[7 * w], W.ext, sizeof(double) * 7);
            else
                ext[7 * w + 6] = 1.0; // (a window without reprojection factors: an identity nobody reads)
            if (W.td) td[w] = *W.td;
        });
        if (icg_reproj_eval_windows(ctx_, n_poses_, poses.data(), ext.data(), n_lm_, inv.data(), td.data(), 1, huber_) != ICG_OK)
            return fail(icg_last_error(ctx_));
    }
    for (auto &W : windows_) W->evaluated = true;
    auto t1 = now();

    // ---- 2: M1 bookkeeping, host factors, compact camera layout per window -------------------------------------------------------------
    struct State {
        bool alive{false}, planned{false};
        MarginalizationInfo::StructuredPlan plan;
    };
    std::vector<State> st(NW);
    const bool force_dense = MarginalizationInfo::denseForced();
    forEachWindow(NW, [&](size_t w) {
        MarginalizationInfo &I = *windows_[w]->info;
        if (!I.updateParameterBlocksIndex() || !I.preMarginalization()) { // (:75-86: nothing to marginalize / an evaluation failed)
            I.isvalid_ = false;
            I.releaseMemory();
            return;
        }
        st[w].alive   = true;
        st[w].planned = !force_dense && I.planStructured(st[w].plan);
    });
    auto t2 = now();

    // ---- 3: assembly + landmark elimination of every planned window, one launch sequence; the landmark diagonals --------------------------
    // (csrc/reproj.hip, schur_impl: reduced systems of up to WindowSolverBatch::kMaxCameraColumns columns; a wider window takes the dense
    // path on its own)
    const int max_camera_columns = WindowSolverBatch::kMaxCameraColumns;
    for (size_t w = 0; w < NW; w++) {
        if (!st[w].planned) continue;
        const Slice &W = *windows_[w];
        int V          = 0;
        for (double *p : W.poses) V += st[w].plan.camera_column_of.count(p) ? 6 : 0;
        V += (W.ext && st[w].plan.camera_column_of.count(W.ext) ? 6 : 0) + (W.td && st[w].plan.camera_column_of.count(W.td) ? 1 : 0);
        if (V > max_camera_columns) st[w].planned = false;
    }
    int P = 0;
    for (size_t w = 0; w < NW; w++)
        if (st[w].planned) P = std::max(P, st[w].plan.P);
    std::vector<double> S, s, hll;
    if (P > 0) {
        std::vector<int32_t> col_pose((size_t) n_poses_, -1), col_ext(NW, -1), col_td(NW, -1);
        for (size_t w = 0; w < NW; w++) {
            if (!st[w].planned) continue; // (its columns stay constant: the window adds nothing to any reduced system that is read)
            const Slice &W = *windows_[w];
            auto col       = [&](const double *p) {
                auto it = st[w].plan.camera_column_of.find(p);
                return it == st[w].plan.camera_column_of.end() ? -1 : it->second;
            };
            for (size_t k = 0; k < W.poses.size(); k++) col_pose[(size_t) W.pose_begin + k] = col(W.poses[k]);
            col_ext[w] = col(W.ext), col_td[w] = col(W.td);
        }
        std::vector<uint8_t> reassemble(NW, 1);
        std::vector<double> damp(NW, 0.0), diag_cc(NW * (size_t) P), cost(NW, 0.0);
        S.assign(NW * (size_t) P * P, 0.0), s.assign(NW * (size_t) P, 0.0), hll.assign((size_t) std::max(n_lm_, 1), 0.0);
        if (icg_reproj_schur_windows(ctx_, P, col_pose.data(), col_ext.data(), col_td.data(), nullptr, reassemble.data(), damp.data(), 0.0, 0.0, S.data(),
                                     s.data(), diag_cc.data(), cost.data()) != ICG_OK ||
            icg_reproj_landmark_diag_windows(ctx_, hll.data()) != ICG_OK)
            return fail(icg_last_error(ctx_));
    }
    auto t3 = now();

    // ---- 4: guard + M3 on the camera block (or the dense M2 + M3), linearization -------------------------------------------------------------
    std::vector<char> structured(NW, 0), good(NW, 0);
    forEachWindow(NW, [&](size_t w) {
        if (!st[w].alive) return;
        Slice &W               = *windows_[w];
        MarginalizationInfo &I = *W.info;
        bool done              = false;
        if (st[w].planned) {
            MarginalizationInfo::StructuredPlan &plan = st[w].plan;
            const int Pw                              = plan.P;
            const double *Sw = &S[w * (size_t) P * P], *sw = &s[w * (size_t) P];
            for (int i = 0; i < Pw; i++) {
                for (int j = 0; j < Pw; j++) plan.H[(size_t) i * Pw + j] += Sw[(size_t) i * P + j];
                plan.b[(size_t) i] += sw[i];
            }
            double mn = W.landmarks.empty() ? 0.0 : hll[(size_t) W.lm_begin];
            for (size_t l = 0; l < W.landmarks.size(); l++) mn = std::min(mn, hll[(size_t) W.lm_begin + l]);
            done = I.finishStructured(plan, mn);
            structured[w] = done ? 1 : 0;
        }
        if (!done) {
            if (!I.constructEquation()) { // (:88-92 with the window's own dense assembly on the device)
                I.isvalid_ = false;
                I.releaseMemory();
                return;
            }
            I.schurElimination();
        }
        I.linearization();
        good[w] = 1;
    });
    // (:99) the factor records of the windows that went through are retired: kept by this batch until clear() / destruction (factors.h
    // MarginalizationInfo::releaseMemory: freeing ~3 300 heap blocks per window in line is 45 of the 90 ms of 256 C2 windows)
    for (size_t w = 0; w < NW; w++)
        if (good[w]) windows_[w]->info->releaseMemoryInto(retired_);
    auto t4 = now();
    for (size_t w = 0; w < NW; w++) {
        windows_[w]->evaluated = false;
        if (good[w]) (structured[w] ? n_structured_ : n_dense_)++;
        if (ok) (*ok)[w] = good[w];
        // (a window that failed on its own — its dense path left a message — is reported through ok[w] and windowError() only: the other
        // windows' priors are good, as they are when every stream marginalizes alone and one of them logs "no valid prior")
        if (!good[w] && st[w].alive && window_error_.empty() && !windows_[w]->err.empty())
            window_error_ = "window " + std::to_string(w) + ": " + windows_[w]->err;
    }
    phase_ms_[0] = ms(t0, t1), phase_ms_[1] = ms(t1, t2), phase_ms_[2] = ms(t2, t3), phase_ms_[3] = ms(t3, t4);
    if (getenv("ICG_MARG_DEBUG"))
        fprintf(stderr, "[marginalization batch] %zu windows (%d structured, %d dense): evaluate %.3f ms, bookkeeping + host factors %.3f ms, assemble + eliminate %.3f ms, M3 + linearize %.3f ms; layout (factor upload + partition) %.3f ms\n",
                NW, n_structured_, n_dense_, phase_ms_[0], phase_ms_[1], phase_ms_[2], phase_ms_[3], layout_ms);
    return true; // (false above: a launch all windows share failed)
}

} // namespace icg
