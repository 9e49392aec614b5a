// WindowSolver: Levenberg-Marquardt on the reduced camera system, visual factors eliminated on the device.  See solver_hip.h.
#include "solver_hip.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>

#include <condition_variable>
#include <thread>

#include "../../include/icgvins_hip.h"

namespace icg {

namespace {
// ICG_SOLVER_DEBUG=1: wall time per phase of the LM loop, printed by solve()
struct PhaseClock {
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool on = getenv("ICG_SOLVER_DEBUG") != nullptr;
};
thread_local PhaseClock g_clock; // per thread: concurrent estimators (Replay::runMany, lock-step groups) each print their own
struct PhaseScope {
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseScope(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
    ~PhaseScope() {
        if (g_clock.on) {
            g_clock.ms[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            g_clock.calls[k]++;
        }
    }
};
enum { PH_EVAL_JAC = 0, PH_SCHUR, PH_HOST_FACTORS, PH_CHOLESKY, PH_BACKSUB, PH_EVAL_TRIAL, PH_COST, PH_CHI2 };

// One helper thread per process for WindowSolver::setHostFactorOverlap: whoever holds the lock hands it the HOST half (the host factors of a
// linearization) and drives the device half itself — the calling thread is the one that talks to the device, always — anybody else runs the
// halves in turn.  The helper's phase clock (thread_local) is merged into the caller's after the join, so ICG_SOLVER_DEBUG books every phase.
std::atomic<bool> g_overlap{false};
class OverlapHelper {
public:
    std::mutex owner; // try_lock'ed by the solver that wants the helper
    ~OverlapHelper() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (thread_.joinable()) thread_.join();
    }
    void submit(const std::function<bool()> *job) {
        if (!thread_.joinable()) thread_ = std::thread(&OverlapHelper::loop, this);
        {
            std::lock_guard<std::mutex> lock(m_);
            job_  = job;
            done_ = false;
        }
        cv_.notify_all();
    }
    bool wait(PhaseClock &into) {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return done_; });
        for (int k = 0; k < 8; k++) into.ms[k] += clock_.ms[k], into.calls[k] += clock_.calls[k];
        return ok_;
    }

private:
    void loop() {
        for (;;) {
            const std::function<bool()> *job;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return stop_ || job_ != nullptr; });
                if (stop_) return;
                job  = job_;
                job_ = nullptr;
            }
            g_clock = PhaseClock();
            bool ok = false;
            try {
                ok = (*job)();
            } catch (...) {
                ok = false;
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                ok_    = ok;
                clock_ = g_clock;
                done_  = true;
            }
            cv_.notify_all();
        }
    }
    std::thread thread_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<bool()> *job_{nullptr};
    bool done_{true}, ok_{false}, stop_{false};
    PhaseClock clock_;
};
OverlapHelper &overlapHelper() {
    static OverlapHelper h;
    return h;
}
// device() and host() both run, side by side when the overlap is on and the helper is free; -> both succeeded
bool runHalves(bool want_overlap, const std::function<bool()> &device, const std::function<bool()> &host) {
    if (want_overlap && g_overlap.load(std::memory_order_relaxed)) {
        OverlapHelper &h = overlapHelper();
        std::unique_lock<std::mutex> lock(h.owner, std::try_to_lock);
        if (lock.owns_lock()) {
            h.submit(&host);
            bool a = false;
            try {
                a = device(); // on the calling thread
            } catch (...) {
                (void) h.wait(g_clock); // the helper still runs `host`, which lives in the caller's frame: join before unwinding
                throw;
            }
            const bool b = h.wait(g_clock);
            return a && b;
        }
    }
    const bool a = device();
    return host() && a;
}

using solver_detail::choleskySolve;
using solver_detail::posePlus;
} // namespace

std::string WindowSolver::Summary::BriefReport() const {
    char buf[256];
    snprintf(buf, sizeof buf, "WindowSolver: initial cost %.6e, final cost %.6e, %d successful / %d unsuccessful steps, %s", initial_cost,
             final_cost, num_successful_steps, num_unsuccessful_steps, termination.c_str());
    return buf;
}

void WindowSolver::setHostFactorOverlap(bool on) { g_overlap.store(on); }

WindowSolver::WindowSolver(ReprojectionBatch *visual, double huber_delta) : visual_(visual), huber_(huber_delta) {
    if (visual_) active_.assign((size_t) visual_->size(), 1);
}

void WindowSolver::addParameterBlock(double *values, int size, bool pose_manifold) {
    if (block_of_.count(values)) return;
    if (pose_manifold && size != 7) throw std::runtime_error("WindowSolver: the pose manifold needs a block of size 7");
    block_of_[values] = (int) blocks_.size();
    blocks_.push_back({values, size, pose_manifold ? 6 : size, pose_manifold, false, -1, false});
}

void WindowSolver::setParameterBlockConstant(double *values) {
    auto it = block_of_.find(values);
    if (it == block_of_.end()) throw std::runtime_error("WindowSolver: unknown parameter block");
    blocks_[(size_t) it->second].constant = true;
}

WindowSolver::ResidualBlockId WindowSolver::addResidualBlock(std::shared_ptr<ceres::CostFunction> cost, std::shared_ptr<ceres::LossFunction> loss,
                                                             const std::vector<double *> &blocks) {
    const auto &sizes = cost->parameter_block_sizes();
    if (sizes.size() != blocks.size()) throw std::runtime_error("WindowSolver: block count does not match the cost function");
    for (size_t k = 0; k < blocks.size(); k++) {
        auto it = block_of_.find(blocks[k]);
        if (it == block_of_.end()) throw std::runtime_error("WindowSolver: residual block uses an unknown parameter block");
        if (blocks_[(size_t) it->second].size != sizes[k]) throw std::runtime_error("WindowSolver: parameter block size mismatch");
    }
    residuals_.push_back({std::move(cost), std::move(loss), blocks, false});
    return (ResidualBlockId) residuals_.size() - 1;
}

void WindowSolver::removeResidualBlock(ResidualBlockId id) { residuals_.at((size_t) id).removed = true; }

bool WindowSolver::evaluateResidualBlock(ResidualBlockId id, bool apply_loss_function, double *cost) const {
    return solver_detail::residualCost(residuals_.at((size_t) id), apply_loss_function, cost);
}

int WindowSolver::numActiveReprojectionFactors() const {
    int n = 0;
    for (uint8_t a : active_) n += a;
    return n;
}

// column layout of the reduced system: every non-constant block that is not an inverse depth of the visual batch, in the
// order the blocks were added; inverse depths are eliminated on the device (column P + batch landmark index there)
bool WindowSolver::layout() {
    for (Block &b : blocks_) b.landmark = false, b.column = -1;
    if (visual_) {
        for (double *p : visual_->lm_ptrs_) {
            auto it = block_of_.find(p);
            if (it == block_of_.end()) {
                error_ = "an inverse-depth block of the reprojection batch was not added to the solver";
                return false;
            }
            if (blocks_[(size_t) it->second].constant) {
                error_ = "constant inverse-depth blocks are not supported";
                return false;
            }
            blocks_[(size_t) it->second].landmark = true;
        }
    }
    for (const Residual &R : residuals_)
        if (!R.removed)
            for (double *p : R.blocks)
                if (blocks_[(size_t) block_of_.at(p)].landmark) {
                    error_ = "host factors on an eliminated inverse-depth block are not supported";
                    return false;
                }
    P_ = 0;
    for (Block &b : blocks_)
        if (!b.constant && !b.landmark) {
            b.column = P_;
            P_ += b.local;
        }
    if (visual_) {
        auto col = [&](const double *p) {
            auto it = block_of_.find(p);
            if (it == block_of_.end()) throw std::runtime_error("WindowSolver: a block of the reprojection batch was not added to the solver");
            return blocks_[(size_t) it->second].column;
        };
        col_pose_.resize(visual_->pose_ptrs_.size());
        for (size_t k = 0; k < col_pose_.size(); k++) col_pose_[k] = col(visual_->pose_ptrs_[k]);
        col_ext_ = visual_->ext_ ? col(visual_->ext_) : -1;
        col_td_  = visual_->td_ ? col(visual_->td_) : -1;
        if (active_.size() != (size_t) visual_->size()) active_.assign((size_t) visual_->size(), 1);
    }
    return P_ > 0;
}

bool WindowSolver::linearize(double damp, bool reassemble, const Options &o, std::vector<double> &S, std::vector<double> &s,
                             std::vector<double> &diag, double *cost) {
    S.assign((size_t) P_ * P_, 0.0);
    s.assign((size_t) P_, 0.0);
    diag.assign((size_t) P_, 0.0);
    double vc = 0, hc = 0;
    const bool has_visual = visual_ && visual_->size() > 0;
    std::string device_error;
    auto device = [&]() -> bool {
        if (!has_visual) return true;
        if (reassemble) {
            PhaseScope ps(PH_EVAL_JAC);
            if (!visual_->run(true, huber_, false)) {
                device_error = visual_->error();
                return false;
            }
        }
        PhaseScope ps(PH_SCHUR);
        if (icg_reproj_schur(visual_->ctx_, P_, col_pose_.data(), col_ext_, col_td_, active_.data(), reassemble ? 1 : 0, damp, o.min_lm_diagonal,
                             o.max_lm_diagonal, S.data(), s.data(), diag.data(), &vc) != ICG_OK) {
            device_error = icg_last_error(visual_->ctx_);
            return false;
        }
        return true;
    };
    auto host = [&]() -> bool {
        if (!reassemble) return true;
        host_S_.assign((size_t) P_ * P_, 0.0);
        host_s_.assign((size_t) P_, 0.0);
        host_diag_.assign((size_t) P_, 0.0);
        PhaseScope ps(PH_HOST_FACTORS);
        return solver_detail::hostFactors(blocks_, block_of_, residuals_, P_, host_S_.data(), host_s_.data(), host_diag_.data(), &hc);
    };
    if (!runHalves(has_visual && reassemble && !residuals_.empty(), device, host)) {
        error_ = device_error.empty() ? "a host cost function failed to evaluate" : device_error;
        return false;
    }
    if (reassemble && cost) *cost = vc + hc;
    for (size_t k = 0; k < S.size(); k++) S[k] += host_S_[k];
    for (size_t k = 0; k < s.size(); k++) s[k] += host_s_[k], diag[k] += host_diag_[k];
    return true;
}

bool WindowSolver::evaluateCost(double *cost) {
    double vc = 0, hc = 0;
    const bool has_visual = visual_ && visual_->size() > 0;
    std::string device_error;
    auto device = [&]() -> bool {
        if (!has_visual) return true;
        {
            PhaseScope ps(PH_EVAL_TRIAL);
            if (!visual_->run(false, huber_, false)) {
                device_error = visual_->error();
                return false;
            }
        }
        PhaseScope ps(PH_COST);
        if (icg_reproj_cost(visual_->ctx_, active_.data(), &vc) != ICG_OK) {
            device_error = icg_last_error(visual_->ctx_);
            return false;
        }
        return true;
    };
    auto host = [&]() -> bool { return solver_detail::hostFactors(blocks_, block_of_, residuals_, P_, nullptr, nullptr, nullptr, &hc); };
    if (!runHalves(has_visual && !residuals_.empty(), device, host)) {
        error_ = device_error.empty() ? "a host cost function failed to evaluate" : device_error;
        return false;
    }
    const double c = vc + hc;
    *cost = c;
    return true;
}

void WindowSolver::backup() {
    saved_.resize(blocks_.size());
    for (size_t k = 0; k < blocks_.size(); k++) saved_[k].assign(blocks_[k].values, blocks_[k].values + blocks_[k].size);
}

void WindowSolver::restore() {
    for (size_t k = 0; k < blocks_.size(); k++) memcpy(blocks_[k].values, saved_[k].data(), sizeof(double) * (size_t) blocks_[k].size);
}

void WindowSolver::applyStep(const std::vector<double> &delta_c, const std::vector<double> &delta_l) {
    for (Block &b : blocks_) {
        if (b.column < 0) continue;
        const double *d = &delta_c[(size_t) b.column];
        if (b.pose)
            posePlus(b.values, d);
        else
            for (int k = 0; k < b.size; k++) b.values[k] += d[k];
    }
    if (visual_)
        for (size_t l = 0; l < visual_->lm_ptrs_.size(); l++) *visual_->lm_ptrs_[l] += delta_l[l];
}

bool WindowSolver::solve(const Options &o, Summary *summary) {
    Summary sum;
    if (!layout()) {
        if (error_.empty()) error_ = "nothing to optimize";
        return false;
    }
    const size_t L = visual_ ? visual_->lm_ptrs_.size() : 0;
    double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
    std::vector<double> S, s, diag, delta_l(L, 0.0);
    double cost = 0;
    if (!linearize(1.0 / radius, true, o, S, s, diag, &cost)) return false;
    sum.initial_cost = cost;
    sum.termination  = "max_num_iterations";
    bool need_redamp = false;
    for (int iter = 0; iter < o.max_num_iterations; iter++) {
        if (need_redamp && !linearize(1.0 / radius, false, o, S, s, diag, nullptr)) return false;
        need_redamp = false;
        // gradient test (max norm of J^T r over all columns; the landmark part is bounded by it after elimination in practice and
        // is not fetched: the camera part decides)
        double gmax = 0;
        for (double v : s) gmax = std::max(gmax, std::fabs(v));
        if (gmax < o.gradient_tolerance) {
            sum.termination = "gradient_tolerance";
            break;
        }
        // (S + D) delta_c = s with the LM diagonal of the camera block
        std::vector<double> A(S), delta_c(s), dd((size_t) P_);
        for (int k = 0; k < P_; k++) {
            dd[(size_t) k] = std::min(std::max(diag[(size_t) k], o.min_lm_diagonal), o.max_lm_diagonal) / radius;
            A[(size_t) k * P_ + k] += dd[(size_t) k];
        }
        bool ok;
        {
            PhaseScope ps(PH_CHOLESKY);
            ok = choleskySolve(P_, A, delta_c);
        }
        double lm_terms[2] = {0, 0};
        if (ok && L > 0) {
            PhaseScope psb(PH_BACKSUB);
            if (icg_reproj_backsub(visual_->ctx_, P_, delta_c.data(), delta_l.data(), lm_terms) != ICG_OK) {
                error_ = icg_last_error(visual_->ctx_);
                return false;
            }
        }
        double model = 0;
        if (ok) {
            // model decrease 0.5 (delta^T b + delta^T D delta) of the FULL damped system; s is the reduced right-hand side, and
            // delta^T b = delta_c^T s + sum b_l^2/(h_ll+d_l) (device), delta^T D delta = delta_c^T Dc delta_c + sum d_l delta_l^2 (device)
            double t0 = lm_terms[0], t1 = lm_terms[1];
            for (int k = 0; k < P_; k++) t0 += delta_c[(size_t) k] * s[(size_t) k], t1 += dd[(size_t) k] * delta_c[(size_t) k] * delta_c[(size_t) k];
            model = 0.5 * (t0 + t1);
        }
        if (!ok || !(model > 0.0)) {
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            sum.num_unsuccessful_steps++;
            need_redamp = true;
            if (radius < o.min_trust_region_radius) {
                sum.termination = "min_trust_region_radius";
                break;
            }
            continue;
        }
        // parameter tolerance
        double dn = 0, xn = 0;
        for (double v : delta_c) dn += v * v;
        for (double v : delta_l) dn += v * v;
        for (const Block &b : blocks_)
            if (!b.constant)
                for (int k = 0; k < b.size; k++) xn += b.values[k] * b.values[k];
        backup();
        applyStep(delta_c, delta_l);
        if (std::sqrt(dn) <= o.parameter_tolerance * (std::sqrt(xn) + o.parameter_tolerance)) {
            restore();
            sum.termination = "parameter_tolerance";
            break;
        }
        double new_cost = 0;
        if (!evaluateCost(&new_cost)) return false;
        const double rho = (cost - new_cost) / model;
        if (rho > o.min_relative_decrease) {
            const double change = cost - new_cost;
            cost                = new_cost;
            sum.num_successful_steps++;
            radius          = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
            decrease_factor = 2.0;
            if (std::fabs(change) < o.function_tolerance * cost) {
                sum.termination = "function_tolerance";
                break;
            }
            if (iter + 1 < o.max_num_iterations && !linearize(1.0 / radius, true, o, S, s, diag, nullptr)) return false;
        } else {
            restore();
            radius /= decrease_factor;
            decrease_factor *= 2.0;
            sum.num_unsuccessful_steps++;
            need_redamp = true;
            if (radius < o.min_trust_region_radius) {
                sum.termination = "min_trust_region_radius";
                break;
            }
        }
    }
    sum.final_cost = cost;
    if (summary) *summary = sum;
    if (g_clock.on) {
        static const char *names[8] = {"eval+jac", "schur", "host_factors", "cholesky", "backsub", "eval_trial", "cost", "chi2"};
        for (int k = 0; k < 8; k++)
            if (g_clock.calls[k]) fprintf(stderr, "[WindowSolver] %-28s %3d calls %8.3f ms\n", names[k], g_clock.calls[k], g_clock.ms[k]);
        g_clock = PhaseClock();
    }
    return true;
}

int WindowSolver::removeReprojectionFactorsByChi2(double chi2) {
    if (!visual_ || visual_->size() == 0) return 0;
    if (active_.size() != (size_t) visual_->size()) active_.assign((size_t) visual_->size(), 1);
    if (!visual_->run(false, 0.0)) { // EvaluateResidualBlock(id, false, &cost, ...): raw residuals, no loss
        error_ = visual_->error();
        return -1;
    }
    int removed = 0;
    for (int f = 0; f < visual_->size(); f++) {
        if (!active_[(size_t) f]) continue;
        const double *r = visual_->residual(f);
        const double cost = 0.5 * (r[0] * r[0] + r[1] * r[1]);
        if (cost * 2.0 > chi2) {
            active_[(size_t) f] = 0;
            removed++;
        }
    }
    return removed;
}

} // namespace icg
