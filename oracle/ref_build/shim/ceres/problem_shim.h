// ORACLE / TEST INFRASTRUCTURE ONLY — the slice of ceres::Problem / ceres::Solver the reference's estimator (ic_gvins.cc) is written against,
// so that ic_gvins.cc compiles UNMODIFIED into oracle/_ref/libref_gvins.so.  Ceres itself is an absent, un-vendored dependency; what is
// restated here is its published trust-region loop (Levenberg-Marquardt diagonal clamp(diag(J^T J), 1e-6, 1e32) / radius, step acceptance
// on rho > 1e-3, radius update radius / max(1/3, 1 - (2 rho - 1)^3) or halving with a doubling factor, function / gradient / parameter
// tolerances 1e-6 / 1e-10 / 1e-8, the robust-loss corrector of corrector.cc) on dense normal equations, with the 1-dimensional blocks that
// never share a residual (the inverse depths) eliminated first — the same algorithm, written independently of, and sharing no code with,
// the product's WindowSolver.  No Jacobi column scaling.  Ownership as in Ceres: the problem deletes cost / loss functions and
// parameterizations it was given.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace ceres {

class LocalParameterization {
public:
    virtual ~LocalParameterization() = default;
    virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double *x, double *jacobian) const                = 0; // row-major GlobalSize x LocalSize
    virtual int GlobalSize() const                                                       = 0;
    virtual int LocalSize() const                                                        = 0;
};

enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

struct ResidualBlockShim {
    CostFunction *cost;
    LossFunction *loss;
    std::vector<double *> blocks;
    bool removed;
};
typedef ResidualBlockShim *ResidualBlockId;

class Problem {
public:
    struct Options {
        bool enable_fast_removal = false;
        EvaluationCallback *evaluation_callback = nullptr; // problem.h: not owned
    };
    Problem() = default;
    explicit Problem(const Options &o) : options_(o) {}
    EvaluationCallback *evaluation_callback() const { return options_.evaluation_callback; }
    Problem(const Problem &)            = delete;
    Problem &operator=(const Problem &) = delete;
    ~Problem() {
        std::set<CostFunction *> costs;
        std::set<LossFunction *> losses;
        std::set<LocalParameterization *> params;
        for (auto *r : residuals_) {
            costs.insert(r->cost);
            if (r->loss) losses.insert(r->loss);
            delete r;
        }
        for (auto &b : blocks_)
            if (b.second.parameterization) params.insert(b.second.parameterization);
        for (auto *c : costs) delete c;
        for (auto *l : losses) delete l;
        for (auto *p : params) delete p;
    }
    void AddParameterBlock(double *values, int size, LocalParameterization *parameterization = nullptr) {
        auto it = blocks_.find(values);
        if (it != blocks_.end()) {
            if (parameterization && it->second.parameterization != parameterization) delete parameterization;
            return;
        }
        Block b;
        b.size = size, b.parameterization = parameterization, b.constant = false, b.order = (int) blocks_.size();
        blocks_[values] = b;
    }
    void SetParameterBlockConstant(double *values) { blocks_.at(values).constant = true; }
    ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &blocks) {
        const auto &sizes = cost->parameter_block_sizes();
        for (size_t k = 0; k < blocks.size(); k++) AddParameterBlock(blocks[k], sizes[k]);
        auto *r = new ResidualBlockShim{cost, loss, blocks, false};
        residuals_.push_back(r);
        return r;
    }
    template <typename... Ts> ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, double *x0, Ts *...xs) {
        return AddResidualBlock(cost, loss, std::vector<double *>{x0, xs...});
    }
    void RemoveResidualBlock(ResidualBlockId id) { id->removed = true; }
    // problem.h: "if an EvaluationCallback is associated with the problem, its PrepareForEvaluation method is called every time this method
    // is called, with new_point = true"
    bool EvaluateResidualBlock(ResidualBlockId id, bool apply_loss_function, double *cost, double *residuals, double **jacobians) const {
        if (options_.evaluation_callback) options_.evaluation_callback->PrepareForEvaluation(jacobians != nullptr, true);
        return EvaluateResidualBlockAssumingParametersUnchanged(id, apply_loss_function, cost, residuals, jacobians);
    }
    bool EvaluateResidualBlockAssumingParametersUnchanged(ResidualBlockId id, bool apply_loss_function, double *cost, double *residuals,
                                                          double **jacobians) const {
        std::vector<double> r((size_t) id->cost->num_residuals());
        if (!id->cost->Evaluate(id->blocks.data(), r.data(), jacobians)) return false;
        double sq = 0;
        for (double v : r) sq += v * v;
        if (apply_loss_function && id->loss) {
            double rho[3];
            id->loss->Evaluate(sq, rho);
            sq = rho[0];
        }
        if (cost) *cost = 0.5 * sq;
        if (residuals) memcpy(residuals, r.data(), sizeof(double) * r.size());
        return true;
    }

private:
    friend class Solver;
    Options options_;
    struct Block {
        int size;
        LocalParameterization *parameterization;
        bool constant;
        int order;
    };
    std::unordered_map<double *, Block> blocks_;
    std::vector<ResidualBlockShim *> residuals_;
};

class Solver {
public:
    struct Options {
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
        LinearSolverType linear_solver_type                = DENSE_QR;
        int max_num_iterations                             = 50;
        int num_threads                                    = 1;
        double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
        double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
        double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    };
    struct Summary {
        double initial_cost = 0, final_cost = 0;
        int num_successful_steps = 0, num_unsuccessful_steps = 0;
        TerminationType termination_type = NO_CONVERGENCE;
        std::string message;
        std::string BriefReport() const {
            char buf[256];
            snprintf(buf, sizeof buf, "Ceres-shim Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
                     num_successful_steps + num_unsuccessful_steps, initial_cost, final_cost, termination_type == CONVERGENCE ? "CONVERGENCE" : "NO_CONVERGENCE");
            return buf;
        }
    };
    void Solve(const Options &o, Problem *problem, Summary *summary) { *summary = Run(o, *problem); }

private:
    struct Var { // one free parameter block
        double *values;
        int size, local;
        LocalParameterization *par;
        int column;   // in the dense (camera) system, or -1
        int landmark; // index among the eliminated 1-d blocks, or -1
    };
    struct System {
        int P = 0, L = 0;
        std::vector<double> Hcc, G, hll, bc, bl; // Hcc P x P, G L x P
        double cost = 0;
    };
    static void corrected(const ResidualBlockShim &R, std::vector<double> &r, std::vector<std::vector<double>> &J, double *cost) {
        // Ceres corrector (corrector.cc) as residual_block_info.h:59-87 applies it
        double sq = 0;
        for (double v : r) sq += v * v;
        if (!R.loss) {
            *cost += 0.5 * sq;
            return;
        }
        double rho[3];
        R.loss->Evaluate(sq, rho);
        *cost += 0.5 * rho[0];
        const double sqrt_rho1 = std::sqrt(rho[1]);
        double residual_scaling, alpha_sq_norm;
        if ((sq == 0.0) || (rho[2] <= 0.0)) {
            residual_scaling = sqrt_rho1;
            alpha_sq_norm    = 0.0;
        } else {
            const double D     = 1.0 + 2.0 * sq * rho[2] / rho[1];
            const double alpha = 1.0 - std::sqrt(D);
            residual_scaling   = sqrt_rho1 / (1 - alpha);
            alpha_sq_norm      = alpha / sq;
        }
        const size_t nr = r.size();
        for (auto &Jb : J) {
            if (Jb.empty()) continue;
            const size_t nc = Jb.size() / nr;
            std::vector<double> rtJ(nc, 0.0);
            for (size_t k = 0; k < nr; k++)
                for (size_t c = 0; c < nc; c++) rtJ[c] += r[k] * Jb[k * nc + c];
            for (size_t k = 0; k < nr; k++)
                for (size_t c = 0; c < nc; c++) Jb[k * nc + c] = sqrt_rho1 * (Jb[k * nc + c] - alpha_sq_norm * r[k] * rtJ[c]);
        }
        for (double &v : r) v *= residual_scaling;
    }
    static bool evaluateCost(Problem &p, double *cost) {
        double c = 0;
        if (p.evaluation_callback()) p.evaluation_callback()->PrepareForEvaluation(false, true); // once per evaluation point
        for (auto *R : p.residuals_) {
            if (R->removed) continue;
            double rc;
            if (!p.EvaluateResidualBlockAssumingParametersUnchanged(R, true, &rc, nullptr, nullptr)) return false;
            c += rc;
        }
        *cost = c;
        return true;
    }
    static bool linearize(Problem &p, const std::unordered_map<double *, int> &var_of, const std::vector<Var> &vars, System &S) {
        const int P = S.P, L = S.L;
        S.Hcc.assign((size_t) P * P, 0.0), S.G.assign((size_t) L * P, 0.0), S.hll.assign((size_t) L, 0.0), S.bc.assign((size_t) P, 0.0), S.bl.assign((size_t) L, 0.0);
        S.cost = 0;
        if (p.evaluation_callback()) p.evaluation_callback()->PrepareForEvaluation(true, true); // once per evaluation point
        for (auto *R : p.residuals_) {
            if (R->removed) continue;
            const int nr      = R->cost->num_residuals();
            const auto &sizes = R->cost->parameter_block_sizes();
            std::vector<double> r((size_t) nr);
            std::vector<std::vector<double>> Jg(R->blocks.size()), Jl(R->blocks.size());
            std::vector<double *> jp(R->blocks.size());
            for (size_t b = 0; b < R->blocks.size(); b++) {
                Jg[b].assign((size_t) nr * sizes[b], 0.0);
                jp[b] = Jg[b].data();
            }
            if (!R->cost->Evaluate(R->blocks.data(), r.data(), jp.data())) return false;
            // local Jacobians of the free blocks
            std::vector<const Var *> vb(R->blocks.size(), nullptr);
            for (size_t b = 0; b < R->blocks.size(); b++) {
                auto it = var_of.find(R->blocks[b]);
                if (it == var_of.end()) continue;
                const Var &v = vars[(size_t) it->second];
                vb[b]        = &v;
                if (v.par) {
                    std::vector<double> Jp((size_t) v.size * v.local);
                    v.par->ComputeJacobian(v.values, Jp.data());
                    Jl[b].assign((size_t) nr * v.local, 0.0);
                    for (int k = 0; k < nr; k++)
                        for (int c = 0; c < v.local; c++) {
                            double s = 0;
                            for (int g = 0; g < v.size; g++) s += Jg[b][(size_t) k * v.size + g] * Jp[(size_t) g * v.local + c];
                            Jl[b][(size_t) k * v.local + c] = s;
                        }
                } else {
                    Jl[b] = Jg[b];
                }
            }
            corrected(*R, r, Jl, &S.cost);
            for (size_t a = 0; a < R->blocks.size(); a++) {
                if (!vb[a]) continue;
                const Var &A = *vb[a];
                for (int x = 0; x < A.local; x++) { // gradient
                    double g = 0;
                    for (int k = 0; k < nr; k++) g += Jl[a][(size_t) k * A.local + x] * r[(size_t) k];
                    if (A.landmark >= 0)
                        S.bl[(size_t) A.landmark] -= g;
                    else
                        S.bc[(size_t) (A.column + x)] -= g;
                }
                for (size_t c = 0; c < R->blocks.size(); c++) {
                    if (!vb[c]) continue;
                    const Var &B = *vb[c];
                    for (int x = 0; x < A.local; x++)
                        for (int y = 0; y < B.local; y++) {
                            double v = 0;
                            for (int k = 0; k < nr; k++) v += Jl[a][(size_t) k * A.local + x] * Jl[c][(size_t) k * B.local + y];
                            if (A.landmark >= 0 && B.landmark >= 0)
                                S.hll[(size_t) A.landmark] += v; // same landmark by construction of the elimination set
                            else if (A.landmark >= 0)
                                S.G[(size_t) A.landmark * P + B.column + y] += v;
                            else if (B.landmark < 0)
                                S.Hcc[(size_t) (A.column + x) * P + B.column + y] += v;
                        }
                }
            }
        }
        return true;
    }
    static bool cholesky(int n, std::vector<double> &A, std::vector<double> &b) {
        for (int j = 0; j < n; j++) {
            double d = A[(size_t) j * n + j];
            for (int k = 0; k < j; k++) d -= A[(size_t) j * n + k] * A[(size_t) j * n + k];
            if (!(d > 0.0) || !std::isfinite(d)) return false;
            d                     = std::sqrt(d);
            A[(size_t) j * n + j] = d;
            for (int i = j + 1; i < n; i++) {
                double v = A[(size_t) i * n + j];
                for (int k = 0; k < j; k++) v -= A[(size_t) i * n + k] * A[(size_t) j * n + k];
                A[(size_t) i * n + j] = v / d;
            }
        }
        for (int i = 0; i < n; i++) {
            double v = b[(size_t) i];
            for (int k = 0; k < i; k++) v -= A[(size_t) i * n + k] * b[(size_t) k];
            b[(size_t) i] = v / A[(size_t) i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double v = b[(size_t) i];
            for (int k = i + 1; k < n; k++) v -= A[(size_t) k * n + i] * b[(size_t) k];
            b[(size_t) i] = v / A[(size_t) i * n + i];
        }
        return true;
    }
    static Summary Run(const Options &o, Problem &p) {
        Summary sum;
        // free blocks that appear in an active residual, in the order they were added
        std::map<int, double *> ordered;
        std::unordered_map<double *, int> uses;
        for (auto *R : p.residuals_)
            if (!R->removed)
                for (double *b : R->blocks) {
                    const auto &B = p.blocks_.at(b);
                    if (!B.constant) ordered[B.order] = b, uses[b]++;
                }
        // elimination set: free 1-d blocks no two of which share a residual (of a clashing pair the more widely used one stays dense)
        std::set<double *> eliminated;
        for (auto &kv : ordered)
            if (p.blocks_.at(kv.second).size == 1) eliminated.insert(kv.second);
        for (auto *R : p.residuals_) {
            if (R->removed) continue;
            std::vector<double *> in;
            for (double *b : R->blocks)
                if (eliminated.count(b)) in.push_back(b);
            while (in.size() > 1) {
                size_t worst = 0;
                for (size_t k = 1; k < in.size(); k++)
                    if (uses[in[k]] > uses[in[worst]]) worst = k;
                eliminated.erase(in[worst]);
                in.erase(in.begin() + (long) worst);
            }
        }
        std::vector<Var> vars;
        std::unordered_map<double *, int> var_of;
        System S;
        for (auto &kv : ordered) {
            const auto &B = p.blocks_.at(kv.second);
            Var v;
            v.values = kv.second, v.size = B.size, v.par = B.parameterization, v.local = B.parameterization ? B.parameterization->LocalSize() : B.size;
            if (eliminated.count(kv.second)) {
                v.column = -1, v.landmark = S.L++;
            } else {
                v.column = S.P, v.landmark = -1;
                S.P += v.local;
            }
            var_of[kv.second] = (int) vars.size();
            vars.push_back(v);
        }
        const int P = S.P, L = S.L;
        if (P + L == 0) {
            sum.termination_type = CONVERGENCE;
            return sum;
        }
        double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
        if (!linearize(p, var_of, vars, S)) {
            sum.termination_type = FAILURE;
            return sum;
        }
        double cost      = S.cost;
        sum.initial_cost = cost;
        auto clampd      = [&](double v) { return std::min(std::max(v, o.min_lm_diagonal), o.max_lm_diagonal); };
        for (int iter = 0; iter < o.max_num_iterations; iter++) {
            double gmax = 0;
            for (double v : S.bc) gmax = std::max(gmax, std::fabs(v));
            for (double v : S.bl) gmax = std::max(gmax, std::fabs(v));
            if (gmax < o.gradient_tolerance) {
                sum.termination_type = CONVERGENCE;
                break;
            }
            std::vector<double> A(S.Hcc), dc(S.bc), dd((size_t) P), dl((size_t) L), inv((size_t) L), delta_l((size_t) L, 0.0);
            for (int l = 0; l < L; l++) {
                dl[(size_t) l]  = clampd(S.hll[(size_t) l]) / radius;
                inv[(size_t) l] = 1.0 / (S.hll[(size_t) l] + dl[(size_t) l]);
            }
            for (int k = 0; k < P; k++) {
                dd[(size_t) k] = clampd(S.Hcc[(size_t) k * P + k]) / radius;
                A[(size_t) k * P + k] += dd[(size_t) k];
            }
            for (int l = 0; l < L; l++) { // Schur complement of the eliminated blocks
                const double *g = &S.G[(size_t) l * P];
                const double w  = inv[(size_t) l];
                for (int x = 0; x < P; x++) {
                    if (g[x] == 0.0) continue;
                    const double gx = g[x] * w;
                    dc[(size_t) x] -= gx * S.bl[(size_t) l];
                    for (int y = 0; y < P; y++)
                        if (g[y] != 0.0) A[(size_t) x * P + y] -= gx * g[y];
                }
            }
            std::vector<double> s_reduced(dc);
            bool ok = P == 0 || cholesky(P, A, dc);
            double model = 0;
            if (ok) {
                double t0 = 0, t1 = 0;
                for (int l = 0; l < L; l++) {
                    double gd = 0;
                    for (int x = 0; x < P; x++) gd += S.G[(size_t) l * P + x] * dc[(size_t) x];
                    delta_l[(size_t) l] = (S.bl[(size_t) l] - gd) * inv[(size_t) l];
                    t0 += delta_l[(size_t) l] * S.bl[(size_t) l];
                    t1 += dl[(size_t) l] * delta_l[(size_t) l] * delta_l[(size_t) l];
                }
                for (int k = 0; k < P; k++) t0 += dc[(size_t) k] * S.bc[(size_t) k], t1 += dd[(size_t) k] * dc[(size_t) k] * dc[(size_t) k];
                model = 0.5 * (t0 + t1);
            }
            if (!ok || !(model > 0.0)) {
                radius /= decrease_factor;
                decrease_factor *= 2.0;
                sum.num_unsuccessful_steps++;
                if (radius < o.min_trust_region_radius) break;
                continue;
            }
            double dn = 0, xn = 0;
            for (double v : dc) dn += v * v;
            for (double v : delta_l) dn += v * v;
            std::vector<std::vector<double>> saved(vars.size());
            for (size_t k = 0; k < vars.size(); k++) {
                const Var &v = vars[k];
                saved[k].assign(v.values, v.values + v.size);
                for (int c = 0; c < v.size; c++) xn += v.values[c] * v.values[c];
                if (v.landmark >= 0) {
                    v.values[0] += delta_l[(size_t) v.landmark];
                } else if (v.par) {
                    std::vector<double> out((size_t) v.size);
                    v.par->Plus(v.values, &dc[(size_t) v.column], out.data());
                    memcpy(v.values, out.data(), sizeof(double) * (size_t) v.size);
                } else {
                    for (int c = 0; c < v.size; c++) v.values[c] += dc[(size_t) (v.column + c)];
                }
            }
            auto restore = [&]() {
                for (size_t k = 0; k < vars.size(); k++) memcpy(vars[k].values, saved[k].data(), sizeof(double) * (size_t) vars[k].size);
            };
            if (std::sqrt(dn) <= o.parameter_tolerance * (std::sqrt(xn) + o.parameter_tolerance)) {
                restore();
                sum.termination_type = CONVERGENCE;
                break;
            }
            double new_cost = 0;
            if (!evaluateCost(p, &new_cost)) {
                restore();
                sum.termination_type = FAILURE;
                break;
            }
            const double rho = (cost - new_cost) / model;
            if (rho > o.min_relative_decrease) {
                const double change = cost - new_cost;
                cost                = new_cost;
                sum.num_successful_steps++;
                radius          = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
                decrease_factor = 2.0;
                if (std::fabs(change) < o.function_tolerance * cost) {
                    sum.termination_type = CONVERGENCE;
                    break;
                }
                if (iter + 1 < o.max_num_iterations && !linearize(p, var_of, vars, S)) {
                    sum.termination_type = FAILURE;
                    break;
                }
            } else {
                restore();
                radius /= decrease_factor;
                decrease_factor *= 2.0;
                sum.num_unsuccessful_steps++;
                if (radius < o.min_trust_region_radius) break;
            }
        }
        sum.final_cost = cost;
        return sum;
    }
};

} // namespace ceres
