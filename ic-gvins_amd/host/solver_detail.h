// Pieces shared by WindowSolver (solver_hip.cc) and WindowSolverBatch (solver_batch_hip.cc): the reduced-system solve, the pose
// manifold step and the accumulation of host-evaluated factors into a window's reduced system.
#pragma once
#include <cmath>
#include <memory>
#include <unordered_map>
#include <vector>

#include "factors.h"

namespace icg {
namespace solver_detail {

struct Block {
    double *values;
    int size, local;
    bool pose, constant;
    int column; // in the reduced (camera) system, -1 for constants and for the eliminated inverse-depth blocks
    bool landmark;
};
struct Residual {
    std::shared_ptr<ceres::CostFunction> cost;
    std::shared_ptr<ceres::LossFunction> loss;
    std::vector<double *> blocks;
    bool removed;
};

// in-place Cholesky solve of the symmetric positive definite n x n system A x = b (row-major, lower triangle used); dense_kernels.cc
bool choleskySolve(int n, std::vector<double> &A, std::vector<double> &b);
// T (nf x nf, upper triangle) += J^T J, g (nf) += J^T r for a dense row-major J (nr x nf); dense_kernels.cc
void accumulateJtJ(int nr, int nf, const double *J, const double *r, double *T, double *g);

// PoseParameterization::Plus (factors/pose_parameterization.h:34-50): p += dp, q = (q * rotvec2quaternion(dtheta)).normalized()
inline void posePlus(double *x, const double *delta) {
    for (int k = 0; k < 3; k++) x[k] += delta[k];
    const double rx = delta[3], ry = delta[4], rz = delta[5];
    const double angle = std::sqrt(rx * rx + ry * ry + rz * rz);
    double ax = rx, ay = ry, az = rz;
    if (angle > 0) ax /= angle, ay /= angle, az /= angle;
    const double sh = std::sin(0.5 * angle), ch = std::cos(0.5 * angle);
    const double bx = sh * ax, by = sh * ay, bz = sh * az, bw = ch;
    const double qx = x[3], qy = x[4], qz = x[5], qw = x[6];
    double nx = qw * bx + qx * bw + qy * bz - qz * by;
    double ny = qw * by + qy * bw + qz * bx - qx * bz;
    double nz = qw * bz + qz * bw + qx * by - qy * bx;
    double nw = qw * bw - qx * bx - qy * by - qz * bz;
    const double nn = std::sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    x[3] = nx / nn, x[4] = ny / nn, x[5] = nz / nn, x[6] = nw / nn;
}

// cost of one host residual block: 0.5 rho(|r|^2) (apply_loss) or 0.5 |r|^2
inline bool residualCost(const Residual &R, bool apply_loss_function, double *cost) {
    std::vector<double> r((size_t) R.cost->num_residuals());
    if (!R.cost->Evaluate(R.blocks.data(), r.data(), nullptr)) return false;
    double s = 0;
    for (double v : r) s += v * v;
    if (apply_loss_function && R.loss) {
        double rho[3];
        R.loss->Evaluate(s, rho);
        s = rho[0];
    }
    *cost = 0.5 * s;
    return true;
}

// host factors of one window: S += J^T J, s -= J^T r (robust-corrected), diag, cost += 0.5 rho(|r|^2); S == nullptr: cost only.
// S is P x P with row stride P.
inline bool hostFactors(const std::vector<Block> &blocks, const std::unordered_map<const double *, int> &block_of, const std::vector<Residual> &residuals,
                        int P, double *S, double *s, double *diag, double *cost) {
    for (const Residual &R : residuals) {
        if (R.removed) continue;
        if (!S) {
            double c;
            if (!residualCost(R, true, &c)) return false;
            *cost += c;
            continue;
        }
        ResidualBlockInfo info(R.cost, nullptr, R.blocks, {});
        if (!info.Evaluate()) return false;
        double sq = 0;
        for (double v : info.residuals()) sq += v * v;
        if (R.loss) { // cost from the raw residual, then the Ceres corrector (residual_block_info.h:59-87)
            double rho[3];
            R.loss->Evaluate(sq, rho);
            *cost += 0.5 * rho[0];
            ResidualBlockInfo corrected(R.cost, R.loss, R.blocks, {});
            if (!corrected.Evaluate()) return false;
            info = corrected;
        } else {
            *cost += 0.5 * sq;
        }
        // J^T J of the factor's free columns: the blocks are gathered into ONE dense row-major Jacobian (nr x n_free) so that the triple
        // loop runs with the residual index outermost and a contiguous, independent inner index (vectorizable as written); every cell is
        // the sum over k in ascending order from zero, as in a cell-by-cell inner product, and is then added to S once.
        const int nr      = R.cost->num_residuals();
        const auto &sizes = R.cost->parameter_block_sizes();
        thread_local std::vector<double> Jd, T;
        thread_local std::vector<int> cols;
        cols.clear();
        for (size_t a = 0; a < R.blocks.size(); a++) {
            const Block &A = blocks[(size_t) block_of.at(R.blocks[a])];
            if (A.column < 0) continue;
            for (int x = 0; x < A.local; x++) cols.push_back(A.column + x);
        }
        const int nf = (int) cols.size();
        if (nf == 0) continue;
        Jd.resize((size_t) nr * nf);
        {
            int c0 = 0;
            for (size_t a = 0; a < R.blocks.size(); a++) {
                const Block &A = blocks[(size_t) block_of.at(R.blocks[a])];
                if (A.column < 0) continue;
                const std::vector<double> &Ja = info.jacobians()[a];
                for (int k = 0; k < nr; k++)
                    for (int x = 0; x < A.local; x++) Jd[(size_t) k * nf + c0 + x] = Ja[(size_t) k * sizes[a] + x];
                c0 += A.local;
            }
        }
        const std::vector<double> &res = info.residuals();
        T.assign((size_t) nf * nf + nf, 0.0);
        double *g = T.data() + (size_t) nf * nf;
        accumulateJtJ(nr, nf, Jd.data(), res.data(), T.data(), g);
        for (int x = 0; x < nf; x++) {
            s[(size_t) cols[(size_t) x]] -= g[x];
            diag[(size_t) cols[(size_t) x]] += T[(size_t) x * nf + x];
            S[(size_t) cols[(size_t) x] * P + cols[(size_t) x]] += T[(size_t) x * nf + x];
            for (int y = x + 1; y < nf; y++) {
                const double v = T[(size_t) x * nf + y];
                S[(size_t) cols[(size_t) x] * P + cols[(size_t) y]] += v;
                S[(size_t) cols[(size_t) y] * P + cols[(size_t) x]] += v;
            }
        }
    }
    return true;
}

} // namespace solver_detail
} // namespace icg
