// R1/R2: batched ReprojectionFactor::Evaluate (+ ResidualBlockInfo robust correction) on gfx950.
//
// Reference: factors/reprojection_factor.h:55-147 (residual + 5 Jacobian blocks),
//            factors/residual_block_info.h:59-87 (Huber corrector used by marginalization).
// Design (HBM-bound, FP64, no MFMA — SURVEY.md §8(d)): one lane per factor, observation constants read
// component-major (15 coalesced 512-B wave loads), shared parameter blocks gathered through L2, the 48 output
// doubles of each factor transposed through LDS (row stride 49 to spread banks) so a 64-factor wave writes its
// r[64x2] and J[64x46] slabs as contiguous 16-B-per-lane stores.
// Algorithmic bytes per factor with Jacobians: 120 (obs) + 12 (indices) + 384 (out) = 516 B.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "dev_math.h"
#include "icg_internal.h"

using namespace icgd;

#define RPJ_TILE 64
#define RPJ_LDS_STRIDE 49

struct rpj_args {
    int n;
    const double *obs;    // 15 x n
    const int32_t *idx_i; // n
    const int32_t *idx_j;
    const int32_t *idx_lm;
    const double *poses; // K x 7
    const double *ext;   // 7
    const double *invdepth;
    double td;
    int want_jac;
    double huber_delta;
    double *out_r; // n x 2
    double *out_J; // n x 46
    // many windows per launch (icg_reproj_eval_windows): factor k belongs to window win[k]; ext is W x 7, tdv has W entries
    const int32_t *win;
    const double *tdv;
};

__device__ __forceinline__ void put23(double *row0, double *row1, const double red[6], const m33 &m, int col0) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
        row0[col0 + j] = red[0] * m.a[0 * 3 + j] + red[1] * m.a[1 * 3 + j] + red[2] * m.a[2 * 3 + j];
        row1[col0 + j] = red[3] * m.a[0 * 3 + j] + red[4] * m.a[1 * 3 + j] + red[5] * m.a[2 * 3 + j];
    }
}

__global__ __launch_bounds__(RPJ_TILE) void k_reproj_eval(rpj_args A) {
    __shared__ double tile[RPJ_TILE * RPJ_LDS_STRIDE];
    const int lane = threadIdx.x;
    const int f0   = blockIdx.x * RPJ_TILE;
    const int k    = f0 + lane;
    double *o      = &tile[lane * RPJ_LDS_STRIDE]; // o[0..1] residual, o[2..47] Jacobians

    if (k < A.n) {
        const size_t n = (size_t) A.n;
        d3 pts0        = mk3(A.obs[0 * n + k], A.obs[1 * n + k], A.obs[2 * n + k]);
        d3 pts1        = mk3(A.obs[3 * n + k], A.obs[4 * n + k], A.obs[5 * n + k]);
        d3 vel0        = mk3(A.obs[6 * n + k], A.obs[7 * n + k], A.obs[8 * n + k]);
        d3 vel1        = mk3(A.obs[9 * n + k], A.obs[10 * n + k], A.obs[11 * n + k]);
        double td0 = A.obs[12 * n + k], td1 = A.obs[13 * n + k];
        double sinfo = 1.0 / A.obs[14 * n + k];

        const double *pi = A.poses + 7 * (size_t) A.idx_i[k];
        const double *pj = A.poses + 7 * (size_t) A.idx_j[k];
        d3 p0  = mk3(pi[0], pi[1], pi[2]);
        dq q0  = q_from_xyzw(pi + 3);
        d3 p1  = mk3(pj[0], pj[1], pj[2]);
        dq q1  = q_from_xyzw(pj + 3);
        const int wk      = A.win ? A.win[k] : 0;
        const double *ext = A.ext + 7 * (size_t) wk;
        d3 tic = mk3(ext[0], ext[1], ext[2]);
        dq qic = q_from_xyzw(ext + 3);
        double id0 = A.invdepth[A.idx_lm[k]];
        double td  = A.win ? A.tdv[wk] : A.td;

        d3 pts_0_td = sub(pts0, scl(td - td0, vel0));
        d3 pts_1_td = sub(pts1, scl(td - td1, vel1));
        d3 pts_c_0  = dvd(pts_0_td, id0);
        d3 pts_b_0  = add(q_rot(qic, pts_c_0), tic);
        d3 pts_n    = add(q_rot(q0, pts_b_0), p0);
        d3 pts_b_1  = q_rot(q_inv(q1), sub(pts_n, p1));
        d3 pts_1    = q_rot(q_inv(qic), sub(pts_b_1, tic));
        double d1   = pts_1.z;

        double r0 = sinfo * (pts_1.x / d1 - pts_1_td.x);
        double r1 = sinfo * (pts_1.y / d1 - pts_1_td.y);
        o[0]      = r0;
        o[1]      = r1;

        if (A.want_jac) {
            m33 cb0n = q_mat(q0);
            m33 cnb1 = m_T(q_mat(q1));
            m33 cbc  = m_T(q_mat(qic));
            double red[6];
            red[0] = sinfo * (1.0 / d1);
            red[1] = sinfo * 0.0;
            red[2] = sinfo * (-pts_1.x / (d1 * d1));
            red[3] = sinfo * 0.0;
            red[4] = sinfo * (1.0 / d1);
            red[5] = sinfo * (-pts_1.y / (d1 * d1));

            double *Ji0 = o + 2, *Ji1 = o + 2 + 7;
            double *Jj0 = o + 16, *Jj1 = o + 16 + 7;
            double *Je0 = o + 30, *Je1 = o + 30 + 7;

            m33 cbc_cnb1  = m_mul(cbc, cnb1);
            m33 ncbc_cnb1 = m_mul(m_neg(cbc), cnb1);
            // pose i
            put23(Ji0, Ji1, red, cbc_cnb1, 0);
            put23(Ji0, Ji1, red, m_mul(m_mul(ncbc_cnb1, cb0n), m_skew(pts_b_0)), 3);
            Ji0[6] = 0;
            Ji1[6] = 0;
            // pose j
            put23(Jj0, Jj1, red, ncbc_cnb1, 0);
            put23(Jj0, Jj1, red, m_mul(cbc, m_skew(pts_b_1)), 3);
            Jj0[6] = 0;
            Jj1[6] = 0;
            // extrinsic
            m33 tmp_r = m_mul(m_mul(cbc_cnb1, cb0n), m_T(cbc));
            put23(Je0, Je1, red, m_mul(cbc, m_sub(m_mul(cnb1, cb0n), m_eye())), 0);
            d3 inner  = sub(m_vec(cnb1, sub(add(m_vec(cb0n, tic), p0), p1)), tic);
            m33 right = m_add(m_add(m_mul(m_neg(tmp_r), m_skew(pts_c_0)), m_skew(m_vec(tmp_r, pts_c_0))),
                              m_skew(m_vec(cbc, inner)));
            put23(Je0, Je1, red, right, 3);
            Je0[6] = 0;
            Je1[6] = 0;
            // inverse depth and td: t = -reduce * tmp_r
            double nred[6];
#pragma unroll
            for (int i = 0; i < 6; i++) nred[i] = -red[i];
            double t0[3], t1[3];
            put23(t0, t1, nred, tmp_r, 0);
            double idsq = id0 * id0;
            o[44]       = (t0[0] * pts_0_td.x + t0[1] * pts_0_td.y + t0[2] * pts_0_td.z) / idsq;
            o[45]       = (t1[0] * pts_0_td.x + t1[1] * pts_0_td.y + t1[2] * pts_0_td.z) / idsq;
            o[46]       = (t0[0] * vel0.x + t0[1] * vel0.y + t0[2] * vel0.z) / id0 + sinfo * vel1.x;
            o[47]       = (t1[0] * vel0.x + t1[1] * vel0.y + t1[2] * vel0.z) / id0 + sinfo * vel1.y;
        }

        if (A.huber_delta > 0) {
            // residual_block_info.h:59-87 with ceres::HuberLoss(delta)
            double a = A.huber_delta, b = a * a;
            double s = r0 * r0 + r1 * r1;
            double rho1, rho2;
            if (s > b) {
                double r = sqrt(s);
                rho1     = fmax(2.2250738585072014e-308, a / r);
                rho2     = -rho1 / (2.0 * s);
            } else {
                rho1 = 1.0;
                rho2 = 0.0;
            }
            double sqrt_rho1 = sqrt(rho1);
            double residual_scaling, alpha_sq_norm;
            if ((s == 0.0) || (rho2 <= 0.0)) {
                residual_scaling = sqrt_rho1;
                alpha_sq_norm    = 0.0;
            } else {
                const double D     = 1.0 + 2.0 * s * rho2 / rho1;
                const double alpha = 1.0 - sqrt(D);
                residual_scaling   = sqrt_rho1 / (1 - alpha);
                alpha_sq_norm      = alpha / s;
            }
            if (A.want_jac) {
                // columns: three 2x7 blocks (row stride 7) and two 2x1 blocks (row stride 1)
#pragma unroll
                for (int blk = 0; blk < 3; blk++) {
                    double *B = o + 2 + 14 * blk;
#pragma unroll
                    for (int c = 0; c < 7; c++) {
                        double j0 = B[c], j1 = B[7 + c];
                        double rtj = r0 * j0 + r1 * j1;
                        B[c]       = sqrt_rho1 * (j0 - alpha_sq_norm * r0 * rtj);
                        B[7 + c]   = sqrt_rho1 * (j1 - alpha_sq_norm * r1 * rtj);
                    }
                }
#pragma unroll
                for (int blk = 0; blk < 2; blk++) {
                    double *B  = o + 44 + 2 * blk;
                    double j0 = B[0], j1 = B[1];
                    double rtj = r0 * j0 + r1 * j1;
                    B[0]       = sqrt_rho1 * (j0 - alpha_sq_norm * r0 * rtj);
                    B[1]       = sqrt_rho1 * (j1 - alpha_sq_norm * r1 * rtj);
                }
            }
            o[0] = r0 * residual_scaling;
            o[1] = r1 * residual_scaling;
        }
    }
    __syncthreads();

    const int nvalid = min(RPJ_TILE, A.n - f0);
    // residual slab: nvalid x 2 doubles, contiguous
    {
        int e = lane * 2;
        if (lane < nvalid) {
            double2 v = make_double2(tile[lane * RPJ_LDS_STRIDE + 0], tile[lane * RPJ_LDS_STRIDE + 1]);
            *reinterpret_cast<double2 *>(A.out_r + (size_t) f0 * 2 + e) = v;
        }
    }
    if (A.want_jac) {
        const int total2 = nvalid * 23; // double2 elements in the J slab
        double *dst      = A.out_J + (size_t) f0 * 46;
        for (int i2 = lane; i2 < total2; i2 += RPJ_TILE) {
            int e = i2 * 2;
            int f = e / 46, c = e - f * 46;
            double2 v = make_double2(tile[f * RPJ_LDS_STRIDE + 2 + c], tile[f * RPJ_LDS_STRIDE + 3 + c]);
            *reinterpret_cast<double2 *>(dst + e) = v;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
static int ensure_factor_capacity(icg_ctx *ctx, int n) {
    if (n <= ctx->factors_cap) return 0;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_obs) (void) hipFree(ctx->d_obs);
    if (ctx->d_fidx) (void) hipFree(ctx->d_fidx);
    if (ctx->d_rJ) (void) hipFree(ctx->d_rJ);
    if (ctx->d_fwin) (void) hipFree(ctx->d_fwin);
    ctx->d_fwin = nullptr;
    ctx->d_obs = nullptr;
    ctx->d_fidx = nullptr;
    ctx->d_rJ = nullptr;
    ctx->factors_cap = 0;
    int cap = n + n / 4 + 64;
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_obs, sizeof(double) * 15 * (size_t) cap));
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_fidx, sizeof(int32_t) * 3 * (size_t) cap));
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_rJ, sizeof(double) * 48 * (size_t) cap));
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_fwin, sizeof(int32_t) * (size_t) cap));
    ctx->factors_cap = cap;
    return 0;
}

// The factor set goes up from PINNED memory: the classic entry point copies the caller's arrays into the context's staging block first,
// icg_reproj_stage_factors hands the block out so that a caller with many windows fills it in place from its own threads (34 MB of
// observations for 256 marginalization windows: as a pageable hipMemcpy they were 9 of the 13 ms MarginalizationBatch::layout took).
static size_t fstage_idx_offset(int n) { return icg_align_up(sizeof(double) * 15 * (size_t) n, 256); }

extern "C" int icg_reproj_stage_factors(icg_ctx *ctx, int n, double **obs_soa, int32_t **idx3) {
    if (!ctx || n < 0 || !obs_soa || !idx3) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t bytes = fstage_idx_offset(n) + sizeof(int32_t) * 3 * (size_t) n + 256;
    if (bytes > ctx->fstage_cap) {
        ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_fstage) (void) hipHostFree(ctx->h_fstage);
        ctx->h_fstage = nullptr, ctx->fstage_cap = 0;
        const size_t cap = bytes + bytes / 4;
        ICG_HIP(ctx, hipHostMalloc((void **) &ctx->h_fstage, cap, hipHostMallocDefault));
        ctx->fstage_cap = cap;
    }
    ctx->fstage_n = n;
    *obs_soa      = reinterpret_cast<double *>(ctx->h_fstage);
    *idx3         = reinterpret_cast<int32_t *>(ctx->h_fstage + fstage_idx_offset(n));
    return ICG_OK;
}

extern "C" int icg_reproj_commit_factors(icg_ctx *ctx) {
    if (!ctx) return ICG_ERR_INVALID;
    const int n = ctx->fstage_n;
    if (n < 0) return icg_fail(ctx, ICG_ERR_INVALID, "no staged factor set: call icg_reproj_stage_factors first");
    ctx->fstage_n = -1;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_factor_capacity(ctx, n);
    if (rc) return rc;
    ctx->n_factors_resident = n;
    ctx->rJ_valid           = 0;
    // a new factor set: the partitions, their assembly plans and resident systems belong to the old one
    for (icg_partition *pt : {&ctx->part_1, &ctx->part_w}) pt->W = 0, pt->plan_valid = false, pt->sys_valid = 0;
    const int32_t *idx3 = reinterpret_cast<const int32_t *>(ctx->h_fstage + fstage_idx_offset(n));
    ctx->h_fidx.assign(idx3, idx3 + 3 * (size_t) n);
    if (n == 0) return ICG_OK;
    // component-major obs is already the device layout; indices packed as 3 x n
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_obs, ctx->h_fstage, sizeof(double) * 15 * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_fidx, idx3, sizeof(int32_t) * 3 * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICG_OK;
}

extern "C" int icg_reproj_set_factors(icg_ctx *ctx, int n, const double *obs_soa, const int32_t *idx_i,
                                      const int32_t *idx_j, const int32_t *idx_lm) {
    if (!ctx || n < 0 || (n > 0 && (!obs_soa || !idx_i || !idx_j || !idx_lm))) return ICG_ERR_INVALID;
    double *so   = nullptr;
    int32_t *si  = nullptr;
    int rc = icg_reproj_stage_factors(ctx, n, &so, &si);
    if (rc) return rc;
    if (n > 0) {
        memcpy(so, obs_soa, sizeof(double) * 15 * (size_t) n);
        memcpy(si, idx_i, sizeof(int32_t) * (size_t) n);
        memcpy(si + n, idx_j, sizeof(int32_t) * (size_t) n);
        memcpy(si + 2 * (size_t) n, idx_lm, sizeof(int32_t) * (size_t) n);
    }
    return icg_reproj_commit_factors(ctx);
}

// r_view / J_view != nullptr: the results are left in the context's pinned staging memory after the device-to-host copy and the views point
// there (valid until the next call on ctx) — the per-factor Evaluate() surface reads them in place, no 1 MB copy-out per window
static int eval_resident_impl(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth, double td,
                              int want_jac, double huber_delta, double *out_r, double *out_J, const double **r_view, const double **J_view) {
    if (!ctx || !poses || !ext || !invdepth || n_poses <= 0 || n_lm <= 0) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int n = ctx->n_factors_resident;
    if (n == 0) return ICG_OK;
    // parameters: poses | ext | invdepth  packed in the staging arena
    size_t pbytes = sizeof(double) * ((size_t) n_poses * 7 + 7 + (size_t) n_lm);
    size_t rbytes = sizeof(double) * 2 * (size_t) n, jbytes = want_jac ? sizeof(double) * 46 * (size_t) n : 0;
    if (int rcd = icg_arena_drain(ctx)) return rcd;
    ctx->arena_off = 0;
    int rc         = icg_arena_reserve(ctx, pbytes + rbytes + jbytes + 4096);
    if (rc) return rc;
    size_t o_par = icg_arena_alloc(ctx, pbytes);
    double *hp   = icg_h<double>(ctx, o_par);
    memcpy(hp, poses, sizeof(double) * 7 * (size_t) n_poses);
    memcpy(hp + 7 * (size_t) n_poses, ext, sizeof(double) * 7);
    memcpy(hp + 7 * (size_t) n_poses + 7, invdepth, sizeof(double) * (size_t) n_lm);
    size_t in_end = ctx->arena_off;
    size_t o_r    = icg_arena_alloc(ctx, rbytes);
    size_t o_J    = want_jac ? icg_arena_alloc(ctx, jbytes) : 0;
    if ((rc = icg_arena_overflow_check(ctx))) return rc;
    if ((rc = icg_arena_h2d(ctx, o_par, in_end))) return rc;

    rpj_args A;
    A.n           = n;
    A.obs         = ctx->d_obs;
    A.idx_i       = ctx->d_fidx;
    A.idx_j       = ctx->d_fidx + n;
    A.idx_lm      = ctx->d_fidx + 2 * (size_t) n;
    double *dp    = icg_d<double>(ctx, o_par);
    A.poses       = dp;
    A.ext         = dp + 7 * (size_t) n_poses;
    A.invdepth    = dp + 7 * (size_t) n_poses + 7;
    A.td          = td;
    A.want_jac    = want_jac;
    A.huber_delta = huber_delta;
    A.win         = nullptr;
    A.tdv         = nullptr;
    // device-resident results (kept for icg_reproj_accumulate_normal): r at d_rJ, J after it
    A.out_r = ctx->d_rJ;
    A.out_J = ctx->d_rJ + 2 * (size_t) ctx->factors_cap;
    {
        icg_prof_scope ps(ctx, "reproj_eval");
        hipLaunchKernelGGL(k_reproj_eval, dim3((n + RPJ_TILE - 1) / RPJ_TILE), dim3(RPJ_TILE), 0, ctx->stream, A);
    }
    ICG_HIP(ctx, hipGetLastError());
    ctx->rJ_valid     = 1;
    ctx->rJ_has_jac   = want_jac;
    ctx->last_huber   = huber_delta;
    ctx->last_n_poses = n_poses;
    ctx->last_n_lm    = n_lm;
    if (out_r || r_view) {
        ICG_HIP(ctx, hipMemcpyAsync(icg_h<double>(ctx, o_r), A.out_r, rbytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if ((out_J || J_view) && want_jac) {
        ICG_HIP(ctx, hipMemcpyAsync(icg_h<double>(ctx, o_J), A.out_J, jbytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    icg_prof_collect(ctx);
    if (out_r) memcpy(out_r, icg_h<double>(ctx, o_r), rbytes);
    if (out_J && want_jac) memcpy(out_J, icg_h<double>(ctx, o_J), jbytes);
    if (r_view) *r_view = icg_h<double>(ctx, o_r);
    if (J_view) *J_view = want_jac ? icg_h<double>(ctx, o_J) : nullptr;
    ctx->arena_off = 0;
    return ICG_OK;
}

extern "C" int icg_reproj_eval_resident(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm,
                                        const double *invdepth, double td, int want_jac, double huber_delta,
                                        double *out_r, double *out_J) {
    return eval_resident_impl(ctx, n_poses, poses, ext, n_lm, invdepth, td, want_jac, huber_delta, out_r, out_J, nullptr, nullptr);
}

extern "C" int icg_reproj_eval_resident_view(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm,
                                             const double *invdepth, double td, int want_jac, double huber_delta,
                                             const double **r_view, const double **J_view) {
    if (!r_view || !J_view) return ICG_ERR_INVALID;
    return eval_resident_impl(ctx, n_poses, poses, ext, n_lm, invdepth, td, want_jac, huber_delta, nullptr, nullptr, r_view, J_view);
}

extern "C" int icg_reproj_eval_batch(icg_ctx *ctx, int n, const double *obs_soa, const int32_t *idx_i,
                                     const int32_t *idx_j, const int32_t *idx_lm, int n_poses, const double *poses,
                                     const double *ext, int n_lm, const double *invdepth, double td, int want_jac,
                                     double huber_delta, double *out_r, double *out_J) {
    int rc = icg_reproj_set_factors(ctx, n, obs_soa, idx_i, idx_j, idx_lm);
    if (rc) return rc;
    return icg_reproj_eval_resident(ctx, n_poses, poses, ext, n_lm, invdepth, td, want_jac, huber_delta, out_r, out_J);
}

// ---- M2 / f1: the normal equations of the resident reprojection factors, assembled in a fixed order ------------------------------------
// Reference: factors/marginalization_info.h:195-230 (constructEquation: H0 += Ji^T Jj over the block pairs of a factor, b0 -= Ji^T e) and
// the DENSE_SCHUR step of GVINS::gvinsOptimization (ic_gvins.cc:1130-1239, 1763-1837): the inverse-depth blocks (1 x 1) go first.
//
// System of one window, N = P + L:   H = [Hcc G^T; G diag(h_ll)]  (row-major N x N: Hcc in rows/columns < P, landmark l in row P + l —
// only its P camera columns and its diagonal element are ever written or read),  b (N),  inv (L) = 1 / (h_ll + d_l).
//
// No atomics anywhere: rounds 1-5 scattered every J^T J product with FP64 atomicAdd (LDS + global), which made every sum depend on the
// arrival order — results equal to rounding only, the lock-step tests could not ask for identical bits, and same-address LDS atomics were
// the cost of the launch (0.8-1.1 ms for 256 windows of 2 700 factors).  Now every output cell is owned by one thread that adds its
// contributions in an order fixed by the window's own factor list (icg_asm_plan):
//   k_asm_runs       one wave per run = the factors of one ordered (reference pose i, observer pose j) pair.  All of them send their
//                    19 camera columns [Ji | Jj | Je | Jtd] (+ the residual as a 20th column: b = -J^T r) to the SAME cells, so the wave
//                    keeps the 20 x 20 product A^T A (A = the run's 2 rows per factor) in registers — lane t owns one 2 x 2 tile of the 55 in
//                    the upper triangle — and walks the run in list order, 16 factors staged in LDS at a time (operands are LDS broadcasts:
//                    4 ds_read_b128 + 8 v_fma_f64 per factor and lane).  Output: 220 doubles per run, one coalesced 32-B store per lane.
//   k_asm_camera     one thread per cell of Hcc (and of bc): gathers the cell from the runs that touch it — (i,j) and (j,i) for a cell
//                    between two poses, row and column p of the pair table for a cell of pose p's diagonal block or against the shared
//                    extrinsic / td block, every run for the (ext|td)^2 block — and stores it.  Cells nobody touches are stored as zero:
//                    no memset of the system (the old path cleared 1 MB per window per launch).
//   k_asm_landmarks  one thread per (landmark, camera column | h_ll | b_l): walks the landmark's factors in list order.
// Algorithmic traffic per launch: J and r once per kernel that needs them (2 x 384 B per factor), 1.76 KB per run out and in, the system
// rows once.  Bound: HBM/L2 streaming of J; the FP64 FMAs (420 per factor) are 2 % of the vector peak.
#define ASM_SUB 16   // factors staged per pass
#define ASM_ROW 40   // doubles per staged factor: two rows of 20 columns [Ji 0..5 | Jj 6..11 | Je 12..17 | Jtd 18 | -r 19]
#define ASM_PART 220 // doubles per run: 55 upper-triangular 2 x 2 tiles of the 20 x 20 product
#define ASM_EXT 0xFFF // owner code of the shared (extrinsic | td) pseudo-block: columns 12..18 of a factor's row

struct win_desc {
    int32_t fac_begin, fac_end, lm_begin, L;
    int64_t sys_off;
    int32_t K, reassemble; // K: poses used by the window's factors (local numbering of the plan)
    double damp;
    int32_t NB, pad; // column blocks of the landmark rows (k_asm_landmarks)
};

// gfx950 has 160 KiB of LDS per CU and one workgroup may own all of it (MI355X_MICROARCH.md, "LDS"); a launch with more dynamic LDS than the
// 64 KiB default needs the function attribute.  Only the batched Cholesky below (P^2 + P doubles per window) asks for that; the limit comes
// from the device the context runs on, so a build for a part with less LDS reports ICG_ERR_CAPACITY instead of failing at launch.
static size_t rpj_lds_limit(icg_ctx *ctx) {
    static std::atomic<int> per_dev[16];
    const int dev = ctx->cfg.device & 15;
    int v         = per_dev[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->cfg.device) != hipSuccess || optin <= 0) optin = 64 * 1024;
        v = optin;
        per_dev[dev].store(v, std::memory_order_relaxed);
    }
    return (size_t) v - 256; // (- the kernels' few static words)
}
template <typename K> static int rpj_allow_lds(icg_ctx *ctx, K kernel, size_t bytes, int slot) {
    static std::atomic<size_t> granted[4][16];
    if (bytes <= 48 * 1024) return 0;
    const int dev = ctx->cfg.device & 15;
    if (granted[slot][dev].load(std::memory_order_relaxed) >= bytes) return 0;
    const size_t lim = rpj_lds_limit(ctx);
    ICG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lim));
    granted[slot][dev].store(lim, std::memory_order_relaxed);
    return 0;
}

// index of (la, lb) inside a run's block: tile (la/2, lb/2) of the upper triangle, element (la&1, lb&1); the diagonal tiles hold both halves
__device__ __forceinline__ int asm_part_index(int la, int lb) {
    int ba = la >> 1, bb = lb >> 1;
    if (ba > bb) {
        const int t = la;
        la = lb, lb = t;
        ba = la >> 1, bb = lb >> 1;
    }
    return (ba * 10 - ((ba * (ba - 1)) >> 1) + (bb - ba)) * 4 + ((la & 1) << 1) + (lb & 1);
}

__global__ __launch_bounds__(256) void k_asm_runs(int n_runs, const int4 *runs, const win_desc *wd, const int32_t *perm, const double *r,
                                                 const double *J, const uint8_t *active, double *part) {
    __shared__ double tiles[4][ASM_SUB * ASM_ROW];
    const int wave = threadIdx.x >> 6, t = threadIdx.x & 63;
    const int ri   = blockIdx.x * 4 + wave;
    if (ri >= n_runs) return;
    const int4 R = runs[ri]; // first, count, li | lj << 16, window
    if (!wd[R.w].reassemble) return;
    double *tile = tiles[wave];
    // this lane's tile of the upper triangle (lanes 55..63 idle along on tile 0 and store nothing)
    int bx = 0, rem = t < 55 ? t : 0, len = 10;
    while (rem >= len) rem -= len, bx++, len--;
    const int by = bx + rem;
    // staging role: lane -> factor t / 4 of the pass, elements (t & 3) + 4 k of its 48 values (J 0..45, r 46..47)
    const int fi = t >> 2, sub = t & 3;
    double v[12];
    auto fetch = [&](int pass) {
        const int k = pass * ASM_SUB + fi;
        bool on     = k < R.y;
        int f       = 0;
        if (on) {
            f = perm[R.x + k];
            if (active && !active[f]) on = false;
        }
        const double *Jf = J + 46 * (size_t) f;
#pragma unroll
        for (int kk = 0; kk < 11; kk++) v[kk] = on ? Jf[sub + 4 * kk] : 0.0;
        v[11] = on ? (sub < 2 ? Jf[44 + sub] : -r[2 * (size_t) f + (sub - 2)]) : 0.0;
    };
    const int npass = (R.y + ASM_SUB - 1) / ASM_SUB;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    fetch(0);
    for (int pass = 0; pass < npass; pass++) {
        // registers -> LDS in the padded two-row layout (the 7th, always-zero column of every 2 x 7 block and the landmark column are dropped)
#pragma unroll
        for (int kk = 0; kk < 12; kk++) {
            const int c = sub + 4 * kk;
            if (c < 42) {
                const int q = c / 7, x = c - 7 * q;
                if (x < 6) tile[fi * ASM_ROW + (q & 1) * 20 + (q >> 1) * 6 + x] = v[kk];
            } else if (c >= 44) {
                tile[fi * ASM_ROW + (c & 1) * 20 + 18 + ((c - 44) >> 1)] = v[kk];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (pass + 1 < npass) fetch(pass + 1); // in flight while this pass is multiplied
#pragma unroll
        for (int g = 0; g < ASM_SUB; g++) {
            const double2 x0 = *reinterpret_cast<const double2 *>(&tile[g * ASM_ROW + 2 * bx]);
            const double2 x1 = *reinterpret_cast<const double2 *>(&tile[g * ASM_ROW + 20 + 2 * bx]);
            const double2 y0 = *reinterpret_cast<const double2 *>(&tile[g * ASM_ROW + 2 * by]);
            const double2 y1 = *reinterpret_cast<const double2 *>(&tile[g * ASM_ROW + 20 + 2 * by]);
            a00 = fma(x1.x, y1.x, fma(x0.x, y0.x, a00));
            a01 = fma(x1.x, y1.y, fma(x0.x, y0.y, a01));
            a10 = fma(x1.y, y1.x, fma(x0.y, y0.x, a10));
            a11 = fma(x1.y, y1.y, fma(x0.y, y0.y, a11));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (t < 55) {
        double *dst = part + (size_t) ri * ASM_PART + 4 * t;
        *reinterpret_cast<double2 *>(dst)     = make_double2(a00, a01);
        *reinterpret_cast<double2 *>(dst + 2) = make_double2(a10, a11);
    }
}

// owner[w * P + a] of a camera column: (local pose << 3) | x for column x of a pose block, (ASM_EXT << 3) | x for the extrinsic (x < 6) and td
// (x == 6), -1 for a column no visual factor of the window touches (host-only blocks, empty tail columns).
// The gathers are chains of additions in a fixed order, but their loads are independent: they are issued eight at a time (a missing run
// contributes +0.0, which leaves every partial sum as it is) — one thread walks up to K^2 runs, and a dependent L2 round trip per run made
// the (ext|td)^2 cells the critical path of the launch.
__device__ __forceinline__ double asm_gather_all(const double *part, int r0, int r1, int idx, double acc) {
    for (int rr = r0; rr < r1; rr += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = rr + u < r1 ? part[(size_t) (rr + u) * ASM_PART + idx] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    return acc;
}
// every run with local pose p as reference (row p of the pair table, element i_r of the run's block) or as observer (column p, element i_o)
__device__ __forceinline__ double asm_gather_pose(const double *part, const int32_t *pr, int Kmax, int K, int p, int i_r, int i_o, double acc) {
    for (int q = 0; q < K; q += 4) {
        int rr[4], ro[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rr[u] = q + u < K ? pr[p * Kmax + q + u] : -1, ro[u] = q + u < K ? pr[(q + u) * Kmax + p] : -1;
        double vr[4], vo[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vr[u] = rr[u] >= 0 ? part[(size_t) rr[u] * ASM_PART + i_r] : 0.0, vo[u] = ro[u] >= 0 ? part[(size_t) ro[u] * ASM_PART + i_o] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) acc += vr[u], acc += vo[u];
    }
    return acc;
}
// grid (ceil((P * P + P) / 256), W)
__global__ __launch_bounds__(256) void k_asm_camera(const win_desc *wd, const int32_t *run_off, const int32_t *pair_run, int Kmax, const int16_t *owner,
                                                   int P, const double *part, double *sys) {
    const int w       = blockIdx.y;
    const win_desc W = wd[w];
    if (!W.reassemble) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= P * P + P) return;
    const int N          = P + W.L, K = W.K;
    double *H            = sys + W.sys_off, *b = H + (size_t) N * N;
    const int16_t *own   = owner + (size_t) w * P;
    const int32_t *pr    = pair_run + (size_t) w * Kmax * Kmax;
    const int r0 = run_off[w], r1 = run_off[w + 1];
    double acc = 0.0;
    if (e < P * P) {
        const int a = e / P, c = e - a * P;
        const int oa = own[a], oc = own[c];
        if (oa >= 0 && oc >= 0) {
            const int pa = oa >> 3, xa = oa & 7, pc = oc >> 3, xc = oc & 7;
            if (pa == ASM_EXT && pc == ASM_EXT) {
                acc = asm_gather_all(part, r0, r1, asm_part_index(12 + xa, 12 + xc), acc);
            } else if (pa != ASM_EXT && pc != ASM_EXT && pa != pc) {
                const int r_ac = pr[pa * Kmax + pc], r_ca = pr[pc * Kmax + pa];
                const double v_ac = r_ac >= 0 ? part[(size_t) r_ac * ASM_PART + asm_part_index(xa, 6 + xc)] : 0.0;
                const double v_ca = r_ca >= 0 ? part[(size_t) r_ca * ASM_PART + asm_part_index(6 + xa, xc)] : 0.0;
                acc = v_ac + v_ca;
            } else {
                // pose p's diagonal block, or pose p against the shared block
                const int p   = pa != ASM_EXT ? pa : pc;
                const int i_r = asm_part_index(pa == ASM_EXT ? 12 + xa : xa, pc == ASM_EXT ? 12 + xc : xc);         // p is the run's reference
                const int i_o = asm_part_index(pa == ASM_EXT ? 12 + xa : 6 + xa, pc == ASM_EXT ? 12 + xc : 6 + xc); // p is the run's observer
                acc = asm_gather_pose(part, pr, Kmax, K, p, i_r, i_o, acc);
            }
        }
        H[(size_t) a * N + c] = acc;
    } else {
        const int a  = e - P * P;
        const int oa = own[a];
        if (oa >= 0) {
            const int pa = oa >> 3, xa = oa & 7;
            if (pa == ASM_EXT)
                acc = asm_gather_all(part, r0, r1, asm_part_index(12 + xa, 19), acc);
            else
                acc = asm_gather_pose(part, pr, Kmax, K, pa, asm_part_index(xa, 19), asm_part_index(6 + xa, 19), acc);
        }
        b[a] = acc;
    }
}

// Landmark rows.  The P camera columns of a window are cut into blocks (host, per call): the six columns of a free pose, the six of the
// extrinsic, runs of up to six columns that no visual factor touches (stored as zeros), and one block for (td column, h_ll, b_l).  A thread
// owns one (landmark, block) pair and walks the landmark's factors ONCE for the whole block — the first version owned single cells and walked
// them once per column (70 % of its iterations found a pose that is neither the factor's reference nor its observer).  A workgroup owns LB
// consecutive landmarks (LB * blocks <= 256); their factors are one contiguous range of the landmark-major list, staged in LDS 64 at a
// time (J and r of a factor = 48 doubles: ONE round trip of independent coalesced loads per pass instead of the dependent lrec -> J -> J
// chain per cell), then added in list order.  An inactive factor is staged as zeros.
#define ASML_FB 64
#define ASM_BLK_TD 0xFFE  // (td column or 0xFFF = none, h_ll, b_l)
#define ASM_BLK_GAP 0xFFD // columns of host-only blocks: zeros
// blocks[w * NBmax + k] = col0 | width << 12 | code << 16 (code: local pose, ASM_EXT, ASM_BLK_TD, ASM_BLK_GAP); grid (ceil(Lmax / LB), W)
__global__ __launch_bounds__(256) void k_asm_landmarks(const win_desc *wd, const int32_t *blocks, int NBmax, int P, int LB, const int32_t *lm_foff,
                                                      const int4 *lrec, const double *r, const double *J, const uint8_t *active, double *sys) {
    __shared__ double st[ASML_FB * 48];
    __shared__ int st_i[ASML_FB], st_j[ASML_FB];
    const int w       = blockIdx.y;
    const win_desc W = wd[w];
    if (!W.reassemble) return;
    const int l0 = blockIdx.x * LB;
    if (l0 >= W.L) return;
    const int t = threadIdx.x, nl = min(LB, W.L - l0), NB = W.NB;
    const int N = P + W.L;
    double *H   = sys + W.sys_off, *b = H + (size_t) N * N;
    const int il = t / NB, ib = t - il * NB;
    const bool has = il < nl;
    int col0 = 0, width = 0, code = ASM_BLK_GAP, fb = 0, fe = 0;
    if (has) {
        const int blk = blocks[(size_t) w * NBmax + ib];
        col0 = blk & 0xFFF, width = (blk >> 12) & 0xF, code = blk >> 16;
        fb = lm_foff[W.lm_begin + l0 + il], fe = lm_foff[W.lm_begin + l0 + il + 1];
    }
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int f_begin = lm_foff[W.lm_begin + l0], f_end = lm_foff[W.lm_begin + l0 + nl];
    const int fi = t >> 2, sub = t & 3;
    for (int c0 = f_begin; c0 < f_end; c0 += ASML_FB) {
        const int nc = min(ASML_FB, f_end - c0);
        __syncthreads(); // (the previous pass has been consumed)
        if (fi < nc) {
            const int4 rec = lrec[c0 + fi]; // factor, local_i, local_j
            const bool on  = !active || active[rec.x];
            const double *Jf = J + 46 * (size_t) rec.x;
#pragma unroll
            for (int kk = 0; kk < 12; kk++) {
                const int c = sub + 4 * kk;
                st[fi * 48 + c] = on ? (c < 46 ? Jf[c] : r[2 * (size_t) rec.x + (c - 46)]) : 0.0;
            }
            if (sub == 0) st_i[fi] = rec.y, st_j[fi] = rec.z;
        }
        __syncthreads();
        if (code == ASM_BLK_GAP) continue;
        const int g1 = min(fe, c0 + nc) - c0;
        for (int g = max(fb, c0) - c0; g < g1; g++) {
            const double *Jf = &st[g * 48];
            const double jl0 = Jf[42], jl1 = Jf[43];
            if (code == ASM_BLK_TD) {
                acc[0] = fma(jl1, Jf[45], fma(jl0, Jf[44], acc[0]));
                acc[1] = fma(jl1, jl1, fma(jl0, jl0, acc[1]));
                acc[2] = fma(jl1, -Jf[47], fma(jl0, -Jf[46], acc[2]));
                continue;
            }
            int o;
            if (code == ASM_EXT)
                o = 28;
            else if (code == st_i[g])
                o = 0;
            else if (code == st_j[g])
                o = 14;
            else
                continue;
#pragma unroll
            for (int x = 0; x < 6; x++) acc[x] = fma(jl1, Jf[o + 7 + x], fma(jl0, Jf[o + x], acc[x]));
        }
    }
    if (!has) return;
    const int l = l0 + il;
    double *row = H + (size_t) (P + l) * N;
    if (code == ASM_BLK_TD) {
        if (col0 != 0xFFF) row[col0] = acc[0];
        row[P + l] = acc[1];
        b[P + l]   = acc[2];
    } else {
#pragma unroll
        for (int x = 0; x < 6; x++)
            if (x < width) row[col0 + x] = acc[x];
    }
}

__global__ void k_schur_inv_w(const win_desc *wd, int P, double *sys, double min_diag, double max_diag) {
    const win_desc W = wd[blockIdx.y];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= W.L) return;
    const int N = P + W.L;
    const double *H = sys + W.sys_off;
    double *inv = sys + W.sys_off + (size_t) N * N + N;
    const double h = H[(size_t) (P + l) * N + P + l];
    // a landmark without any active factor has an empty row: it is left where it is (delta_l = 0)
    inv[l] = h > 0.0 ? 1.0 / (h + fmin(fmax(h, min_diag), max_diag) * W.damp) : 0.0;
}

// S = Hcc - G^T diag(inv) G,  s = bc - G^T (inv b_l),  diag = diag(Hcc).
// Rounds 1-5 ran 16 x 16 output tiles of one thread per element, every tile re-reading its two G panels from memory with two barriers per 16
// landmarks: 157-199 us for 256 windows of a 0.35 GFLOP contraction.  Now a workgroup owns (up to 256 of) the 4 x 4 register tiles of the
// LOWER triangle of one window's S: the G rows of 32 landmarks are staged in LDS once per pass and every thread reads its row and column
// quadruples from there (4 ds_read_b128 per landmark for 16 FMAs).  The upper triangle is the mirror image of the lower one (S is
// symmetric; the factorizations read rows >= columns): written as such when the caller wants the full matrix, not at all otherwise.
// Landmarks are added in index order: the value of a window does not depend on the batch it is reduced in.
#define SCH_LT 32 // landmark rows per pass (fewer for wide systems: LT * 4 TQ <= 3 072 elements, twelve per thread)
#define SCH_PRE 12
// grid (ceil(NT / 256), W), NT = TQ (TQ + 1) / 2 lower tiles, TQ = ceil(P / 4); dynamic LDS: LT * 4 TQ doubles (G) + 2 LT (inv, b_l)
__global__ __launch_bounds__(256) void k_schur_reduce_w(const win_desc *wd, int P, int LT, const double *sys, double *S, double *s, double *diag,
                                                       int lower_only) {
    extern __shared__ double sm[];
    const win_desc W = wd[blockIdx.y];
    const int L = W.L, N = P + L, TQ = (P + 3) >> 2, PP = 4 * TQ, NT = (TQ * (TQ + 1)) >> 1;
    const double *H = sys + W.sys_off, *b = H + (size_t) N * N, *inv = b + N;
    double *g = sm, *sw = sm + LT * PP, *swb = sw + LT;
    const int t = threadIdx.x, tid = blockIdx.x * 256 + t;
    // tile (ti, tj), tj <= ti, from the triangular index
    int ti = (int) ((sqrtf(8.0f * (float) tid + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= tid) ti++;
    while (ti * (ti + 1) / 2 > tid) ti--;
    const int tj     = tid - ti * (ti + 1) / 2;
    const bool owner = tid < NT;
    double acc[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int cc = 0; cc < 4; cc++) acc[rr][cc] = 0.0;
    double accs[2] = {0.0, 0.0}; // s entries t and t + 256 (the window's first workgroup; P <= 512)
    // staging: element e = t + 256 k of a pass is row e / PP, column e % PP of the slice; the next pass is fetched into registers while
    // this one is multiplied (one workgroup per CU at 256 windows: nobody else would hide the round trip)
    int pl[SCH_PRE], pc[SCH_PRE];
    double pre[SCH_PRE];
#pragma unroll
    for (int k = 0; k < SCH_PRE; k++) {
        const int e = t + 256 * k;
        pl[k] = e / PP, pc[k] = e - pl[k] * PP;
        if (e >= LT * PP) pl[k] = -1;
    }
    auto fetch = [&](int l0) {
#pragma unroll
        for (int k = 0; k < SCH_PRE; k++) {
            pre[k] = 0.0;
            if (pl[k] >= 0 && l0 + pl[k] < L && pc[k] < P) pre[k] = H[(size_t) (P + l0 + pl[k]) * N + pc[k]];
        }
    };
    fetch(0);
    for (int l0 = 0; l0 < L; l0 += LT) {
        const int nl = min(LT, L - l0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCH_PRE; k++)
            if (pl[k] >= 0) g[t + 256 * k] = pre[k];
        if (t < LT) sw[t] = t < nl ? inv[l0 + t] : 0.0, swb[t] = t < nl ? b[P + l0 + t] : 0.0;
        __syncthreads();
        if (l0 + LT < L) fetch(l0 + LT);
        if (owner) {
            for (int l = 0; l < nl; l++) {
                const double wl  = sw[l];
                const double2 a0 = *reinterpret_cast<const double2 *>(&g[l * PP + 4 * ti]), a1 = *reinterpret_cast<const double2 *>(&g[l * PP + 4 * ti + 2]);
                const double2 b0 = *reinterpret_cast<const double2 *>(&g[l * PP + 4 * tj]), b1 = *reinterpret_cast<const double2 *>(&g[l * PP + 4 * tj + 2]);
                const double av[4] = {a0.x * wl, a0.y * wl, a1.x * wl, a1.y * wl}, bv[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
                for (int rr = 0; rr < 4; rr++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) acc[rr][cc] = fma(av[rr], bv[cc], acc[rr][cc]);
            }
        }
        if (blockIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int i = t + 256 * k;
                if (i < P)
                    for (int l = 0; l < nl; l++) accs[k] = fma(g[l * PP + i] * sw[l], swb[l], accs[k]);
            }
        }
    }
    if (owner) {
        double *Sw = S + (size_t) blockIdx.y * P * P;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const int i = 4 * ti + rr, j = 4 * tj + cc;
                if (i >= P || j > i) continue; // (cells above the diagonal inside a diagonal tile are mirrors too)
                const double v       = H[(size_t) i * N + j] - acc[rr][cc];
                Sw[(size_t) i * P + j] = v;
                if (!lower_only && j < i) Sw[(size_t) j * P + i] = v;
            }
    }
    if (blockIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = t + 256 * k;
            if (i < P) s[(size_t) blockIdx.y * P + i] = b[i] - accs[k], diag[(size_t) blockIdx.y * P + i] = H[(size_t) i * N + i];
        }
    }
}

// one wave per landmark (global index): delta_l = (b_l - G_l . delta_c) * inv_l; lterms[l][2] = b_l^2 / (h_ll + d_l), d_l delta_l^2 — the
// landmark's part of the LM model decrease 0.5 (delta^T b + delta^T D delta): with the reduced right-hand side s, delta^T b =
// delta_c^T s + sum lterms[.][0], so the step-quality ratio is formed without moving G or b_l to the host
__global__ __launch_bounds__(64) void k_schur_backsub_w(const win_desc *wd, const int32_t *lm_win, int lm_base, int P, const double *sys,
                                                        const double *delta_c, double *delta_l, double *lterms, double min_diag, double max_diag) {
    const int lg = lm_base + blockIdx.x, wi = lm_win ? lm_win[lg] : 0;
    const win_desc W = wd[wi];
    const int l = lg - W.lm_begin, N = P + W.L;
    const double *H = sys + W.sys_off, *b = H + (size_t) N * N, *inv = b + N;
    const double *dc = delta_c + (size_t) wi * P;
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += 64) acc += H[(size_t) (P + l) * N + i] * dc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) {
        const double bl = b[P + l], wv = inv[l];
        const double d  = (bl - acc) * wv;
        delta_l[lg]     = d;
        double t0 = 0.0, t1 = 0.0;
        if (wv > 0.0) {
            const double dl = fmin(fmax(H[(size_t) (P + l) * N + P + l], min_diag), max_diag) * W.damp; // the damping that went into inv
            t0 = bl * bl * wv, t1 = dl * d * d;
        }
        lterms[2 * (size_t) lg] = t0, lterms[2 * (size_t) lg + 1] = t1;
    }
}

// sum of 256 per-thread partial sums in a fixed tree: butterfly inside each wave, the four wave sums in order
__device__ __forceinline__ double block_sum_256(double v, double *sh4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
}

// one workgroup per window: terms[w] = the window's landmark terms, thread t adds landmarks t, t + 256, ... in that order
__global__ __launch_bounds__(256) void k_terms_reduce_w(const win_desc *wd, const double *lterms, double *terms) {
    __shared__ double sh[2][4];
    const win_desc W = wd[blockIdx.x];
    double t0 = 0.0, t1 = 0.0;
    for (int l = threadIdx.x; l < W.L; l += 256) t0 += lterms[2 * (size_t) (W.lm_begin + l)], t1 += lterms[2 * (size_t) (W.lm_begin + l) + 1];
    t0 = block_sum_256(t0, sh[0]);
    t1 = block_sum_256(t1, sh[1]);
    if (threadIdx.x == 0) terms[2 * blockIdx.x] = t0, terms[2 * blockIdx.x + 1] = t1;
}

// 0.5 * sum rho(|r|^2) of the window's active factors from the resident (possibly Huber-corrected) residuals: the corrector leaves
// |r_c|^2 = rho'(s) s, i.e. s for inliers and a sqrt(s) > a^2 for outliers, so rho(s) = 2 a sqrt(s) - a^2 = 2 |r_c|^2 - a^2.
// One workgroup per window, thread t adds factors fac_begin + t, + 256, ... in that order, then the fixed tree: the value of a window does
// not depend on the batch it is evaluated in.
__global__ __launch_bounds__(256) void k_reproj_cost_w(const win_desc *wd, const double *r, const uint8_t *active, double huber, double *out) {
    __shared__ double sh[4];
    const win_desc W = wd[blockIdx.x];
    double acc = 0.0;
    for (int f = W.fac_begin + threadIdx.x; f < W.fac_end; f += 256) {
        if (active && !active[f]) continue;
        const double r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
        double q = r0 * r0 + r1 * r1;
        if (huber > 0.0 && q > huber * huber) q = 2.0 * q - huber * huber;
        acc += 0.5 * q;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

static int ensure_sys_capacity(icg_ctx *ctx, size_t doubles) {
    if (doubles <= ctx->sys_cap) return 0;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_sys) (void) hipFree(ctx->d_sys);
    ctx->d_sys   = nullptr;
    ctx->sys_cap = 0;
    ctx->part_1.sys_valid = ctx->part_w.sys_valid = 0;
    size_t cap   = doubles + doubles / 4;
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_sys, sizeof(double) * cap));
    ctx->sys_cap = cap;
    return 0;
}

// ---- the assembly plan of a partition (host, once per factor set / partition) ---------------------------------------------------------------
static int asm_plan_build(icg_ctx *ctx, icg_partition &pt) {
    const int W = pt.W, n = ctx->n_factors_resident, n_lm = pt.lm_off[(size_t) W];
    icg_asm_plan &pl = pt.plan;
    pt.plan_valid    = false;
    if ((int) ctx->h_fidx.size() != 3 * n) return icg_fail(ctx, ICG_ERR_INVALID, "no resident factors");
    const int32_t *ii = ctx->h_fidx.data(), *jj = ii + n, *ll = jj + n;
    int max_pose = -1;
    for (int f = 0; f < n; f++) {
        if (ii[f] < 0 || jj[f] < 0) return icg_fail(ctx, ICG_ERR_INVALID, "factor %d: negative pose index", f);
        if (ii[f] == jj[f]) return icg_fail(ctx, ICG_ERR_INVALID, "factor %d: reference and observer pose are the same block (%d)", f, ii[f]);
        max_pose = std::max(max_pose, std::max((int) ii[f], (int) jj[f]));
    }
    std::vector<int32_t> pose_win((size_t) (max_pose + 1), -1), g2l((size_t) (max_pose + 1), -1), used;
    std::vector<int32_t> perm((size_t) std::max(n, 1)), runs, lrec(4 * (size_t) std::max(n, 1)), lm_foff((size_t) n_lm + 1, 0), cnt;
    pl.run_off.assign((size_t) W + 1, 0);
    pl.pose_off.assign((size_t) W + 1, 0);
    pl.pose_glob.clear();
    pl.Kmax = 1;
    for (int w = 0; w < W; w++) {
        const int f0 = pt.fac_off[(size_t) w], f1 = pt.fac_off[(size_t) w + 1], l0 = pt.lm_off[(size_t) w], l1 = pt.lm_off[(size_t) w + 1];
        used.clear();
        for (int f = f0; f < f1; f++)
            for (int32_t p : {ii[f], jj[f]}) {
                int32_t &pw = pose_win[(size_t) p];
                if (pw >= 0 && pw != w) return icg_fail(ctx, ICG_ERR_INVALID, "pose %d is used by windows %d and %d", (int) p, (int) pw, w);
                if (pw < 0) pw = w, used.push_back(p);
            }
        std::sort(used.begin(), used.end());
        const int K = (int) used.size();
        if (K >= ASM_EXT) return icg_fail(ctx, ICG_ERR_CAPACITY, "window %d uses %d poses (limit %d)", w, K, ASM_EXT - 1);
        for (int k = 0; k < K; k++) g2l[(size_t) used[(size_t) k]] = k;
        pl.pose_glob.insert(pl.pose_glob.end(), used.begin(), used.end());
        pl.pose_off[(size_t) w + 1] = (int32_t) pl.pose_glob.size();
        pl.Kmax                      = std::max(pl.Kmax, K);
        // runs: stable counting sort of the window's factors by the ordered local pose pair
        cnt.assign((size_t) K * K + 1, 0);
        for (int f = f0; f < f1; f++) cnt[(size_t) g2l[(size_t) ii[f]] * K + g2l[(size_t) jj[f]] + 1]++;
        for (size_t k = 0; k < (size_t) K * K; k++) {
            if (cnt[k + 1] > 0) {
                runs.push_back(f0 + cnt[k]), runs.push_back(cnt[k + 1]);
                runs.push_back((int32_t) (k / (size_t) K) | ((int32_t) (k % (size_t) K) << 16)), runs.push_back(w);
            }
            cnt[k + 1] += cnt[k];
        }
        for (int f = f0; f < f1; f++) perm[(size_t) f0 + (size_t) cnt[(size_t) g2l[(size_t) ii[f]] * K + g2l[(size_t) jj[f]]]++] = f;
        pl.run_off[(size_t) w + 1] = (int32_t) (runs.size() / 4);
        // landmark-major records: stable counting sort by landmark
        for (int f = f0; f < f1; f++) {
            if (ll[f] < l0 || ll[f] >= l1) return icg_fail(ctx, ICG_ERR_INVALID, "factor %d: landmark %d outside its window's range [%d, %d)", f, (int) ll[f], l0, l1);
            lm_foff[(size_t) ll[f] + 1]++;
        }
    }
    for (int l = 0; l < n_lm; l++) lm_foff[(size_t) l + 1] += lm_foff[(size_t) l];
    {
        std::vector<int32_t> pos(lm_foff.begin(), lm_foff.end() - 1);
        for (int w = 0; w < W; w++) {
            // (g2l of a pose is its number inside its own window: poses are not shared between windows)
            for (int f = pt.fac_off[(size_t) w]; f < pt.fac_off[(size_t) w + 1]; f++) {
                int32_t *rec = &lrec[4 * (size_t) pos[(size_t) ll[f]]++];
                rec[0] = f, rec[1] = g2l[(size_t) ii[f]], rec[2] = g2l[(size_t) jj[f]], rec[3] = 0;
            }
        }
    }
    pl.n_runs = (int) (runs.size() / 4);
    std::vector<int32_t> pair_run((size_t) W * pl.Kmax * pl.Kmax, -1);
    for (int k = 0; k < pl.n_runs; k++) {
        const int32_t lilj = runs[4 * (size_t) k + 2], w = runs[4 * (size_t) k + 3];
        pair_run[((size_t) w * pl.Kmax + (size_t) (lilj & 0xFFFF)) * pl.Kmax + (size_t) (lilj >> 16)] = k;
    }
    // one device allocation, 256-byte aligned sections
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t b_perm = icg_align_up(sizeof(int32_t) * (size_t) std::max(n, 1), 256), b_runs = icg_align_up(sizeof(int32_t) * std::max<size_t>(runs.size(), 4), 256),
                 b_roff = icg_align_up(sizeof(int32_t) * ((size_t) W + 1), 256), b_pair = icg_align_up(sizeof(int32_t) * pair_run.size(), 256),
                 b_lrec = icg_align_up(sizeof(int32_t) * lrec.size(), 256), b_lmf = icg_align_up(sizeof(int32_t) * lm_foff.size(), 256);
    const size_t total = b_perm + b_runs + b_roff + b_pair + b_lrec + b_lmf;
    if (total > pl.buf_cap) {
        ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (pl.d_buf) (void) hipFree(pl.d_buf);
        pl.d_buf = nullptr, pl.buf_cap = 0;
        ICG_HIP(ctx, hipMalloc((void **) &pl.d_buf, total + total / 4));
        pl.buf_cap = total + total / 4;
    }
    char *p       = pl.d_buf;
    pl.d_perm     = reinterpret_cast<int32_t *>(p), p += b_perm;
    pl.d_runs     = reinterpret_cast<int32_t *>(p), p += b_runs;
    pl.d_run_off  = reinterpret_cast<int32_t *>(p), p += b_roff;
    pl.d_pair_run = reinterpret_cast<int32_t *>(p), p += b_pair;
    pl.d_lrec     = reinterpret_cast<int32_t *>(p), p += b_lrec;
    pl.d_lm_foff  = reinterpret_cast<int32_t *>(p);
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (a launch of the previous plan may still read the buffer)
    if (n) ICG_HIP(ctx, hipMemcpyAsync(pl.d_perm, perm.data(), sizeof(int32_t) * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    if (!runs.empty()) ICG_HIP(ctx, hipMemcpyAsync(pl.d_runs, runs.data(), sizeof(int32_t) * runs.size(), hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipMemcpyAsync(pl.d_run_off, pl.run_off.data(), sizeof(int32_t) * ((size_t) W + 1), hipMemcpyHostToDevice, ctx->stream));
    if (!pair_run.empty()) ICG_HIP(ctx, hipMemcpyAsync(pl.d_pair_run, pair_run.data(), sizeof(int32_t) * pair_run.size(), hipMemcpyHostToDevice, ctx->stream));
    if (n) ICG_HIP(ctx, hipMemcpyAsync(pl.d_lrec, lrec.data(), sizeof(int32_t) * 4 * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipMemcpyAsync(pl.d_lm_foff, lm_foff.data(), sizeof(int32_t) * lm_foff.size(), hipMemcpyHostToDevice, ctx->stream));
    if ((size_t) pl.n_runs * ASM_PART > pl.part_cap) {
        if (pl.d_part) (void) hipFree(pl.d_part);
        pl.d_part = nullptr, pl.part_cap = 0;
        const size_t cap = (size_t) pl.n_runs * ASM_PART + (size_t) pl.n_runs * ASM_PART / 4;
        ICG_HIP(ctx, hipMalloc((void **) &pl.d_part, sizeof(double) * cap));
        pl.part_cap = cap;
    }
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (the host vectors go out of scope)
    pt.plan_valid = true;
    return ICG_OK;
}

// the implicit partition behind the single-window entry points: every resident factor, landmarks 0 .. n_lm - 1
static int single_partition(icg_ctx *ctx, int n_lm) {
    icg_partition &pt = ctx->part_1;
    const int n       = ctx->n_factors_resident;
    if (pt.plan_valid && pt.W == 1 && pt.fac_off[1] == n && pt.lm_off[1] == n_lm) return ICG_OK;
    pt.W = 1;
    pt.fac_off = {0, n}, pt.lm_off = {0, n_lm};
    pt.sys_valid = 0;
    return asm_plan_build(ctx, pt);
}

static void build_win_desc(const icg_partition &pt, const uint8_t *reassemble, const double *damp, std::vector<win_desc> &out) {
    const int W = pt.W;
    out.resize((size_t) W);
    for (int w = 0; w < W; w++) {
        win_desc &d = out[(size_t) w];
        d.fac_begin = pt.fac_off[(size_t) w], d.fac_end = pt.fac_off[(size_t) w + 1];
        d.lm_begin = pt.lm_off[(size_t) w], d.L = pt.lm_off[(size_t) w + 1] - pt.lm_off[(size_t) w];
        d.sys_off    = pt.sys_off.size() == (size_t) W + 1 ? pt.sys_off[(size_t) w] : 0;
        d.K          = pt.plan_valid ? pt.plan.pose_off[(size_t) w + 1] - pt.plan.pose_off[(size_t) w] : 0;
        d.reassemble = reassemble ? reassemble[w] : 1;
        d.damp       = damp ? damp[w] : (pt.damp.size() == (size_t) W ? pt.damp[(size_t) w] : 0.0);
        d.NB = d.pad = 0;
    }
}

// Assembly (for the windows with reassemble[w] != 0) + landmark elimination of every window of the partition.
// S_view != nullptr: the reduced systems are written by the reduction kernel straight into the context's pinned staging memory (zero-copy)
// and *S_view points there — no device-to-host copy and no 9 MB copy-out per LM step at 256 windows; valid until the next call on ctx.
// S and S_view both null: the reduced systems stay on the device (ctx->d_redS).
static int schur_impl(icg_ctx *ctx, icg_partition &pt, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                      const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, double *S, const double **S_view, bool resident,
                      double *s, double *diag_cc, double *cost) {
    const bool tdbg = getenv("ICG_ABI_DEBUG") != nullptr;
    auto tnow       = [] { return std::chrono::steady_clock::now(); };
    auto t_begin    = tnow();
    const int W = pt.W, n = ctx->n_factors_resident;
    const icg_asm_plan &pl = pt.plan;
    bool any_new = false;
    for (int w = 0; w < W; w++) any_new |= reassemble[w] != 0;
    if (any_new && (!ctx->rJ_valid || !ctx->rJ_has_jac)) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Jacobians: evaluate with want_jac first");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (P > 512) return icg_fail(ctx, ICG_ERR_CAPACITY, "reduced systems of more than 512 camera columns are not supported (%d)", P);
    const int TQ = (P + 3) / 4, NT = TQ * (TQ + 1) / 2;
    const int red_LT     = std::max(1, std::min(SCH_LT, (256 * SCH_PRE) / (4 * TQ)));
    const size_t red_lds = sizeof(double) * ((size_t) red_LT * 4 * TQ + 2 * (size_t) red_LT); // <= 24.5 KB
    if (int rca = rpj_allow_lds(ctx, k_schur_reduce_w, red_lds, 1)) return rca;
    // system layout
    if (!pt.sys_valid || pt.sys_P != P) {
        for (int w = 0; w < W; w++)
            if (!reassemble[w]) return icg_fail(ctx, ICG_ERR_INVALID, "window %d: nothing resident of size %d to re-damp", w, P);
        pt.sys_off.assign((size_t) W + 1, 0);
        for (int w = 0; w < W; w++) {
            const int64_t N = P + (pt.lm_off[(size_t) w + 1] - pt.lm_off[(size_t) w]);
            pt.sys_off[(size_t) w + 1] = pt.sys_off[(size_t) w] + N * N + N + (N - P);
        }
        pt.damp.assign((size_t) W, 0.0);
    }
    int rc = ensure_sys_capacity(ctx, (size_t) pt.sys_off[(size_t) W] + 8);
    if (rc) return rc;
    icg_partition &other = &pt == &ctx->part_1 ? ctx->part_w : ctx->part_1;
    other.sys_valid      = 0; // (d_sys is shared: whatever the other partition left there is overwritten)
    pt.sys_valid         = 0;
    // owner of every camera column of every window (k_asm_camera / k_asm_landmarks)
    std::vector<int16_t> owner((size_t) W * P, (int16_t) -1);
    int Lmax = 1;
    for (int w = 0; w < W; w++) {
        Lmax        = std::max(Lmax, pt.lm_off[(size_t) w + 1] - pt.lm_off[(size_t) w]);
        int16_t *ow = &owner[(size_t) w * P];
        auto claim  = [&](int col, int width, int code, const char *what) -> int {
            if (col < 0) return 0;
            if (col + width > P) return icg_fail(ctx, ICG_ERR_INVALID, "window %d: %s column %d outside the reduced system (%d)", w, what, col, P);
            for (int x = 0; x < width; x++) {
                if (ow[col + x] != -1) return icg_fail(ctx, ICG_ERR_INVALID, "window %d: camera column %d is claimed by two blocks", w, col + x);
                ow[col + x] = (int16_t) ((code << 3) | (code == ASM_EXT && width == 1 ? 6 : x));
            }
            return 0;
        };
        for (int k = pl.pose_off[(size_t) w]; k < pl.pose_off[(size_t) w + 1]; k++) {
            const int g = pl.pose_glob[(size_t) k];
            if (g >= ctx->last_n_poses) return icg_fail(ctx, ICG_ERR_INVALID, "pose %d of the factors is beyond the %d evaluated poses", g, ctx->last_n_poses);
            if ((rc = claim(col_pose[g], 6, k - pl.pose_off[(size_t) w], "pose"))) return rc;
        }
        if ((rc = claim(col_ext[w], 6, ASM_EXT, "extrinsic"))) return rc;
        if ((rc = claim(col_td[w], 1, ASM_EXT, "td"))) return rc;
    }
    for (int w = 0; w < W; w++)
        if (reassemble[w] || damp[w] != pt.damp[(size_t) w]) pt.damp[(size_t) w] = damp[w];
    std::vector<win_desc> wd;
    build_win_desc(pt, reassemble, damp, wd);
    // column blocks of the landmark rows: owned blocks start where their owner's column 0 sits, unowned columns in runs of up to six
    std::vector<std::vector<int32_t>> blk((size_t) W);
    int NBmax = 1;
    for (int w = 0; w < W; w++) {
        const int16_t *ow = &owner[(size_t) w * P];
        std::vector<int32_t> &B = blk[(size_t) w];
        for (int a = 0; a < P;) {
            const int o = ow[a];
            if (o < 0) {
                int wdt = 1;
                while (a + wdt < P && wdt < 6 && ow[a + wdt] < 0) wdt++;
                B.push_back(a | (wdt << 12) | (ASM_BLK_GAP << 16));
                a += wdt;
            } else if ((o >> 3) == ASM_EXT && (o & 7) == 6) {
                a += 1; // td: part of the (td, h_ll, b_l) block below
            } else {
                B.push_back(a | (6 << 12) | ((o >> 3) << 16)); // (claim() laid the six columns of a pose / the extrinsic down contiguously)
                a += 6;
            }
        }
        B.push_back((col_td[w] >= 0 ? col_td[w] : 0xFFF) | (1 << 12) | (ASM_BLK_TD << 16));
        wd[(size_t) w].NB = (int32_t) B.size();
        NBmax             = std::max(NBmax, (int) B.size());
    }
    if (NBmax > 256) return icg_fail(ctx, ICG_ERR_CAPACITY, "a window's camera columns fall into %d blocks (limit 256)", NBmax);
    std::vector<int32_t> blocks((size_t) W * NBmax, 0);
    for (int w = 0; w < W; w++) std::copy(blk[(size_t) w].begin(), blk[(size_t) w].end(), blocks.begin() + (size_t) w * NBmax);
    icg_call c(ctx);
    rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(int16_t) * owner.size() + sizeof(int32_t) * blocks.size() + (size_t) n +
                   sizeof(double) * ((size_t) W * ((size_t) P * P + 2 * (size_t) P + 1)) + 8192);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const int16_t *d_own = c.in(owner.data(), owner.size());
    const int32_t *d_blk = c.in(blocks.data(), blocks.size());
    const uint8_t *d_act = active ? c.in(active, (size_t) n) : nullptr;
    auto t_prep = tnow();
    if ((rc = c.seal())) return rc;
    // (the zero-copy region is allocated LAST: finish() copies ONE device range back that spans all mirrored outputs, and must not run
    // over memory the kernel wrote through the host mapping)
    double *d_cost = c.out(any_new ? cost : (double *) nullptr, (size_t) W);
    double *d_S    = (S_view || resident) ? nullptr : c.out(S, (size_t) W * P * P);
    double *d_s    = c.out(s, (size_t) W * P);
    double *d_dg   = c.out(diag_cc, (size_t) W * P); // user pointer may be null: still a valid device scratch
    if (S_view) {
        d_S     = c.out_zc((double *) nullptr, (size_t) W * P * P);
        *S_view = d_S;
    }
    if (resident) {
        if ((size_t) W * P * P > ctx->redS_cap) {
            ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_redS) (void) hipFree(ctx->d_redS);
            ctx->d_redS = nullptr, ctx->redS_cap = 0;
            ICG_HIP(ctx, hipMalloc((void **) &ctx->d_redS, sizeof(double) * (size_t) W * P * P));
            ctx->redS_cap = (size_t) W * P * P;
        }
        d_S = ctx->d_redS;
    }
    const double *d_r = ctx->d_rJ, *d_J = ctx->d_rJ + 2 * (size_t) ctx->factors_cap;
    ICG_LAUNCH_GUARD(c);
    if (any_new) {
        icg_prof_scope ps(ctx, "reproj_normal");
        if (pl.n_runs > 0)
            hipLaunchKernelGGL(k_asm_runs, dim3((unsigned) ((pl.n_runs + 3) / 4)), dim3(256), 0, ctx->stream, pl.n_runs, reinterpret_cast<const int4 *>(pl.d_runs),
                               d_wd, (const int32_t *) pl.d_perm, d_r, d_J, d_act, pl.d_part);
        hipLaunchKernelGGL(k_asm_camera, dim3((unsigned) ((P * P + P + 255) / 256), W), dim3(256), 0, ctx->stream, d_wd, (const int32_t *) pl.d_run_off,
                           (const int32_t *) pl.d_pair_run, pl.Kmax, d_own, P, (const double *) pl.d_part, ctx->d_sys);
        const int LB = std::max(1, 256 / NBmax); // landmarks per workgroup: one (landmark, block) pair per thread
        hipLaunchKernelGGL(k_asm_landmarks, dim3((unsigned) ((Lmax + LB - 1) / LB), W), dim3(256), 0, ctx->stream, d_wd, d_blk, NBmax, P, LB,
                           (const int32_t *) pl.d_lm_foff, reinterpret_cast<const int4 *>(pl.d_lrec), d_r, d_J, d_act, ctx->d_sys);
    }
    {
        icg_prof_scope ps(ctx, "schur_reduce");
        hipLaunchKernelGGL(k_schur_inv_w, dim3((Lmax + 255) / 256, W), dim3(256), 0, ctx->stream, d_wd, P, ctx->d_sys, min_diag, max_diag);
        hipLaunchKernelGGL(k_schur_reduce_w, dim3((unsigned) ((NT + 255) / 256), W), dim3(256), red_lds, ctx->stream, d_wd, P, red_LT, (const double *) ctx->d_sys, d_S, d_s,
                           d_dg, (S_view || resident) ? 1 : 0);
        // the cost belongs to the linearization point: only meaningful while the resident residuals are the ones assembled
        if (any_new) hipLaunchKernelGGL(k_reproj_cost_w, dim3(W), dim3(256), 0, ctx->stream, d_wd, d_r, d_act, ctx->last_huber, d_cost);
    }
    ICG_HIP(ctx, hipGetLastError());
    auto t_launch = tnow();
    if (tdbg) (void) hipStreamSynchronize(ctx->stream);
    auto t_kernels = tnow();
    if ((rc = c.finish())) return rc;
    if (tdbg) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[icg_reproj_schur] W=%d: host prep %.3f, h2d+launch %.3f, kernels %.3f, d2h+copy-out %.3f ms\n", W, ms(t_begin, t_prep),
                ms(t_prep, t_launch), ms(t_launch, t_kernels), ms(t_kernels, tnow()));
    }
    pt.sys_P = P, pt.sys_valid = 1;
    ctx->sys_min_diag = min_diag, ctx->sys_max_diag = max_diag;
    return ICG_OK;
}

static int backsub_impl(icg_ctx *ctx, icg_partition &pt, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    const int W = pt.W, n_lm = pt.lm_off[(size_t) W];
    if (n_lm == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<win_desc> wd;
    build_win_desc(pt, nullptr, nullptr, wd);
    icg_call c(ctx);
    int rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(double) * ((size_t) W * P + 3 * (size_t) n_lm + 2 * (size_t) W) + 4096);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const double *d_dc   = c.in(delta_c, (size_t) W * P);
    if ((rc = c.seal())) return rc;
    double *d_dl = c.out(delta_l, (size_t) n_lm);
    double *d_tm = c.out(lm_terms, 2 * (size_t) W);
    double *d_lt = c.out((double *) nullptr, 2 * (size_t) n_lm);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_backsub");
        hipLaunchKernelGGL(k_schur_backsub_w, dim3(n_lm), dim3(64), 0, ctx->stream, d_wd, W > 1 ? (const int32_t *) ctx->d_lmwin : (const int32_t *) nullptr, 0, P,
                           (const double *) ctx->d_sys, d_dc, d_dl, d_lt, ctx->sys_min_diag, ctx->sys_max_diag);
        hipLaunchKernelGGL(k_terms_reduce_w, dim3(W), dim3(256), 0, ctx->stream, d_wd, (const double *) d_lt, d_tm);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

static int cost_impl(icg_ctx *ctx, icg_partition &pt, const uint8_t *active, double *cost) {
    const int W = pt.W, n = ctx->n_factors_resident;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<win_desc> wd;
    build_win_desc(pt, nullptr, nullptr, wd);
    icg_call c(ctx);
    int rc = c.reserve(sizeof(win_desc) * (size_t) W + (size_t) n + sizeof(double) * (size_t) W + 4096);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const uint8_t *d_act = active ? c.in(active, (size_t) n) : nullptr;
    if ((rc = c.seal())) return rc;
    double *d_cost = c.out(cost, (size_t) W);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_cost");
        hipLaunchKernelGGL(k_reproj_cost_w, dim3(W), dim3(256), 0, ctx->stream, d_wd, (const double *) ctx->d_rJ, d_act, ctx->last_huber, d_cost);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

// h_ll of every landmark (global landmark order of the partition) from the systems left resident by the last assembly: the diagonal
// element (P + l, P + l) of each window's block
__global__ void k_lm_diag_w(const win_desc *wd, int P, const double *sys, double *h_ll) {
    const win_desc W = wd[blockIdx.y];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= W.L) return;
    const size_t N = (size_t) P + W.L;
    h_ll[W.lm_begin + l] = sys[W.sys_off + (size_t) (P + l) * N + P + l];
}

static int landmark_diag_impl(icg_ctx *ctx, icg_partition &pt, double *h_ll) {
    const int W = pt.W, P = pt.sys_P, n_lm = pt.lm_off[(size_t) W];
    if (n_lm == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<win_desc> wd;
    build_win_desc(pt, nullptr, nullptr, wd);
    int Lmax = 1;
    for (int w = 0; w < W; w++) Lmax = std::max(Lmax, (int) wd[(size_t) w].L);
    icg_call c(ctx);
    int rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(double) * (size_t) n_lm + 4096);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    if ((rc = c.seal())) return rc;
    double *d_out = c.out(h_ll, (size_t) n_lm);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_reduce");
        hipLaunchKernelGGL(k_lm_diag_w, dim3((Lmax + 255) / 256, W), dim3(256), 0, ctx->stream, d_wd, P, (const double *) ctx->d_sys, d_out);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

// ---- single window: every resident factor ----------------------------------------------------------------------------------------------
extern "C" int icg_reproj_schur(icg_ctx *ctx, int P, const int32_t *col_pose, int32_t col_ext, int32_t col_td, const uint8_t *active,
                                int reassemble, double damp, double min_diag, double max_diag, double *S, double *s, double *diag_cc,
                                double *cost) {
    if (!ctx || P <= 0 || !col_pose || !S || !s) return ICG_ERR_INVALID;
    if (ctx->n_factors_resident == 0) return icg_fail(ctx, ICG_ERR_INVALID, "no resident factors");
    if (reassemble && (!ctx->rJ_valid || !ctx->rJ_has_jac))
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident Jacobians: call icg_reproj_eval_resident with want_jac first");
    if (!reassemble && (!ctx->part_1.sys_valid || ctx->part_1.sys_P != P))
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident normal equations of size %d to re-damp", P);
    int rc = single_partition(ctx, ctx->last_n_lm);
    if (rc) return rc;
    const uint8_t re = reassemble ? 1 : 0;
    return schur_impl(ctx, ctx->part_1, P, col_pose, &col_ext, &col_td, active, &re, &damp, min_diag, max_diag, S, nullptr, false, s, diag_cc,
                      reassemble ? cost : nullptr);
}

extern "C" int icg_reproj_landmark_diag(icg_ctx *ctx, double *h_ll) {
    if (!ctx || !h_ll) return ICG_ERR_INVALID;
    if (!ctx->part_1.sys_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Schur system: call icg_reproj_schur first");
    return landmark_diag_impl(ctx, ctx->part_1, h_ll);
}

extern "C" int icg_reproj_backsub(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    if (!ctx || !delta_c || !delta_l) return ICG_ERR_INVALID;
    if (!ctx->part_1.sys_valid || ctx->part_1.sys_P != P) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Schur system of size %d: call icg_reproj_schur first", P);
    return backsub_impl(ctx, ctx->part_1, P, delta_c, delta_l, lm_terms);
}

extern "C" int icg_reproj_cost(icg_ctx *ctx, const uint8_t *active, double *cost) {
    if (!ctx || !cost) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident residuals: call icg_reproj_eval_resident first");
    *cost = 0.0;
    if (ctx->n_factors_resident == 0) return ICG_OK;
    // (the cost needs the factor range only: a one-window descriptor without a plan)
    icg_partition &pt = ctx->part_1;
    if (pt.W != 1 || pt.fac_off.size() != 2 || pt.fac_off[1] != ctx->n_factors_resident) {
        pt.W = 1, pt.fac_off = {0, ctx->n_factors_resident}, pt.lm_off = {0, ctx->last_n_lm};
        pt.plan_valid = false, pt.sys_valid = 0;
    }
    return cost_impl(ctx, pt, active, cost);
}

// M2 for a caller-defined dense layout (MarginalizationInfo::constructEquation, factors/marginalization_info.h:195-230): the system is
// assembled in the compact layout above (free poses in pose order, then extrinsic, then td; landmark l in row V + l) by the same kernels
// and spread into the caller's local_size x local_size matrix on the host — every cell has one source, no sum is formed there.
extern "C" int icg_reproj_accumulate_normal(icg_ctx *ctx, int local_size, const int32_t *col_pose, int32_t col_ext,
                                            const int32_t *col_lm, int32_t col_td, double *H0, double *b0) {
    if (!ctx || local_size <= 0 || !col_pose || !col_lm || !H0 || !b0) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid || !ctx->rJ_has_jac) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Jacobians: call icg_reproj_eval_* with want_jac first");
    const int n = ctx->n_factors_resident, L = ctx->last_n_lm;
    if (n == 0) return ICG_OK;
    auto inside = [&](int col, int width) { return col < 0 || col + width <= local_size; };
    std::vector<int32_t> vcol((size_t) ctx->last_n_poses, -1), vmap;
    for (int k = 0; k < ctx->last_n_poses; k++)
        if (col_pose[k] >= 0) {
            if (!inside(col_pose[k], 6)) return icg_fail(ctx, ICG_ERR_INVALID, "pose %d: column %d outside the system (%d)", k, col_pose[k], local_size);
            vcol[(size_t) k] = (int32_t) vmap.size();
            for (int x = 0; x < 6; x++) vmap.push_back(col_pose[k] + x);
        }
    if (!inside(col_ext, 6) || !inside(col_td, 1)) return icg_fail(ctx, ICG_ERR_INVALID, "ext/td column outside the system (%d)", local_size);
    int vext = -1, vtd = -1;
    if (col_ext >= 0) {
        vext = (int) vmap.size();
        for (int x = 0; x < 6; x++) vmap.push_back(col_ext + x);
    }
    if (col_td >= 0) vtd = (int) vmap.size(), vmap.push_back(col_td);
    for (int l = 0; l < L; l++)
        if (!inside(col_lm[l], 1)) return icg_fail(ctx, ICG_ERR_INVALID, "landmark %d: column %d outside the system (%d)", l, col_lm[l], local_size);
    const int V = std::max(1, (int) vmap.size());
    int rc = single_partition(ctx, L);
    if (rc) return rc;
    const size_t N = (size_t) V + L;
    std::vector<double> S((size_t) V * V), s((size_t) V), sys(N * N + N);
    const uint8_t re  = 1;
    const double zero = 0.0;
    if ((rc = schur_impl(ctx, ctx->part_1, V, vcol.data(), &vext, &vtd, nullptr, &re, &zero, 0.0, 0.0, S.data(), nullptr, false, s.data(), nullptr, nullptr))) return rc;
    ICG_HIP(ctx, hipMemcpyAsync(sys.data(), ctx->d_sys, sizeof(double) * (N * N + N), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = icg_stream_wait(ctx))) return rc;
    ctx->part_1.sys_valid = 0; // (a by-product: not a system the Schur entry points may re-damp)
    const size_t LS = (size_t) local_size;
    const double *H = sys.data(), *b = H + N * N;
    for (size_t a = 0; a < vmap.size(); a++) {
        for (size_t c = 0; c < vmap.size(); c++) H0[(size_t) vmap[a] * LS + (size_t) vmap[c]] += H[a * N + c];
        b0[(size_t) vmap[a]] += b[a];
    }
    for (int l = 0; l < L; l++) {
        if (col_lm[l] < 0) continue;
        const size_t cl = (size_t) col_lm[l], row = ((size_t) V + (size_t) l) * N;
        for (size_t a = 0; a < vmap.size(); a++) {
            H0[cl * LS + (size_t) vmap[a]] += H[row + a];
            H0[(size_t) vmap[a] * LS + cl] += H[row + a];
        }
        H0[cl * LS + cl] += H[row + (size_t) V + (size_t) l];
        b0[cl] += b[(size_t) V + (size_t) l];
    }
    return ICG_OK;
}

// ---- f1, many windows per launch ------------------------------------------------------------------------------------------------
// One solver in flight per stream is bounded by the runtime's rate of small launches and copies (~100 per window and solve, DESIGN.md
// §6).  Here the windows of many streams advance in lock-step: ONE evaluation, ONE assembly, ONE reduction, ONE back-substitution
// call per LM step for all of them.  The resident factor set is partitioned into W windows (factors sorted by window, landmarks
// contiguous per window, poses indexed globally); every window has its own extrinsic / td, its own reduced system of the common
// size P and its own damping.  Window w's system lives at d_sys + sys_off[w]: H (N_w x N_w, N_w = P + L_w) | b (N_w) | inv (L_w).
extern "C" int icg_reproj_set_windows(icg_ctx *ctx, int n_windows, const int32_t *fac_off, const int32_t *lm_off) {
    if (!ctx || n_windows <= 0 || !fac_off || !lm_off) return ICG_ERR_INVALID;
    const int n = ctx->n_factors_resident;
    if (fac_off[0] != 0 || fac_off[n_windows] != n) return icg_fail(ctx, ICG_ERR_INVALID, "fac_off must cover the %d resident factors", n);
    if (lm_off[0] != 0) return icg_fail(ctx, ICG_ERR_INVALID, "lm_off must start at 0");
    for (int w = 0; w < n_windows; w++)
        if (fac_off[w + 1] < fac_off[w] || lm_off[w + 1] < lm_off[w]) return icg_fail(ctx, ICG_ERR_INVALID, "window %d: offsets not monotone", w);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int n_lm = lm_off[n_windows];
    if (n_lm > ctx->lmwin_cap) {
        ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_lmwin) (void) hipFree(ctx->d_lmwin);
        ctx->d_lmwin   = nullptr;
        ctx->lmwin_cap = 0;
        ICG_HIP(ctx, hipMalloc((void **) &ctx->d_lmwin, sizeof(int32_t) * (size_t) (n_lm + n_lm / 4 + 64)));
        ctx->lmwin_cap = n_lm + n_lm / 4 + 64;
    }
    std::vector<int32_t> fwin((size_t) n), lwin((size_t) std::max(n_lm, 1));
    for (int w = 0; w < n_windows; w++) {
        for (int f = fac_off[w]; f < fac_off[w + 1]; f++) fwin[(size_t) f] = w;
        for (int l = lm_off[w]; l < lm_off[w + 1]; l++) lwin[(size_t) l] = w;
    }
    if (n) ICG_HIP(ctx, hipMemcpyAsync(ctx->d_fwin, fwin.data(), sizeof(int32_t) * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    if (n_lm) ICG_HIP(ctx, hipMemcpyAsync(ctx->d_lmwin, lwin.data(), sizeof(int32_t) * (size_t) n_lm, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    icg_partition &pt = ctx->part_w;
    pt.W              = n_windows;
    pt.fac_off.assign(fac_off, fac_off + n_windows + 1);
    pt.lm_off.assign(lm_off, lm_off + n_windows + 1);
    pt.sys_valid = 0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc       = asm_plan_build(ctx, pt);
    if (rc) pt.W = 0;
    if (getenv("ICG_ABI_DEBUG"))
        fprintf(stderr, "[icg_reproj_set_windows] W=%d n=%d: assembly plan %.3f ms\n", n_windows, n,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

extern "C" int icg_reproj_eval_windows(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth,
                                       const double *td, int want_jac, double huber_delta) {
    if (!ctx || !poses || !ext || !invdepth || !td || n_poses <= 0 || n_lm <= 0) return ICG_ERR_INVALID;
    if (ctx->part_w.W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    if (n_lm != ctx->part_w.lm_off[(size_t) ctx->part_w.W]) return icg_fail(ctx, ICG_ERR_INVALID, "n_lm does not match the window partition");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int n = ctx->n_factors_resident, W = ctx->part_w.W;
    if (n == 0) return ICG_OK;
    icg_call c(ctx);
    int rc = c.reserve(sizeof(double) * ((size_t) n_poses * 7 + 8 * (size_t) W + (size_t) n_lm) + 4096);
    if (rc) return rc;
    rpj_args A;
    A.n           = n;
    A.obs         = ctx->d_obs;
    A.idx_i       = ctx->d_fidx;
    A.idx_j       = ctx->d_fidx + n;
    A.idx_lm      = ctx->d_fidx + 2 * (size_t) n;
    A.poses       = c.in(poses, 7 * (size_t) n_poses);
    A.ext         = c.in(ext, 7 * (size_t) W);
    A.invdepth    = c.in(invdepth, (size_t) n_lm);
    A.tdv         = c.in(td, (size_t) W);
    A.td          = 0.0;
    A.win         = ctx->d_fwin;
    A.want_jac    = want_jac;
    A.huber_delta = huber_delta;
    A.out_r       = ctx->d_rJ;
    A.out_J       = ctx->d_rJ + 2 * (size_t) ctx->factors_cap;
    if ((rc = c.seal())) return rc;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_eval");
        hipLaunchKernelGGL(k_reproj_eval, dim3((n + RPJ_TILE - 1) / RPJ_TILE), dim3(RPJ_TILE), 0, ctx->stream, A);
    }
    ICG_HIP(ctx, hipGetLastError());
    ctx->rJ_valid     = 1;
    ctx->rJ_has_jac   = want_jac;
    ctx->last_huber   = huber_delta;
    ctx->last_n_poses = n_poses;
    ctx->last_n_lm    = n_lm;
    return c.finish_async(); // nothing comes back: the assembly / cost call that follows is stream-ordered behind the evaluation
}

static int windows_args_ok(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *reassemble,
                           const double *damp, double *s) {
    if (!ctx || P <= 0 || !col_pose || !col_ext || !col_td || !reassemble || !damp || !s) return ICG_ERR_INVALID;
    if (ctx->part_w.W <= 0 || !ctx->part_w.plan_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    return ICG_OK;
}

// Problem-setup companion of the batched calls: sizes the resident window systems (W x ((P + L_w)^2 + ...) doubles of device memory) and the
// staging arena of the largest per-step call for reduced systems of size P, so that the first LM step of a solve does not pay a device
// allocation and a pinned re-allocation (4 ms at 256 C2 windows).  A hint: the calls themselves still grow what they need.
extern "C" int icg_reproj_reserve_windows(icg_ctx *ctx, int P) {
    if (!ctx || P <= 0) return ICG_ERR_INVALID;
    const icg_partition &pt = ctx->part_w;
    const int W = pt.W, n = ctx->n_factors_resident;
    if (W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    size_t doubles = 0;
    for (int w = 0; w < W; w++) {
        const size_t N = (size_t) P + (size_t) (pt.lm_off[(size_t) w + 1] - pt.lm_off[(size_t) w]);
        doubles += N * N + N + (N - (size_t) P);
    }
    int rc = ensure_sys_capacity(ctx, doubles + 8);
    if (rc) return rc;
    icg_call c(ctx);
    return c.reserve(sizeof(win_desc) * (size_t) W + (sizeof(int16_t) + sizeof(int32_t)) * (size_t) W * (size_t) P + (size_t) n +
                     sizeof(double) * ((size_t) W * ((size_t) P * P + 2 * (size_t) P + 1)) + 8192);
}

extern "C" int icg_reproj_schur_windows(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                        const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag, double max_diag,
                                        double *S, double *s, double *diag_cc, double *cost) {
    if (!S) return ICG_ERR_INVALID;
    if (int rc = windows_args_ok(ctx, P, col_pose, col_ext, col_td, reassemble, damp, s)) return rc;
    return schur_impl(ctx, ctx->part_w, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, S, nullptr, false, s, diag_cc, cost);
}

extern "C" int icg_reproj_schur_windows_view(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                             const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag,
                                             double max_diag, const double **S_view, double *s, double *diag_cc, double *cost) {
    if (!S_view) return ICG_ERR_INVALID;
    if (int rc = windows_args_ok(ctx, P, col_pose, col_ext, col_td, reassemble, damp, s)) return rc;
    return schur_impl(ctx, ctx->part_w, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, nullptr, S_view, false, s, diag_cc, cost);
}

extern "C" int icg_reproj_schur_windows_resident(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                                 const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag,
                                                 double max_diag, double *s, double *diag_cc, double *cost) {
    if (int rc = windows_args_ok(ctx, P, col_pose, col_ext, col_td, reassemble, damp, s)) return rc;
    return schur_impl(ctx, ctx->part_w, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, nullptr, nullptr, true, s, diag_cc, cost);
}

// ---- reduced systems solved on the device -------------------------------------------------------------------------------------------
// packed lower triangle: row i of a P x P symmetric matrix occupies entries i(i+1)/2 .. i(i+1)/2 + i
__global__ __launch_bounds__(256) void k_hostS_scatter(int tri, const int32_t *win_idx, const double *packed, double *hostS) {
    const double *src = packed + (size_t) blockIdx.x * tri;
    double *dst       = hostS + (size_t) win_idx[blockIdx.x] * tri;
    for (int k = threadIdx.x; k < tri; k += 256) dst[k] = src[k];
}

// One workgroup per window: A = lower(S_w + hostS_w) + diag(dd_w) in LDS, right-looking Cholesky (the whole trailing update of a column
// step spread over the 256 threads), forward substitution by wave 0 (a wave reduction per row), backward substitution as a column
// sweep.  ok[w] = 0 when a pivot is not positive / finite (dense_kernels.cc choleskySolve returns false there: the caller re-damps).
// Columns >= Pw[w] of a window are empty (zero rows): only the leading Pw x Pw block is factored, the rest of delta_c is zero.
__global__ __launch_bounds__(256) void k_chol_solve_w(int P, const int32_t *Pw, const uint8_t *stepped, const double *S, const double *hostS,
                                                      const double *rhs, const double *dd, double *delta_c, uint8_t *ok) {
    extern __shared__ double sh[];
    const int w = blockIdx.x, t = threadIdx.x, n = Pw[w];
    double *A = sh;          // n x n, row stride n (lower triangle used)
    double *b = sh + P * P;  // n
    __shared__ int failed;
    double *dcw = delta_c + (size_t) w * P;
    for (int k = t; k < P; k += 256) dcw[k] = 0.0;
    if (!stepped[w] || n <= 0) {
        if (t == 0) ok[w] = 0;
        return;
    }
    const double *Sw = S + (size_t) w * P * P, *Hw = hostS + (size_t) w * ((size_t) P * (P + 1) / 2);
    for (int e = t; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        if (j <= i) A[i * n + j] = Sw[(size_t) i * P + j] + Hw[(size_t) i * (i + 1) / 2 + j] + (i == j ? dd[(size_t) w * P + i] : 0.0);
    }
    for (int k = t; k < n; k += 256) b[k] = rhs[(size_t) w * P + k];
    if (t == 0) failed = 0;
    __syncthreads();
    for (int k = 0; k < n; k++) {
        const double d = A[k * n + k];
        if (!(d > 0.0) || !isfinite(d)) {
            if (t == 0) failed = 1;
            break; // uniform: every thread reads the same pivot
        }
        const double piv = sqrt(d);
        __syncthreads(); // everybody has read the pivot before it is overwritten
        for (int i = k + t; i < n; i += 256) A[i * n + k] = (i == k) ? piv : A[i * n + k] / piv;
        __syncthreads();
        // trailing update of the lower triangle: A[i][j] -= L[i][k] * L[j][k], k < j <= i < n
        const int m = n - k - 1;
        for (int e = t; e < m * m; e += 256) {
            const int a = e / m, c = e - a * m;
            if (c <= a) A[(k + 1 + a) * n + (k + 1 + c)] -= A[(k + 1 + a) * n + k] * A[(k + 1 + c) * n + k];
        }
        __syncthreads();
    }
    __syncthreads();
    if (failed) {
        if (t == 0) ok[w] = 0;
        return;
    }
    if (t < 64) { // wave 0: L y = b, then L^T x = y
        for (int r = 0; r < n; r++) {
            double acc = 0.0;
            for (int k = t; k < r; k += 64) acc += A[r * n + k] * b[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (t == 0) b[r] = (b[r] - acc) / A[r * n + r];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        for (int r = n - 1; r >= 0; r--) {
            const double x = b[r] / A[r * n + r];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (t == 0) b[r] = x;
            for (int k = t; k < r; k += 64) b[k] -= A[r * n + k] * x;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        for (int k = t; k < n; k += 64) dcw[k] = b[k];
        if (t == 0) ok[w] = 1;
    }
}

// packed[k]: the lower triangle (P(P+1)/2 doubles, row by row) of the host factors' contribution to the reduced system of window
// win_idx[k]; stays on the device until replaced (a re-damped step re-uses it)
extern "C" int icg_reproj_set_host_part_windows(icg_ctx *ctx, int P, int n_upd, const int32_t *win_idx, const double *packed) {
    if (!ctx || P <= 0 || n_upd < 0 || (n_upd > 0 && (!win_idx || !packed))) return ICG_ERR_INVALID;
    const int W = ctx->part_w.W;
    if (W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t tri = (size_t) P * (P + 1) / 2;
    if (ctx->hostS_P != P || ctx->hostS_W != W || (size_t) W * tri > ctx->hostS_cap) {
        ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((size_t) W * tri > ctx->hostS_cap) {
            if (ctx->d_hostS) (void) hipFree(ctx->d_hostS);
            ctx->d_hostS = nullptr, ctx->hostS_cap = 0;
            ICG_HIP(ctx, hipMalloc((void **) &ctx->d_hostS, sizeof(double) * (size_t) W * tri));
            ctx->hostS_cap = (size_t) W * tri;
        }
        ICG_HIP(ctx, hipMemsetAsync(ctx->d_hostS, 0, sizeof(double) * (size_t) W * tri, ctx->stream));
        ctx->hostS_P = P, ctx->hostS_W = W;
    }
    if (n_upd == 0) return ICG_OK;
    for (int k = 0; k < n_upd; k++)
        if (win_idx[k] < 0 || win_idx[k] >= W) return icg_fail(ctx, ICG_ERR_INVALID, "window index out of range");
    icg_call c(ctx);
    int rc = c.reserve(sizeof(int32_t) * (size_t) n_upd + sizeof(double) * (size_t) n_upd * tri + 4096);
    if (rc) return rc;
    const int32_t *d_wi = c.in(win_idx, (size_t) n_upd);
    const double *d_pk  = c.in(packed, (size_t) n_upd * tri);
    if ((rc = c.seal())) return rc;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_host_part");
        hipLaunchKernelGGL(k_hostS_scatter, dim3(n_upd), dim3(256), 0, ctx->stream, (int) tri, d_wi, d_pk, ctx->d_hostS);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

// For every window with stepped[w] != 0: (S_w + hostS_w + diag(dd_w)) delta_c_w = rhs_w on the leading Pw[w] columns (batched Cholesky in
// LDS: P^2 + P doubles per window have to fit one workgroup's LDS — P <= 142 on gfx950), then the landmark back-substitution of
// icg_reproj_backsub_windows with those steps, in one call: per LM step only rhs, dd go up and delta_c, ok, delta_l and the two
// model-decrease sums come back; the P x P systems never leave the device.
extern "C" int icg_reproj_solve_backsub_windows(icg_ctx *ctx, int P, const int32_t *Pw, const uint8_t *stepped, const double *rhs,
                                                const double *dd, double *delta_c, uint8_t *ok, double *delta_l, double *lm_terms) {
    if (!ctx || P <= 0 || !Pw || !stepped || !rhs || !dd || !delta_c || !ok) return ICG_ERR_INVALID;
    icg_partition &pt = ctx->part_w;
    if (!pt.sys_valid || pt.sys_P != P || !ctx->d_redS)
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident reduced systems of size %d: call icg_reproj_schur_windows_resident first", P);
    const int W = pt.W, n_lm = pt.lm_off[(size_t) W];
    const size_t lds = sizeof(double) * ((size_t) P * P + (size_t) P);
    if (lds > rpj_lds_limit(ctx))
        return icg_fail(ctx, ICG_ERR_CAPACITY, "reduced system of %d columns does not fit the LDS tile of the batched Cholesky (%zu bytes per workgroup)", P, rpj_lds_limit(ctx));
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (int rca = rpj_allow_lds(ctx, k_chol_solve_w, lds, 2)) return rca;
    int rc = icg_reproj_set_host_part_windows(ctx, P, 0, nullptr, nullptr); // (allocates a zero host part if none was ever set)
    if (rc) return rc;
    std::vector<win_desc> wd;
    build_win_desc(pt, nullptr, nullptr, wd);
    icg_call c(ctx);
    rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(int32_t) * (size_t) W + 2 * (size_t) W + sizeof(double) * (3 * (size_t) W * P + 3 * (size_t) n_lm + 2 * (size_t) W) + 8192);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const int32_t *d_pw  = c.in(Pw, (size_t) W);
    const uint8_t *d_st  = c.in(stepped, (size_t) W);
    const double *d_rhs  = c.in(rhs, (size_t) W * P);
    const double *d_dd   = c.in(dd, (size_t) W * P);
    if ((rc = c.seal())) return rc;
    const bool want_l = n_lm > 0 && delta_l;
    double *d_dl   = want_l ? c.out(delta_l, (size_t) n_lm) : nullptr;
    double *d_tm   = c.out(want_l ? lm_terms : (double *) nullptr, 2 * (size_t) W);
    double *d_dco  = c.out(delta_c, (size_t) W * P);
    uint8_t *d_ok  = c.out(ok, (size_t) W);
    double *d_lt   = want_l ? c.out((double *) nullptr, 2 * (size_t) n_lm) : nullptr;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_cholesky");
        hipLaunchKernelGGL(k_chol_solve_w, dim3(W), dim3(256), lds, ctx->stream, P, d_pw, d_st, (const double *) ctx->d_redS, (const double *) ctx->d_hostS,
                           d_rhs, d_dd, d_dco, d_ok);
    }
    if (want_l) {
        icg_prof_scope ps(ctx, "schur_backsub");
        hipLaunchKernelGGL(k_schur_backsub_w, dim3(n_lm), dim3(64), 0, ctx->stream, d_wd, (const int32_t *) ctx->d_lmwin, 0, P, (const double *) ctx->d_sys,
                           (const double *) d_dco, d_dl, d_lt, ctx->sys_min_diag, ctx->sys_max_diag);
        hipLaunchKernelGGL(k_terms_reduce_w, dim3(W), dim3(256), 0, ctx->stream, d_wd, (const double *) d_lt, d_tm);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_reproj_landmark_diag_windows(icg_ctx *ctx, double *h_ll) {
    if (!ctx || !h_ll) return ICG_ERR_INVALID;
    if (!ctx->part_w.sys_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident window systems: call icg_reproj_schur_windows first");
    return landmark_diag_impl(ctx, ctx->part_w, h_ll);
}

extern "C" int icg_reproj_backsub_windows(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    if (!ctx || !delta_c || !delta_l) return ICG_ERR_INVALID;
    if (!ctx->part_w.sys_valid || ctx->part_w.sys_P != P)
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident window systems of size %d: call icg_reproj_schur_windows first", P);
    return backsub_impl(ctx, ctx->part_w, P, delta_c, delta_l, lm_terms);
}

extern "C" int icg_reproj_cost_windows(icg_ctx *ctx, const uint8_t *active, double *cost) {
    if (!ctx || !cost) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident residuals: call icg_reproj_eval_windows first");
    if (ctx->part_w.W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition");
    return cost_impl(ctx, ctx->part_w, active, cost);
}

// the resident residuals of the last evaluation (n x 2), e.g. for the per-factor chi-square test after icg_reproj_eval_windows
__global__ void k_reproj_chi2(int n, const double *r, double chi2, const uint8_t *active_in, uint8_t *active_out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const double r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
    const double cost = 0.5 * (r0 * r0 + r1 * r1); // EvaluateResidualBlock(id, false, &cost, ...) (ic_gvins.cc:1278)
    active_out[f]     = (active_in[f] && !(cost * 2.0 > chi2)) ? 1 : 0;
}

extern "C" int icg_reproj_chi2_cull(icg_ctx *ctx, double chi2, uint8_t *active) {
    if (!ctx || !active) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident residuals");
    const int n = ctx->n_factors_resident;
    if (n == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    int rc = c.reserve(2 * (size_t) n + 4096);
    if (rc) return rc;
    const uint8_t *d_in = c.in_zc(active, (size_t) n);
    uint8_t *d_out      = c.out_zc(active, (size_t) n);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_chi2");
        hipLaunchKernelGGL(k_reproj_chi2, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, (const double *) ctx->d_rJ, chi2, d_in, d_out);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_reproj_fetch_residuals(icg_ctx *ctx, double *out_r) {
    if (!ctx || !out_r) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident residuals");
    const int n = ctx->n_factors_resident;
    if (n == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ICG_HIP(ctx, hipMemcpyAsync(out_r, ctx->d_rJ, sizeof(double) * 2 * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
    return icg_stream_wait(ctx);
}
