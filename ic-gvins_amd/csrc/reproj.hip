// R1/R2: batched ReprojectionFactor::Evaluate (+ ResidualBlockInfo robust correction) on gfx950.
//
// Reference: factors/reprojection_factor.h:55-147 (residual + 5 Jacobian blocks),
//            factors/residual_block_info.h:59-87 (Huber corrector used by marginalization).
// Design (HBM-bound, FP64, no MFMA — SURVEY.md §8(d)): one lane per factor, observation constants read
// component-major (15 coalesced 512-B wave loads), shared parameter blocks gathered through L2, the 48 output
// doubles of each factor transposed through LDS (row stride 49 to spread banks) so a 64-factor wave writes its
// r[64x2] and J[64x46] slabs as contiguous 16-B-per-lane stores.
// Algorithmic bytes per factor with Jacobians: 120 (obs) + 12 (indices) + 384 (out) = 516 B.
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

extern "C" int icg_reproj_set_factors(icg_ctx *ctx, int n, const double *obs_soa, const int32_t *idx_i,
                                      const int32_t *idx_j, const int32_t *idx_lm) {
    if (!ctx || n < 0 || (n > 0 && (!obs_soa || !idx_i || !idx_j || !idx_lm))) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_factor_capacity(ctx, n);
    if (rc) return rc;
    ctx->n_factors_resident = n;
    ctx->rJ_valid           = 0;
    ctx->sys_valid          = 0;
    ctx->n_windows          = 0;
    ctx->wsys_valid         = 0;
    if (n == 0) return ICG_OK;
    // component-major obs is already the device layout; indices packed as 3 x n
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_obs, obs_soa, sizeof(double) * 15 * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_fidx, idx_i, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_fidx + n, idx_j, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipMemcpyAsync(ctx->d_fidx + 2 * (size_t) n, idx_lm, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICG_OK;
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

// ---- M2: MarginalizationInfo::constructEquation for the resident reprojection factors -------------------------------
// Reference: factors/marginalization_info.h:195-230 — H0 += Ji^T Jj over all block pairs of a factor, b0 -= Ji^T e.
// One lane per factor.  Entries that every factor of the launch shares (extrinsic x extrinsic, extrinsic x td, td x td
// and their b0 rows) are first reduced in LDS with ds_add_f64 and flushed with ONE global atomic per entry and
// workgroup; pose / landmark blocks go straight to hardware FP64 atomics (addresses differ between factors).
// Summation order is not fixed -> results equal the sequential sum to ~1e-15 relative (tolerance-tested, not bit-tested).
#define NRM_BLOCK 256

// gfx950 has 160 KiB of LDS per CU and one workgroup may own all of it (MI355X_MICROARCH.md, "LDS"); a launch with more dynamic LDS than the
// 64 KiB default needs the function attribute.  The camera-block tiles of the assembly kernels and the batched Cholesky are sized by the
// window (97 free columns for the 15-keyframe windows of BASELINE configs[3] = 76 KB), so their launches go through this.
static constexpr size_t RPJ_LDS_LIMIT = 160 * 1024 - 256; // (- the kernels' few static words)
template <typename K> static int rpj_allow_lds(icg_ctx *ctx, K kernel, size_t bytes, int slot) {
    static std::atomic<size_t> granted[4][16];
    if (bytes <= 48 * 1024) return 0;
    const int dev = ctx->cfg.device & 15;
    if (granted[slot][dev].load(std::memory_order_relaxed) >= bytes) return 0;
    ICG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) RPJ_LDS_LIMIT));
    granted[slot][dev].store(RPJ_LDS_LIMIT, std::memory_order_relaxed);
    return 0;
}

__global__ __launch_bounds__(NRM_BLOCK) void k_reproj_normal(int n, const double *r, const double *J, const int32_t *idx_i,
                                                             const int32_t *idx_j, const int32_t *idx_lm,
                                                             const int32_t *col_pose, int col_ext, const int32_t *col_lm,
                                                             int col_td, int L, double *H, double *b, const uint8_t *active,
                                                             int lm_base) {
    __shared__ double sh[7 * 7 + 7]; // shared (ext|td) x (ext|td) block and its b rows
    const int t = threadIdx.x;
    for (int e = t; e < 56; e += NRM_BLOCK) sh[e] = 0.0;
    __syncthreads();
    const int f = blockIdx.x * NRM_BLOCK + t;
    if (f < n && (!active || active[f])) {
        const double *Jf = J + 46 * (size_t) f;
        const double r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
        // col_lm == nullptr: landmark l sits at column lm_base + l (the Schur layout of icg_reproj_schur)
        const int col[5] = {col_pose[idx_i[f]], col_pose[idx_j[f]], col_ext, col_lm ? col_lm[idx_lm[f]] : lm_base + idx_lm[f], col_td};
        const int sz[5]  = {6, 6, 6, 1, 1};
        const int off[5] = {0, 14, 28, 42, 44};
        const int ld[5]  = {7, 7, 7, 1, 1};
        const int shoff[5] = {-1, -1, 0, -1, 6}; // position of the block inside the shared 7-vector
#pragma unroll
        for (int a = 0; a < 5; a++) {
            if (col[a] < 0) continue;
#pragma unroll
            for (int bb = 0; bb < 5; bb++) {
                if (col[bb] < 0) continue;
                const bool shared_pair = shoff[a] >= 0 && shoff[bb] >= 0;
                for (int x = 0; x < sz[a]; x++)
                    for (int y = 0; y < sz[bb]; y++) {
                        double v = Jf[off[a] + x] * Jf[off[bb] + y] + Jf[off[a] + ld[a] + x] * Jf[off[bb] + ld[bb] + y];
                        if (shared_pair)
                            atomicAdd(&sh[(shoff[a] + x) * 7 + shoff[bb] + y], v);
                        else
                            unsafeAtomicAdd(&H[(size_t) (col[a] + x) * L + col[bb] + y], v);
                    }
            }
            for (int x = 0; x < sz[a]; x++) {
                double v = -(Jf[off[a] + x] * r0 + Jf[off[a] + ld[a] + x] * r1);
                if (shoff[a] >= 0)
                    atomicAdd(&sh[49 + shoff[a] + x], v);
                else
                    unsafeAtomicAdd(&b[col[a] + x], v);
            }
        }
    }
    __syncthreads();
    // flush the shared block
    for (int e = t; e < 56; e += NRM_BLOCK) {
        double v = sh[e];
        if (v == 0.0) continue;
        if (e < 49) {
            int x = e / 7, y = e - x * 7;
            int cx = x < 6 ? col_ext + x : col_td, cy = y < 6 ? col_ext + y : col_td;
            if ((x < 6 ? col_ext : col_td) >= 0 && (y < 6 ? col_ext : col_td) >= 0) unsafeAtomicAdd(&H[(size_t) cx * L + cy], v);
        } else {
            int x = e - 49;
            int cx = x < 6 ? col_ext + x : col_td;
            if ((x < 6 ? col_ext : col_td) >= 0) unsafeAtomicAdd(&b[cx], v);
        }
    }
}

extern "C" int icg_reproj_accumulate_normal(icg_ctx *ctx, int local_size, const int32_t *col_pose, int32_t col_ext,
                                            const int32_t *col_lm, int32_t col_td, double *H0, double *b0) {
    if (!ctx || local_size <= 0 || !col_pose || !col_lm || !H0 || !b0) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid || !ctx->rJ_has_jac) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Jacobians: call icg_reproj_eval_* with want_jac first");
    const int n = ctx->n_factors_resident;
    if (n == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t L = (size_t) local_size;
    icg_call c(ctx);
    int rc = c.reserve(sizeof(int32_t) * ((size_t) ctx->last_n_poses + ctx->last_n_lm) + sizeof(double) * (L * L + L) * 2 + 4096);
    if (rc) return rc;
    const int32_t *d_cp = c.in(col_pose, (size_t) ctx->last_n_poses);
    const int32_t *d_cl = c.in(col_lm, (size_t) ctx->last_n_lm);
    if ((rc = c.seal())) return rc;
    std::vector<double> hH(L * L), hb(L);
    double *d_H = c.out(hH.data(), L * L);
    double *d_b = c.out(hb.data(), L);
    ICG_HIP(ctx, hipMemsetAsync(d_H, 0, sizeof(double) * L * L, ctx->stream));
    ICG_HIP(ctx, hipMemsetAsync(d_b, 0, sizeof(double) * L, ctx->stream));
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_normal");
        hipLaunchKernelGGL(k_reproj_normal, dim3((n + NRM_BLOCK - 1) / NRM_BLOCK), dim3(NRM_BLOCK), 0, ctx->stream, n,
                           (const double *) ctx->d_rJ, (const double *) (ctx->d_rJ + 2 * (size_t) ctx->factors_cap),
                           (const int32_t *) ctx->d_fidx, (const int32_t *) (ctx->d_fidx + n),
                           (const int32_t *) (ctx->d_fidx + 2 * (size_t) n), d_cp, (int) col_ext, d_cl, (int) col_td, local_size, d_H, d_b,
                           (const uint8_t *) nullptr, 0);
    }
    ICG_HIP(ctx, hipGetLastError());
    if ((rc = c.finish())) return rc;
    for (size_t i = 0; i < L * L; i++) H0[i] += hH[i];
    for (size_t i = 0; i < L; i++) b0[i] += hb[i];
    return ICG_OK;
}


// ---- f1 (SURVEY.md §8 "next" row): device-side Schur complement of the visual factors ----------------------------------------
// The Gauss-Newton / Levenberg-Marquardt step of GVINS::gvinsOptimization (ic_gvins.cc:1130-1239, Ceres DENSE_SCHUR) eliminates
// the inverse-depth blocks (1x1 each) first.  With the robust-corrected r/J of the last icg_reproj_eval_resident call still
// resident, k_reproj_normal assembles  H = [Hcc G^T; G diag(h_ll)], b = -J^T r  in the column layout (camera columns 0..P-1,
// landmark l at P+l), and the kernels below reduce it:   S = Hcc - G^T diag(1/(h_ll + d_l)) G,   s = bc - G^T (b_l/(h_ll+d_l)),
// d_l = clamp(h_ll, min_diag, max_diag) * damp  (the LM diagonal of the eliminated block).  G, h_ll, b_l stay resident for the
// back-substitution  delta_l = (b_l - G_l . delta_c) / (h_ll + d_l).
// Sizes: P <= ~160 (10 poses x 6 + extrinsic 6 + td 1 [+ anything the host adds]), L = 300..500: a (P+L)^2 FP64 system of
// 1.7-3.4 MB that never leaves the device; only S (P x P), s and the diagonal cross PCIe per iteration.
#define SCH_T 16

__global__ void k_schur_inv(int P, int L, int N, const double *H, double damp, double min_diag, double max_diag, double *inv) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double h = H[(size_t) (P + l) * N + P + l];
    // a landmark without any active factor has an empty row: it is left where it is (delta_l = 0)
    inv[l] = h > 0.0 ? 1.0 / (h + fmin(fmax(h, min_diag), max_diag) * damp) : 0.0;
}

__global__ __launch_bounds__(SCH_T *SCH_T) void k_schur_reduce(int P, int L, int N, const double *H, const double *b, const double *inv,
                                                               double *S, double *s, double *diag) {
    __shared__ double gi[SCH_T][SCH_T + 1], gj[SCH_T][SCH_T + 1], w[SCH_T];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i = blockIdx.y * SCH_T + ty, j = blockIdx.x * SCH_T + tx;
    double acc = 0.0, accs = 0.0;
    for (int l0 = 0; l0 < L; l0 += SCH_T) {
        // rows l0..l0+15 of G, the 16 columns of this tile's i range and j range (coalesced along the row)
        const int l = l0 + ty;
        const int ci = blockIdx.y * SCH_T + tx;
        gi[ty][tx] = (l < L && ci < P) ? H[(size_t) (P + l) * N + ci] : 0.0;
        gj[ty][tx] = (l < L && j < P) ? H[(size_t) (P + l) * N + j] : 0.0;
        if (ty == 0) w[tx] = (l0 + tx < L) ? inv[l0 + tx] : 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCH_T; k++) {
            acc += gi[k][ty] * w[k] * gj[k][tx];
            if (blockIdx.x == 0 && tx == 0) accs += gi[k][ty] * w[k] * ((l0 + k < L) ? b[P + l0 + k] : 0.0);
        }
        __syncthreads();
    }
    if (i < P && j < P) S[(size_t) i * P + j] = H[(size_t) i * N + j] - acc;
    if (blockIdx.x == 0 && tx == 0 && i < P) {
        s[i]    = b[i] - accs;
        diag[i] = H[(size_t) i * N + i];
    }
}

// one wave per landmark: delta_l = (b_l - G_l . delta_c) * inv_l
// terms (2 doubles, pre-zeroed): sum b_l^2 / (h_ll + d_l) and sum d_l delta_l^2 — the landmark part of the LM model decrease
// 0.5 (delta^T b + delta^T D delta): with the reduced right-hand side s, delta^T b = delta_c^T s + terms[0], so the step-quality
// ratio can be formed without moving G or b_l to the host
__global__ __launch_bounds__(64) void k_schur_backsub(int P, int L, int N, const double *H, const double *b, const double *inv,
                                                      const double *delta_c, double *delta_l, double *terms, double damp, double min_diag,
                                                      double max_diag) {
    const int l = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += 64) acc += H[(size_t) (P + l) * N + i] * delta_c[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) {
        const double bl = b[P + l], w = inv[l];
        const double d  = (bl - acc) * w;
        delta_l[l]      = d;
        if (w > 0.0) {
            const double dl = fmin(fmax(H[(size_t) (P + l) * N + P + l], min_diag), max_diag) * damp; // the damping that went into inv
            unsafeAtomicAdd(&terms[0], bl * bl * w);
            unsafeAtomicAdd(&terms[1], dl * d * d);
        }
    }
}

// 0.5 * sum rho(|r|^2) of the active factors from the resident (possibly Huber-corrected) residuals: the corrector leaves
// |r_c|^2 = rho'(s) s, i.e. s for inliers and a sqrt(s) > a^2 for outliers, so rho(s) = 2 a sqrt(s) - a^2 = 2 |r_c|^2 - a^2
__global__ __launch_bounds__(256) void k_reproj_cost(int n, const double *r, const uint8_t *active, double huber, double *out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < n; f += gridDim.x * 256) {
        if (active && !active[f]) continue;
        const double r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
        double q = r0 * r0 + r1 * r1;
        if (huber > 0.0 && q > huber * huber) q = 2.0 * q - huber * huber;
        acc += 0.5 * q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// Assembly for the Schur layout with the camera-camera block privatised in LDS.  k_reproj_normal sends every J^T J entry to a
// global FP64 atomic; in a sliding window ~270 factors share each pose block and ALL factors share the extrinsic/td block, so
// those atomics serialise in L2 (measured: ~0.9 ms for 2 651 factors).  Here each workgroup accumulates its 256 factors into an
// LDS copy of the compact camera system (V = 6 x poses + 7 columns touched by visual factors, V^2 doubles), the fully shared
// (ext|td)^2 block is first reduced across the wave with shuffles, and one pass of global atomics per workgroup flushes the tile.
// Landmark rows (G_l, h_ll, b_l: 21 values per factor, <= ~10 factors per landmark) go straight to global atomics.
__global__ __launch_bounds__(NRM_BLOCK) void k_reproj_normal_schur(int n, const double *r, const double *J, const int32_t *idx_i,
                                                                   const int32_t *idx_j, const int32_t *idx_lm, const int32_t *vcol_pose,
                                                                   int vcol_ext, int vcol_td, const int32_t *vmap, int V, int P, int N,
                                                                   double *H, double *b, const uint8_t *active) {
    extern __shared__ double sm[]; // Hs[V*V] | bs[V]
    double *Hs = sm, *bs = sm + (size_t) V * V;
    const int t = threadIdx.x;
    for (int e = t; e < V * V + V; e += NRM_BLOCK) sm[e] = 0.0;
    __syncthreads();
    const int f   = blockIdx.x * NRM_BLOCK + t;
    const bool on = f < n && (!active || active[f]);
    // the factor's 19 camera columns: pose_i (6), pose_j (6), ext (6), td (1); compact column or -1
    double j0[19], j1[19];
    int cc[19];
    double r0 = 0.0, r1 = 0.0, jl0 = 0.0, jl1 = 0.0;
    int lm = 0;
    if (on) {
        const double *Jf = J + 46 * (size_t) f;
        r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
        const int ci = vcol_pose[idx_i[f]], cj = vcol_pose[idx_j[f]];
#pragma unroll
        for (int x = 0; x < 6; x++) {
            j0[x] = Jf[x], j1[x] = Jf[7 + x], cc[x] = ci < 0 ? -1 : ci + x;
            j0[6 + x] = Jf[14 + x], j1[6 + x] = Jf[21 + x], cc[6 + x] = cj < 0 ? -1 : cj + x;
            j0[12 + x] = Jf[28 + x], j1[12 + x] = Jf[35 + x], cc[12 + x] = vcol_ext < 0 ? -1 : vcol_ext + x;
        }
        j0[18] = Jf[44], j1[18] = Jf[45], cc[18] = vcol_td;
        jl0 = Jf[42], jl1 = Jf[43];
        lm  = idx_lm[f];
    } else {
#pragma unroll
        for (int x = 0; x < 19; x++) j0[x] = j1[x] = 0.0, cc[x] = -1;
    }
    // pose rows against all 19 columns, and the shared rows against the pose columns: LDS atomics
#pragma unroll
    for (int x = 0; x < 19; x++) {
        if (cc[x] < 0) continue;
#pragma unroll
        for (int y = 0; y < 19; y++) {
            if (x >= 12 && y >= 12) continue; // (ext|td)^2: wave-reduced below
            if (cc[y] < 0) continue;
            atomicAdd(&Hs[cc[x] * V + cc[y]], j0[x] * j0[y] + j1[x] * j1[y]);
        }
        if (x < 12) atomicAdd(&bs[cc[x]], -(j0[x] * r0 + j1[x] * r1));
    }
    // shared block: every active lane contributes to the same 49 + 7 addresses -> butterfly over the wave, lane 0 adds
#pragma unroll
    for (int x = 12; x < 19; x++) {
#pragma unroll
        for (int y = 12; y < 20; y++) { // y == 19: the right-hand side entry
            double v = (y < 19) ? j0[x] * j0[y] + j1[x] * j1[y] : -(j0[x] * r0 + j1[x] * r1);
            if (!on) v = 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((t & 63) == 0) {
                const int cx = (x < 18) ? (vcol_ext < 0 ? -1 : vcol_ext + x - 12) : vcol_td;
                const int cy = (y < 18) ? (vcol_ext < 0 ? -1 : vcol_ext + y - 12) : (y == 18 ? vcol_td : 0);
                if (cx >= 0 && cy >= 0 && v != 0.0) {
                    if (y < 19)
                        atomicAdd(&Hs[cx * V + cy], v);
                    else
                        atomicAdd(&bs[cx], v);
                }
            }
        }
    }
    // landmark row: G_l (camera columns), h_ll, b_l
    if (on) {
        double *row = H + (size_t) (P + lm) * N;
#pragma unroll
        for (int x = 0; x < 19; x++)
            if (cc[x] >= 0) unsafeAtomicAdd(&row[vmap[cc[x]]], jl0 * j0[x] + jl1 * j1[x]);
        unsafeAtomicAdd(&row[P + lm], jl0 * jl0 + jl1 * jl1);
        unsafeAtomicAdd(&b[P + lm], -(jl0 * r0 + jl1 * r1));
    }
    __syncthreads();
    for (int e = t; e < V * V; e += NRM_BLOCK) {
        const double v = Hs[e];
        if (v != 0.0) unsafeAtomicAdd(&H[(size_t) vmap[e / V] * N + vmap[e % V]], v);
    }
    for (int e = t; e < V; e += NRM_BLOCK) {
        const double v = bs[e];
        if (v != 0.0) unsafeAtomicAdd(&b[vmap[e]], v);
    }
}

static int ensure_sys_capacity(icg_ctx *ctx, size_t doubles) {
    if (doubles <= ctx->sys_cap) return 0;
    ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_sys) (void) hipFree(ctx->d_sys);
    ctx->d_sys   = nullptr;
    ctx->sys_cap = 0;
    size_t cap   = doubles + doubles / 4;
    ICG_HIP(ctx, hipMalloc((void **) &ctx->d_sys, sizeof(double) * cap));
    ctx->sys_cap = cap;
    return 0;
}

extern "C" int icg_reproj_schur(icg_ctx *ctx, int P, const int32_t *col_pose, int32_t col_ext, int32_t col_td, const uint8_t *active,
                                int reassemble, double damp, double min_diag, double max_diag, double *S, double *s, double *diag_cc,
                                double *cost) {
    if (!ctx || P <= 0 || !col_pose || !S || !s) return ICG_ERR_INVALID;
    if (reassemble && (!ctx->rJ_valid || !ctx->rJ_has_jac))
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident Jacobians: call icg_reproj_eval_resident with want_jac first");
    if (!reassemble && (!ctx->sys_valid || ctx->sys_P != P))
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident normal equations of size %d to re-damp", P);
    const int n = ctx->n_factors_resident, L = ctx->last_n_lm;
    if (n == 0) return icg_fail(ctx, ICG_ERR_INVALID, "no resident factors");
    for (int k = 0; k < ctx->last_n_poses; k++)
        if (col_pose[k] >= 0 && col_pose[k] + 6 > P) return icg_fail(ctx, ICG_ERR_INVALID, "pose %d: column %d outside the reduced system (%d)", k, col_pose[k], P);
    if ((col_ext >= 0 && col_ext + 6 > P) || (col_td >= 0 && col_td + 1 > P)) return icg_fail(ctx, ICG_ERR_INVALID, "ext/td column outside the reduced system");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t N = (size_t) P + L;
    int rc         = ensure_sys_capacity(ctx, N * N + N + (size_t) L + 8);
    if (rc) return rc;
    double *d_H = ctx->d_sys, *d_b = d_H + N * N, *d_inv = d_b + N, *d_cost = d_inv + L;
    icg_call c(ctx);
    rc = c.reserve(sizeof(int32_t) * (8 * (size_t) ctx->last_n_poses + 16) + (size_t) n + sizeof(double) * ((size_t) P * P + 2 * (size_t) P + 1) + 4096);
    if (rc) return rc;
    const int32_t *d_cp = c.in(col_pose, (size_t) ctx->last_n_poses);
    const uint8_t *d_act = active ? c.in(active, (size_t) n) : nullptr;
    // compact camera space of the visual factors: free poses (6 columns each), then ext, then td
    std::vector<int32_t> vcol_pose((size_t) ctx->last_n_poses, -1), vmap;
    for (int k = 0; k < ctx->last_n_poses; k++)
        if (col_pose[k] >= 0) {
            vcol_pose[(size_t) k] = (int32_t) vmap.size();
            for (int x = 0; x < 6; x++) vmap.push_back(col_pose[k] + x);
        }
    int vcol_ext = -1, vcol_td = -1;
    if (col_ext >= 0) {
        vcol_ext = (int) vmap.size();
        for (int x = 0; x < 6; x++) vmap.push_back(col_ext + x);
    }
    if (col_td >= 0) {
        vcol_td = (int) vmap.size();
        vmap.push_back(col_td);
    }
    const int V = (int) vmap.size();
    if (V == 0) return icg_fail(ctx, ICG_ERR_INVALID, "every camera block of the visual factors is constant");
    const int32_t *d_vp = c.in(vcol_pose.data(), vcol_pose.size());
    const int32_t *d_vm = c.in(vmap.data(), vmap.size());
    if ((rc = c.seal())) return rc;
    double *d_S = c.out(S, (size_t) P * P);
    double *d_s = c.out(s, (size_t) P);
    double *d_dg = c.out(diag_cc, (size_t) P); // user pointer may be null: still a valid device scratch
    double *d_co = c.out(cost, 1);
    const double *d_r = ctx->d_rJ, *d_J = ctx->d_rJ + 2 * (size_t) ctx->factors_cap;
    ICG_LAUNCH_GUARD(c);
    if (reassemble) {
        ICG_HIP(ctx, hipMemsetAsync(d_H, 0, sizeof(double) * (N * N + N + (size_t) L + 1), ctx->stream));
        icg_prof_scope ps(ctx, "reproj_normal");
        const size_t lds = sizeof(double) * ((size_t) V * V + V);
        if (lds <= RPJ_LDS_LIMIT) {
            if ((rc = rpj_allow_lds(ctx, k_reproj_normal_schur, lds, 0))) return rc;
            hipLaunchKernelGGL(k_reproj_normal_schur, dim3((n + NRM_BLOCK - 1) / NRM_BLOCK), dim3(NRM_BLOCK), lds, ctx->stream, n, d_r, d_J,
                               (const int32_t *) ctx->d_fidx, (const int32_t *) (ctx->d_fidx + n), (const int32_t *) (ctx->d_fidx + 2 * (size_t) n),
                               d_vp, vcol_ext, vcol_td, d_vm, V, P, (int) N, d_H, d_b, d_act);
        } else { // more than 142 free camera columns (23 free poses): the camera block does not fit a CU's LDS
            hipLaunchKernelGGL(k_reproj_normal, dim3((n + NRM_BLOCK - 1) / NRM_BLOCK), dim3(NRM_BLOCK), 0, ctx->stream, n, d_r, d_J,
                               (const int32_t *) ctx->d_fidx, (const int32_t *) (ctx->d_fidx + n), (const int32_t *) (ctx->d_fidx + 2 * (size_t) n),
                               d_cp, (int) col_ext, (const int32_t *) nullptr, (int) col_td, (int) N, d_H, d_b, d_act, P);
        }
    }
    {
        icg_prof_scope ps(ctx, "schur_reduce");
        hipLaunchKernelGGL(k_schur_inv, dim3((L + 255) / 256), dim3(256), 0, ctx->stream, P, L, (int) N, (const double *) d_H, damp, min_diag,
                           max_diag, d_inv);
        hipLaunchKernelGGL(k_schur_reduce, dim3((P + SCH_T - 1) / SCH_T, (P + SCH_T - 1) / SCH_T), dim3(SCH_T, SCH_T), 0, ctx->stream, P, L,
                           (int) N, (const double *) d_H, (const double *) d_b, (const double *) d_inv, d_S, d_s, d_dg);
        // the cost belongs to the linearization point: only meaningful while the resident residuals are the ones assembled
        if (reassemble)
            hipLaunchKernelGGL(k_reproj_cost, dim3(std::min(64, (n + 255) / 256)), dim3(256), 0, ctx->stream, n, d_r, d_act, ctx->last_huber,
                               d_cost);
    }
    ICG_HIP(ctx, hipGetLastError());
    ICG_HIP(ctx, hipMemcpyAsync(d_co, d_cost, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = c.finish())) return rc;
    ctx->sys_P = P, ctx->sys_L = L, ctx->sys_valid = 1;
    ctx->sys_damp = damp, ctx->sys_min_diag = min_diag, ctx->sys_max_diag = max_diag;
    return ICG_OK;
}

extern "C" int icg_reproj_landmark_diag(icg_ctx *ctx, double *h_ll) {
    if (!ctx || !h_ll) return ICG_ERR_INVALID;
    if (!ctx->sys_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Schur system: call icg_reproj_schur first");
    const int P = ctx->sys_P, L = ctx->sys_L;
    if (L == 0) return ICG_OK;
    const size_t N = (size_t) P + L;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    // the diagonal (P+l, P+l) of the row-major N x N matrix: one 8-byte column with a pitch of (N+1) doubles
    ICG_HIP(ctx, hipMemcpy2DAsync(h_ll, sizeof(double), ctx->d_sys + (size_t) P * N + P, sizeof(double) * (N + 1), sizeof(double), (size_t) L,
                                  hipMemcpyDeviceToHost, ctx->stream));
    return icg_stream_wait(ctx);
}

extern "C" int icg_reproj_backsub(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    if (!ctx || !delta_c || !delta_l) return ICG_ERR_INVALID;
    if (!ctx->sys_valid || ctx->sys_P != P) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Schur system of size %d: call icg_reproj_schur first", P);
    const int L = ctx->sys_L;
    const size_t N = (size_t) P + L;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const double *d_H = ctx->d_sys, *d_b = d_H + N * N, *d_inv = d_b + N;
    icg_call c(ctx);
    int rc = c.reserve(sizeof(double) * ((size_t) P + L + 2) + 4096);
    if (rc) return rc;
    const double *d_dc = c.in(delta_c, (size_t) P);
    const double zeros[2] = {0.0, 0.0};
    double *d_tm = c.inout(zeros, lm_terms, 2); // device accumulators, pre-zeroed
    if ((rc = c.seal())) return rc;
    double *d_dl = c.out(delta_l, (size_t) L);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_backsub");
        hipLaunchKernelGGL(k_schur_backsub, dim3(L), dim3(64), 0, ctx->stream, P, L, (int) N, d_H, d_b, d_inv, d_dc, d_dl, d_tm, ctx->sys_damp,
                           ctx->sys_min_diag, ctx->sys_max_diag);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_reproj_cost(icg_ctx *ctx, const uint8_t *active, double *cost) {
    if (!ctx || !cost) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident residuals: call icg_reproj_eval_resident first");
    const int n = ctx->n_factors_resident;
    *cost       = 0.0;
    if (n == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    int rc = c.reserve((size_t) n + 64 + 4096);
    if (rc) return rc;
    const uint8_t *d_act = active ? c.in(active, (size_t) n) : nullptr;
    const double zero = 0.0;
    double *d_acc = c.inout(&zero, cost, 1); // device accumulator, pre-zeroed
    if ((rc = c.seal())) return rc;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_cost");
        hipLaunchKernelGGL(k_reproj_cost, dim3(std::min(64, (n + 255) / 256)), dim3(256), 0, ctx->stream, n, (const double *) ctx->d_rJ, d_act,
                           ctx->last_huber, d_acc);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}


// ---- f1, many windows per launch ------------------------------------------------------------------------------------------------
// One solver in flight per stream is bounded by the runtime's rate of small launches and copies (~100 per window and solve, DESIGN.md
// §6).  Here the windows of many streams advance in lock-step: ONE evaluation, ONE assembly, ONE reduction, ONE back-substitution
// launch per LM step for all of them.  The resident factor set is partitioned into W windows (factors sorted by window, landmarks
// contiguous per window, poses indexed globally); every window has its own extrinsic / td, its own reduced system of the common
// size P and its own damping.  Window w's system lives at d_sys + w_sys_off[w]: H (N_w x N_w, N_w = P + L_w) | b (N_w) | inv (L_w).
#define LM_SLOTS 64 // landmark slots of one assembly workgroup (256 factors ~ 30 landmarks when the list is landmark-major)
struct win_desc {
    int32_t fac_begin, fac_end, lm_begin, L;
    int64_t sys_off;
    int32_t vcol_ext, vcol_td, V, reassemble;
    double damp;
};

__global__ __launch_bounds__(NRM_BLOCK) void k_reproj_normal_schur_w(const win_desc *wd, const int32_t *blk_win, const int32_t *blk_first,
                                                                     const double *r, const double *J, const int32_t *idx_i, const int32_t *idx_j,
                                                                     const int32_t *idx_lm, const int32_t *vcol_pose, const int32_t *vmap_all, int Vmax,
                                                                     int P, double *sys, const uint8_t *active) {
    extern __shared__ double sm[]; // Hs[V*V] | bs[V]
    const win_desc W = wd[blk_win[blockIdx.x]];
    if (!W.reassemble) return; // uniform per workgroup
    const int V = W.V, N = P + W.L;
    const int32_t *vmap = vmap_all + (size_t) blk_win[blockIdx.x] * Vmax;
    double *H = sys + W.sys_off, *b = H + (size_t) N * N;
    double *Hs = sm, *bs = sm + (size_t) V * V;
    const int t = threadIdx.x;
    for (int e = t; e < V * V + V; e += NRM_BLOCK) sm[e] = 0.0;
    __syncthreads();
    const int f   = blk_first[blockIdx.x] + t;
    const bool on = f < W.fac_end && (!active || active[f]);
    double j0[19], j1[19];
    int cc[19];
    double r0 = 0.0, r1 = 0.0, jl0 = 0.0, jl1 = 0.0;
    int lm = 0;
    if (on) {
        const double *Jf = J + 46 * (size_t) f;
        r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
        const int ci = vcol_pose[idx_i[f]], cj = vcol_pose[idx_j[f]];
#pragma unroll
        for (int x = 0; x < 6; x++) {
            j0[x] = Jf[x], j1[x] = Jf[7 + x], cc[x] = ci < 0 ? -1 : ci + x;
            j0[6 + x] = Jf[14 + x], j1[6 + x] = Jf[21 + x], cc[6 + x] = cj < 0 ? -1 : cj + x;
            j0[12 + x] = Jf[28 + x], j1[12 + x] = Jf[35 + x], cc[12 + x] = W.vcol_ext < 0 ? -1 : W.vcol_ext + x;
        }
        j0[18] = Jf[44], j1[18] = Jf[45], cc[18] = W.vcol_td;
        jl0 = Jf[42], jl1 = Jf[43];
        lm  = idx_lm[f] - W.lm_begin;
    } else {
#pragma unroll
        for (int x = 0; x < 19; x++) j0[x] = j1[x] = 0.0, cc[x] = -1;
    }
    // J^T J is symmetric: only the pairs x <= y of a factor's 19 camera columns are accumulated (half the LDS atomics, which bound this
    // kernel); the flush adds every cell and its mirror cell into both halves of the window's matrix
#pragma unroll
    for (int x = 0; x < 19; x++) {
        if (cc[x] < 0) continue;
#pragma unroll
        for (int y = x; y < 19; y++) {
            if (x >= 12 && y >= 12) continue;
            if (cc[y] < 0) continue;
            atomicAdd(&Hs[cc[x] * V + cc[y]], j0[x] * j0[y] + j1[x] * j1[y]);
        }
        if (x < 12) atomicAdd(&bs[cc[x]], -(j0[x] * r0 + j1[x] * r1));
    }
#pragma unroll
    for (int x = 12; x < 19; x++) {
#pragma unroll
        for (int y = x; y < 20; y++) {
            double v = (y < 19) ? j0[x] * j0[y] + j1[x] * j1[y] : -(j0[x] * r0 + j1[x] * r1);
            if (!on) v = 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((t & 63) == 0) {
                const int cx = (x < 18) ? (W.vcol_ext < 0 ? -1 : W.vcol_ext + x - 12) : W.vcol_td;
                const int cy = (y < 18) ? (W.vcol_ext < 0 ? -1 : W.vcol_ext + y - 12) : (y == 18 ? W.vcol_td : 0);
                if (cx >= 0 && cy >= 0 && v != 0.0) {
                    if (y < 19)
                        atomicAdd(&Hs[cx * V + cy], v);
                    else
                        atomicAdd(&bs[cx], v);
                }
            }
        }
    }
    // Landmark rows (G_l, h_ll, b_l).  A landmark's factors are neighbours in the factor list (the window builder adds them landmark by
    // landmark) and share its reference pose, the extrinsic and td: those 15 sums (6 reference-pose columns, 6 + 1 shared columns, h_ll,
    // b_l) are first accumulated in LDS per (landmark, reference pose) slot of this workgroup and flushed with ONE global atomic each —
    // ~9 factors per landmark made 21 global FP64 atomics per factor the dominant cost of the batched assembly (rocprofv3, round 1).
    // The observer-pose columns are unique per factor and go straight to memory.  Landmarks outside the workgroup's slot range (an
    // unsorted factor list) take the direct path: placement only, the sums are the same.
    double *lrow = bs + V; // LM_SLOTS x 16: [0..5] reference pose, [6..11] ext, [12] td, [13] h_ll, [14] b_l, [15] ci (slot owner check)
    int *lslot_ci = reinterpret_cast<int *>(lrow + LM_SLOTS * 16);
    for (int e = t; e < LM_SLOTS * 16; e += NRM_BLOCK) lrow[e] = 0.0;
    for (int e = t; e < LM_SLOTS; e += NRM_BLOCK) lslot_ci[e] = -2;
    __shared__ int lm_base;
    if (t == 0) lm_base = idx_lm[blk_first[blockIdx.x]] - W.lm_begin; // landmark of the workgroup's first factor
    __syncthreads();
    const int slot = lm - lm_base;
    bool in_lds    = false;
    if (on) {
        double *row = H + (size_t) (P + lm) * N;
        const int ci = cc[0]; // compact column of the reference pose (-1: constant)
        if (slot >= 0 && slot < LM_SLOTS) { // claim the slot for this (landmark, reference pose); a different owner -> direct path
            const int prev = atomicCAS(&lslot_ci[slot], -2, ci);
            in_lds         = prev == -2 || prev == ci;
        }
        if (in_lds) {
            double *L = lrow + slot * 16;
#pragma unroll
            for (int x = 0; x < 6; x++)
                if (cc[x] >= 0) atomicAdd(&L[x], jl0 * j0[x] + jl1 * j1[x]);
#pragma unroll
            for (int x = 12; x < 19; x++)
                if (cc[x] >= 0) atomicAdd(&L[x - 6], jl0 * j0[x] + jl1 * j1[x]);
            atomicAdd(&L[13], jl0 * jl0 + jl1 * jl1);
            atomicAdd(&L[14], -(jl0 * r0 + jl1 * r1));
#pragma unroll
            for (int x = 6; x < 12; x++)
                if (cc[x] >= 0) unsafeAtomicAdd(&row[vmap[cc[x]]], jl0 * j0[x] + jl1 * j1[x]);
        } else {
#pragma unroll
            for (int x = 0; x < 19; x++)
                if (cc[x] >= 0) unsafeAtomicAdd(&row[vmap[cc[x]]], jl0 * j0[x] + jl1 * j1[x]);
            unsafeAtomicAdd(&row[P + lm], jl0 * jl0 + jl1 * jl1);
            unsafeAtomicAdd(&b[P + lm], -(jl0 * r0 + jl1 * r1));
        }
    }
    __syncthreads();
    // flush the landmark slots: thread e -> (slot e / 16, value e % 16)
    for (int e = t; e < LM_SLOTS * 16; e += NRM_BLOCK) {
        const int sl = e >> 4, k = e & 15, ci = lslot_ci[sl];
        if (ci == -2 || k == 15) continue;
        const double v = lrow[e];
        if (v == 0.0) continue;
        const int l = lm_base + sl;
        double *row = H + (size_t) (P + l) * N;
        if (k < 6) {
            if (ci >= 0) unsafeAtomicAdd(&row[vmap[ci + k]], v);
        } else if (k < 12) {
            unsafeAtomicAdd(&row[vmap[W.vcol_ext + (k - 6)]], v);
        } else if (k == 12) {
            unsafeAtomicAdd(&row[vmap[W.vcol_td]], v);
        } else if (k == 13) {
            unsafeAtomicAdd(&row[P + l], v);
        } else {
            unsafeAtomicAdd(&b[P + l], v);
        }
    }
    for (int e = t; e < V * V; e += NRM_BLOCK) {
        const int ca = e / V, cb = e - ca * V;
        const double v = Hs[e] + (ca != cb ? Hs[cb * V + ca] : 0.0); // the cell and its mirror (x <= y accumulation above)
        if (v != 0.0) unsafeAtomicAdd(&H[(size_t) vmap[ca] * N + vmap[cb]], v);
    }
    for (int e = t; e < V; e += NRM_BLOCK) {
        const double v = bs[e];
        if (v != 0.0) unsafeAtomicAdd(&b[vmap[e]], v);
    }
}

// zeroes the (H | b) part of every window that is re-assembled (one workgroup column per window)
__global__ void k_sys_clear_w(const win_desc *wd, int P, double *sys) {
    const win_desc W = wd[blockIdx.y];
    if (!W.reassemble) return;
    const size_t N = (size_t) P + W.L, total = N * N + N;
    double *H = sys + W.sys_off;
    for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t) gridDim.x * blockDim.x) H[e] = 0.0;
}

__global__ void k_schur_inv_w(const win_desc *wd, int P, double *sys, double min_diag, double max_diag) {
    const win_desc W = wd[blockIdx.y];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= W.L) return;
    const int N = P + W.L;
    const double *H = sys + W.sys_off;
    double *inv = sys + W.sys_off + (size_t) N * N + N;
    const double h = H[(size_t) (P + l) * N + P + l];
    inv[l] = h > 0.0 ? 1.0 / (h + fmin(fmax(h, min_diag), max_diag) * W.damp) : 0.0;
}

__global__ __launch_bounds__(SCH_T *SCH_T) void k_schur_reduce_w(const win_desc *wd, int P, const double *sys, double *S, double *s, double *diag,
                                                                int lower_only) {
    __shared__ double gi[SCH_T][SCH_T + 1], gj[SCH_T][SCH_T + 1], w[SCH_T];
    // the reduced systems are symmetric and the factorization reads rows >= columns only: the tiles strictly above the diagonal are neither
    // computed nor written when the caller asks for that (40 % of the tiles, and of the bytes that cross PCIe in the zero-copy form, at P = 67)
    if (lower_only && blockIdx.x > blockIdx.y) return;
    const win_desc W = wd[blockIdx.z];
    const int L = W.L, N = P + L;
    const double *H = sys + W.sys_off, *b = H + (size_t) N * N, *inv = b + N;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int i = blockIdx.y * SCH_T + ty, j = blockIdx.x * SCH_T + tx;
    double acc = 0.0, accs = 0.0;
    for (int l0 = 0; l0 < L; l0 += SCH_T) {
        const int l  = l0 + ty;
        const int ci = blockIdx.y * SCH_T + tx;
        gi[ty][tx] = (l < L && ci < P) ? H[(size_t) (P + l) * N + ci] : 0.0;
        gj[ty][tx] = (l < L && j < P) ? H[(size_t) (P + l) * N + j] : 0.0;
        if (ty == 0) w[tx] = (l0 + tx < L) ? inv[l0 + tx] : 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCH_T; k++) {
            acc += gi[k][ty] * w[k] * gj[k][tx];
            if (blockIdx.x == 0 && tx == 0) accs += gi[k][ty] * w[k] * ((l0 + k < L) ? b[P + l0 + k] : 0.0);
        }
        __syncthreads();
    }
    double *Sw = S + (size_t) blockIdx.z * P * P;
    if (i < P && j < P) Sw[(size_t) i * P + j] = H[(size_t) i * N + j] - acc;
    if (blockIdx.x == 0 && tx == 0 && i < P) {
        s[(size_t) blockIdx.z * P + i]    = b[i] - accs;
        diag[(size_t) blockIdx.z * P + i] = H[(size_t) i * N + i];
    }
}

// one wave per landmark (global index); terms[w][2] pre-zeroed
__global__ __launch_bounds__(64) void k_schur_backsub_w(const win_desc *wd, const int32_t *lm_win, int P, const double *sys, const double *delta_c,
                                                        double *delta_l, double *terms, double min_diag, double max_diag) {
    const int lg = blockIdx.x, wi = lm_win[lg];
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
        if (wv > 0.0) {
            const double dl = fmin(fmax(H[(size_t) (P + l) * N + P + l], min_diag), max_diag) * W.damp;
            unsafeAtomicAdd(&terms[2 * wi], bl * bl * wv);
            unsafeAtomicAdd(&terms[2 * wi + 1], dl * d * d);
        }
    }
}

// grid (chunks, W): cost[w] += 0.5 sum rho over the window's active factors
__global__ __launch_bounds__(256) void k_reproj_cost_w(const win_desc *wd, const double *r, const uint8_t *active, double huber, double *out) {
    __shared__ double sh[4];
    const win_desc W = wd[blockIdx.y];
    double acc = 0.0;
    for (int f = W.fac_begin + blockIdx.x * 256 + threadIdx.x; f < W.fac_end; f += gridDim.x * 256) {
        if (active && !active[f]) continue;
        const double r0 = r[2 * (size_t) f], r1 = r[2 * (size_t) f + 1];
        double q = r0 * r0 + r1 * r1;
        if (huber > 0.0 && q > huber * huber) q = 2.0 * q - huber * huber;
        acc += 0.5 * q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(&out[blockIdx.y], sh[0] + sh[1] + sh[2] + sh[3]);
}

extern "C" int icg_reproj_set_windows(icg_ctx *ctx, int n_windows, const int32_t *fac_off, const int32_t *lm_off) {
    if (!ctx || n_windows <= 0 || !fac_off || !lm_off) return ICG_ERR_INVALID;
    const int n = ctx->n_factors_resident;
    if (fac_off[0] != 0 || fac_off[n_windows] != n) return icg_fail(ctx, ICG_ERR_INVALID, "fac_off must cover the %d resident factors", n);
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
    // pose -> window from the resident index arrays (one read-back per partition)
    std::vector<int32_t> h_idx((size_t) 2 * std::max(n, 1));
    if (n) {
        ICG_HIP(ctx, hipMemcpyAsync(h_idx.data(), ctx->d_fidx, sizeof(int32_t) * 2 * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
        ICG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    int max_pose = -1;
    for (int f = 0; f < 2 * n; f++) max_pose = std::max(max_pose, (int) h_idx[(size_t) f]);
    ctx->w_pose_win.assign((size_t) (max_pose + 1), -1);
    for (int w = 0; w < n_windows; w++)
        for (int f = fac_off[w]; f < fac_off[w + 1]; f++) {
            for (int side = 0; side < 2; side++) {
                int32_t &pw = ctx->w_pose_win[(size_t) h_idx[(size_t) side * n + f]];
                if (pw >= 0 && pw != w) return icg_fail(ctx, ICG_ERR_INVALID, "pose %d is used by windows %d and %d", (int) h_idx[(size_t) side * n + f], pw, w);
                pw = w;
            }
        }
    ctx->n_windows = n_windows;
    ctx->w_fac_off.assign(fac_off, fac_off + n_windows + 1);
    ctx->w_lm_off.assign(lm_off, lm_off + n_windows + 1);
    ctx->wsys_valid = 0;
    return ICG_OK;
}

extern "C" int icg_reproj_eval_windows(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth,
                                       const double *td, int want_jac, double huber_delta) {
    if (!ctx || !poses || !ext || !invdepth || !td || n_poses <= 0 || n_lm <= 0) return ICG_ERR_INVALID;
    if (ctx->n_windows <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    if (n_lm != ctx->w_lm_off[(size_t) ctx->n_windows]) return icg_fail(ctx, ICG_ERR_INVALID, "n_lm does not match the window partition");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int n = ctx->n_factors_resident, W = ctx->n_windows;
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
    return c.finish();
}

// builds the per-window descriptors (host) for the current partition; reassemble / damp may be null (all re-assembled / keep damping)
static void build_win_desc(icg_ctx *ctx, int P, const int32_t *vcol_ext, const int32_t *vcol_td, const int32_t *V, const uint8_t *reassemble,
                           const double *damp, std::vector<win_desc> &out) {
    const int W = ctx->n_windows;
    out.resize((size_t) W);
    for (int w = 0; w < W; w++) {
        win_desc &d = out[(size_t) w];
        d.fac_begin = ctx->w_fac_off[(size_t) w], d.fac_end = ctx->w_fac_off[(size_t) w + 1];
        d.lm_begin = ctx->w_lm_off[(size_t) w], d.L = ctx->w_lm_off[(size_t) w + 1] - ctx->w_lm_off[(size_t) w];
        d.sys_off    = ctx->w_sys_off[(size_t) w];
        d.vcol_ext   = vcol_ext ? vcol_ext[w] : -1;
        d.vcol_td    = vcol_td ? vcol_td[w] : -1;
        d.V          = V ? V[w] : 0;
        d.reassemble = reassemble ? reassemble[w] : 1;
        d.damp       = damp ? damp[w] : ctx->w_damp[(size_t) w];
    }
}

// S_view != nullptr: the reduced systems are written by the reduction kernel straight into the context's pinned staging memory (zero-copy)
// and *S_view points there — no device-to-host copy and no 9 MB copy-out per LM step at 256 windows; valid until the next call on ctx
// S and S_view both null: the reduced systems stay on the device (ctx->d_redS)
static int schur_windows_impl(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                              const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, double *S, const double **S_view,
                              double *s, double *diag_cc, double *cost) {
    if (!ctx || P <= 0 || !col_pose || !col_ext || !col_td || !reassemble || !damp || !s) return ICG_ERR_INVALID;
    const bool tdbg = getenv("ICG_ABI_DEBUG") != nullptr;
    auto tnow       = [] { return std::chrono::steady_clock::now(); };
    auto t_begin    = tnow();
    const int W = ctx->n_windows, n = ctx->n_factors_resident;
    if (W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    bool any_new = false;
    for (int w = 0; w < W; w++) any_new |= reassemble[w] != 0;
    if (any_new && (!ctx->rJ_valid || !ctx->rJ_has_jac)) return icg_fail(ctx, ICG_ERR_INVALID, "no resident Jacobians: call icg_reproj_eval_windows first");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    // system layout
    if (!ctx->wsys_valid || ctx->wsys_P != P) {
        for (int w = 0; w < W; w++)
            if (!reassemble[w]) return icg_fail(ctx, ICG_ERR_INVALID, "window %d: nothing resident to re-damp", w);
        ctx->w_sys_off.assign((size_t) W + 1, 0);
        for (int w = 0; w < W; w++) {
            const int64_t N = P + (ctx->w_lm_off[(size_t) w + 1] - ctx->w_lm_off[(size_t) w]);
            ctx->w_sys_off[(size_t) w + 1] = ctx->w_sys_off[(size_t) w] + N * N + N + (N - P);
        }
        ctx->w_damp.assign((size_t) W, 0.0);
    }
    int rc = ensure_sys_capacity(ctx, (size_t) ctx->w_sys_off[(size_t) W] + 8);
    if (rc) return rc;
    // compact camera space per window (free poses in pose order, then ext, then td)
    std::vector<int32_t> vcol_pose((size_t) ctx->last_n_poses, -1), Vw((size_t) W, 0), vce((size_t) W, -1), vct((size_t) W, -1);
    std::vector<std::vector<int32_t>> vmaps((size_t) W);
    // window of every pose (recorded at icg_reproj_set_windows from the resident index arrays): a pose column is compact within
    // the window whose factors use the pose
    std::vector<int32_t> pose_win((size_t) ctx->last_n_poses, -1);
    for (int k = 0; k < ctx->last_n_poses && k < (int) ctx->w_pose_win.size(); k++) pose_win[(size_t) k] = ctx->w_pose_win[(size_t) k];
    for (int k = 0; k < ctx->last_n_poses; k++) {
        const int w = pose_win[(size_t) k];
        if (w < 0 || col_pose[k] < 0) continue;
        if (col_pose[k] + 6 > P) return icg_fail(ctx, ICG_ERR_INVALID, "pose %d: column %d outside the reduced system (%d)", k, col_pose[k], P);
        vcol_pose[(size_t) k] = (int32_t) vmaps[(size_t) w].size();
        for (int x = 0; x < 6; x++) vmaps[(size_t) w].push_back(col_pose[k] + x);
    }
    int Vmax = 1;
    for (int w = 0; w < W; w++) {
        if ((col_ext[w] >= 0 && col_ext[w] + 6 > P) || (col_td[w] >= 0 && col_td[w] + 1 > P))
            return icg_fail(ctx, ICG_ERR_INVALID, "window %d: ext/td column outside the reduced system", w);
        if (col_ext[w] >= 0) {
            vce[(size_t) w] = (int32_t) vmaps[(size_t) w].size();
            for (int x = 0; x < 6; x++) vmaps[(size_t) w].push_back(col_ext[w] + x);
        }
        if (col_td[w] >= 0) {
            vct[(size_t) w] = (int32_t) vmaps[(size_t) w].size();
            vmaps[(size_t) w].push_back(col_td[w]);
        }
        Vw[(size_t) w] = (int32_t) vmaps[(size_t) w].size();
        Vmax           = std::max(Vmax, (int) Vw[(size_t) w]);
    }
    const size_t lds = sizeof(double) * ((size_t) Vmax * Vmax + Vmax + LM_SLOTS * 16) + sizeof(int) * LM_SLOTS;
    if (lds > RPJ_LDS_LIMIT)
        return icg_fail(ctx, ICG_ERR_CAPACITY, "a window has %d free camera columns: more than the LDS tile of the batched assembly holds (138)", Vmax);
    if ((rc = rpj_allow_lds(ctx, k_reproj_normal_schur_w, lds, 1))) return rc;
    std::vector<int32_t> vmap_all((size_t) W * Vmax, 0);
    for (int w = 0; w < W; w++) std::copy(vmaps[(size_t) w].begin(), vmaps[(size_t) w].end(), vmap_all.begin() + (size_t) w * Vmax);
    for (int w = 0; w < W; w++)
        if (reassemble[w] || damp[w] != ctx->w_damp[(size_t) w]) ctx->w_damp[(size_t) w] = damp[w];
    std::vector<win_desc> wd;
    build_win_desc(ctx, P, vce.data(), vct.data(), Vw.data(), reassemble, damp, wd);
    // block tables of the assembly launch
    std::vector<int32_t> blk_win, blk_first;
    int Lmax = 1;
    for (int w = 0; w < W; w++) {
        Lmax = std::max(Lmax, (int) wd[(size_t) w].L);
        if (!reassemble[w]) continue;
        for (int f = wd[(size_t) w].fac_begin; f < wd[(size_t) w].fac_end; f += NRM_BLOCK) {
            blk_win.push_back(w);
            blk_first.push_back(f);
        }
    }
    icg_call c(ctx);
    rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(int32_t) * (vcol_pose.size() + vmap_all.size() + 2 * blk_win.size() + 16) + (size_t) n +
                   sizeof(double) * ((size_t) W * ((size_t) P * P + 2 * (size_t) P + 1)) + 8192);
    if (rc) return rc;
    const win_desc *d_wd  = c.in(wd.data(), (size_t) W);
    const int32_t *d_vp   = c.in(vcol_pose.data(), vcol_pose.size());
    const int32_t *d_vm   = c.in(vmap_all.data(), vmap_all.size());
    const int32_t *d_bw   = blk_win.empty() ? nullptr : c.in(blk_win.data(), blk_win.size());
    const int32_t *d_bf   = blk_first.empty() ? nullptr : c.in(blk_first.data(), blk_first.size());
    const uint8_t *d_act  = active ? c.in(active, (size_t) n) : nullptr;
    std::vector<double> zeros((size_t) W, 0.0);
    double *d_cost = c.inout(zeros.data(), any_new ? cost : (double *) nullptr, (size_t) W);
    auto t_prep = tnow();
    if ((rc = c.seal())) return rc;
    // (the zero-copy region is allocated LAST: finish() copies ONE device range back that spans all mirrored outputs, and must not run
    // over memory the kernel wrote through the host mapping)
    const bool resident = !S && !S_view; // the reduced systems stay on the device (solved there: icg_reproj_solve_backsub_windows)
    double *d_S  = (S_view || resident) ? nullptr : c.out(S, (size_t) W * P * P);
    double *d_s  = c.out(s, (size_t) W * P);
    double *d_dg = c.out(diag_cc, (size_t) W * P);
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
        hipLaunchKernelGGL(k_sys_clear_w, dim3(32, W), dim3(256), 0, ctx->stream, d_wd, P, ctx->d_sys);
        hipLaunchKernelGGL(k_reproj_normal_schur_w, dim3((unsigned) blk_win.size()), dim3(NRM_BLOCK), lds, ctx->stream, d_wd, d_bw, d_bf, d_r, d_J,
                           (const int32_t *) ctx->d_fidx, (const int32_t *) (ctx->d_fidx + n), (const int32_t *) (ctx->d_fidx + 2 * (size_t) n), d_vp,
                           d_vm, Vmax, P, ctx->d_sys, d_act);
    }
    {
        icg_prof_scope ps(ctx, "schur_reduce");
        hipLaunchKernelGGL(k_schur_inv_w, dim3((Lmax + 255) / 256, W), dim3(256), 0, ctx->stream, d_wd, P, ctx->d_sys, min_diag, max_diag);
        hipLaunchKernelGGL(k_schur_reduce_w, dim3((P + SCH_T - 1) / SCH_T, (P + SCH_T - 1) / SCH_T, W), dim3(SCH_T, SCH_T), 0, ctx->stream, d_wd, P,
                           (const double *) ctx->d_sys, d_S, d_s, d_dg, (S_view || resident) ? 1 : 0);
        if (any_new) {
            // cost only of the windows that were re-assembled is meaningful; the others keep their previous value on the host side
            hipLaunchKernelGGL(k_reproj_cost_w, dim3(4, W), dim3(256), 0, ctx->stream, d_wd, d_r, d_act, ctx->last_huber, d_cost);
        }
    }
    ICG_HIP(ctx, hipGetLastError());
    auto t_launch = tnow();
    if (tdbg) {
        (void) hipStreamSynchronize(ctx->stream);
    }
    auto t_kernels = tnow();
    if ((rc = c.finish())) return rc;
    if (tdbg) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[icg_reproj_schur_windows] W=%d: host prep %.3f, h2d+launch %.3f, kernels %.3f, d2h+copy-out %.3f ms\n", W, ms(t_begin, t_prep),
                ms(t_prep, t_launch), ms(t_launch, t_kernels), ms(t_kernels, tnow()));
    }
    ctx->wsys_P = P, ctx->wsys_valid = 1;
    ctx->sys_min_diag = min_diag, ctx->sys_max_diag = max_diag;
    return ICG_OK;
}

// Problem-setup companion of the batched calls: sizes the resident window systems (W x ((P + L_w)^2 + ...) doubles of device memory) and the
// staging arena of the largest per-step call for reduced systems of size P, so that the first LM step of a solve does not pay a device
// allocation and a pinned re-allocation (4 ms at 256 C2 windows).  A hint: the calls themselves still grow what they need.
extern "C" int icg_reproj_reserve_windows(icg_ctx *ctx, int P) {
    if (!ctx || P <= 0) return ICG_ERR_INVALID;
    const int W = ctx->n_windows, n = ctx->n_factors_resident;
    if (W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition: call icg_reproj_set_windows first");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    size_t doubles = 0;
    for (int w = 0; w < W; w++) {
        const size_t N = (size_t) P + (size_t) (ctx->w_lm_off[(size_t) w + 1] - ctx->w_lm_off[(size_t) w]);
        doubles += N * N + N + (N - (size_t) P);
    }
    int rc = ensure_sys_capacity(ctx, doubles + 8);
    if (rc) return rc;
    icg_call c(ctx);
    return c.reserve(sizeof(win_desc) * (size_t) W + sizeof(int32_t) * ((size_t) ctx->last_n_poses + (size_t) W * (size_t) P + 2 * ((size_t) n / NRM_BLOCK + (size_t) W) + 16) +
                     (size_t) n + sizeof(double) * ((size_t) W * ((size_t) P * P + 2 * (size_t) P + 1)) + 8192);
}

extern "C" int icg_reproj_schur_windows(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                        const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag, double max_diag,
                                        double *S, double *s, double *diag_cc, double *cost) {
    if (!S) return ICG_ERR_INVALID;
    return schur_windows_impl(ctx, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, S, nullptr, s, diag_cc, cost);
}

extern "C" int icg_reproj_schur_windows_view(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                             const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag,
                                             double max_diag, const double **S_view, double *s, double *diag_cc, double *cost) {
    if (!S_view) return ICG_ERR_INVALID;
    return schur_windows_impl(ctx, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, nullptr, S_view, s, diag_cc, cost);
}

extern "C" int icg_reproj_schur_windows_resident(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                                 const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag,
                                                 double max_diag, double *s, double *diag_cc, double *cost) {
    return schur_windows_impl(ctx, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, nullptr, nullptr, s, diag_cc, cost);
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
    const int W = ctx->n_windows;
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
// LDS, P <= 142), then the landmark back-substitution of icg_reproj_backsub_windows with those steps, in one call: per LM step only rhs, dd
// go up and delta_c, ok, delta_l and the two model-decrease sums come back; the P x P systems never leave the device.
extern "C" int icg_reproj_solve_backsub_windows(icg_ctx *ctx, int P, const int32_t *Pw, const uint8_t *stepped, const double *rhs,
                                                const double *dd, double *delta_c, uint8_t *ok, double *delta_l, double *lm_terms) {
    if (!ctx || P <= 0 || !Pw || !stepped || !rhs || !dd || !delta_c || !ok) return ICG_ERR_INVALID;
    if (!ctx->wsys_valid || ctx->wsys_P != P || !ctx->d_redS)
        return icg_fail(ctx, ICG_ERR_INVALID, "no resident reduced systems of size %d: call icg_reproj_schur_windows_resident first", P);
    const int W = ctx->n_windows, n_lm = ctx->w_lm_off[(size_t) W];
    const size_t lds = sizeof(double) * ((size_t) P * P + (size_t) P);
    if (lds > RPJ_LDS_LIMIT) return icg_fail(ctx, ICG_ERR_CAPACITY, "reduced system of %d columns does not fit the LDS tile of the batched Cholesky (<= 142)", P);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (int rca = rpj_allow_lds(ctx, k_chol_solve_w, lds, 2)) return rca;
    int rc = icg_reproj_set_host_part_windows(ctx, P, 0, nullptr, nullptr); // (allocates a zero host part if none was ever set)
    if (rc) return rc;
    std::vector<win_desc> wd;
    build_win_desc(ctx, P, nullptr, nullptr, nullptr, nullptr, nullptr, wd);
    icg_call c(ctx);
    rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(int32_t) * (size_t) W + 2 * (size_t) W + sizeof(double) * (3 * (size_t) W * P + (size_t) n_lm + 2 * (size_t) W) + 8192);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const int32_t *d_pw  = c.in(Pw, (size_t) W);
    const uint8_t *d_st  = c.in(stepped, (size_t) W);
    const double *d_rhs  = c.in(rhs, (size_t) W * P);
    const double *d_dd   = c.in(dd, (size_t) W * P);
    std::vector<double> zeros(2 * (size_t) W, 0.0);
    double *d_tm = c.inout(zeros.data(), lm_terms, 2 * (size_t) W);
    if ((rc = c.seal())) return rc;
    double *d_dl   = n_lm > 0 && delta_l ? c.out(delta_l, (size_t) n_lm) : nullptr;
    double *d_dco  = c.out(delta_c, (size_t) W * P);
    uint8_t *d_ok  = c.out(ok, (size_t) W);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_cholesky");
        hipLaunchKernelGGL(k_chol_solve_w, dim3(W), dim3(256), lds, ctx->stream, P, d_pw, d_st, (const double *) ctx->d_redS, (const double *) ctx->d_hostS,
                           d_rhs, d_dd, d_dco, d_ok);
    }
    if (d_dl) {
        icg_prof_scope ps(ctx, "schur_backsub");
        hipLaunchKernelGGL(k_schur_backsub_w, dim3(n_lm), dim3(64), 0, ctx->stream, d_wd, (const int32_t *) ctx->d_lmwin, P, (const double *) ctx->d_sys,
                           (const double *) d_dco, d_dl, d_tm, ctx->sys_min_diag, ctx->sys_max_diag);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

// h_ll of every landmark (global landmark order of the partition) from the window systems left resident by the last
// icg_reproj_schur_windows*: the diagonal (P + l, P + l) of each window's N x N block
__global__ void k_lm_diag_w(const win_desc *wd, int P, const double *sys, double *h_ll) {
    const win_desc W = wd[blockIdx.y];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= W.L) return;
    const size_t N = (size_t) P + W.L;
    h_ll[W.lm_begin + l] = sys[W.sys_off + (size_t) (P + l) * N + P + l];
}

extern "C" int icg_reproj_landmark_diag_windows(icg_ctx *ctx, double *h_ll) {
    if (!ctx || !h_ll) return ICG_ERR_INVALID;
    if (!ctx->wsys_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident window systems: call icg_reproj_schur_windows first");
    const int W = ctx->n_windows, P = ctx->wsys_P, n_lm = ctx->w_lm_off[(size_t) W];
    if (n_lm == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<win_desc> wd;
    build_win_desc(ctx, P, nullptr, nullptr, nullptr, nullptr, nullptr, wd);
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

extern "C" int icg_reproj_backsub_windows(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    if (!ctx || !delta_c || !delta_l) return ICG_ERR_INVALID;
    if (!ctx->wsys_valid || ctx->wsys_P != P) return icg_fail(ctx, ICG_ERR_INVALID, "no resident window systems of size %d: call icg_reproj_schur_windows first", P);
    const int W = ctx->n_windows, n_lm = ctx->w_lm_off[(size_t) W];
    if (n_lm == 0) return ICG_OK;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<win_desc> wd;
    build_win_desc(ctx, P, nullptr, nullptr, nullptr, nullptr, nullptr, wd);
    icg_call c(ctx);
    int rc = c.reserve(sizeof(win_desc) * (size_t) W + sizeof(double) * ((size_t) W * P + (size_t) n_lm + 2 * (size_t) W) + 4096);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const double *d_dc   = c.in(delta_c, (size_t) W * P);
    std::vector<double> zeros(2 * (size_t) W, 0.0);
    double *d_tm = c.inout(zeros.data(), lm_terms, 2 * (size_t) W);
    if ((rc = c.seal())) return rc;
    double *d_dl = c.out(delta_l, (size_t) n_lm);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "schur_backsub");
        hipLaunchKernelGGL(k_schur_backsub_w, dim3(n_lm), dim3(64), 0, ctx->stream, d_wd, (const int32_t *) ctx->d_lmwin, P, (const double *) ctx->d_sys, d_dc,
                           d_dl, d_tm, ctx->sys_min_diag, ctx->sys_max_diag);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_reproj_cost_windows(icg_ctx *ctx, const uint8_t *active, double *cost) {
    if (!ctx || !cost) return ICG_ERR_INVALID;
    if (!ctx->rJ_valid) return icg_fail(ctx, ICG_ERR_INVALID, "no resident residuals: call icg_reproj_eval_windows first");
    const int W = ctx->n_windows, n = ctx->n_factors_resident;
    if (W <= 0) return icg_fail(ctx, ICG_ERR_INVALID, "no window partition");
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (ctx->w_sys_off.size() != (size_t) W + 1) ctx->w_sys_off.assign((size_t) W + 1, 0);
    if (ctx->w_damp.size() != (size_t) W) ctx->w_damp.assign((size_t) W, 0.0);
    std::vector<win_desc> wd;
    build_win_desc(ctx, 0, nullptr, nullptr, nullptr, nullptr, nullptr, wd);
    icg_call c(ctx);
    int rc = c.reserve(sizeof(win_desc) * (size_t) W + (size_t) n + sizeof(double) * (size_t) W + 4096);
    if (rc) return rc;
    const win_desc *d_wd = c.in(wd.data(), (size_t) W);
    const uint8_t *d_act = active ? c.in(active, (size_t) n) : nullptr;
    std::vector<double> zeros((size_t) W, 0.0);
    double *d_cost = c.inout(zeros.data(), cost, (size_t) W);
    if ((rc = c.seal())) return rc;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "reproj_cost");
        hipLaunchKernelGGL(k_reproj_cost_w, dim3(4, W), dim3(256), 0, ctx->stream, d_wd, (const double *) ctx->d_rJ, d_act, ctx->last_huber, d_cost);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
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
