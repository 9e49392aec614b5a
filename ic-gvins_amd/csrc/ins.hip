// f4: the two INS steps immediately in front of the tracker, batched over independent streams.
//
//   k_ins_mechanize    MISC::insMechanization (misc.cc:151-206) applied sample by sample: bias (and optional scale-factor)
//                      compensation, two-sample coning/sculling terms, Earth-rotation/Coriolis terms when iswithearth, attitude
//                      update with renormalisation, trapezoidal position update.  Strictly sequential inside a stream
//                      (every step reads the state the previous one wrote), embarrassingly parallel across streams: one lane
//                      per stream, the state lives in registers for the whole series.
//   k_ins_camera_pose  MISC::statePoseInterpolation (misc.cc:85-100) + stateToCameraPose (:102-108): the INS pose prior of a
//                      frame from the two window states that bracket its time stamp (the bracket search itself,
//                      getInsWindowIndex :30-65, is compare-only work on the host-resident window and stays in the host layer).
// Both are latency-bound FP64 scalar work (64 B in per IMU sample / 128 B per query): reported as samples/s, not against the
// HBM roofline.  sin/cos/atan2 come from the device math library: results agree with the CPU restatement to ~1e-15 relative
// per step, not bit for bit (tests: 1e-12 after 400 samples).
#include "dev_math.h"
#include "icg_internal.h"

using namespace icgd;

namespace {
struct ins_state {
    double time;
    d3 p;
    dq q;
    d3 v, bg, ba, sg, sa;
};
__device__ __forceinline__ ins_state load_state(const double *s) {
    ins_state st;
    st.time = s[0];
    st.p    = mk3(s[1], s[2], s[3]);
    st.q    = dq{s[4], s[5], s[6], s[7]};
    st.v    = mk3(s[8], s[9], s[10]);
    st.bg   = mk3(s[11], s[12], s[13]);
    st.ba   = mk3(s[14], s[15], s[16]);
    st.sg   = mk3(s[17], s[18], s[19]);
    st.sa   = mk3(s[20], s[21], s[22]);
    return st;
}
__device__ __forceinline__ void store_state(const ins_state &st, double *s) {
    s[0] = st.time;
    s[1] = st.p.x, s[2] = st.p.y, s[3] = st.p.z;
    s[4] = st.q.x, s[5] = st.q.y, s[6] = st.q.z, s[7] = st.q.w;
    s[8] = st.v.x, s[9] = st.v.y, s[10] = st.v.z;
    s[11] = st.bg.x, s[12] = st.bg.y, s[13] = st.bg.z;
    s[14] = st.ba.x, s[15] = st.ba.y, s[16] = st.ba.z;
    s[17] = st.sg.x, s[18] = st.sg.y, s[19] = st.sg.z;
    s[20] = st.sa.x, s[21] = st.sa.y, s[22] = st.sa.z;
}
__device__ __forceinline__ d3 mul3(d3 a, d3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
} // namespace

__global__ __launch_bounds__(64) void k_ins_mechanize(int n_streams, const int32_t *offsets, const double *imu, const double *cfg,
                                                      double *states, double *traj) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_streams) return;
    const int begin = offsets[s], n = offsets[s + 1] - offsets[s];
    const d3 gravity = mk3(cfg[0], cfg[1], cfg[2]), iewn = mk3(cfg[3], cfg[4], cfg[5]);
    const bool withearth = cfg[6] != 0.0, withscale = cfg[7] != 0.0;
    ins_state st = load_state(states + 23 * (size_t) s);
    if (traj && n > 0) store_state(st, traj + 23 * (size_t) begin);
    const d3 og = mk3(1.0 - st.sg.x, 1.0 - st.sg.y, 1.0 - st.sg.z), oa = mk3(1.0 - st.sa.x, 1.0 - st.sa.y, 1.0 - st.sa.z);
    // Earth-rotation compensation quaternion for a constant dt is recomputed per sample like the reference (dt may vary)
    for (int k = 1; k < n; k++) {
        const double *pp = imu + 8 * (size_t) (begin + k - 1), *pc = imu + 8 * (size_t) (begin + k);
        const double pre_dt = pp[1], dt = pc[1];
        d3 pre_dtheta = sub(mk3(pp[2], pp[3], pp[4]), scl(pre_dt, st.bg)), pre_dvel = sub(mk3(pp[5], pp[6], pp[7]), scl(pre_dt, st.ba));
        d3 cur_dtheta = sub(mk3(pc[2], pc[3], pc[4]), scl(dt, st.bg)), cur_dvel = sub(mk3(pc[5], pc[6], pc[7]), scl(dt, st.ba));
        if (withscale) { // misc.cc:161-168
            cur_dtheta = mul3(cur_dtheta, og), cur_dvel = mul3(cur_dvel, oa);
            pre_dtheta = mul3(pre_dtheta, og), pre_dvel = mul3(pre_dvel, oa);
        }
        st.time   = pc[0];
        d3 dvfb   = add(add(cur_dvel, scl(0.5, crs(cur_dtheta, cur_dvel))),
                        scl(1.0 / 12.0, add(crs(pre_dtheta, cur_dvel), crs(pre_dvel, cur_dtheta)))); // :174-175
        d3 dtheta = add(cur_dtheta, scl(1.0 / 12.0, crs(pre_dtheta, cur_dtheta)));                    // :176
        d3 dvel;
        if (withearth) { // :181-193
            d3 dv_cor_g = scl(dt, sub(gravity, scl(2.0, crs(iewn, st.v))));
            d3 dnn      = scl(dt, mk3(-iewn.x, -iewn.y, -iewn.z));
            dq qnn      = rotvec2quat(dnn);
            m33 half    = m_scale(m_add(m_eye(), q_mat(qnn)), 0.5);
            dvel        = add(m_vec(m_mul(half, q_mat(st.q)), dvfb), dv_cor_g);
            st.q        = q_normalized(q_mul(q_mul(qnn, st.q), rotvec2quat(dtheta)));
        } else { // :194-200
            dvel = add(m_vec(q_mat(st.q), dvfb), scl(dt, gravity));
            st.q = q_normalized(q_mul(st.q, rotvec2quat(dtheta)));
        }
        st.p = add(st.p, add(scl(dt, st.v), scl(0.5 * dt, dvel))); // :203
        st.v = add(st.v, dvel);                                    // :205
        if (traj) store_state(st, traj + 23 * (size_t) (begin + k));
    }
    store_state(st, states + 23 * (size_t) s);
}

// brackets: n x 16 = (time, p3, q4) of the state before and after the query time; interp[i] == 0 -> the first state as is
__global__ __launch_bounds__(64) void k_ins_camera_pose(int n, const double *brackets, const int32_t *interp, const double *pose_b_c,
                                                        const double *times, double *pose12) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double *b = brackets + 16 * (size_t) i;
    d3 p = mk3(b[1], b[2], b[3]);
    dq q = dq{b[4], b[5], b[6], b[7]};
    if (interp[i]) { // misc.cc:85-100
        const double t0 = b[0], t1 = b[8];
        d3 p1 = mk3(b[9], b[10], b[11]);
        dq q1 = dq{b[12], b[13], b[14], b[15]};
        d3 dp = sub(p1, p);
        dq dqq = q_mul(q_inv(q1), q);
        d3 rvec = quat2rotvec(dqq);
        double scale = (times[i] - t0) / (t1 - t0);
        rvec = scl(scale, rvec);
        dqq  = rotvec2quat(rvec);
        p    = add(p, scl(scale, dp));
        q    = q_normalized(q_mul(q, q_inv(dqq)));
    }
    m33 R = q_mat(q), Rb; // misc.cc:102-108
#pragma unroll
    for (int k = 0; k < 9; k++) Rb.a[k] = pose_b_c[k];
    d3 t   = add(p, m_vec(R, mk3(pose_b_c[9], pose_b_c[10], pose_b_c[11])));
    m33 Rc = m_mul(R, Rb);
    double *o = pose12 + 12 * (size_t) i;
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = Rc.a[k];
    o[9] = t.x, o[10] = t.y, o[11] = t.z;
}

extern "C" int icg_ins_mechanize_batch(icg_ctx *ctx, int n_streams, const int32_t *offsets, const double *imu, const double *cfg8,
                                       double *states23, double *traj23) {
    if (!ctx || n_streams < 0) return ICG_ERR_INVALID;
    if (n_streams == 0) return ICG_OK;
    if (!offsets || !imu || !cfg8 || !states23) return ICG_ERR_INVALID;
    const int total = offsets[n_streams];
    for (int s = 0; s < n_streams; s++)
        if (offsets[s + 1] < offsets[s]) return icg_fail(ctx, ICG_ERR_INVALID, "stream %d: offsets not monotone", s);
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    int rc = c.reserve((size_t) total * (64 + (traj23 ? 184 : 0)) + (size_t) n_streams * (184 * 2 + 8) + 1024);
    if (rc) return rc;
    const int32_t *d_off = c.in(offsets, (size_t) n_streams + 1);
    const double *d_imu  = c.in(imu, 8 * (size_t) total);
    const double *d_cfg  = c.in(cfg8, 8);
    double *d_st         = c.inout(states23, states23, 23 * (size_t) n_streams); // read and written in place
    if ((rc = c.seal())) return rc;
    double *d_traj = traj23 ? c.out(traj23, 23 * (size_t) total) : nullptr;
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "ins_mechanize");
        hipLaunchKernelGGL(k_ins_mechanize, dim3((n_streams + 63) / 64), dim3(64), 0, ctx->stream, n_streams, d_off, d_imu, d_cfg, d_st,
                           d_traj);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}

extern "C" int icg_ins_camera_pose_batch(icg_ctx *ctx, int n, const double *brackets16, const int32_t *interp, const double *pose_b_c12,
                                         const double *times, double *pose12_out) {
    if (!ctx || n < 0) return ICG_ERR_INVALID;
    if (n == 0) return ICG_OK;
    if (!brackets16 || !interp || !pose_b_c12 || !times || !pose12_out) return ICG_ERR_INVALID;
    ICG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    icg_call c(ctx);
    int rc = c.reserve((size_t) n * (128 + 4 + 8 + 96) + 1024);
    if (rc) return rc;
    const double *d_b   = c.in_zc(brackets16, 16 * (size_t) n);
    const int32_t *d_i  = c.in_zc(interp, (size_t) n);
    const double *d_pbc = c.in_zc(pose_b_c12, 12);
    const double *d_t   = c.in_zc(times, (size_t) n);
    double *d_o         = c.out_zc(pose12_out, 12 * (size_t) n);
    ICG_LAUNCH_GUARD(c);
    {
        icg_prof_scope ps(ctx, "ins_camera_pose");
        hipLaunchKernelGGL(k_ins_camera_pose, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, n, d_b, d_i, d_pbc, d_t, d_o);
    }
    ICG_HIP(ctx, hipGetLastError());
    return c.finish();
}
