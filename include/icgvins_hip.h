/*
 * icgvins_hip.h — C ABI of the MI355X-native IC-GVINS hot path (libicgvins_hip.so).
 *
 * This is the drop-in boundary B3 of SURVEY.md §8(b): plain pointers and sizes, no C++/torch types, no exceptions.
 * Every entry point names the reference interface it replaces (paths relative to /root/reference/ic_gvins/ic_gvins/).
 *
 * Conventions
 *   - return value: 0 = ICG_OK, <0 = error code; icg_last_error(ctx) gives a message.
 *   - an icg_ctx owns one HIP stream + all device buffers for one (GPU, image size); it is NOT thread-safe;
 *     distinct contexts are independent.  Caller owns every host buffer passed in.
 *   - every call is batched: "jobs" index frame *slots* (device-resident CLAHE image + pyramid), so one call can
 *     serve many independent camera streams at once (the unit that fills an MI355X; SURVEY.md §0 "scale honesty").
 *   - calls are synchronous unless stated: results are in the host buffers when the call returns.
 *   - points are float pairs (x,y) == cv::Point2f; status bytes are 0/1 == the reference's vector<uint8_t>.
 *   - there is NO CPU fallback: without a HIP device icg_ctx_create fails with ICG_ERR_NODEVICE.
 */
#ifndef ICGVINS_HIP_H
#define ICGVINS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icg_ctx icg_ctx;

enum {
    ICG_OK           = 0,
    ICG_ERR_INVALID  = -1, /* bad argument */
    ICG_ERR_HIP      = -2, /* HIP runtime error (see icg_last_error) */
    ICG_ERR_NOMEM    = -3,
    ICG_ERR_NODEVICE = -4, /* no usable gfx950 device: the product path refuses to run */
    ICG_ERR_CAPACITY = -5  /* batch larger than the capacity given at icg_ctx_create */
};

/* Camera intrinsics/distortion, tracking/camera.h:88-93 (fx,fy,cx,cy,skew ; k1,k2,p1,p2,k3). */
typedef struct icg_camera {
    double fx, fy, cx, cy, skew;
    double k1, k2, p1, p2, k3;
} icg_camera;

typedef struct icg_ctx_config {
    int device;     /* HIP device ordinal */
    int width;      /* image width  (camera_->width())  */
    int height;     /* image height (camera_->height()) */
    int n_slots;    /* frame slots kept resident (>= 2 per stream: frame_pre_ and frame_cur_, +1 per ref frame) */
    int max_batch;  /* max frames per icg_frames_preprocess / icg_detect call */
    int max_points; /* max points per point-batched call */
    int max_factors;/* max reprojection factors per icg_reproj_eval_batch call (0 = back-end unused) */
} icg_ctx_config;

/* ---- context ----------------------------------------------------------------------------------------- */
int icg_ctx_create(const icg_ctx_config *cfg, icg_ctx **out);
void icg_ctx_destroy(icg_ctx *ctx);
const char *icg_last_error(const icg_ctx *ctx);
int icg_ctx_sync(icg_ctx *ctx);
/* How a call waits for its kernels (no reference counterpart; the reference's OpenCV calls are synchronous CPU code):
 *   ICG_WAIT_SPIN   busy-wait on the stream: lowest latency, occupies a host core (default; one tracker per process)
 *   ICG_WAIT_POLL   query + sleep sleep_us between queries: the core is free for other contexts' threads (many contexts
 *                   per host core, see host/tracking_batch.h)
 * The environment variable ICG_WAIT_MODE=spin|poll[:us] overrides the setting of every context. */
#define ICG_WAIT_SPIN 0
#define ICG_WAIT_POLL 1
int icg_ctx_set_wait_mode(icg_ctx *ctx, int mode, int sleep_us);
/* hipStream_t of the context (for callers that want to record their own HIP events on it). */
void *icg_ctx_stream(icg_ctx *ctx);
int icg_set_camera(icg_ctx *ctx, const icg_camera *cam);
const char *icg_version(void);
/* number of pyramid levels built for the context's image size (== maxLevel+1 of calcOpticalFlowPyrLK). */
int icg_pyramid_levels(const icg_ctx *ctx);

/* Per-kernel timing with HIP events recorded on the context stream around every launch of `kernel_name`
 * (e.g. "lk_track_fb"); used by bench.py for the roofline figures.  enable=0 disables and clears. */
int icg_prof_enable(icg_ctx *ctx, int enable);
int icg_prof_get(icg_ctx *ctx, const char *kernel_name, int *launches, double *total_ms);
/* names of profiled kernels, '\n' separated, into buf */
int icg_prof_names(icg_ctx *ctx, char *buf, int buflen);

/* Device scratch for callers that keep inputs resident in HBM (bench harness): plain hipMalloc/hipFree/hipMemcpy. */
int icg_dev_alloc(icg_ctx *ctx, size_t bytes, void **dptr);
int icg_dev_free(icg_ctx *ctx, void *dptr);
int icg_dev_upload(icg_ctx *ctx, void *dptr, const void *host, size_t bytes);
int icg_dev_download(icg_ctx *ctx, void *host, const void *dptr, size_t bytes);

/* ---- F1: Tracking::preprocessing (tracking/tracking.cc:107-142) ----------------------------------------
 * For each job k: BGR->gray if channels==3 (tracking.cc:112), CLAHE(3.0, 21x21) (tracking.cc:63,139), then the
 * LK pyramid (the buildOpticalFlowPyramid every calcOpticalFlowPyrLK call redoes, tracking.cc:385-393) into slot
 * slots[k].  images[k] points to host memory (src_on_device=0) or device memory (src_on_device=1).
 * hist_mean (optional, n doubles) receives calculateHistigram() of the gray image (tracking.cc:88-105). */
int icg_frames_preprocess(icg_ctx *ctx, int n, const int32_t *slots, const uint8_t *const *images, int stride,
                          int channels, int src_on_device, double *hist_mean);
/* copy pyramid level `level` of a slot back to the host (Frame::image() == level 0, after CLAHE). */
int icg_frame_download(icg_ctx *ctx, int slot, int level, uint8_t *dst, int dst_stride);

/* ---- F2: cv::calcOpticalFlowPyrLK as called at tracking.cc:385-393 / 487-496 ---------------------------
 * win 21x21, maxLevel 3, (COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4.
 * next_pts: initial flow in, result out.  status/err as OpenCV (err may be NULL). */
int icg_lk_track(icg_ctx *ctx, int n, const int32_t *prev_slot, const int32_t *next_slot, const float *prev_pts,
                 float *next_pts, uint8_t *status, float *err);

/* ---- F2+F3(+F4): the fused forward/backward track of tracking.cc:380-403 and :482-506 --------------------
 * forward LK prev->next from guess_pts, backward LK next->prev from prev_pts, then
 * status = fwd && bwd && !isOnBorder(fwd) (tracking.cc:847-849) && ptsDistance(bwd, prev) < 0.5 (:841-845).
 * out_undist (optional) = Camera::undistortPoints(out_pts) (camera.cc:72-74), needs icg_set_camera.
 * keep_idx/n_keep (optional) = order-preserving compaction indices, i.e. what reduceVector (tracking.cc:831-839)
 * keeps, computed on device. */
int icg_lk_track_fb(icg_ctx *ctx, int n, const int32_t *prev_slot, const int32_t *next_slot, const float *prev_pts,
                    const float *guess_pts, float *out_pts, uint8_t *status, float *out_undist, int32_t *keep_idx,
                    int32_t *n_keep);

/* ---- F4: Camera point maps (tracking/camera.cc) -------------------------------------------------------- */
int icg_undistort_points(icg_ctx *ctx, int n, float *pts);                 /* camera.cc:72-74  in place */
int icg_distort_points(icg_ctx *ctx, int n, float *pts);                   /* camera.cc:76-89  in place */
/* ---- F5: INS-aided predictions --------------------------------------------------------------------------
 * map points: world2pixel(pos, pose_cur) then distortPoints (tracking.cc:367-378).  pose = R(9 row-major), t(3);
 * pose_idx[k] selects the pose of point k (one pose per stream). */
int icg_predict_mappoints(icg_ctx *ctx, int n, const double *pw, const int32_t *pose_idx, int n_poses,
                          const double *poses12, float *pts_out);
/* reference features: distortCameraPoint(R_cur^T R_pre * pixel2cam(undistort(p))) (tracking.cc:465-479).
 * rot_idx[k] selects r_cur_pre (n_rots x 9 row-major, already R_cur^T*R_pre). */
int icg_predict_rotation(icg_ctx *ctx, int n, const float *pts_in, const int32_t *rot_idx, int n_rots,
                         const double *rots9, float *pts_out);

/* ---- F6: cv::findFundamentalMat(FM_RANSAC, thresh, conf, mask) as used at tracking.cc:547-555 -----------
 * Batched over independent point sets: set s owns points [offsets[s], offsets[s+1]).  Sets with fewer than 15
 * points are left untouched (mask = 1), as the reference skips them.  mask: 0/1 per point. */
int icg_fm_ransac(icg_ctx *ctx, int n_sets, const int32_t *offsets, const float *pts1, const float *pts2,
                  double thresh, double conf, uint8_t *mask);
/* the same masks from ONE launch: every set's whole RANSAC run — subset draws from the set's cv::RNG, seven-point solves, scoring, the
 * best / niters recurrence of RANSACPointSetRegistrator::run — inside the workgroup that owns the set (csrc/ransac.hip k_fm_ransac_sets);
 * the device-resident tracker runs the same kernel on its segments.  Sets of more than 1024 points go through icg_fm_ransac. */
int icg_fm_ransac_device(icg_ctx *ctx, int n_sets, const int32_t *offsets, const float *pts1, const float *pts2, double thresh,
                         double conf, uint8_t *mask);

/* ---- F7: Tracking::featuresDetection (tracking.cc:576-688) ----------------------------------------------
 * Gridded cv::goodFeaturesToTrack(quality 0.01, minDistance, mask) + cv::cornerSubPix((5,5),(-1,-1),(20,0.01))
 * per block ROI, for n jobs at once.  Job k detects on slot slots[k]; its mask discs (cv::circle radius
 * min_dist, tracking.cc:609-620) are centred on mask_pts[mask_off[k]..mask_off[k+1]); its per-block quotas
 * (track_max_block_features_ - features_cnts[b], tracking.cc:629) are quota[k*n_blocks + b] (<=0: skip block).
 * Output: out_pts holds up to max_per_job points per job in block order with block origin already added
 * (tracking.cc:669-685); out_count[k] = points found; out_block (optional) = block id per point. */
typedef struct icg_detect_grid {
    int block_cols, block_rows; /* block_cols_, block_rows_  tracking.cc:66-67 */
    int block_w, block_h;       /* block_indexs_[0]          tracking.cc:71-73 */
    int min_dist;               /* track_min_pixel_distance_ tracking.cc:85    */
    int max_per_block;          /* track_max_block_features_ tracking.cc:81    */
} icg_detect_grid;
int icg_detect(icg_ctx *ctx, int n, const int32_t *slots, const icg_detect_grid *grid, const int32_t *mask_off,
               const float *mask_pts, const int32_t *quota, int max_per_job, float *out_pts, int32_t *out_count,
               int32_t *out_block);

/* ---- F8: Tracking::triangulatePoint (tracking.cc:800-811), batched ---------------------------------------
 * T0/T1: per-point indices into Tcw (n_T x 12, row-major 3x4 == pose2Tcw(pose).topRows<3>()); pc0/pc1: pixel2cam. */
int icg_triangulate(icg_ctx *ctx, int n, const int32_t *T0_idx, const int32_t *T1_idx, int n_T, const double *Tcw12,
                    const double *pc0, const double *pc1, double *pw);

/* ---- R1: ReprojectionFactor::Evaluate (factors/reprojection_factor.h:55-147), batched ---------------------
 * obs_soa: 15 x n doubles, component-major: pts0[3], pts1[3], vel0[3], vel1[3], td0, td1, std (ctor :42-53).
 * poses: K x 7 [px,py,pz,qx,qy,qz,qw] (parameters[0], parameters[1]); ext: 7 (parameters[2]);
 * invdepth: L (parameters[3]); td (parameters[4]).
 * out_r: n x 2.  out_J (want_jac): n x 46 = J_pose_i[2x7], J_pose_j[2x7], J_ext[2x7] row-major (7th col 0),
 * J_invdepth[2], J_td[2] — exactly the blocks Ceres hands to Evaluate.
 * huber_delta > 0 additionally applies ResidualBlockInfo's robust correction (factors/residual_block_info.h:59-87),
 * i.e. R2 as used by marginalization; 0 = plain Evaluate. */
int icg_reproj_eval_batch(icg_ctx *ctx, int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j,
                          const int32_t *idx_lm, int n_poses, const double *poses, const double *ext, int n_lm,
                          const double *invdepth, double td, int want_jac, double huber_delta, double *out_r,
                          double *out_J);
/* Same with the static part (obs, indices) already resident: upload once per Ceres problem, evaluate many times. */
int icg_reproj_set_factors(icg_ctx *ctx, int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j,
                           const int32_t *idx_lm);
/* The same upload in two steps for callers that assemble a large factor set from many threads (the windows of many streams): the context
 * hands out PINNED host memory for n factors — *obs_soa: 15 x n doubles, component-major as above; *idx3: 3 x n int32, idx_i | idx_j |
 * idx_lm — the caller fills it in place, icg_reproj_commit_factors uploads it (equivalent to icg_reproj_set_factors on the same content).
 * The pointers are valid until the next stage / set call on ctx. */
int icg_reproj_stage_factors(icg_ctx *ctx, int n, double **obs_soa, int32_t **idx3);
int icg_reproj_commit_factors(icg_ctx *ctx);
int icg_reproj_eval_resident(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm,
                             const double *invdepth, double td, int want_jac, double huber_delta, double *out_r,
                             double *out_J);
/* the same evaluation with r (n x 2) and J (n x 46) left in the context's pinned staging memory: *r_view / *J_view are valid until the
 * next call on ctx.  ReprojectionBatch (the ceres::EvaluationCallback of boundary B1) hands each factor's Evaluate() its slice of the views. */
int icg_reproj_eval_resident_view(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth,
                                  double td, int want_jac, double huber_delta, const double **r_view, const double **J_view);

/* ---- M2: MarginalizationInfo::constructEquation (factors/marginalization_info.h:195-230) for the reprojection
 * factors of the last icg_reproj_eval_* call (Jacobians still resident): accumulates H0 += J^T J, b0 -= J^T e into
 * a dense (local_size x local_size) system on device and adds it to the host arrays H0/b0.
 * col_pose[k] (n_poses), col_ext, col_lm[l] (n_lm), col_td: local column index of each parameter block, -1 = constant. */
int icg_reproj_accumulate_normal(icg_ctx *ctx, int local_size, const int32_t *col_pose, int32_t col_ext,
                                 const int32_t *col_lm, int32_t col_td, double *H0, double *b0);

/* ---- f1 (SURVEY.md §8 "next" row): device-side landmark elimination for the Gauss-Newton / LM step of
 * GVINS::gvinsOptimization (ic_gvins.cc:1130-1239: Ceres LEVENBERG_MARQUARDT + DENSE_SCHUR).  Works on the robust-corrected r/J
 * left resident by the last icg_reproj_eval_resident(want_jac=1): assembles H = J^T J, b = -J^T r of the ACTIVE factors on the
 * device (camera columns 0..P-1 given by col_pose/col_ext/col_td, -1 = constant block; inverse depth l at column P+l), then
 * eliminates the 1x1 inverse-depth blocks:  S = Hcc - G^T diag(1/(h_ll+d_l)) G,  s = bc - G^T (b_l/(h_ll+d_l)),
 * d_l = clamp(h_ll, min_diag, max_diag) * damp (LM diagonal of the eliminated block; damp = 1/trust-region radius).
 * Outputs: S (P x P row-major), s (P), diag_cc (P, diagonal of Hcc before elimination: the host needs it for the LM diagonal
 * of the camera block; may be NULL), cost (0.5 sum rho(|r|^2) of the active factors at the linearization point; may be NULL).
 * reassemble = 0 re-uses the resident H, b with a new damp (after a rejected step; cost is not touched).
 * The full system stays on the device for icg_reproj_backsub:  delta_l = (b_l - G_l . delta_c) / (h_ll + d_l)  (n_lm values);
 * lm_terms (2 doubles, may be NULL) = sum b_l^2/(h_ll+d_l), sum d_l delta_l^2: the landmark part of the LM model decrease
 * 0.5 (delta^T b + delta^T D delta), where delta^T b = delta_c^T s + lm_terms[0].
 * icg_reproj_cost: the same cost for the CURRENT resident residuals (e.g. after a want_jac=0 evaluation at a trial point). */
int icg_reproj_schur(icg_ctx *ctx, int P, const int32_t *col_pose, int32_t col_ext, int32_t col_td, const uint8_t *active,
                     int reassemble, double damp, double min_diag, double max_diag, double *S, double *s, double *diag_cc,
                     double *cost);
int icg_reproj_backsub(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms);
int icg_reproj_cost(icg_ctx *ctx, const uint8_t *active, double *cost);
/* h_ll (n_lm values) of the system left resident by the last icg_reproj_schur: the diagonal of the inverse-depth block BEFORE damping.
 * M3 (factors/marginalization_info.h:170-192) uses it to decide whether the landmark block may be eliminated by plain reciprocals
 * (all h_ll well above the reference's 1e-8 eigenvalue floor) or has to go through the dense pseudo-inverse. */
int icg_reproj_landmark_diag(icg_ctx *ctx, double *h_ll);

/* f1, many windows per launch: the windows of many camera streams advance through their LM steps together, ONE evaluation / assembly /
 * reduction / back-substitution launch per step for all of them (a solver per stream is bounded by the runtime's launch rate).
 * icg_reproj_set_windows partitions the resident factor set: factors [fac_off[w], fac_off[w+1]) and landmarks [lm_off[w], lm_off[w+1])
 * belong to window w (factors sorted by window, landmarks contiguous per window, every pose used by one window only; poses and
 * landmarks keep their global indices).  The *_windows calls mirror icg_reproj_eval_resident / _schur / _backsub / _cost with one
 * extrinsic (ext: W x 7) and td per window, one reduced system of the common size P per window (S: W x P x P, s / diag_cc: W x P; a
 * window simply leaves the columns it does not use empty), per-window damping and per-window reassemble flags (0 = keep the window's
 * resident H, b and only apply the new damping).  cost (W) is written for the re-assembled windows only (0 elsewhere); lm_terms is
 * W x 2.  col_pose[k] is the column of pose k INSIDE its window's system. */
int icg_reproj_set_windows(icg_ctx *ctx, int n_windows, const int32_t *fac_off, const int32_t *lm_off);
int icg_reproj_eval_windows(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth,
                            const double *td, int want_jac, double huber_delta);
int icg_reproj_schur_windows(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                             const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, double *S, double *s,
                             double *diag_cc, double *cost);
/* the same call with the W x P x P reduced systems left where the reduction kernel writes them (the context's pinned staging memory):
 * *S_view is valid until the next call on ctx.  Saves the device-to-host copy and the copy-out of 9 MB per LM step at 256 C2 windows.
 * Only the lower triangle (rows >= columns) is written (the systems are symmetric; a Cholesky factorization reads nothing else): the
 * elements above the diagonal are undefined. */
int icg_reproj_schur_windows_view(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                                  const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, const double **S_view, double *s,
                                  double *diag_cc, double *cost);
/* The reduced camera systems of icg_reproj_schur_windows solved ON THE DEVICE (what Ceres' DENSE_SCHUR does per LM iteration on the host:
 * ic_gvins.cc:1143-1146, 1183, 1217): icg_reproj_schur_windows_resident leaves the W lower-triangular P x P systems in device memory (s,
 * diag_cc and cost come back as before); icg_reproj_set_host_part_windows uploads, for the listed windows, the packed lower triangle
 * (P(P+1)/2 doubles, row by row) of what the HOST-evaluated factors (preintegration, marginalization prior, priors) add to the system — it
 * stays until replaced, a re-damped step re-uses it; icg_reproj_solve_backsub_windows factors (S_w + host_w + diag(dd_w)) by a batched
 * Cholesky (one workgroup per window, LDS, P <= 88) for every window with stepped[w] != 0, solves for delta_c (leading Pw[w] columns,
 * rhs = the window's full gradient), sets ok[w] = 0 where a pivot is not positive (the caller re-damps, as it does when its own
 * factorization fails), and runs the landmark back-substitution of icg_reproj_backsub_windows with those steps in the same call. */
int icg_reproj_schur_windows_resident(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td,
                                      const uint8_t *active, const uint8_t *reassemble, const double *damp, double min_diag, double max_diag,
                                      double *s, double *diag_cc, double *cost);
int icg_reproj_set_host_part_windows(icg_ctx *ctx, int P, int n_upd, const int32_t *win_idx, const double *packed);
int icg_reproj_solve_backsub_windows(icg_ctx *ctx, int P, const int32_t *Pw, const uint8_t *stepped, const double *rhs, const double *dd,
                                     double *delta_c, uint8_t *ok, double *delta_l, double *lm_terms);
/* problem setup: pre-sizes the resident window systems and the staging memory for reduced systems of size P (a hint; optional) */
int icg_reproj_reserve_windows(icg_ctx *ctx, int P);
int icg_reproj_backsub_windows(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms);
int icg_reproj_cost_windows(icg_ctx *ctx, const uint8_t *active, double *cost);
/* h_ll of every landmark of the partition (n_lm values, global landmark order) from the window systems left resident by the last
 * icg_reproj_schur_windows*: icg_reproj_landmark_diag for many windows — the conditioning guard of the batched M3
 * (factors/marginalization_info.h:170-192 for the marginalizations of many streams, host/marg_batch.h). */
int icg_reproj_landmark_diag_windows(icg_ctx *ctx, double *h_ll);
/* the resident residuals of the last evaluation (n x 2 doubles): per-factor tests (chi-square culling) after a resident evaluation */
int icg_reproj_fetch_residuals(icg_ctx *ctx, double *out_r);
/* GVINS::removeReprojectionFactorsByChi2 (ic_gvins.cc:1269-1297) on the resident residuals of a want_jac = 0, huber = 0 evaluation:
 * factor f stays active iff it was active and NOT  (0.5 |r_f|^2) * 2.0 > chi2  (the reference's test on EvaluateResidualBlock's cost).
 * active (n bytes) is updated in place; only the flags cross the link instead of 16 bytes of residual per factor. */
int icg_reproj_chi2_cull(icg_ctx *ctx, double chi2, uint8_t *active);

/* ---- P1: preintegration inner loop (preintegration/preintegration_base.cc:39-70, preintegration_earth.cc:205-303,
 * preintegration_normal.cc:183-232), batched over independent intervals.
 * imu: total x 8 doubles (time, dt, dtheta[3], dvel[3]); interval s owns samples [offsets[s], offsets[s+1]) with
 * sample 0 = imu0.  state0: n x 16 (p3, q4 xyzw, v3, bg3, ba3).  params: 9 doubles
 * (gyr_arw, acc_vrw, gyr_bias_std, acc_bias_std, corr_time, gravity, iewn[3]).  variant: 0 Normal, 1 Earth.
 * outputs per interval: cur_state 16, delta_state 16, jac 225, cov 225, delta_time 1.
 * pn (optional, total x 4): the Earth variant's pn_ list (preintegration_earth.cc:235) — (dt, position) after sample
 * index of interval s is stored at row offsets[s] + index - 1. */
int icg_preint_batch(icg_ctx *ctx, int variant, int n_intervals, const int32_t *offsets, const double *imu,
                     const double *state0, const double *params, double *cur_state, double *delta_state, double *jac,
                     double *cov, double *delta_time, double *pn);

/* ---- f3 (SURVEY.md §8 "next" row): per-observation arithmetic of GVINS::gvinsOutlierCulling (ic_gvins.cc:1035-1128) and
 * GVINS::parametersStatistic (ic_gvins.cc:930-1033).  Observation i = landmark lm_idx[i] (world position pw, n_lm x 3) seen in
 * keyframe pose_idx[i] (poses12: n_poses x 12, R row-major camera->world | t) at the undistorted key point pix[i]:
 *   err_out[i]  = |Camera::reprojectionError(pose, pw, pp)|                (tracking/camera.cc:153-157)
 *   good_out[i] = Tracking::isGoodToTrack(pp, pose, pw, scale, depth_scale)  (tracking/tracking.cc:813-829) with
 *                 max_error = reprojection_error_std * scale, min_depth = MapPoint::NEAREST_DEPTH,
 *                 max_depth = MapPoint::FARTHEST_DEPTH * depth_scale.
 * The camera set with icg_set_camera is used. */
int icg_reproj_error_batch(icg_ctx *ctx, int n, const int32_t *pose_idx, const int32_t *lm_idx, int n_poses, const double *poses12,
                           int n_lm, const double *pw, const float *pix, double max_error, double min_depth, double max_depth,
                           double *err_out, uint8_t *good_out);

/* ---- f4 (SURVEY.md §8 "next" row): the INS steps in front of the tracker, batched over independent streams -----------
 * Layouts: imu rows of 8 doubles (time, dt, dtheta[3], dvel[3]); state rows of 23 doubles (time, p3, q4 xyzw, v3, bg3, ba3,
 * sg3, sa3 = IntegrationState, preintegration/integration_state.h:35-52 without the odometer fields);
 * cfg8 = gravity[3], iewn[3], iswithearth, iswithscale (IntegrationConfiguration, integration_state.h:91-99).
 *
 * icg_ins_mechanize_batch: MISC::insMechanization (misc.cc:151-206) applied in sequence to the samples
 * offsets[s]+1 .. offsets[s+1]-1 of stream s (sample offsets[s] is imu_pre of the first step), starting from states23[s]
 * (updated in place to the state after the last sample).  traj23 (optional, total x 23): row offsets[s] = the start state,
 * row offsets[s]+k = the state after sample k — i.e. the (IMU, state) window the reference keeps in ins_window_
 * (ic_gvins.cc:270-290) and re-propagates in MISC::redoInsMechanization (misc.cc:208-261). */
int icg_ins_mechanize_batch(icg_ctx *ctx, int n_streams, const int32_t *offsets, const double *imu, const double *cfg8,
                            double *states23, double *traj23);
/* icg_ins_camera_pose_batch: the INS pose prior of n frames, MISC::getCameraPoseFromInsWindow (misc.cc:67-83) after its
 * bracket search: brackets16[i] = (time, p3, q4) of the window states before and after times[i]; interp[i] != 0 ->
 * statePoseInterpolation (misc.cc:85-100), interp[i] == 0 -> the first state as is (the reference's fallback to the newest
 * state when the time is outside the window); then stateToCameraPose (misc.cc:102-108) with the body->camera extrinsic
 * pose_b_c12 (R row-major 9, t 3).  pose12_out: n x 12, same layout (camera->world). */
int icg_ins_camera_pose_batch(icg_ctx *ctx, int n, const double *brackets16, const int32_t *interp, const double *pose_b_c12,
                              const double *times, double *pose12_out);

/* ---- device-resident tracker of a stream group (round 4) ---------------------------------------------------------------------
 * Replaces, for n independent camera streams at once, everything Tracking::track (tracking/tracking.cc:144-245) does BETWEEN its image
 * primitives: the state machine (first frame / initializing / tracking / lost), trackMappoint (:351-455), trackReferenceFrame (:457-574),
 * reduceVector (:831-845), the parallax sums and keyframe decision (:263-307, 873-922), triangulation bookkeeping (:690-798), the detection
 * lists (:576-688), and — as the throughput harness does for GVINS — the sliding-window side effects on the map (map.cc:27-127).  The
 * per-stream state (frames with their feature rows in the reference container's iteration order, map points, candidate lists, window) is
 * one flat block per stream in HBM; a step is ONE chain of launches on the context's stream — stage kernel, primitive, stage kernel, ... —
 * with the work lists of every primitive left in device memory by the stage kernel before it, and ONE wait at the end.  The host never
 * builds a list, never sizes a grid from results and never reads the tracker's state unless it asks for a block.
 *
 * The stage bodies are the functions of ic-gvins_amd/host/track_core.h, compiled for gfx950; the same source compiled for the host is the
 * CPU twin the parity tests pin against the track table and the reference's own tracker.  icg_tracker_config mirrors tc::Cfg of that file
 * (static_assert in csrc/tracker.hip); a block is sizeof(tc::Stream) bytes = icg_tracker_block_bytes(). */
typedef struct icg_tracker icg_tracker;
typedef struct icg_tracker_config {
    double fx, fy, cx, cy, skew, k1, k2, p1, p2, k3; /* camera (icg_camera) */
    int32_t width, height;
    int32_t track_max_features, check_histogram, window_size, pad0;
    double track_min_parallax, reprojection_error_std, track_max_interval; /* track_max_interval already x 0.95 (tracking.cc:57) */
    int32_t block_cols, block_rows, block_cnts, block_w, block_h, max_block_features, min_pixel_distance, max_per_job; /* tracking.cc:66-85 */
    uint64_t stream_id_base;
} icg_tracker_config;
typedef struct icg_tracker_result { /* per stream, after a step */
    int32_t active;          /* the stream had a frame in this step */
    int32_t state;           /* TrackState of the frame (tracking.h:38-44) */
    int32_t is_new_keyframe; /* Tracking::isNewKeyFrame() */
    int32_t overflow;        /* != 0: a capacity of the block was exceeded (the step fails) */
    int32_t n_features, n_candidates, window_keyframes, landmarks;
    uint64_t frames, keyframes, tracked_sum, digest; /* running statistics / digest of everything index-like the stream produced */
    uint64_t frame_id, keyframe_id, mappoint_id, last_input_fid; /* id factories (frame.cc:37-53, mappoint.cc:45-49) */
    int32_t need_detect_a;   /* the stream's next frame starts with a detection (first frame / initialization without candidates) */
    int32_t n_log;           /* landmark-container operations logged in the block since the last drain */
    int32_t lk_points, detect_jobs, ransac_sets, tri_points; /* work this frame handed to the primitives */
    /* the frame's line of tracking.txt (tracking.cc:236-238, 309-315), when it has one: stamp, dt, parallax, translation, rotation [deg] as
     * kept at the keyframe decision, and the frame's feature count before the window keeper ran; the caller appends its own time cost */
    int32_t log_valid, log_features;
    double log_data[5];
} icg_tracker_result;

/* buckets_after[k], k = 0..n_buckets_after-1: bucket count of a fresh std::unordered_map<ulong, T> after k insertions on the host's standard
 * library (the reference container whose iteration order the feature rows keep); n_buckets_after must cover the row capacity + 2. */
int icg_tracker_create(icg_ctx *ctx, int n_streams, const icg_tracker_config *cfg, const uint32_t *buckets_after, int n_buckets_after,
                       icg_tracker **out);
void icg_tracker_destroy(icg_tracker *t);
size_t icg_tracker_block_bytes(void);
/* One frame per stream: images[k] (stride / channels as icg_frames_preprocess; NULL = stream k idles this step), its stamp and INS pose
 * prior (R row-major 9, t 3).  Returns after the whole chain has run; results[k] is filled for every stream. */
int icg_tracker_step(icg_tracker *t, const uint8_t *const *images, int stride, int channels, int images_on_device, const double *stamps,
                     const double *poses12, icg_tracker_result *results);
/* the block of one stream, to / from host memory (B2 view, dumps, optimizer write-back); both wait for the context's stream */
int icg_tracker_download(icg_tracker *t, int stream, void *block);
int icg_tracker_upload(icg_tracker *t, int stream, const void *block);
/* the landmark-container history of a stream has been replayed by the host: restart it (n_log = 0) */
int icg_tracker_reset_log(icg_tracker *t, int stream);
/* The same without the block: the first counts[k] entries (16 bytes each: id u64, map-point index u32, op i32 — 1 insert, 0 erase) of the
 * history of streams[k] are copied to out + k * entry_stride entries and the history restarts; one wait for all n_req streams.  counts[k]
 * is the n_log the stream's last result reported. */
int icg_tracker_fetch_logs(icg_tracker *t, int n_req, const int32_t *streams, const int32_t *counts, void *out, int entry_stride);

#ifdef __cplusplus
}
#endif
#endif /* ICGVINS_HIP_H */
