// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement (plain C++/C ABI) of the reference algorithms on the IC-GVINS hot path.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the product library
// (libicgvins_hip.so / libicgvins_host.so) never links, loads or calls anything from this directory.
// Every function cites the reference file:line it follows in its .cc file.
// PARITY STATUS: "parity unpinned" at the OpenCV boundary (front-end rows F1-F8: the reference has no tests and OpenCV is
// absent offline).  The in-tree back-end math (R1/R2 reprojection + robust corrector, M1-M4 marginalization, P1/P2
// preintegration, both variants) IS pinned against the reference's own sources: oracle/ref_build compiles them unmodified
// against interface shims into oracle/_ref/, golden outputs are committed under tests/golden/ (generators next to them).
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// ---- factors (orc_reproj.cc) -------------------------------------------------------------------------
void orc_reproj_eval_batch(int n, const double *obs_soa /*15 x n*/, const int32_t *idx_i, const int32_t *idx_j,
                           const int32_t *idx_lm, const double *poses /*K x 7*/, const double *ext /*7*/,
                           const double *invdepth /*L*/, double td, int want_jac, double *out_r /*n x 2*/,
                           double *out_J /*n x 46*/);
void orc_reproj_eval_one(const double *obs15, const double *pose_i, const double *pose_j, const double *ext,
                         double invdepth, double td, int want_jac, double *r2, double *J46);
void orc_huber_correct_2x46(int n, double huber_delta, double *r, double *J);

// ---- image ops (orc_image.cc) ------------------------------------------------------------------------
void orc_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray, int gstride);
double orc_histogram_mean(const uint8_t *img, int w, int h, int stride);
void orc_clahe(const uint8_t *src, int w, int h, int stride, double clip_limit, int tiles, uint8_t *dst, int dstride,
               uint8_t *lut_out);
void orc_pyrdown(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride);
int orc_pyramid_levels(int w, int h, int max_level, int win);
void orc_scharr(const uint8_t *src, int w, int h, int stride, int16_t *deriv);

// ---- optical flow (orc_lk.cc) ------------------------------------------------------------------------
void orc_lk_track(const uint8_t *prev, const uint8_t *next, int w, int h, int stride, int n, const float *prev_pts,
                  float *next_pts, uint8_t *status, float *err);
void orc_lk_track_fb(const uint8_t *prev, const uint8_t *next, int w, int h, int stride, int n, const float *prev_pts,
                     const float *guess_pts, float *out_pts, uint8_t *status);

// ---- camera model (orc_camera.cc) --------------------------------------------------------------------
void orc_undistort_points(const double *cam10, int n, float *pts);
void orc_distort_points(const double *cam10, int n, float *pts);
void orc_distort_camera_points(const double *cam10, int n, const double *pc, float *pts);
void orc_pixel2cam(const double *cam10, int n, const float *pts, double *pc);
void orc_world2cam(const double *pose12, int n, const double *pw, double *pc);
void orc_world2pixel(const double *cam10, const double *pose12, int n, const double *pw, float *pts);
void orc_predict_rotation(const double *cam10, const double *R_cur, const double *R_pre, int n, const float *pts_in,
                          float *pts_out);

void orc_reproj_error_batch(const double *cam10, int n, const int32_t *pose_idx, const int32_t *lm_idx, const double *poses12,
                            const double *pw, const float *pix, double max_error, double min_depth, double max_depth, double *err_out,
                            uint8_t *good_out);

// ---- detection (orc_detect.cc) -----------------------------------------------------------------------
void orc_draw_filled_circle(uint8_t *mask, int w, int h, int stride, int cx, int cy, int radius, uint8_t value);
void orc_min_eigen_map(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw, int rh, float *eig);
int orc_good_features(const uint8_t *img, int w, int h, int stride, const uint8_t *mask, int mstride, int rx, int ry,
                      int rw, int rh, int max_corners, double quality, double min_dist, float *corners);
void orc_corner_subpix(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw, int rh, int n,
                       float *corners);
void orc_circle_halfwidths(int radius, int *hw /* radius+1 */);
void orc_subpix_mask(float *mask121);
int orc_detect(const uint8_t *img, int w, int h, int stride, const int *grid6, int n_mask, const float *mask_pts,
               const int *quota, int max_out, float *out_pts, int *out_block);

// ---- RANSAC (orc_ransac.cc) --------------------------------------------------------------------------
int orc_find_fundamental_ransac(int n, const float *pts1, const float *pts2, double thresh, double conf,
                                uint8_t *mask, double *F_out, int *iters_out);
int orc_seven_point(const double *m1 /*7x2*/, const double *m2 /*7x2*/, double *F /*up to 3 x 9*/);
int orc_fm_score(const double *F, int n, const float *pts1, const float *pts2, double thresh, uint8_t *mask);
int orc_ransac_subsets(int n_points, const float *pts1, const float *pts2, int n_hyp, int32_t *idx_out);
int orc_have_collinear_points(const float *pts, int count);
int orc_solve_cubic(const double *coeffs4, double *roots3);

// ---- triangulation (orc_triang.cc) -------------------------------------------------------------------
void orc_triangulate_point(const double *T0 /*3x4 row-major*/, const double *T1, const double *pc0, const double *pc1,
                           double *pw);

// ---- marginalization (orc_marg.cc) -------------------------------------------------------------------
int orc_sym_eigen(int n, const double *A, double *evals, double *evecs);
void orc_reproj_accumulate_normal(int n, const double *r, const double *J, const int32_t *idx_i, const int32_t *idx_j,
                                  const int32_t *idx_lm, const int32_t *col_pose, int col_ext, const int32_t *col_lm, int col_td,
                                  int local_size, double *H0, double *b0);
int orc_marginalize(int n_total, int m, const double *H0, const double *b0, double eps, double *J0 /*r x r*/,
                    double *e0 /*r*/, double *Hp /*r x r*/, double *bp /*r*/);
void orc_marg_factor_eval(int r, int n_blocks, const int *block_size, const int *block_index /*local idx - m*/,
                          const double *x0_concat, const double *x_concat, const double *J0, const double *e0,
                          double *residuals, double *jac_concat /* r x sum(global sizes), block-major, may be NULL */);

// ---- preintegration (orc_preint.cc) ------------------------------------------------------------------
void orc_preint_integrate(int variant /*0 normal, 1 earth*/, int n_imu, const double *imu /*n x 8: time,dt,dtheta3,dvel3*/,
                          const double *state0 /*p3 q4(xyzw) v3 bg3 ba3 = 16*/, const double *params /*gyr_arw,acc_vrw,gbstd,abstd,corr_time,gravity,iewn3*/,
                          double *cur_state /*16*/, double *delta_state /*16: p q v bg ba*/, double *jac /*15x15*/,
                          double *cov /*15x15*/, double *delta_time, double *pn /* (n-1) x 4: dt,p */);
void orc_preint_evaluate(int variant, const double *delta_state, const double *jac, const double *cov, double delta_time,
                         const double *gravity3, const double *iewn3, int n_pn, const double *pn, const double *q0_xyzw,
                         const double *pose0, const double *mix0, const double *pose1, const double *mix1,
                         double *residuals /*15*/, double *jacobians /*15x7,15x9,15x7,15x9 concatenated, may be NULL*/);

// ---- landmark elimination of one GN/LM step (orc_solve.cc; SURVEY.md §8 f1, parity unpinned: Ceres absent) --------
void orc_schur_reduce(int P, int L, const double *H /*(P+L)^2*/, const double *b, double damp, double min_diag, double max_diag,
                      double *S, double *s, double *diag_cc, double *inv /*L*/);
void orc_schur_backsub(int P, int L, const double *H, const double *b, const double *inv, double damp, double min_diag, double max_diag,
                       const double *delta_c, double *delta_l, double *lm_terms);
double orc_reproj_cost(int n, const double *r, const uint8_t *active, double huber);

// ---- INS helpers in front of the tracker (orc_ins.cc) ---------------------------------
// imu rows of 8 (time, dt, dtheta3, dvel3); state rows of 23 (time, p3, q4 xyzw, v3, bg3, ba3, sg3, sa3);
// cfg8 = gravity3, iewn3, iswithearth, iswithscale; poses 12 = R row-major 9, t 3
void orc_ins_mechanize(const double *cfg8, int n_imu, const double *imu, double *state23 /*in/out*/,
                       double *traj /* (n_imu-1) x 23, may be NULL */);
int64_t orc_ins_window_index(int n_win, const double *imu, double time);
int orc_ins_camera_pose(int n_win, const double *imu, const double *states, const double *pose_b_c12, double time,
                        double *pose12);
int orc_imu_series(int n_win, const double *imu, double start, double end, int cap, double *series /* cap x 8 */);
int orc_redo_ins(const double *cfg8, const double *updated_state23, int reserved, int n_win, double *imu, double *states);
// rows of MISC::writeNavResult (misc.cc:417-499): nav[11], errrow[<=14], traj[8]; returns the number of err values
int orc_nav_result_rows(const double *origin3, int iswithscale, const double *state23, double sodo, double *nav, double *errrow, double *traj);

#ifdef __cplusplus
}
#endif
