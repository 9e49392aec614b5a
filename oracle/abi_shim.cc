// ORACLE — TEST INFRASTRUCTURE ONLY.
// The C ABI of include/icgvins_hip.h implemented on the CPU restatement (orc_*), so that the SAME host layer
// (ic-gvins_amd/host/*.cc: Tracking, TrackingBatch, ...) can be linked a second time against the oracle:
//   oracle/libicgvins_host_oracle.so = host sources + this shim + orc_*.o
// Used by tests/ for end-to-end parity (HIP-backed vs oracle-backed streams must produce identical track ids,
// states and digests) and by bench.py's cpu_baseline leg (kind "port").  It is never linked into, loaded by or
// shipped with the product libraries (libicgvins_hip.so / libicgvins_host.so).
#include <cmath>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../include/icgvins_hip.h"
#include "oracle.h"

struct icg_ctx {
    icg_ctx_config cfg;
    std::string err;
    std::vector<std::vector<uint8_t>> slots; // CLAHE image (level 0), w*h
    icg_camera cam{};
    bool has_cam = false;
    int threads  = 1;
};

static std::string g_err;

template <typename F> static void parallel_for(int n, int threads, F &&f) {
    if (threads <= 1 || n < 2) {
        for (int i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n) break;
            f(i);
        }
    };
    std::vector<std::thread> th;
    int nt = std::min(threads, n);
    for (int t = 1; t < nt; t++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
}

extern "C" {

const char *icg_version(void) { return "icgvins ORACLE shim (CPU restatement, test infrastructure)"; }
const char *icg_last_error(const icg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int icg_ctx_create(const icg_ctx_config *cfg, icg_ctx **out) {
    if (!cfg || !out) return ICG_ERR_INVALID;
    icg_ctx *c = new icg_ctx();
    c->cfg     = *cfg;
    c->slots.resize((size_t) cfg->n_slots);
    const char *t = getenv("ICG_ORACLE_THREADS");
    c->threads    = t ? std::max(1, atoi(t)) : 1;
    *out          = c;
    return ICG_OK;
}
void icg_ctx_destroy(icg_ctx *ctx) { delete ctx; }
int icg_ctx_sync(icg_ctx *) { return ICG_OK; }
int icg_ctx_set_wait_mode(icg_ctx *, int, int) { return ICG_OK; }
void *icg_ctx_stream(icg_ctx *) { return nullptr; }
int icg_set_camera(icg_ctx *ctx, const icg_camera *cam) {
    ctx->cam     = *cam;
    ctx->has_cam = true;
    return ICG_OK;
}
int icg_pyramid_levels(const icg_ctx *ctx) { return orc_pyramid_levels(ctx->cfg.width, ctx->cfg.height, 3, 21); }
int icg_prof_enable(icg_ctx *, int) { return ICG_OK; }
int icg_prof_get(icg_ctx *, const char *, int *l, double *ms) {
    if (l) *l = 0;
    if (ms) *ms = 0;
    return ICG_OK;
}
int icg_prof_names(icg_ctx *, char *buf, int n) {
    if (buf && n > 0) buf[0] = 0;
    return ICG_OK;
}

int icg_frames_preprocess(icg_ctx *ctx, int n, const int32_t *slots, const uint8_t *const *images, int stride, int channels,
                          int src_on_device, double *hist_mean) {
    if (src_on_device) {
        ctx->err = "oracle shim: device images are not supported";
        return ICG_ERR_INVALID;
    }
    const int w = ctx->cfg.width, h = ctx->cfg.height;
    parallel_for(n, ctx->threads, [&](int k) {
        std::vector<uint8_t> gray((size_t) w * h);
        if (channels == 3)
            orc_bgr2gray(images[k], w, h, stride, gray.data(), w);
        else
            for (int y = 0; y < h; y++) memcpy(&gray[(size_t) y * w], images[k] + (size_t) y * stride, (size_t) w);
        if (hist_mean) hist_mean[k] = orc_histogram_mean(gray.data(), w, h, w);
        auto &dst = ctx->slots[(size_t) slots[k]];
        dst.resize((size_t) w * h);
        orc_clahe(gray.data(), w, h, w, 3.0, 21, dst.data(), w, nullptr);
    });
    return ICG_OK;
}

int icg_frame_download(icg_ctx *ctx, int slot, int level, uint8_t *dst, int dst_stride) {
    int w = ctx->cfg.width, h = ctx->cfg.height;
    std::vector<uint8_t> cur = ctx->slots[(size_t) slot], nxt;
    for (int l = 0; l < level; l++) {
        nxt.resize((size_t) ((w + 1) / 2) * ((h + 1) / 2));
        orc_pyrdown(cur.data(), w, h, w, nxt.data(), (w + 1) / 2);
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        cur.swap(nxt);
    }
    for (int y = 0; y < h; y++) memcpy(dst + (size_t) y * dst_stride, &cur[(size_t) y * w], (size_t) w);
    return ICG_OK;
}

int icg_lk_track(icg_ctx *ctx, int n, const int32_t *prev_slot, const int32_t *next_slot, const float *prev_pts,
                 float *next_pts, uint8_t *status, float *err) {
    const int w = ctx->cfg.width, h = ctx->cfg.height;
    std::map<std::pair<int, int>, std::vector<int>> groups;
    for (int i = 0; i < n; i++) groups[{prev_slot[i], next_slot[i]}].push_back(i);
    std::vector<std::pair<std::pair<int, int>, std::vector<int>>> G(groups.begin(), groups.end());
    parallel_for((int) G.size(), ctx->threads, [&](int g) {
        auto &idx = G[(size_t) g].second;
        int m     = (int) idx.size();
        std::vector<float> pp(2 * (size_t) m), np(2 * (size_t) m), e((size_t) m);
        std::vector<uint8_t> st((size_t) m);
        for (int k = 0; k < m; k++) {
            pp[2 * k] = prev_pts[2 * idx[k]], pp[2 * k + 1] = prev_pts[2 * idx[k] + 1];
            np[2 * k] = next_pts[2 * idx[k]], np[2 * k + 1] = next_pts[2 * idx[k] + 1];
        }
        orc_lk_track(ctx->slots[(size_t) G[g].first.first].data(), ctx->slots[(size_t) G[g].first.second].data(), w, h, w, m,
                     pp.data(), np.data(), st.data(), e.data());
        for (int k = 0; k < m; k++) {
            next_pts[2 * idx[k]] = np[2 * k], next_pts[2 * idx[k] + 1] = np[2 * k + 1];
            status[idx[k]] = st[k];
            if (err) err[idx[k]] = e[k];
        }
    });
    return ICG_OK;
}

int icg_lk_track_fb(icg_ctx *ctx, int n, const int32_t *prev_slot, const int32_t *next_slot, const float *prev_pts,
                    const float *guess_pts, float *out_pts, uint8_t *status, float *out_undist, int32_t *keep_idx,
                    int32_t *n_keep) {
    const int w = ctx->cfg.width, h = ctx->cfg.height;
    std::map<std::pair<int, int>, std::vector<int>> groups;
    for (int i = 0; i < n; i++) groups[{prev_slot[i], next_slot[i]}].push_back(i);
    std::vector<std::pair<std::pair<int, int>, std::vector<int>>> G(groups.begin(), groups.end());
    parallel_for((int) G.size(), ctx->threads, [&](int g) {
        auto &idx = G[(size_t) g].second;
        int m     = (int) idx.size();
        std::vector<float> pp(2 * (size_t) m), gs(2 * (size_t) m), op(2 * (size_t) m);
        std::vector<uint8_t> st((size_t) m);
        for (int k = 0; k < m; k++) {
            pp[2 * k] = prev_pts[2 * idx[k]], pp[2 * k + 1] = prev_pts[2 * idx[k] + 1];
            gs[2 * k] = guess_pts[2 * idx[k]], gs[2 * k + 1] = guess_pts[2 * idx[k] + 1];
        }
        orc_lk_track_fb(ctx->slots[(size_t) G[g].first.first].data(), ctx->slots[(size_t) G[g].first.second].data(), w, h, w, m,
                        pp.data(), gs.data(), op.data(), st.data());
        for (int k = 0; k < m; k++) {
            out_pts[2 * idx[k]] = op[2 * k], out_pts[2 * idx[k] + 1] = op[2 * k + 1];
            status[idx[k]] = st[k];
        }
    });
    if (out_undist) {
        memcpy(out_undist, out_pts, sizeof(float) * 2 * (size_t) n);
        orc_undistort_points(&ctx->cam.fx, n, out_undist);
    }
    if (keep_idx) {
        int c = 0;
        for (int i = 0; i < n; i++)
            if (status[i]) keep_idx[c++] = i;
        *n_keep = c;
    }
    return ICG_OK;
}

int icg_undistort_points(icg_ctx *ctx, int n, float *pts) {
    orc_undistort_points(&ctx->cam.fx, n, pts);
    return ICG_OK;
}
int icg_distort_points(icg_ctx *ctx, int n, float *pts) {
    orc_distort_points(&ctx->cam.fx, n, pts);
    return ICG_OK;
}

int icg_reproj_error_batch(icg_ctx *ctx, int n, const int32_t *pose_idx, const int32_t *lm_idx, int, const double *poses12, int, const double *pw,
                           const float *pix, double max_error, double min_depth, double max_depth, double *err_out, uint8_t *good_out) {
    orc_reproj_error_batch(&ctx->cam.fx, n, pose_idx, lm_idx, poses12, pw, pix, max_error, min_depth, max_depth, err_out, good_out);
    return ICG_OK;
}

int icg_fm_ransac(icg_ctx *ctx, int n_sets, const int32_t *offsets, const float *pts1, const float *pts2, double thresh,
                  double conf, uint8_t *mask) {
    parallel_for(n_sets, ctx->threads, [&](int s) {
        int b = offsets[s], n = offsets[s + 1] - offsets[s];
        if (n < 15) {
            for (int i = 0; i < n; i++) mask[b + i] = 1;
            return;
        }
        orc_find_fundamental_ransac(n, pts1 + 2 * (size_t) b, pts2 + 2 * (size_t) b, thresh, conf, mask + b, nullptr, nullptr);
    });
    return ICG_OK;
}

int icg_fm_ransac_device(icg_ctx *ctx, int n_sets, const int32_t *offsets, const float *pts1, const float *pts2, double thresh, double conf,
                         uint8_t *mask) {
    return icg_fm_ransac(ctx, n_sets, offsets, pts1, pts2, thresh, conf, mask);
}

int icg_detect(icg_ctx *ctx, int n, const int32_t *slots, const icg_detect_grid *grid, const int32_t *mask_off,
               const float *mask_pts, const int32_t *quota, int max_per_job, float *out_pts, int32_t *out_count,
               int32_t *out_block) {
    const int w = ctx->cfg.width, h = ctx->cfg.height;
    const int g6[6] = {grid->block_cols, grid->block_rows, grid->block_w, grid->block_h, grid->min_dist, grid->max_per_block};
    const int nblk  = grid->block_cols * grid->block_rows;
    parallel_for(n, ctx->threads, [&](int k) {
        std::vector<int> q((size_t) nblk);
        for (int b = 0; b < nblk; b++) q[(size_t) b] = std::min(quota[(size_t) k * nblk + b], grid->max_per_block);
        out_count[k] = orc_detect(ctx->slots[(size_t) slots[k]].data(), w, h, w, g6, mask_off[k + 1] - mask_off[k],
                                  mask_pts + 2 * (size_t) mask_off[k], q.data(), max_per_job,
                                  out_pts + (size_t) k * max_per_job * 2, out_block ? out_block + (size_t) k * max_per_job : nullptr);
    });
    return ICG_OK;
}

int icg_triangulate(icg_ctx *, int n, const int32_t *T0_idx, const int32_t *T1_idx, int, const double *Tcw12, const double *pc0,
                    const double *pc1, double *pw) {
    for (int i = 0; i < n; i++)
        orc_triangulate_point(Tcw12 + 12 * (size_t) T0_idx[i], Tcw12 + 12 * (size_t) T1_idx[i], pc0 + 3 * (size_t) i,
                              pc1 + 3 * (size_t) i, pw + 3 * (size_t) i);
    return ICG_OK;
}

int icg_reproj_eval_batch(icg_ctx *, int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j,
                          const int32_t *idx_lm, int, const double *poses, const double *ext, int, const double *invdepth,
                          double td, int want_jac, double huber_delta, double *out_r, double *out_J) {
    orc_reproj_eval_batch(n, obs_soa, idx_i, idx_j, idx_lm, poses, ext, invdepth, td, want_jac, out_r, out_J);
    if (huber_delta > 0) orc_huber_correct_2x46(n, huber_delta, out_r, want_jac ? out_J : nullptr);
    return ICG_OK;
}

} // extern "C"

// ---- back-end entry points on the oracle (resident factor state kept per context) -------------------------------------
#include <mutex>
#include <unordered_map>
namespace {
struct shim_backend {
    int n = 0, n_poses = 0, n_lm = 0;
    std::vector<double> obs, r, J;
    std::vector<int32_t> ii, jj, ll;
    std::vector<double> stage_obs; // icg_reproj_stage_factors / _commit_factors
    std::vector<int32_t> stage_idx;
    int stage_n = -1;
    double huber = 0.0;
    // f1: resident normal equations
    int P = 0;
    std::vector<double> H, b, inv;
    double damp = 0, min_diag = 0, max_diag = 0;
    // many windows per launch
    int W = 0;
    std::vector<int32_t> fac_off, lm_off;
    std::vector<std::vector<double>> wH, wb, winv;
    std::vector<double> S_view; // icg_reproj_schur_windows_view: the reduced systems stay here until the next call
    std::vector<double> redS, hostS; // icg_reproj_schur_windows_resident / _set_host_part_windows: resident systems, packed host parts
    int hostS_P = 0;
    std::vector<double> wdamp;
    int wP = 0;
};
std::unordered_map<icg_ctx *, shim_backend> g_backend_map;
std::mutex g_backend_mutex;
// element references stay valid across insertions (node-based container); only the lookup/insert itself needs the lock
struct backend_accessor {
    shim_backend &operator[](icg_ctx *ctx) {
        std::lock_guard<std::mutex> lock(g_backend_mutex);
        return g_backend_map[ctx];
    }
} g_backend;
} // namespace

extern "C" {

int icg_reproj_set_factors(icg_ctx *ctx, int n, const double *obs_soa, const int32_t *idx_i, const int32_t *idx_j,
                           const int32_t *idx_lm) {
    shim_backend &B = g_backend[ctx];
    B.n = n;
    B.obs.assign(obs_soa, obs_soa + 15 * (size_t) n);
    B.ii.assign(idx_i, idx_i + n);
    B.jj.assign(idx_j, idx_j + n);
    B.ll.assign(idx_lm, idx_lm + n);
    return ICG_OK;
}

int icg_reproj_stage_factors(icg_ctx *ctx, int n, double **obs_soa, int32_t **idx3) {
    if (!ctx || n < 0 || !obs_soa || !idx3) return ICG_ERR_INVALID;
    shim_backend &B = g_backend[ctx];
    B.stage_obs.assign(15 * (size_t) n + 1, 0.0);
    B.stage_idx.assign(3 * (size_t) n + 1, 0);
    B.stage_n = n;
    *obs_soa = B.stage_obs.data(), *idx3 = B.stage_idx.data();
    return ICG_OK;
}
int icg_reproj_commit_factors(icg_ctx *ctx) {
    shim_backend &B = g_backend[ctx];
    if (B.stage_n < 0) return ICG_ERR_INVALID;
    const int n = B.stage_n;
    B.stage_n   = -1;
    return icg_reproj_set_factors(ctx, n, B.stage_obs.data(), B.stage_idx.data(), B.stage_idx.data() + n, B.stage_idx.data() + 2 * (size_t) n);
}

int icg_reproj_eval_resident(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth,
                             double td, int want_jac, double huber_delta, double *out_r, double *out_J) {
    shim_backend &B = g_backend[ctx];
    B.n_poses = n_poses;
    B.n_lm    = n_lm;
    B.r.assign(2 * (size_t) B.n, 0.0);
    B.J.assign(46 * (size_t) B.n, 0.0);
    orc_reproj_eval_batch(B.n, B.obs.data(), B.ii.data(), B.jj.data(), B.ll.data(), poses, ext, invdepth, td, want_jac, B.r.data(),
                          B.J.data());
    if (huber_delta > 0) orc_huber_correct_2x46(B.n, huber_delta, B.r.data(), want_jac ? B.J.data() : nullptr);
    B.huber = huber_delta;
    if (out_r) memcpy(out_r, B.r.data(), sizeof(double) * B.r.size());
    if (out_J && want_jac) memcpy(out_J, B.J.data(), sizeof(double) * B.J.size());
    return ICG_OK;
}

int icg_reproj_eval_resident_view(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth, double td,
                                  int want_jac, double huber_delta, const double **r_view, const double **J_view) {
    int rc = icg_reproj_eval_resident(ctx, n_poses, poses, ext, n_lm, invdepth, td, want_jac, huber_delta, nullptr, nullptr);
    shim_backend &B = g_backend[ctx];
    *r_view = B.r.data();
    *J_view = want_jac ? B.J.data() : nullptr;
    return rc;
}

int icg_reproj_accumulate_normal(icg_ctx *ctx, int local_size, const int32_t *col_pose, int32_t col_ext, const int32_t *col_lm,
                                 int32_t col_td, double *H0, double *b0) {
    shim_backend &B = g_backend[ctx];
    orc_reproj_accumulate_normal(B.n, B.r.data(), B.J.data(), B.ii.data(), B.jj.data(), B.ll.data(), col_pose, col_ext, col_lm, col_td,
                                 local_size, H0, b0);
    return ICG_OK;
}

int icg_reproj_schur(icg_ctx *ctx, int P, const int32_t *col_pose, int32_t col_ext, int32_t col_td, const uint8_t *active, int reassemble,
                     double damp, double min_diag, double max_diag, double *S, double *s, double *diag_cc, double *cost) {
    shim_backend &B = g_backend[ctx];
    const int L = B.n_lm;
    const size_t N = (size_t) P + L;
    if (reassemble) {
        std::vector<int32_t> ii, jj, ll, col_lm((size_t) L);
        std::vector<double> r, J;
        for (int l = 0; l < L; l++) col_lm[(size_t) l] = P + l;
        for (int f = 0; f < B.n; f++) {
            if (active && !active[f]) continue;
            ii.push_back(B.ii[(size_t) f]), jj.push_back(B.jj[(size_t) f]), ll.push_back(B.ll[(size_t) f]);
            r.insert(r.end(), B.r.begin() + 2 * (size_t) f, B.r.begin() + 2 * (size_t) f + 2);
            J.insert(J.end(), B.J.begin() + 46 * (size_t) f, B.J.begin() + 46 * (size_t) f + 46);
        }
        B.P = P;
        B.H.assign(N * N, 0.0);
        B.b.assign(N, 0.0);
        B.inv.assign((size_t) L, 0.0);
        orc_reproj_accumulate_normal((int) ii.size(), r.data(), J.data(), ii.data(), jj.data(), ll.data(), col_pose, col_ext, col_lm.data(),
                                     col_td, (int) N, B.H.data(), B.b.data());
        if (cost) *cost = orc_reproj_cost(B.n, B.r.data(), active, B.huber);
    } else if (B.P != P || B.H.size() != N * N) {
        return ICG_ERR_INVALID;
    }
    orc_schur_reduce(P, L, B.H.data(), B.b.data(), damp, min_diag, max_diag, S, s, diag_cc, B.inv.data());
    B.damp = damp, B.min_diag = min_diag, B.max_diag = max_diag;
    return ICG_OK;
}

int icg_reproj_landmark_diag(icg_ctx *ctx, double *h_ll) {
    shim_backend &B = g_backend[ctx];
    if (B.H.empty()) return ICG_ERR_INVALID;
    const size_t N = (size_t) B.P + B.n_lm;
    for (int l = 0; l < B.n_lm; l++) h_ll[l] = B.H[((size_t) B.P + l) * N + B.P + l];
    return ICG_OK;
}

int icg_reproj_backsub(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    shim_backend &B = g_backend[ctx];
    if (B.P != P || B.H.empty()) return ICG_ERR_INVALID;
    orc_schur_backsub(P, B.n_lm, B.H.data(), B.b.data(), B.inv.data(), B.damp, B.min_diag, B.max_diag, delta_c, delta_l, lm_terms);
    return ICG_OK;
}

int icg_reproj_cost(icg_ctx *ctx, const uint8_t *active, double *cost) {
    shim_backend &B = g_backend[ctx];
    *cost = orc_reproj_cost(B.n, B.r.data(), active, B.huber);
    return ICG_OK;
}

int icg_reproj_set_windows(icg_ctx *ctx, int n_windows, const int32_t *fac_off, const int32_t *lm_off) {
    shim_backend &B = g_backend[ctx];
    if (fac_off[0] != 0 || fac_off[n_windows] != B.n) return ICG_ERR_INVALID;
    B.W = n_windows;
    B.fac_off.assign(fac_off, fac_off + n_windows + 1);
    B.lm_off.assign(lm_off, lm_off + n_windows + 1);
    B.wH.assign((size_t) n_windows, {}), B.wb.assign((size_t) n_windows, {}), B.winv.assign((size_t) n_windows, {});
    B.wdamp.assign((size_t) n_windows, 0.0);
    B.wP = 0;
    return ICG_OK;
}

int icg_reproj_eval_windows(icg_ctx *ctx, int n_poses, const double *poses, const double *ext, int n_lm, const double *invdepth, const double *td,
                            int want_jac, double huber_delta) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0) return ICG_ERR_INVALID;
    B.n_poses = n_poses, B.n_lm = n_lm, B.huber = huber_delta;
    B.r.assign(2 * (size_t) B.n, 0.0);
    B.J.assign(46 * (size_t) B.n, 0.0);
    for (int w = 0; w < B.W; w++) {
        const int f0 = B.fac_off[(size_t) w], nf = B.fac_off[(size_t) w + 1] - f0;
        if (nf == 0) continue;
        std::vector<double> obs(15 * (size_t) nf); // the component-major slab of this window
        for (int c = 0; c < 15; c++) memcpy(&obs[(size_t) c * nf], &B.obs[(size_t) c * B.n + f0], sizeof(double) * (size_t) nf);
        orc_reproj_eval_batch(nf, obs.data(), &B.ii[(size_t) f0], &B.jj[(size_t) f0], &B.ll[(size_t) f0], poses, ext + 7 * (size_t) w, invdepth, td[w],
                              want_jac, &B.r[2 * (size_t) f0], &B.J[46 * (size_t) f0]);
        if (huber_delta > 0) orc_huber_correct_2x46(nf, huber_delta, &B.r[2 * (size_t) f0], want_jac ? &B.J[46 * (size_t) f0] : nullptr);
    }
    return ICG_OK;
}

int icg_reproj_schur_windows(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                             const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, double *S, double *s, double *diag_cc,
                             double *cost) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0) return ICG_ERR_INVALID;
    if (B.wP != P)
        for (int w = 0; w < B.W; w++)
            if (!reassemble[w]) return ICG_ERR_INVALID;
    B.wP = P, B.min_diag = min_diag, B.max_diag = max_diag;
    for (int w = 0; w < B.W; w++) {
        const int f0 = B.fac_off[(size_t) w], f1 = B.fac_off[(size_t) w + 1], l0 = B.lm_off[(size_t) w], L = B.lm_off[(size_t) w + 1] - l0;
        const size_t N = (size_t) P + L;
        if (cost) cost[w] = 0.0;
        if (reassemble[w]) {
            std::vector<int32_t> ii, jj, ll, col_lm((size_t) B.n_lm, -1);
            std::vector<double> r, J;
            for (int l = 0; l < L; l++) col_lm[(size_t) (l0 + l)] = P + l;
            for (int f = f0; f < f1; f++) {
                if (active && !active[f]) continue;
                ii.push_back(B.ii[(size_t) f]), jj.push_back(B.jj[(size_t) f]), ll.push_back(B.ll[(size_t) f]);
                r.insert(r.end(), B.r.begin() + 2 * (size_t) f, B.r.begin() + 2 * (size_t) f + 2);
                J.insert(J.end(), B.J.begin() + 46 * (size_t) f, B.J.begin() + 46 * (size_t) f + 46);
            }
            B.wH[(size_t) w].assign(N * N, 0.0);
            B.wb[(size_t) w].assign(N, 0.0);
            B.winv[(size_t) w].assign((size_t) L, 0.0);
            orc_reproj_accumulate_normal((int) ii.size(), r.data(), J.data(), ii.data(), jj.data(), ll.data(), col_pose, col_ext[w], col_lm.data(),
                                         col_td[w], (int) N, B.wH[(size_t) w].data(), B.wb[(size_t) w].data());
            if (cost) cost[w] = orc_reproj_cost(f1 - f0, &B.r[2 * (size_t) f0], active ? active + f0 : nullptr, B.huber);
        }
        B.wdamp[(size_t) w] = damp[w];
        orc_schur_reduce(P, L, B.wH[(size_t) w].data(), B.wb[(size_t) w].data(), damp[w], min_diag, max_diag, S + (size_t) w * P * P, s + (size_t) w * P,
                         diag_cc ? diag_cc + (size_t) w * P : nullptr, B.winv[(size_t) w].data());
    }
    return ICG_OK;
}

int icg_reproj_schur_windows_view(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                                  const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, const double **S_view, double *s,
                                  double *diag_cc, double *cost) {
    shim_backend &B = g_backend[ctx];
    B.S_view.assign((size_t) std::max(0, B.W) * P * P, 0.0);
    *S_view = B.S_view.data();
    return icg_reproj_schur_windows(ctx, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, B.S_view.data(), s, diag_cc, cost);
}

int icg_reproj_reserve_windows(icg_ctx *, int P) { return P > 0 ? ICG_OK : ICG_ERR_INVALID; } // nothing to pre-size on the CPU

// CPU statement of the device-side reduced solve (same interface: the checker build of the host layer drives the same code path)
int icg_reproj_schur_windows_resident(icg_ctx *ctx, int P, const int32_t *col_pose, const int32_t *col_ext, const int32_t *col_td, const uint8_t *active,
                                      const uint8_t *reassemble, const double *damp, double min_diag, double max_diag, double *s, double *diag_cc,
                                      double *cost) {
    shim_backend &B = g_backend[ctx];
    B.redS.assign((size_t) std::max(0, B.W) * P * P, 0.0);
    return icg_reproj_schur_windows(ctx, P, col_pose, col_ext, col_td, active, reassemble, damp, min_diag, max_diag, B.redS.data(), s, diag_cc, cost);
}

int icg_reproj_set_host_part_windows(icg_ctx *ctx, int P, int n_upd, const int32_t *win_idx, const double *packed) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0 || P <= 0) return ICG_ERR_INVALID;
    const size_t tri = (size_t) P * (P + 1) / 2;
    if (B.hostS_P != P || B.hostS.size() != (size_t) B.W * tri) B.hostS.assign((size_t) B.W * tri, 0.0), B.hostS_P = P;
    for (int k = 0; k < n_upd; k++) {
        if (win_idx[k] < 0 || win_idx[k] >= B.W) return ICG_ERR_INVALID;
        memcpy(&B.hostS[(size_t) win_idx[k] * tri], packed + (size_t) k * tri, sizeof(double) * tri);
    }
    return ICG_OK;
}

int icg_reproj_solve_backsub_windows(icg_ctx *ctx, int P, const int32_t *Pw, const uint8_t *stepped, const double *rhs, const double *dd,
                                     double *delta_c, uint8_t *ok, double *delta_l, double *lm_terms) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0 || B.wP != P || B.redS.size() != (size_t) B.W * P * P) return ICG_ERR_INVALID;
    int rc = icg_reproj_set_host_part_windows(ctx, P, 0, nullptr, nullptr);
    if (rc) return rc;
    const size_t tri = (size_t) P * (P + 1) / 2;
    for (int w = 0; w < B.W; w++) {
        double *dc = delta_c + (size_t) w * P;
        std::fill(dc, dc + P, 0.0);
        ok[w]       = 0;
        const int n = Pw[w];
        if (!stepped[w] || n <= 0) continue;
        std::vector<double> A((size_t) n * n, 0.0), b(rhs + (size_t) w * P, rhs + (size_t) w * P + n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++)
                A[(size_t) i * n + j] = B.redS[(size_t) w * P * P + (size_t) i * P + j] + B.hostS[(size_t) w * tri + (size_t) i * (i + 1) / 2 + j] +
                                        (i == j ? dd[(size_t) w * P + i] : 0.0);
        bool good = true; // right-looking Cholesky, the order of the device kernel
        for (int k = 0; k < n && good; k++) {
            const double d = A[(size_t) k * n + k];
            if (!(d > 0.0) || !std::isfinite(d)) {
                good = false;
                break;
            }
            const double piv = std::sqrt(d);
            A[(size_t) k * n + k] = piv;
            for (int i = k + 1; i < n; i++) A[(size_t) i * n + k] /= piv;
            for (int i = k + 1; i < n; i++)
                for (int j = k + 1; j <= i; j++) A[(size_t) i * n + j] -= A[(size_t) i * n + k] * A[(size_t) j * n + k];
        }
        if (!good) continue;
        for (int r = 0; r < n; r++) {
            double acc = 0.0;
            for (int k = 0; k < r; k++) acc += A[(size_t) r * n + k] * b[(size_t) k];
            b[(size_t) r] = (b[(size_t) r] - acc) / A[(size_t) r * n + r];
        }
        for (int r = n - 1; r >= 0; r--) {
            const double x = b[(size_t) r] / A[(size_t) r * n + r];
            b[(size_t) r]  = x;
            for (int k = 0; k < r; k++) b[(size_t) k] -= A[(size_t) r * n + k] * x;
        }
        std::copy(b.begin(), b.end(), dc);
        ok[w] = 1;
    }
    if (delta_l) return icg_reproj_backsub_windows(ctx, P, delta_c, delta_l, lm_terms);
    return ICG_OK;
}

int icg_reproj_backsub_windows(icg_ctx *ctx, int P, const double *delta_c, double *delta_l, double *lm_terms) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0 || B.wP != P) return ICG_ERR_INVALID;
    for (int w = 0; w < B.W; w++) {
        const int l0 = B.lm_off[(size_t) w], L = B.lm_off[(size_t) w + 1] - l0;
        orc_schur_backsub(P, L, B.wH[(size_t) w].data(), B.wb[(size_t) w].data(), B.winv[(size_t) w].data(), B.wdamp[(size_t) w], B.min_diag, B.max_diag,
                          delta_c + (size_t) w * P, delta_l + l0, lm_terms ? lm_terms + 2 * (size_t) w : nullptr);
    }
    return ICG_OK;
}

int icg_reproj_landmark_diag_windows(icg_ctx *ctx, double *h_ll) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0 || B.wP <= 0 || !h_ll) return ICG_ERR_INVALID;
    for (int w = 0; w < B.W; w++) {
        const int l0 = B.lm_off[(size_t) w], L = B.lm_off[(size_t) w + 1] - l0;
        const size_t N = (size_t) B.wP + L;
        for (int l = 0; l < L; l++) h_ll[l0 + l] = B.wH[(size_t) w][((size_t) B.wP + l) * N + B.wP + l];
    }
    return ICG_OK;
}

int icg_reproj_chi2_cull(icg_ctx *ctx, double chi2, uint8_t *active) {
    shim_backend &B = g_backend[ctx];
    for (int f = 0; f < B.n; f++) {
        const double r0 = B.r[2 * (size_t) f], r1 = B.r[2 * (size_t) f + 1];
        const double cost = 0.5 * (r0 * r0 + r1 * r1);
        if (cost * 2.0 > chi2) active[f] = 0;
    }
    return ICG_OK;
}

int icg_reproj_fetch_residuals(icg_ctx *ctx, double *out_r) {
    shim_backend &B = g_backend[ctx];
    memcpy(out_r, B.r.data(), sizeof(double) * B.r.size());
    return ICG_OK;
}

int icg_reproj_cost_windows(icg_ctx *ctx, const uint8_t *active, double *cost) {
    shim_backend &B = g_backend[ctx];
    if (B.W <= 0) return ICG_ERR_INVALID;
    for (int w = 0; w < B.W; w++) {
        const int f0 = B.fac_off[(size_t) w], f1 = B.fac_off[(size_t) w + 1];
        cost[w] = orc_reproj_cost(f1 - f0, &B.r[2 * (size_t) f0], active ? active + f0 : nullptr, B.huber);
    }
    return ICG_OK;
}

int icg_preint_batch(icg_ctx *, int variant, int n_intervals, const int32_t *offsets, const double *imu, const double *state0,
                     const double *params, double *cur_state, double *delta_state, double *jac, double *cov, double *delta_time,
                     double *pn) {
    for (int s = 0; s < n_intervals; s++) {
        int b = offsets[s], n = offsets[s + 1] - offsets[s];
        std::vector<double> pnl((size_t) 4 * std::max(n - 1, 1));
        orc_preint_integrate(variant, n, imu + 8 * (size_t) b, state0 + 16 * (size_t) s, params, cur_state + 16 * (size_t) s,
                             delta_state + 16 * (size_t) s, jac + 225 * (size_t) s, cov + 225 * (size_t) s, delta_time + s, pnl.data());
        if (pn && n > 1) memcpy(pn + 4 * (size_t) b, pnl.data(), sizeof(double) * 4 * (size_t) (n - 1));
    }
    return ICG_OK;
}

int icg_ins_mechanize_batch(icg_ctx *, int n_streams, const int32_t *offsets, const double *imu, const double *cfg8, double *states23,
                            double *traj23) {
    for (int s = 0; s < n_streams; s++) {
        int b = offsets[s], n = offsets[s + 1] - offsets[s];
        if (traj23 && n > 0) memcpy(traj23 + 23 * (size_t) b, states23 + 23 * (size_t) s, sizeof(double) * 23);
        orc_ins_mechanize(cfg8, n, imu + 8 * (size_t) b, states23 + 23 * (size_t) s, (traj23 && n > 1) ? traj23 + 23 * (size_t) (b + 1) : nullptr);
    }
    return ICG_OK;
}

int icg_ins_camera_pose_batch(icg_ctx *, int n, const double *brackets16, const int32_t *interp, const double *pose_b_c12,
                              const double *times, double *pose12_out) {
    for (int i = 0; i < n; i++) { // a two-state window reproduces the bracketed interpolation of orc_ins_camera_pose
        const double *b = brackets16 + 16 * (size_t) i;
        double imu[16] = {0}, st[46] = {0};
        imu[0] = b[0], imu[8] = b[8];
        st[0]  = b[0], st[23] = b[8];
        memcpy(st + 1, b + 1, sizeof(double) * 7);
        memcpy(st + 24, b + 9, sizeof(double) * 7);
        if (interp[i]) {
            // times[i] lies in [t0, t1): index 1 of the two-state window
            double t = times[i];
            if (!(b[0] <= t && t < b[8])) return ICG_ERR_INVALID;
            orc_ins_camera_pose(2, imu, st, pose_b_c12, t, pose12_out + 12 * (size_t) i);
        } else {
            orc_ins_camera_pose(1, imu, st, pose_b_c12, b[0] - 1.0, pose12_out + 12 * (size_t) i); // outside -> newest state as is
        }
    }
    return ICG_OK;
}

} // extern "C"

// ---- icg_tracker_* on the CPU (TEST INFRASTRUCTURE, like the rest of this file) -------------------------------------------------------------
// The device-resident tracker's ABI with host memory as "device memory": the blocks are heap memory, a step runs the SAME stage bodies
// (ic-gvins_amd/host/track_core.h, here compiled by g++) between the shim's primitives, stream after stream.  With it the host executor of
// the device engine (TrackingBatch::stepDevice: download / import / view / absorb / upload, log drains, statistics) runs in the CPU suite,
// and the GPU tests have a checker that speaks the same ABI.
#include "../ic-gvins_amd/host/track_core.h"

static_assert(sizeof(icg_tracker_config) == sizeof(tc::Cfg), "icg_tracker_config mirrors tc::Cfg");

struct icg_tracker {
    icg_ctx *ctx;
    int n;
    tc::Cfg cfg;
    icg_detect_grid grid;
    std::vector<uint32_t> buckets;
    std::vector<tc::Stream *> streams;
    struct Arena {
        int32_t pre_slot = -1, lk_count = 0, rs_count = 0, tri_count = 0, tri_n_tcw = 0, det_slot = -1, det_mask_count = 0, det_count = 0;
        double pre_hist = 0;
        std::vector<int32_t> lk_prev_slot, lk_next_slot, tri_T0, tri_T1, det_quota;
        std::vector<tc::P2f> lk_prev, lk_guess, lk_out, lk_undist, rs_p1, rs_p2, det_mask_pts, det_out;
        std::vector<uint8_t> lk_status, rs_mask;
        std::vector<double> tri_Tcw, tri_pc0, tri_pc1, tri_pw;
    };
    std::vector<Arena> arena;
    tc::Scratch scratch;
};

static tc::Io shim_io(icg_tracker::Arena &a, int lk_base) {
    tc::Io io;
    memset(&io, 0, sizeof io);
    io.pre_slot = &a.pre_slot, io.pre_hist = &a.pre_hist;
    io.lk_count = &a.lk_count, io.lk_prev_slot = a.lk_prev_slot.data(), io.lk_next_slot = a.lk_next_slot.data();
    io.lk_prev = a.lk_prev.data(), io.lk_guess = a.lk_guess.data(), io.lk_out = a.lk_out.data(), io.lk_undist = a.lk_undist.data();
    io.lk_status = a.lk_status.data(), io.lk_base = lk_base;
    io.rs_count = &a.rs_count, io.rs_p1 = a.rs_p1.data(), io.rs_p2 = a.rs_p2.data(), io.rs_mask = a.rs_mask.data();
    io.tri_count = &a.tri_count, io.tri_n_tcw = &a.tri_n_tcw, io.tri_T0 = a.tri_T0.data(), io.tri_T1 = a.tri_T1.data();
    io.tri_Tcw = a.tri_Tcw.data(), io.tri_pc0 = a.tri_pc0.data(), io.tri_pc1 = a.tri_pc1.data(), io.tri_pw = a.tri_pw.data();
    io.det_slot = &a.det_slot, io.det_quota = a.det_quota.data(), io.det_mask_count = &a.det_mask_count;
    io.det_mask_pts = a.det_mask_pts.data(), io.det_count = &a.det_count, io.det_out = a.det_out.data();
    return io;
}

extern "C" {

size_t icg_tracker_block_bytes(void) { return sizeof(tc::Stream); }

void icg_tracker_destroy(icg_tracker *t) {
    if (!t) return;
    for (tc::Stream *s : t->streams) free(s);
    delete t;
}

int icg_tracker_create(icg_ctx *ctx, int n_streams, const icg_tracker_config *cfg, const uint32_t *buckets_after, int n_buckets_after, icg_tracker **out) {
    if (!ctx || !cfg || !buckets_after || !out || n_streams <= 0 || n_buckets_after < tc::MAX_ROWS + 2) return ICG_ERR_INVALID;
    if (ctx->cfg.n_slots < tc::MAX_SLOTS * n_streams || cfg->block_cnts > tc::MAX_BLOCKS || cfg->max_per_job + 64 > tc::MAX_ROWS) return ICG_ERR_CAPACITY;
    icg_tracker *t = new icg_tracker;
    t->ctx = ctx, t->n = n_streams;
    memcpy(&t->cfg, cfg, sizeof(tc::Cfg));
    t->grid.block_cols = cfg->block_cols, t->grid.block_rows = cfg->block_rows, t->grid.block_w = cfg->block_w, t->grid.block_h = cfg->block_h;
    t->grid.min_dist = cfg->min_pixel_distance, t->grid.max_per_block = cfg->max_block_features;
    t->buckets.assign(buckets_after, buckets_after + n_buckets_after);
    t->arena.resize((size_t) n_streams);
    for (int s = 0; s < n_streams; s++) {
        tc::Stream *S = static_cast<tc::Stream *>(calloc(1, sizeof(tc::Stream)));
        if (!S) {
            icg_tracker_destroy(t);
            return ICG_ERR_NOMEM;
        }
        tc::stream_init(*S, s * tc::MAX_SLOTS);
        t->streams.push_back(S);
        auto &a = t->arena[(size_t) s];
        const size_t R = tc::MAX_ROWS;
        a.lk_prev_slot.resize(R), a.lk_next_slot.resize(R), a.lk_prev.resize(R), a.lk_guess.resize(R), a.lk_out.resize(R), a.lk_undist.resize(R);
        a.lk_status.resize(R), a.rs_mask.resize(R), a.rs_p1.resize(R), a.rs_p2.resize(R), a.tri_T0.resize(R), a.tri_T1.resize(R);
        a.tri_Tcw.resize(12 * tc::MAX_TCW), a.tri_pc0.resize(3 * R), a.tri_pc1.resize(3 * R), a.tri_pw.resize(3 * R);
        a.det_quota.resize(tc::MAX_BLOCKS), a.det_mask_pts.resize(R), a.det_out.resize(R);
    }
    *out = t;
    return ICG_OK;
}

int icg_tracker_step(icg_tracker *t, const uint8_t *const *images, int stride, int channels, int images_on_device, const double *stamps,
                     const double *poses12, icg_tracker_result *results) {
    if (!t || !images || !stamps || !poses12 || !results) return ICG_ERR_INVALID;
    icg_ctx *ctx     = t->ctx;
    const tc::Cfg &C = t->cfg;
    int overflow     = 0;
    for (int s = 0; s < t->n; s++) {
        tc::Stream &S = *t->streams[(size_t) s];
        auto &a       = t->arena[(size_t) s];
        icg_tracker_result &r = results[s];
        memset(&r, 0, sizeof r);
        int work[4] = {0, 0, 0, 0};
        if (images[s]) {
            tc::Io io = shim_io(a, s * tc::MAX_ROWS);
            tc::Pose pose;
            memcpy(pose.R, poses12 + 12 * (size_t) s, sizeof(double) * 12);
            tc::stage_begin_frame(S, io, stamps[s], pose, (tc::u64) (uintptr_t) images[s]);
            int rc = icg_frames_preprocess(ctx, 1, &a.pre_slot, &images[s], stride, channels, images_on_device, C.check_histogram ? &a.pre_hist : nullptr);
            if (rc) return rc;
            auto detect = [&]() -> int {
                a.det_count = 0;
                if (a.det_slot < 0) return ICG_OK;
                const int32_t moff[2] = {0, a.det_mask_count};
                std::vector<float> out((size_t) C.max_per_job * 2);
                int32_t cnt = 0;
                int rcd = icg_detect(ctx, 1, &a.det_slot, &t->grid, moff, reinterpret_cast<const float *>(a.det_mask_pts.data()), a.det_quota.data(),
                                     C.max_per_job, out.data(), &cnt, nullptr);
                if (rcd) return rcd;
                a.det_count = cnt;
                memcpy(a.det_out.data(), out.data(), sizeof(float) * 2 * (size_t) cnt);
                work[1]++;
                return ICG_OK;
            };
            tc::stage_on_preprocess(S, C, io, t->scratch);
            if ((rc = detect())) return rc;
            tc::stage_on_detect_a(S, C, io, t->scratch);
            work[0] = a.lk_count;
            if (a.lk_count > 0) {
                rc = icg_lk_track_fb(ctx, a.lk_count, a.lk_prev_slot.data(), a.lk_next_slot.data(), reinterpret_cast<const float *>(a.lk_prev.data()),
                                     reinterpret_cast<const float *>(a.lk_guess.data()), reinterpret_cast<float *>(a.lk_out.data()), a.lk_status.data(),
                                     reinterpret_cast<float *>(a.lk_undist.data()), nullptr, nullptr);
                if (rc) return rc;
            }
            tc::stage_on_lk(S, C, io, t->buckets.data(), t->scratch);
            if (a.rs_count > 0) {
                work[2]               = 1;
                const int32_t off[2] = {0, a.rs_count};
                for (int k = 0; k < a.rs_count; k++) a.rs_mask[(size_t) k] = 1;
                rc = icg_fm_ransac(ctx, 1, off, reinterpret_cast<const float *>(a.rs_p1.data()), reinterpret_cast<const float *>(a.rs_p2.data()),
                                   C.reprojection_error_std, 0.99, a.rs_mask.data());
                if (rc) return rc;
            }
            tc::stage_on_ransac(S, C, io, t->scratch);
            work[3] = a.tri_count;
            if (a.tri_count > 0) {
                rc = icg_triangulate(ctx, a.tri_count, a.tri_T0.data(), a.tri_T1.data(), a.tri_n_tcw, a.tri_Tcw.data(), a.tri_pc0.data(), a.tri_pc1.data(),
                                     a.tri_pw.data());
                if (rc) return rc;
            }
            tc::stage_on_triangulate(S, C, io, t->buckets.data(), t->scratch);
            if ((rc = detect())) return rc;
            tc::stage_on_detect_b(S, C, io);
            r.log_valid    = (S.log_valid && S.result == tc::TRACK_TRACKING && S.mode == tc::M_TRACK && S.lost_reset != 2) ? 1 : 0;
            r.log_features = S.cur >= 0 ? S.frame[S.cur].n_rows : 0;
            tc::stage_end_frame(S, C);
            r.active = 1;
        }
        for (int k = 0; k < 5; k++) r.log_data[k] = S.log_data[k];
        r.state = S.result, r.is_new_keyframe = S.isnewkeyframe, r.overflow = S.overflow;
        r.n_features = S.cur >= 0 ? S.frame[S.cur].n_rows : 0, r.n_candidates = S.n_new, r.window_keyframes = S.n_map_kf, r.landmarks = S.n_landmarks;
        r.frames = S.frames, r.keyframes = S.keyframes, r.tracked_sum = S.tracked_sum, r.digest = S.digest;
        r.frame_id = S.frame_id, r.keyframe_id = S.keyframe_id, r.mappoint_id = S.mappoint_id, r.last_input_fid = S.last_input_fid;
        r.need_detect_a = (S.isinitializing && (S.ref < 0 || S.n_ref == 0)) ? 1 : 0;
        r.n_log = S.n_log;
        r.lk_points = work[0], r.detect_jobs = work[1], r.ransac_sets = work[2], r.tri_points = work[3];
        overflow |= S.overflow;
    }
    if (overflow) {
        ctx->err = "tracker block capacity exceeded";
        return ICG_ERR_CAPACITY;
    }
    return ICG_OK;
}

int icg_tracker_download(icg_tracker *t, int stream, void *block) {
    if (!t || !block || stream < 0 || stream >= t->n) return ICG_ERR_INVALID;
    memcpy(block, t->streams[(size_t) stream], sizeof(tc::Stream));
    return ICG_OK;
}
int icg_tracker_upload(icg_tracker *t, int stream, const void *block) {
    if (!t || !block || stream < 0 || stream >= t->n) return ICG_ERR_INVALID;
    memcpy(t->streams[(size_t) stream], block, sizeof(tc::Stream));
    return ICG_OK;
}
int icg_tracker_fetch_logs(icg_tracker *t, int n_req, const int32_t *streams, const int32_t *counts, void *out, int entry_stride) {
    if (!t || n_req < 0) return ICG_ERR_INVALID;
    for (int k = 0; k < n_req; k++) {
        tc::Stream &S = *t->streams[(size_t) streams[k]];
        if (counts[k] != S.n_log || counts[k] > entry_stride) return ICG_ERR_INVALID;
        memcpy((char *) out + sizeof(tc::LmLog) * (size_t) k * entry_stride, S.log, sizeof(tc::LmLog) * (size_t) counts[k]);
        S.n_log = 0;
    }
    return ICG_OK;
}
int icg_tracker_reset_log(icg_tracker *t, int stream) {
    if (!t || stream < 0 || stream >= t->n) return ICG_ERR_INVALID;
    t->streams[(size_t) stream]->n_log = 0;
    return ICG_OK;
}

} // extern "C"
