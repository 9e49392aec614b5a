"""Column layout of a marginalization built from a synthetic reprojection window:
the oldest keyframe's pose and the landmarks referenced in it are marginalized, everything else is retained."""
import numpy as np

import reproj_data as rd


def make_problem(n_lm=60, n_kf=6, seed=0, estimate_ext=True, estimate_td=True):
    w = rd.make_window(n_lm, n_kf, seed=seed, pixel_noise=0.7)
    # keep only factors whose reference frame is the oldest keyframe (ic_gvins.cc:1554-1610)
    keep = w["idx_i"] == 0
    obs = w["obs_soa"][:, keep]
    ii, jj, ll = w["idx_i"][keep], w["idx_j"][keep], w["idx_lm"][keep]
    lms = np.unique(ll)
    col_pose = np.full(n_kf, -1, np.int32)
    col_lm = np.full(n_lm, -1, np.int32)
    c = 0
    col_pose[0] = c  # marginalized: pose 0 (local size 6) ...
    c += 6
    for l in lms:    # ... and its landmarks
        col_lm[l] = c
        c += 1
    m = c
    for k in range(1, n_kf):
        if np.any(jj == k):
            col_pose[k] = c
            c += 6
    col_ext = c if estimate_ext else -1
    c += 6 if estimate_ext else 0
    col_td = c if estimate_td else -1
    c += 1 if estimate_td else 0
    return dict(w=w, obs=obs, ii=ii, jj=jj, ll=ll, col_pose=col_pose, col_lm=col_lm, col_ext=col_ext, col_td=col_td, m=m,
                local_size=c)
