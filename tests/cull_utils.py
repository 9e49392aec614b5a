"""SURVEY.md §8 row f3 (outlier culling / window statistics): test data, the golden arithmetic cases and an independent Python
restatement of the decision loops of GVINS::gvinsOutlierCulling (ic_gvins.cc:1035-1128) and parametersStatistic (:930-1033)
that works on the raw landmark-graph dump of the host layer (icgh_batch_landmark_table)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "cull_ref_golden.npz")
REPROJ_STD = 1.5
SCALES = [(1.0, 1.0), (3.0, 1.0), (1.0, 3.0)]  # (scale, depth_scale): triangulation gate, culling gate, wide depth gate
NEAREST, FARTHEST = 1.0, 200.0


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_observations(n_poses=6, n_lm=240, seed=3):
    """poses around the origin, landmarks from 0.5 m to 700 m in front (both depth gates are crossed), key points = projection
    + noise between 0 and ~8 px (both error gates are crossed)"""
    import harness as H
    w, h = 1280, 720
    cam = np.asarray(H.camera_for(w, h), np.float64)
    rng = np.random.RandomState(seed)
    poses = []
    for k in range(n_poses):
        R = H._rot_yp(rng.normal(0, 0.05), rng.normal(0, 0.03))
        poses.append(np.concatenate([R.ravel(), rng.normal(0, 0.5, 3)]))
    poses12 = np.ascontiguousarray(np.stack(poses))
    depth = np.concatenate([rng.uniform(0.5, 1.5, 20), rng.uniform(3, 60, n_lm - 60), rng.uniform(150, 250, 20), rng.uniform(500, 700, 20)])
    xy = rng.uniform([-0.5, -0.28], [0.5, 0.28], (n_lm, 2))
    pw = np.ascontiguousarray(np.stack([xy[:, 0] * depth, xy[:, 1] * depth, depth], axis=1))
    pose_idx, lm_idx, pix = [], [], []
    fx, fy, cx, cy, skew = cam[0], cam[1], cam[2], cam[3], cam[4]
    for l in range(n_lm):
        for k in rng.choice(n_poses, rng.randint(1, n_poses + 1), replace=False):
            R, t = poses12[k, :9].reshape(3, 3), poses12[k, 9:]
            pc = R.T @ (pw[l] - t)
            u = fx * pc[0] / pc[2] + skew * pc[1] / pc[2] + cx
            v = fy * pc[1] / pc[2] + cy
            noise = rng.normal(0, 1, 2) * rng.choice([0.3, 1.2, 4.0, 8.0])
            pose_idx.append(k), lm_idx.append(l), pix.append([u + noise[0], v + noise[1]])
    return dict(cam=cam, w=w, h=h, poses12=poses12, pw=pw, pose_idx=np.array(pose_idx, np.int32), lm_idx=np.array(lm_idx, np.int32),
                pix=np.ascontiguousarray(pix, np.float32))


def oracle_eval(oracle, d, scale, dscale):
    n = len(d["pose_idx"])
    err, good = np.zeros(n), np.zeros(n, np.uint8)
    oracle.lib.orc_reproj_error_batch(_p(d["cam"]), n, _p(d["pose_idx"]), _p(d["lm_idx"]), _p(d["poses12"]), _p(d["pw"]), _p(d["pix"]),
                                      C.c_double(REPROJ_STD * scale), C.c_double(NEAREST), C.c_double(FARTHEST * dscale), _p(err), _p(good))
    return err, good


# ---- landmark graph of a stream (host layer) --------------------------------------------------------------------------------------
def landmark_table(sb, stream, max_lm=4096, max_obs=200000):
    lib = sb.lib
    lm_id, lm_pos = np.zeros(max_lm, np.uint64), np.zeros((max_lm, 3))
    lm_flags, lm_ref, lm_off = np.zeros(max_lm, np.int32), np.zeros(max_lm, np.uint64), np.zeros(max_lm + 1, np.int32)
    ob_frame, ob_flags = np.zeros(max_obs, np.uint64), np.zeros(max_obs, np.int32)
    ob_pose, ob_pix = np.zeros((max_obs, 12)), np.zeros((max_obs, 2), np.float32)
    n = lib.icgh_batch_landmark_table(C.c_void_p(sb.h_), stream, max_lm, max_obs, _p(lm_id), _p(lm_pos), _p(lm_flags), _p(lm_ref), _p(lm_off),
                                      _p(ob_frame), _p(ob_flags), _p(ob_pose), _p(ob_pix))
    assert n >= 0, n
    no = int(lm_off[n])
    return dict(id=lm_id[:n].copy(), pos=lm_pos[:n].copy(), outlier=(lm_flags[:n] & 1).astype(bool), ref_frame=lm_ref[:n].copy(), off=lm_off[:n + 1].copy(),
                obs_frame=ob_frame[:no].copy(), obs_flags=ob_flags[:no].copy(), obs_pose=ob_pose[:no].copy(), obs_pix=ob_pix[:no].copy())


def run_culling(sb, mode, lists, std=REPROJ_STD):
    n = sb.n
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    flat = np.ascontiguousarray(np.concatenate(lists) if off[-1] else np.zeros(1), np.uint64)
    out5, stats5 = np.zeros((n, 5), np.int32), np.zeros((n, 5))
    err = C.create_string_buffer(512)
    rc = sb.lib.icgh_batch_culling(C.c_void_p(sb.h_), int(mode), _p(off), _p(flat), C.c_double(std), _p(out5), _p(stats5), err, 512)
    assert rc == 0, (rc, err.value)
    return out5 if mode == 0 else stats5


def _eligible_obs(T, l):
    """observations of landmark l the reference would use (ic_gvins.cc:1061-1069): live feature, not an outlier, frame is a keyframe
    still in the map"""
    return [k for k in range(T["off"][l], T["off"][l + 1]) if (T["obs_flags"][k] & 1) == 0 and (T["obs_flags"][k] & 2) == 0
            and (T["obs_flags"][k] & 8) != 0]


def expected_culling(oracle, cam, T, in_list, std=REPROJ_STD):
    """-> counters [mappoint outliers, feature outliers, num1, num2, num3], set of removed landmark ids, set of (landmark id,
    observation index) features flagged outlier"""
    inl = set(int(x) for x in in_list)
    c = [0, 0, 0, 0, 0]
    removed, flagged = set(), set()
    for l in range(len(T["id"])):
        if T["outlier"][l] or int(T["id"][l]) not in inl:
            continue
        obs = _eligible_obs(T, l)
        errors = []
        if obs:
            d = dict(cam=np.asarray(cam, np.float64), pose_idx=np.arange(len(obs), dtype=np.int32), lm_idx=np.zeros(len(obs), np.int32),
                     poses12=np.ascontiguousarray(T["obs_pose"][obs]), pw=np.ascontiguousarray(T["pos"][l:l + 1]), pix=np.ascontiguousarray(T["obs_pix"][obs]))
            err, good = oracle_eval(oracle, d, 3.0, 1.0)
            for j, k in enumerate(obs):
                if not good[j]:
                    flagged.add((int(T["id"][l]), k - int(T["off"][l])))
                    if T["obs_frame"][k] == T["ref_frame"][l]:
                        c[0] += 1
                        c[2] += 1
                        removed.add(int(T["id"][l]))
                        break
                    c[1] += 1
                else:
                    errors.append(err[j])
        if len(errors) < 2:
            c[0] += 1
            c[3] += 1
            removed.add(int(T["id"][l]))
        elif sum(errors) / len(errors) > std:
            c[0] += 1
            c[4] += 1
            removed.add(int(T["id"][l]))
    return c, removed, flagged


def expected_statistics(oracle, cam, T, in_list):
    inl = set(int(x) for x in in_list)
    means = []
    for l in range(len(T["id"])):
        if T["outlier"][l] or int(T["id"][l]) not in inl:
            continue
        obs = _eligible_obs(T, l)
        if not obs:
            continue
        d = dict(cam=np.asarray(cam, np.float64), pose_idx=np.arange(len(obs), dtype=np.int32), lm_idx=np.zeros(len(obs), np.int32),
                 poses12=np.ascontiguousarray(T["obs_pose"][obs]), pw=np.ascontiguousarray(T["pos"][l:l + 1]), pix=np.ascontiguousarray(T["obs_pix"][obs]))
        err, _ = oracle_eval(oracle, d, 1.0, 1.0)
        means.append(sum(err) / len(err))
    n = len(means)
    if not means:
        means = [0.0]
    m = np.array(means)
    return [m.min(), m.max(), m.sum() / len(m), np.sqrt((m * m).sum() / len(m)), n]
