"""Synthetic sliding-window reprojection batches (SURVEY.md §8(d) C3 shape: L landmarks x (K-1) observers)."""
import numpy as np


def quat_from_rotvec(rv):
    a = np.linalg.norm(rv)
    if a < 1e-12:
        return np.array([0, 0, 0, 1.0])
    ax = rv / a
    return np.array([*(np.sin(a / 2) * ax), np.cos(a / 2)])  # x,y,z,w


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_window(n_landmarks=300, n_kf=10, seed=0, pixel_noise=0.5, f=787.0):
    """Returns dict with obs_soa(15,n), idx_i, idx_j, idx_lm, poses(K,7), ext(7), invdepth(L), td."""
    rng = np.random.RandomState(seed)
    # body poses: forward motion along x with small yaw; body frame front-right-down, camera looks along body x
    poses = np.zeros((n_kf, 7))
    for k in range(n_kf):
        p = np.array([1.0 * k, 0.05 * np.sin(k), 0.02 * k])
        q = quat_from_rotvec(np.array([0.01 * k, -0.02 * np.sin(k), 0.03 * k]))
        poses[k, :3] = p
        poses[k, 3:] = q
    # extrinsic: q_b_c from the reference yaml (config/gvins.yaml:79-80)
    qic = np.array([0.497766, 0.502679, 0.501396, 0.498141])
    qic /= np.linalg.norm(qic)
    tic = np.array([0.074, -0.030, 0.128])
    ext = np.concatenate([tic, qic])
    Ric = quat_to_R(qic)
    td = 0.003
    sigma = 1.5 / f
    obs, ii, jj, ll = [], [], [], []
    invdepth = np.zeros(n_landmarks)
    for l in range(n_landmarks):
        ref = rng.randint(0, n_kf - 1) if l % 3 else 0
        depth = rng.uniform(5, 50)
        invdepth[l] = 1.0 / depth
        pts0 = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.35, 0.35), 1.0])
        vel0 = np.array([rng.normal(0, 0.05), rng.normal(0, 0.05), 0.0])
        td0 = 0.001
        R0 = quat_to_R(poses[ref, 3:])
        pw = R0 @ (Ric @ (pts0 * depth) + tic) + poses[ref, :3]
        for k in range(n_kf):
            if k == ref:
                continue
            Rk = quat_to_R(poses[k, 3:])
            pc = Ric.T @ (Rk.T @ (pw - poses[k, :3]) - tic)
            if pc[2] < 0.5:
                continue
            pts1 = np.array([pc[0] / pc[2] + rng.normal(0, pixel_noise / f), pc[1] / pc[2] + rng.normal(0, pixel_noise / f), 1.0])
            vel1 = np.array([rng.normal(0, 0.05), rng.normal(0, 0.05), 0.0])
            td1 = 0.002
            obs.append(np.concatenate([pts0, pts1, vel0, vel1, [td0, td1, sigma]]))
            ii.append(ref)
            jj.append(k)
            ll.append(l)
    obs = np.array(obs)
    return dict(obs_soa=np.ascontiguousarray(obs.T), idx_i=np.array(ii, np.int32), idx_j=np.array(jj, np.int32),
                idx_lm=np.array(ll, np.int32), poses=poses, ext=ext, invdepth=invdepth, td=td)


def pose_plus(pose, delta):
    """PoseParameterization::Plus (factors/pose_parameterization.h:34-49): p + dp ; q * dq(rotvec)."""
    out = pose.copy()
    out[:3] += delta[:3]
    dq = quat_from_rotvec(delta[3:6])
    x1, y1, z1, w1 = pose[3:]
    x2, y2, z2, w2 = dq
    q = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                  w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    out[3:] = q / np.linalg.norm(q)
    return out
