"""A visual-inertial sliding window for the f1 integration test: K IMU intervals chained by preintegration (so the true states are
exactly consistent with the IMU data), landmarks observed from the resulting body poses through the camera extrinsic."""
import ctypes as C

import numpy as np

import preint_data as pd
import reproj_data as rd


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_vio_window(oracle, n_intervals=6, per=40, n_lm=120, seed=0, pixel_noise=0.3, f=787.0):
    rng = np.random.RandomState(seed)
    imu_all = pd.make_interval(n_intervals * per + 1, seed=seed, noise=True)
    offsets = (np.arange(n_intervals + 1) * (per + 1)).astype(np.int32)
    imu = np.concatenate([imu_all[k * per:k * per + per + 1] for k in range(n_intervals)])  # interval k+1 starts with the last sample of k
    states = [pd.state()]
    for k in range(n_intervals):
        states.append(oracle.preint_integrate(0, imu_all[k * per:k * per + per + 1], states[-1], pd.PARAMS)["cur"])
    states = np.stack(states)
    K = n_intervals + 1
    qic = np.array([0.497766, 0.502679, 0.501396, 0.498141])
    qic /= np.linalg.norm(qic)
    tic = np.array([0.074, -0.030, 0.128])
    ext = np.concatenate([tic, qic])
    Ric = rd.quat_to_R(qic)
    sigma = 1.5 / f
    obs, ii, jj, ll, inv = [], [], [], [], []
    for l in range(n_lm):
        ref = rng.randint(0, K - 1) if l % 3 else 0
        depth = rng.uniform(4, 40)
        pts0 = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3), 1.0])
        R0 = rd.quat_to_R(states[ref, 3:7])
        pw = R0 @ (Ric @ (pts0 * depth) + tic) + states[ref, :3]
        rows = []
        for k in range(K):
            if k == ref:
                continue
            Rk = rd.quat_to_R(states[k, 3:7])
            pc = Ric.T @ (Rk.T @ (pw - states[k, :3]) - tic)
            if pc[2] < 0.5 or abs(pc[0] / pc[2]) > 0.8 or abs(pc[1] / pc[2]) > 0.5:
                continue
            pts1 = np.array([pc[0] / pc[2] + rng.normal(0, pixel_noise / f), pc[1] / pc[2] + rng.normal(0, pixel_noise / f), 1.0])
            rows.append((k, np.concatenate([pts0, pts1, np.zeros(3), np.zeros(3), [0.0, 0.0, sigma]])))
        if len(rows) < 2:
            continue
        lm = len(inv)
        inv.append(1.0 / depth)
        for k, o in rows:
            obs.append(o), ii.append(ref), jj.append(k), ll.append(lm)
    return dict(offsets=offsets, imu=np.ascontiguousarray(imu), states=states, ext=ext, td=0.0, invdepth=np.array(inv),
                obs=np.ascontiguousarray(np.array(obs).T), ii=np.array(ii, np.int32), jj=np.array(jj, np.int32), ll=np.array(ll, np.int32))


def perturbed_start(W, seed=0):
    rng = np.random.RandomState(500 + seed)
    s = W["states"].copy()
    for k in range(1, len(s)):  # state 0 carries the prior
        s[k, :7] = rd.pose_plus(s[k, :7], rng.normal(0, [0.05] * 3 + [0.01] * 3))
        s[k, 7:10] += rng.normal(0, 0.1, 3)
    inv = W["invdepth"] * (1 + rng.normal(0, 0.15, len(W["invdepth"])))
    return s, inv


def host_solve_vio(lib, W, states, inv, prior_weight=100.0, huber=1.0, iters=25):
    st, inv = np.ascontiguousarray(states).copy(), np.ascontiguousarray(inv).copy()
    ext, td = W["ext"].copy(), np.array([W["td"]])
    summ = np.zeros(4)
    err = C.create_string_buffer(512)
    n = W["obs"].shape[1]
    rc = lib.icgh_backend_solve_vio(len(W["offsets"]) - 1, _p(W["offsets"]), _p(W["imu"]), _p(np.ascontiguousarray(pd.PARAMS)), _p(st), n, _p(W["obs"]),
                                    _p(W["ii"]), _p(W["jj"]), _p(W["ll"]), _p(ext), len(inv), _p(inv), _p(td),
                                    _p(np.ascontiguousarray(W["states"][0, :7])), _p(np.ascontiguousarray(W["states"][0, 7:])), C.c_double(prior_weight), C.c_double(huber), int(iters), _p(summ), err, 512)
    assert rc == 0, (rc, err.value)
    return st, inv, summ
