"""test problems and an independent dense Levenberg-Marquardt in numpy for the window optimizer.

`dense_lm` follows the same published trust-region rules as icg::WindowSolver (host/solver_hip.cc) but never forms a Schur
complement: it assembles the FULL damped normal equations (camera + inverse-depth columns) from the oracle's per-factor
residuals/Jacobians and solves them with numpy.  Agreement of the two therefore checks the device-side elimination, the
back-substitution, the model-decrease bookkeeping and the step control — not merely that two copies of one code path agree.
"""
import ctypes as C

import numpy as np

import reproj_data as rd


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_problem(n_lm=60, n_kf=6, seed=0, n_outliers=0, perturb=1.0):
    """truth window + noisy observations (+ gross outliers) + a perturbed starting point + pose priors at the truth"""
    w = rd.make_window(n_lm, n_kf, seed=seed, pixel_noise=0.3)
    rng = np.random.RandomState(1000 + seed)
    obs = w["obs_soa"].copy()
    n = obs.shape[1]
    bad = rng.choice(n, n_outliers, replace=False) if n_outliers else np.zeros(0, int)
    obs[3, bad] += rng.choice([-1, 1], len(bad)) * rng.uniform(0.03, 0.08, len(bad))  # 25-60 px at f = 787
    start = dict(poses=np.stack([rd.pose_plus(p, perturb * rng.normal(0, [0.05] * 3 + [0.01] * 3)) for p in w["poses"]]),
                 ext=rd.pose_plus(w["ext"], perturb * rng.normal(0, [0.005] * 3 + [0.003] * 3)),
                 invdepth=w["invdepth"] * (1 + perturb * rng.normal(0, 0.15, len(w["invdepth"]))), td=w["td"] + perturb * 0.002)
    return dict(obs=obs, ii=w["idx_i"], jj=w["idx_j"], ll=w["idx_lm"], truth=w, start=start, prior=w["poses"].copy(), outliers=np.sort(bad))


# ---- the host layer (capi icgh_backend_solve) ---------------------------------------------------------------------------------
def host_solve(lib, P, prior_weight=30.0, huber=1.0, ext_const=False, td_const=False, iters1=6, iters2=18, chi2=5.991):
    s = P["start"]
    poses, ext, inv, td = s["poses"].copy(), s["ext"].copy(), s["invdepth"].copy(), np.array([s["td"]])
    n = P["obs"].shape[1]
    summ, active = np.zeros(10), np.zeros(n, np.uint8)
    err = C.create_string_buffer(512)
    obs = np.ascontiguousarray(P["obs"])
    rc = lib.icgh_backend_solve(n, _p(obs), _p(np.ascontiguousarray(P["ii"], np.int32)), _p(np.ascontiguousarray(P["jj"], np.int32)),
                                _p(np.ascontiguousarray(P["ll"], np.int32)), poses.shape[0], _p(poses), _p(ext), len(inv), _p(inv), _p(td),
                                _p(np.ascontiguousarray(P["prior"])), C.c_double(prior_weight), C.c_double(huber), int(ext_const), int(td_const),
                                int(iters1), int(iters2), C.c_double(chi2), _p(summ), _p(active), err, 512)
    assert rc == 0, (rc, err.value)
    return dict(poses=poses, ext=ext, invdepth=inv, td=float(td[0]), summary=summ[:8], active=active, solve_ms=float(summ[8]), setup_ms=float(summ[9]))


def host_solve_batch(lib, problems, prior_weight=30.0, huber=1.0, ext_const=False, td_const=False, iters1=6, iters2=18, chi2=5.991):
    """all problems in ONE WindowSolverBatch (lock-step LM); -> list of result dicts like host_solve, and the solve wall time in ms"""
    W = len(problems)
    fac_off = np.concatenate([[0], np.cumsum([P["obs"].shape[1] for P in problems])]).astype(np.int32)
    pose_off = np.concatenate([[0], np.cumsum([P["start"]["poses"].shape[0] for P in problems])]).astype(np.int32)
    lm_off = np.concatenate([[0], np.cumsum([len(P["start"]["invdepth"]) for P in problems])]).astype(np.int32)
    obs = np.ascontiguousarray(np.concatenate([P["obs"] for P in problems], axis=1))
    ii = np.ascontiguousarray(np.concatenate([P["ii"] for P in problems]), np.int32)
    jj = np.ascontiguousarray(np.concatenate([P["jj"] for P in problems]), np.int32)
    ll = np.ascontiguousarray(np.concatenate([P["ll"] for P in problems]), np.int32)
    poses = np.ascontiguousarray(np.concatenate([P["start"]["poses"] for P in problems]))
    ext = np.ascontiguousarray(np.stack([P["start"]["ext"] for P in problems]))
    inv = np.ascontiguousarray(np.concatenate([P["start"]["invdepth"] for P in problems]))
    td = np.array([P["start"]["td"] for P in problems], np.float64)
    prior = np.ascontiguousarray(np.concatenate([P["prior"] for P in problems]))
    summ, ms = np.zeros((W, 8)), C.c_double(0)
    err = C.create_string_buffer(512)
    rc = lib.icgh_backend_solve_batch(W, _p(fac_off), _p(pose_off), _p(lm_off), _p(obs), _p(ii), _p(jj), _p(ll), _p(poses), _p(ext), _p(inv), _p(td),
                                      _p(prior), C.c_double(prior_weight), C.c_double(huber), int(ext_const), int(td_const), int(iters1), int(iters2),
                                      C.c_double(chi2), _p(summ), C.byref(ms), err, 512)
    assert rc == 0, (rc, err.value)
    out = []
    for w in range(W):
        out.append(dict(poses=poses[pose_off[w]:pose_off[w + 1]], ext=ext[w], invdepth=inv[lm_off[w]:lm_off[w + 1]], td=float(td[w]), summary=summ[w]))
    return out, ms.value


def host_solve_throughput(lib, P, threads, repeat, prior_weight=30.0, huber=1.0, iters1=6, iters2=18, chi2=5.991):
    """-> windows per second with `threads` solvers in flight (each its own device context), problem construction excluded"""
    s = P["start"]
    n = P["obs"].shape[1]
    err = C.create_string_buffer(512)
    lib.icgh_backend_solve_throughput.restype = C.c_double
    sec = lib.icgh_backend_solve_throughput(n, _p(np.ascontiguousarray(P["obs"])), _p(np.ascontiguousarray(P["ii"], np.int32)),
                                            _p(np.ascontiguousarray(P["jj"], np.int32)), _p(np.ascontiguousarray(P["ll"], np.int32)), s["poses"].shape[0],
                                            _p(np.ascontiguousarray(s["poses"])), _p(np.ascontiguousarray(s["ext"])), len(s["invdepth"]),
                                            _p(np.ascontiguousarray(s["invdepth"])), C.c_double(s["td"]), _p(np.ascontiguousarray(P["prior"])),
                                            C.c_double(prior_weight), C.c_double(huber), int(iters1), int(iters2), C.c_double(chi2), int(threads), int(repeat),
                                            err, 512)
    assert sec > 0, (sec, err.value)
    return threads * repeat / sec


# ---- independent dense LM -------------------------------------------------------------------------------------------------------
def _prior_eval(x, x0, w):
    """PosePriorFactor of host/capi.cc: r = w [p - p0; 2 vec(q0^-1 q)], J (6x6 tangent) = w diag(1,1,1, dq_w, dq_w, dq_w)"""
    n2 = float(x0[3:] @ x0[3:])
    ax, ay, az, aw = -x0[3] / n2, -x0[4] / n2, -x0[5] / n2, x0[6] / n2
    bx, by, bz, bw = x[3:]
    dq = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                   aw * bw - ax * bx - ay * by - az * bz])
    r = w * np.concatenate([x[:3] - x0[:3], 2.0 * dq[:3]])
    J = np.zeros((6, 6))
    J[:3, :3] = w * np.eye(3)
    J[3:, 3:] = w * dq[3] * np.eye(3)
    return r, J


class DenseLM:
    def __init__(self, oracle, P, prior_weight, huber, ext_const, td_const):
        self.o, self.P, self.w, self.huber = oracle, P, prior_weight, huber
        s = P["start"]
        self.poses, self.ext, self.inv, self.td = s["poses"].copy(), s["ext"].copy(), s["invdepth"].copy(), float(s["td"])
        K, L = self.poses.shape[0], len(self.inv)
        self.K, self.L = K, L
        # column layout: poses (6 each), ext (6), td (1), then inverse depths — any order works for a dense solve
        c = 0
        self.col_pose = np.arange(K) * 6
        c = 6 * K
        self.col_ext = -1 if ext_const else c
        c += 0 if ext_const else 6
        self.col_td = -1 if td_const else c
        c += 0 if td_const else 1
        self.Pc = c
        self.col_lm = c + np.arange(L)
        self.N = c + L
        self.active = np.ones(P["obs"].shape[1], bool)

    def _eval(self, huber):
        P = self.P
        return self.o.reproj_eval(P["obs"], P["ii"], P["jj"], P["ll"], self.poses, self.ext, self.inv, self.td, huber=huber)

    def cost(self):
        r, _ = self._eval(0.0)
        s = (r * r).sum(axis=1)[self.active]
        a = self.huber
        rho = np.where(s > a * a, 2 * a * np.sqrt(s) - a * a, s) if a > 0 else s
        c = 0.5 * rho.sum()
        for k in range(self.K):
            rk, _ = _prior_eval(self.poses[k], self.P["prior"][k], self.w)
            c += 0.5 * rk @ rk
        return c

    def normal(self):
        r, J = self._eval(self.huber)
        H, b = np.zeros((self.N, self.N)), np.zeros(self.N)
        P = self.P
        for f in np.nonzero(self.active)[0]:
            cols, blocks = [], []
            Jf = J[f]
            for c0, blk in ((self.col_pose[P["ii"][f]], Jf[0:14].reshape(2, 7)[:, :6]), (self.col_pose[P["jj"][f]], Jf[14:28].reshape(2, 7)[:, :6]),
                            (self.col_ext, Jf[28:42].reshape(2, 7)[:, :6]), (self.col_lm[P["ll"][f]], Jf[42:44].reshape(2, 1)),
                            (self.col_td, Jf[44:46].reshape(2, 1))):
                if c0 >= 0:
                    cols.extend(range(c0, c0 + blk.shape[1]))
                    blocks.append(blk)
            Jrow = np.concatenate(blocks, axis=1)
            H[np.ix_(cols, cols)] += Jrow.T @ Jrow
            b[cols] -= Jrow.T @ r[f]
        for k in range(self.K):
            rk, Jk = _prior_eval(self.poses[k], self.P["prior"][k], self.w)
            c0 = self.col_pose[k]
            H[c0:c0 + 6, c0:c0 + 6] += Jk.T @ Jk
            b[c0:c0 + 6] -= Jk.T @ rk
        return H, b

    def apply(self, d):
        for k in range(self.K):
            self.poses[k] = rd.pose_plus(self.poses[k], d[self.col_pose[k]:self.col_pose[k] + 6])
        if self.col_ext >= 0:
            self.ext = rd.pose_plus(self.ext, d[self.col_ext:self.col_ext + 6])
        if self.col_td >= 0:
            self.td += d[self.col_td]
        self.inv = self.inv + d[self.col_lm]

    def state(self):
        return self.poses.copy(), self.ext.copy(), self.inv.copy(), self.td

    def set_state(self, st):
        self.poses, self.ext, self.inv, self.td = st[0].copy(), st[1].copy(), st[2].copy(), st[3]

    def solve(self, iters, radius0=1e4):
        radius, dec = radius0, 2.0
        cost = self.cost()
        init = cost
        good = bad = 0
        H, b = self.normal()
        for it in range(iters):
            if np.abs(b[:self.Pc]).max() < 1e-10:
                break
            dg = np.diag(H).copy()
            has = dg > 0  # landmarks whose factors were all removed keep an empty row (delta = 0), as on the device
            D = np.clip(dg, 1e-6, 1e32) / radius
            A = H + np.diag(D)
            idx = np.nonzero(has | (np.arange(self.N) < self.Pc))[0]
            d = np.zeros(self.N)
            try:
                np.linalg.cholesky(A[np.ix_(idx, idx)])
                d[idx] = np.linalg.solve(A[np.ix_(idx, idx)], b[idx])
                model = 0.5 * (d @ b + d[idx] @ (D[idx] * d[idx]))
                ok = model > 0
            except np.linalg.LinAlgError:
                ok = False
            if not ok:
                radius /= dec
                dec *= 2
                bad += 1
                continue
            xn = np.sqrt((self.poses ** 2).sum() + (0 if self.col_ext < 0 else (self.ext ** 2).sum()) + (self.inv ** 2).sum()
                         + (0 if self.col_td < 0 else self.td ** 2))
            if np.linalg.norm(d) <= 1e-8 * (xn + 1e-8):
                break
            st = self.state()
            self.apply(d)
            new = self.cost()
            rho = (cost - new) / model
            if rho > 1e-3:
                change = cost - new
                cost = new
                good += 1
                radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3))
                dec = 2.0
                if abs(change) < 1e-6 * cost:
                    break
                if it + 1 < iters:
                    H, b = self.normal()
            else:
                self.set_state(st)
                radius /= dec
                dec *= 2
                bad += 1
        return init, cost, good, bad

    def chi2_cull(self, chi2):
        r, _ = self._eval(0.0)
        s = (r * r).sum(axis=1)
        kill = self.active & (s > chi2)
        self.active &= ~kill
        return int(kill.sum())


def dense_solve(oracle, P, prior_weight=30.0, huber=1.0, ext_const=False, td_const=False, iters1=6, iters2=18, chi2=5.991):
    lm = DenseLM(oracle, P, prior_weight, huber, ext_const, td_const)
    i1, c1, g1, b1 = lm.solve(iters1)
    summ = [i1, c1, c1, g1, b1, 0, 0, 0]
    if chi2 > 0:
        removed = lm.chi2_cull(chi2)
        _, c2, g2, b2 = lm.solve(iters2)
        summ[2], summ[5], summ[6], summ[7] = c2, g2, b2, removed
    return dict(poses=lm.poses, ext=lm.ext, invdepth=lm.inv, td=lm.td, summary=np.array(summ, float), active=lm.active.astype(np.uint8))
