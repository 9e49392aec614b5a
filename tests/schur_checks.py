"""Shared bodies for the ABI-level checks of SURVEY.md §8 row f1 (icg_reproj_schur / icg_reproj_backsub / icg_reproj_cost):
run on the oracle shim (CPU) and on the HIP library (GPU) through the same icgvins.Context wrapper."""
import numpy as np

import reproj_data as rd


def full_system(oracle, w, col_pose, col_ext, col_td, P, huber, active=None):
    """(P+L)^2 normal equations from the oracle's per-factor Jacobians (numpy assembly, independent of orc_solve.cc)"""
    r, J = oracle.reproj_eval(w["obs_soa"], w["idx_i"], w["idx_j"], w["idx_lm"], w["poses"], w["ext"], w["invdepth"], w["td"], huber=huber)
    L = len(w["invdepth"])
    N = P + L
    H, b = np.zeros((N, N)), np.zeros(N)
    n = len(r)
    act = np.ones(n, bool) if active is None else np.asarray(active, bool)
    for f in np.nonzero(act)[0]:
        cols, blocks = [], []
        for c0, blk in ((col_pose[w["idx_i"][f]], J[f, 0:14].reshape(2, 7)[:, :6]), (col_pose[w["idx_j"][f]], J[f, 14:28].reshape(2, 7)[:, :6]),
                        (col_ext, J[f, 28:42].reshape(2, 7)[:, :6]), (P + w["idx_lm"][f], J[f, 42:44].reshape(2, 1)), (col_td, J[f, 44:46].reshape(2, 1))):
            if c0 >= 0:
                cols.extend(range(c0, c0 + blk.shape[1]))
                blocks.append(blk)
        Jr = np.concatenate(blocks, axis=1)
        H[np.ix_(cols, cols)] += Jr.T @ Jr
        b[cols] -= Jr.T @ r[f]
    s = (r * r).sum(axis=1)
    rho = np.where(s > huber * huber, 2 * s - huber * huber, s) if huber > 0 else s  # r is Huber-corrected: |r_c|^2 = a sqrt(s) for outliers
    return H, b, 0.5 * rho[act].sum()


def check_schur(ctx, oracle, tol=1e-9):
    w = rd.make_window(90, 7, seed=5, pixel_noise=2.0)
    # perturb so that residuals are not tiny and some factors sit on the Huber branch
    rng = np.random.RandomState(2)
    w["invdepth"] = w["invdepth"] * (1 + rng.normal(0, 0.2, len(w["invdepth"])))
    K, L = w["poses"].shape[0], len(w["invdepth"])
    n = w["obs_soa"].shape[1]
    ctx.reproj_set_factors(w["obs_soa"], w["idx_i"], w["idx_j"], w["idx_lm"])
    for huber, ext_const, td_const, pose0_const, mask in [(1.0, False, False, False, False), (0.0, True, False, True, False), (1.5, False, True, False, True)]:
        col_pose, c = np.full(K, -1, np.int32), 0
        for k in range(K):
            if k == 0 and pose0_const:
                continue
            col_pose[k] = c
            c += 6
        col_ext = -1 if ext_const else c
        c += 0 if ext_const else 6
        col_td = -1 if td_const else c
        c += 0 if td_const else 1
        P = c + 3  # three spare camera columns that no visual factor touches (host-only blocks in a real window)
        active = None
        if mask:
            active = (rng.uniform(0, 1, n) > 0.2).astype(np.uint8)
            active[w["idx_lm"] == 4] = 0  # a landmark that loses every factor
        ctx.reproj_eval_resident(w["poses"], w["ext"], w["invdepth"], w["td"], huber=huber, fetch=False)
        H, b, cost = full_system(oracle, w, col_pose, col_ext, col_td, P, huber, active)
        for damp in (0.0 if not mask else 1e-4, 1e-4, 3.0):
            S, s, dg, cst = ctx.reproj_schur(P, col_pose, col_ext, col_td, active=active, reassemble=(damp != 3.0), damp=damp)
            hll = np.diag(H)[P:]
            inv = np.where(hll > 0, 1.0 / (hll + np.clip(hll, 1e-6, 1e32) * damp), 0.0)
            G = H[P:, :P]
            S_exp = H[:P, :P] - G.T @ (inv[:, None] * G)
            s_exp = b[:P] - G.T @ (inv * b[P:])
            sc = max(1.0, np.abs(S_exp).max())
            assert np.abs(S - S_exp).max() < tol * sc, (huber, damp, np.abs(S - S_exp).max() / sc)
            assert np.abs(s - s_exp).max() < tol * max(1.0, np.abs(s_exp).max())
            assert np.abs(dg - np.diag(H)[:P]).max() < tol * sc
            if damp != 3.0:
                assert abs(cst - cost) < tol * max(1.0, cost), (cst, cost)
            # reduced solve + device back-substitution == dense solve of the full damped system
            Dc = np.clip(np.diag(H)[:P], 1e-6, 1e32) * max(damp, 1e-3)
            dc = np.linalg.solve(S + np.diag(Dc), s)
            dl, terms = ctx.reproj_backsub(P, dc, L)
            keep = np.concatenate([np.ones(P, bool), hll > 0])
            Dfull = np.concatenate([Dc, np.clip(hll, 1e-6, 1e32) * damp])
            A = (H + np.diag(Dfull))[np.ix_(keep, keep)]
            if damp > 0 or np.linalg.cond(A) < 1e12:
                d_exp = np.zeros(P + L)
                d_exp[keep] = np.linalg.solve(A, b[keep])
                scale = max(1e-12, np.abs(d_exp).max())
                assert np.abs(dc - d_exp[:P]).max() < 1e-6 * scale and np.abs(dl - d_exp[P:]).max() < 1e-6 * scale, (huber, damp)
            assert np.all(dl[hll == 0] == 0)
            assert abs(terms[0] - (b[P:] ** 2 * inv).sum()) < tol * max(1.0, abs(terms[0]))
            dll = np.where(hll > 0, np.clip(hll, 1e-6, 1e32) * damp, 0.0)
            assert abs(terms[1] - (dll * dl * dl).sum()) < 1e-9 * max(1e-30, abs(terms[1])) + 1e-300
        # cost of a residual-only evaluation at another point
        w2 = dict(w)
        w2["invdepth"] = w["invdepth"] * 1.01
        ctx.reproj_eval_resident(w2["poses"], w2["ext"], w2["invdepth"], w2["td"], want_jac=False, huber=huber, fetch=False)
        _, _, cost2 = full_system(oracle, w2, col_pose, col_ext, col_td, P, huber, active)
        assert abs(ctx.reproj_cost(active) - cost2) < tol * max(1.0, cost2)
