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


def check_schur_any_factor_order(ctx, oracle, tol=1e-9, refuses_same_block=True):
    """The assembly makes no assumption about the factor list (csrc/reproj.hip, asm_plan_build sorts by pose pair and by landmark itself):
    a shuffled list, a landmark whose factors name DIFFERENT reference poses, the same (landmark, observer) pair twice, poses used as
    reference by some factors and as observer by others — all equal the numpy assembly of the same list; and a commit through the staged
    upload equals the classic upload bit for bit.  A factor whose reference and observer are the same block is refused."""
    w = rd.make_window(60, 6, seed=9, pixel_noise=2.0)
    rng = np.random.RandomState(4)
    n0 = w["obs_soa"].shape[1]
    perm = rng.permutation(n0)
    dup = perm[:7]  # seven factors once more (same landmark, same pose pair, same observation)
    order = np.concatenate([perm, dup])
    w2 = dict(w, obs_soa=np.ascontiguousarray(w["obs_soa"][:, order]), idx_i=w["idx_i"][order].copy(), idx_j=w["idx_j"][order].copy(), idx_lm=w["idx_lm"][order].copy())
    # landmark 3's factors get another reference pose here and there (any pose that is not the observer)
    K, L = w["poses"].shape[0], len(w["invdepth"])
    for f in np.nonzero(w2["idx_lm"] == 3)[0][::2]:
        w2["idx_i"][f] = (w2["idx_j"][f] + 1 + rng.randint(0, K - 1)) % K
        assert w2["idx_i"][f] != w2["idx_j"][f]
    n = len(order)
    col_pose = np.arange(K, dtype=np.int32) * 6
    col_pose[2] = -1  # one constant pose in the middle
    col_pose[3:] -= 6
    P = 6 * (K - 1) + 7
    col_ext, col_td = 6 * (K - 1), 6 * (K - 1) + 6
    active = (rng.uniform(0, 1, n) > 0.1).astype(np.uint8)
    out = []
    for upload in (ctx.reproj_set_factors, ctx.reproj_set_factors_staged):
        upload(w2["obs_soa"], w2["idx_i"], w2["idx_j"], w2["idx_lm"])
        ctx.reproj_eval_resident(w2["poses"], w2["ext"], w2["invdepth"], w2["td"], huber=1.0, fetch=False)
        out.append(ctx.reproj_schur(P, col_pose, col_ext, col_td, active=active, damp=1e-4))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b))  # same kernels on the same resident data: the same bits
    S, s, dg, cst = out[0]
    H, b, cost = full_system(oracle, w2, col_pose, col_ext, col_td, P, 1.0, active)
    hll = np.diag(H)[P:]
    inv = np.where(hll > 0, 1.0 / (hll + np.clip(hll, 1e-6, 1e32) * 1e-4), 0.0)
    G = H[P:, :P]
    S_exp = H[:P, :P] - G.T @ (inv[:, None] * G)
    s_exp = b[:P] - G.T @ (inv * b[P:])
    sc = max(1.0, np.abs(S_exp).max())
    assert np.abs(S - S_exp).max() < tol * sc, np.abs(S - S_exp).max() / sc
    assert np.abs(s - s_exp).max() < tol * max(1.0, np.abs(s_exp).max())
    assert np.abs(dg - np.diag(H)[:P]).max() < tol * sc and abs(cst - cost) < tol * max(1.0, cost)
    if refuses_same_block:  # (the device path; the CPU shim forms both triangles on their own)
        assert np.array_equal(S, S.T)  # the upper triangle is the mirror image of the lower one
    if not refuses_same_block:  # (the CPU shim sums whatever it is given; Ceres itself refuses duplicate blocks in a residual)
        return
    bad = dict(w2, idx_i=w2["idx_i"].copy())
    bad["idx_i"][5] = bad["idx_j"][5]
    ctx.reproj_set_factors(bad["obs_soa"], bad["idx_i"], bad["idx_j"], bad["idx_lm"])
    ctx.reproj_eval_resident(bad["poses"], bad["ext"], bad["invdepth"], bad["td"], huber=1.0, fetch=False)
    try:
        ctx.reproj_schur(P, col_pose, col_ext, col_td, active=active, damp=1e-4)
    except Exception as e:  # icgvins.IcgError
        assert "same block" in str(e), e
    else:
        raise AssertionError("a factor with reference == observer was assembled")


def check_schur_windows(make_ctx, tol=1e-9):
    """the many-windows-per-launch entry points against per-window calls of the single-window entry points on the same library
    (those are checked against numpy above): different window sizes, per-window extrinsic / td, constant blocks, a factor mask, a
    second call that re-damps some windows and re-assembles the others at a new linearization point"""
    rng = np.random.RandomState(7)
    specs = [(40, 5), (90, 8), (60, 6), (25, 4), (75, 7)]
    wins = []
    for k, (n_lm, n_kf) in enumerate(specs):
        w = rd.make_window(n_lm, n_kf, seed=20 + k, pixel_noise=2.0)
        w["invdepth"] = w["invdepth"] * (1 + rng.normal(0, 0.2, n_lm))
        w["ext"] = rd.pose_plus(w["ext"], rng.normal(0, [0.01] * 3 + [0.005] * 3))
        w["td"] = 0.003 + 0.001 * k
        wins.append(w)
    W = len(wins)
    pose_off = np.concatenate([[0], np.cumsum([w["poses"].shape[0] for w in wins])])
    lm_off = np.concatenate([[0], np.cumsum([len(w["invdepth"]) for w in wins])]).astype(np.int32)
    fac_off = np.concatenate([[0], np.cumsum([w["obs_soa"].shape[1] for w in wins])]).astype(np.int32)
    obs = np.concatenate([w["obs_soa"] for w in wins], axis=1)
    ii = np.concatenate([w["idx_i"] + pose_off[k] for k, w in enumerate(wins)]).astype(np.int32)
    jj = np.concatenate([w["idx_j"] + pose_off[k] for k, w in enumerate(wins)]).astype(np.int32)
    ll = np.concatenate([w["idx_lm"] + lm_off[k] for k, w in enumerate(wins)]).astype(np.int32)
    poses = np.concatenate([w["poses"] for w in wins])
    ext = np.stack([w["ext"] for w in wins])
    inv = np.concatenate([w["invdepth"] for w in wins])
    td = np.array([w["td"] for w in wins])
    n = obs.shape[1]
    # columns: window k keeps pose 0 constant when k is odd, ext constant when k % 3 == 0, td constant when k % 2 == 0
    col_pose, col_ext, col_td, Pw = np.full(len(poses), -1, np.int32), np.full(W, -1, np.int32), np.full(W, -1, np.int32), []
    for k, w in enumerate(wins):
        c = 0
        for p in range(w["poses"].shape[0]):
            if p == 0 and k % 2 == 1:
                continue
            col_pose[pose_off[k] + p] = c
            c += 6
        if k % 3 != 0:
            col_ext[k] = c
            c += 6
        if k % 2 != 0:
            col_td[k] = c
            c += 1
        Pw.append(c)
    P = max(Pw) + 2
    active = (rng.uniform(0, 1, n) > 0.15).astype(np.uint8)
    huber = 1.0
    damp1 = np.array([1e-4, 0.0, 1e-3, 1e-4, 1e-2])

    def single(k, poses_k, ext_k, inv_k, td_k, damp, reassemble_ctx=None):
        ctx = reassemble_ctx or make_ctx()
        w = wins[k]
        if reassemble_ctx is None:
            ctx.reproj_set_factors(w["obs_soa"], w["idx_i"], w["idx_j"], w["idx_lm"])
        ctx.reproj_eval_resident(poses_k, ext_k, inv_k, td_k, huber=huber, fetch=False)
        cp = col_pose[pose_off[k]:pose_off[k + 1]]
        S, s, dg, cost = ctx.reproj_schur(P, cp, col_ext[k], col_td[k], active=active[fac_off[k]:fac_off[k + 1]], damp=damp)
        return ctx, S, s, dg, cost

    ctxb = make_ctx()
    ctxb.reproj_set_factors(obs, ii, jj, ll)
    ctxb.reproj_set_windows(fac_off, lm_off)
    ctxb.reproj_eval_windows(poses, ext, inv, td, huber=huber)
    S, s, dg, cost = ctxb.reproj_schur_windows(P, col_pose, col_ext, col_td, active=active, damp=damp1)
    # the zero-copy form of the same call: the lower triangle, s, diag and the costs equal the copying form
    ctxb.reproj_reserve_windows(P)
    Sv, sv, dgv, costv = ctxb.reproj_schur_windows_view(P, col_pose, col_ext, col_td, active=active, damp=damp1)
    lower_tiles = np.arange(P)[:, None] >= np.arange(P)[None, :]  # (the view defines rows >= columns only)
    for k in range(W):
        sc = max(1.0, np.abs(S[k]).max())
        assert np.abs(np.where(lower_tiles, Sv[k] - S[k], 0.0)).max() < tol * sc, k
        assert np.abs(sv[k] - s[k]).max() < tol * max(1.0, np.abs(s[k]).max()) and np.abs(dgv[k] - dg[k]).max() < tol * sc, k
        assert abs(costv[k] - cost[k]) < tol * max(1.0, cost[k]), k
    dc = rng.normal(0, 1e-3, (W, P))
    dl, terms = ctxb.reproj_backsub_windows(P, dc, len(inv))
    singles = []
    for k in range(W):
        c1, S1, s1, dg1, cost1 = single(k, wins[k]["poses"], wins[k]["ext"], wins[k]["invdepth"], wins[k]["td"], damp1[k])
        sc = max(1.0, np.abs(S1).max())
        assert np.abs(S[k] - S1).max() < tol * sc and np.abs(s[k] - s1).max() < tol * max(1.0, np.abs(s1).max()), k
        assert np.abs(dg[k] - dg1).max() < tol * sc and abs(cost[k] - cost1) < tol * max(1.0, cost1), k
        dl1, t1 = c1.reproj_backsub(P, dc[k], len(wins[k]["invdepth"]))
        assert np.abs(dl[lm_off[k]:lm_off[k + 1]] - dl1).max() < tol * max(1e-12, np.abs(dl1).max()), k
        assert np.abs(terms[k] - t1).max() < tol * max(1.0, np.abs(t1).max()), k
        singles.append(c1)
    # second round: windows 0 and 3 only get a new damping (rejected step), the others are re-linearized at a moved point
    re = np.array([0, 1, 1, 0, 1], np.uint8)
    damp2 = damp1 * np.array([4.0, 1.0, 0.5, 8.0, 1.0]) + np.array([0, 1e-5, 0, 0, 0])
    inv2 = inv * (1 + 0.01 * rng.normal(0, 1, len(inv)))
    ctxb.reproj_eval_windows(poses, ext, inv2, td, huber=huber)
    costs_now = ctxb.reproj_cost_windows(active)
    S2, s2, dg2, cost2 = ctxb.reproj_schur_windows(P, col_pose, col_ext, col_td, active=active, reassemble=re, damp=damp2)
    for k in range(W):
        cp = col_pose[pose_off[k]:pose_off[k + 1]]
        if re[k]:
            _, S1, s1, dg1, cost1 = single(k, wins[k]["poses"], wins[k]["ext"], inv2[lm_off[k]:lm_off[k + 1]], wins[k]["td"], damp2[k], singles[k])
            assert abs(cost2[k] - cost1) < tol * max(1.0, cost1) and abs(costs_now[k] - cost1) < tol * max(1.0, cost1), k
        else:
            S1, s1, dg1, _ = singles[k].reproj_schur(P, cp, col_ext[k], col_td[k], active=active[fac_off[k]:fac_off[k + 1]], reassemble=False, damp=damp2[k])
        sc = max(1.0, np.abs(S1).max())
        assert np.abs(S2[k] - S1).max() < tol * sc and np.abs(s2[k] - s1).max() < tol * max(1.0, np.abs(s1).max()), (k, re[k])
    for c in singles + [ctxb]:
        c.close()
