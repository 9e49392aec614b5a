"""Known-answer tests pinning the ORACLE's marginalization restatement: eigen-solver vs numpy, normal-equation assembly
vs a dense numpy J^T J, the identities J0^T J0 = Hp and J0^T e0 = -bp, and the prior factor."""
import numpy as np

import marg_data as md
import reproj_data as rd


def test_sym_eigen_matches_numpy(oracle):
    rng = np.random.RandomState(0)
    for n in (1, 2, 7, 45, 157):
        A = rng.normal(size=(n, n))
        A = A @ A.T + np.diag(rng.uniform(0, 1e-3, n))
        ev, V = oracle.sym_eigen(A)
        ref = np.linalg.eigvalsh(A)
        assert np.allclose(ev, ref, rtol=1e-10, atol=1e-10 * ref.max())
        assert np.allclose(V @ np.diag(ev) @ V.T, A, atol=1e-9 * np.abs(A).max())
        assert np.allclose(V.T @ V, np.eye(n), atol=1e-10)


def test_normal_equations_match_dense_numpy(oracle):
    P = md.make_problem()
    w = P["w"]
    r, J = oracle.reproj_eval(P["obs"], P["ii"], P["jj"], P["ll"], w["poses"], w["ext"], w["invdepth"], w["td"], huber=1.0)
    H, b = oracle.reproj_accumulate_normal(r, J, P["ii"], P["jj"], P["ll"], P["col_pose"], P["col_ext"], P["col_lm"], P["col_td"], P["local_size"])
    n, L = len(r), P["local_size"]
    D = np.zeros((2 * n, L))
    for f in range(n):
        blocks = [(P["col_pose"][P["ii"][f]], J[f, 0:14].reshape(2, 7)[:, :6]), (P["col_pose"][P["jj"][f]], J[f, 14:28].reshape(2, 7)[:, :6]),
                  (P["col_ext"], J[f, 28:42].reshape(2, 7)[:, :6]), (P["col_lm"][P["ll"][f]], J[f, 42:44].reshape(2, 1)),
                  (P["col_td"], J[f, 44:46].reshape(2, 1))]
        for c0, B in blocks:
            if c0 >= 0:
                D[2 * f:2 * f + 2, c0:c0 + B.shape[1]] += B
    assert np.allclose(H, D.T @ D, rtol=1e-10, atol=1e-8)
    assert np.allclose(b, -D.T @ r.ravel(), rtol=1e-10, atol=1e-8)
    assert np.allclose(H, H.T)


def test_marginalization_identities(oracle):
    P = md.make_problem()
    w = P["w"]
    r, J = oracle.reproj_eval(P["obs"], P["ii"], P["jj"], P["ll"], w["poses"], w["ext"], w["invdepth"], w["td"], huber=1.0)
    H, b = oracle.reproj_accumulate_normal(r, J, P["ii"], P["jj"], P["ll"], P["col_pose"], P["col_ext"], P["col_lm"], P["col_td"], P["local_size"])
    # a prior keeps the marginalized pose observable, like the IMU/prior factors do in the real window
    H[:6, :6] += np.eye(6) * 1e4
    J0, e0, Hp, bp = oracle.marginalize(H, b, P["m"])
    m = P["m"]
    Hmm, Hmr, Hrr = H[:m, :m], H[:m, m:], H[m:, m:]
    ev, V = np.linalg.eigh(0.5 * (Hmm + Hmm.T))
    inv = V @ np.diag(np.where(ev > 1e-8, 1 / np.where(ev > 1e-8, ev, 1), 0)) @ V.T
    assert np.allclose(Hp, Hrr - Hmr.T @ inv @ Hmr, rtol=1e-8, atol=1e-6)
    assert np.allclose(bp, b[m:] - Hmr.T @ inv @ b[:m], rtol=1e-8, atol=1e-6)
    # J0^T J0 = Hp on the retained eigen-space (eigenvalues <= 1e-8 are dropped by construction)
    evp, Vp = np.linalg.eigh(0.5 * (Hp + Hp.T))
    Hp_trunc = Vp @ np.diag(np.where(evp > 1e-8, evp, 0)) @ Vp.T
    assert np.allclose(J0.T @ J0, Hp_trunc, rtol=1e-7, atol=1e-6 * np.abs(Hp).max())
    proj = Vp @ np.diag((evp > 1e-8).astype(float)) @ Vp.T
    assert np.allclose(J0.T @ e0, -proj @ bp, rtol=1e-6, atol=1e-6 * np.abs(bp).max())


def test_marginalization_factor_evaluate(oracle):
    rng = np.random.RandomState(3)
    sizes = [7, 9, 7, 9, 7, 1]              # pose, mix, pose, mix, extrinsic, td (global sizes)
    local = [6 if s == 7 else s for s in sizes]
    index = np.concatenate([[0], np.cumsum(local)[:-1]]).astype(np.int32)
    r = int(np.sum(local))
    J0 = rng.normal(size=(r, r))
    e0 = rng.normal(size=r)
    x0 = []
    for s in sizes:
        if s == 7:
            q = rd.quat_from_rotvec(rng.normal(0, 0.3, 3))
            x0.append(np.concatenate([rng.normal(size=3), q]))
        else:
            x0.append(rng.normal(size=s))
    x0c = np.concatenate(x0)
    res, jac = oracle.marg_factor_eval(sizes, index, x0c, x0c, J0, e0)
    assert np.allclose(res, e0)  # dx = 0 at the linearization point
    # perturb: residual = e0 + J0 dx with dx = (dp, rotvec) for poses (first order)
    d = rng.normal(0, 1e-4, r)
    x = []
    for s, i, xb in zip(sizes, index, x0):
        if s == 7:
            x.append(rd.pose_plus(xb, d[i:i + 6]))
        else:
            x.append(xb + d[i:i + s])
    res2, jac2 = oracle.marg_factor_eval(sizes, index, x0c, np.concatenate(x), J0, e0)
    assert np.allclose(res2, e0 + J0 @ d, atol=1e-9)
    # Jacobian blocks: columns of J0, zero-padded 7th pose column
    off = 0
    for s, i in zip(sizes, index):
        B = jac[off:off + r * s].reshape(r, s)
        ls = 6 if s == 7 else s
        assert np.array_equal(B[:, :ls], J0[:, i:i + ls])
        if s == 7:
            assert np.all(B[:, 6] == 0)
        off += r * s
    # q and -q describe the same rotation (sign fix at marginalization_factor.h:68-72)
    xneg = [xb.copy() for xb in x]
    xneg[0][3:] *= -1
    res3, _ = oracle.marg_factor_eval(sizes, index, x0c, np.concatenate(xneg), J0, e0, want_jac=False)
    assert np.allclose(res3, res2, atol=1e-12)
