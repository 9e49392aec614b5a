"""ctypes drivers for the back-end entry points of the host layer (host/capi.cc: icgh_backend_*), plus the shared test
bodies that are run twice: on the oracle-backed host library (CPU, not gpu) and on the product library (gpu)."""
import ctypes as C

import numpy as np

import marg_data as md
import preint_data as pd
import reproj_data as rd


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def backend_reproj(lib, w):
    obs = _f64(w["obs_soa"])
    n = obs.shape[1]
    r, J = np.zeros((n, 2)), np.zeros((n, 46))
    err = C.create_string_buffer(512)
    poses, inv = _f64(w["poses"]), _f64(w["invdepth"])
    rc = lib.icgh_backend_reproj(n, _p(obs), _p(_i32(w["idx_i"])), _p(_i32(w["idx_j"])), _p(_i32(w["idx_lm"])), poses.shape[0],
                                 _p(poses), _p(_f64(w["ext"])), inv.shape[0], _p(inv), C.c_double(w["td"]), _p(r), _p(J), err, 512)
    assert rc == 0, (rc, err.value)
    return r, J


def check_reproj_costfunction_surface(lib, oracle):
    w = rd.make_window(120, 7, seed=11)
    r, J = backend_reproj(lib, w)
    r_exp, J_exp = oracle.reproj_eval(w["obs_soa"], w["idx_i"], w["idx_j"], w["idx_lm"], w["poses"], w["ext"], w["invdepth"], w["td"])
    assert np.abs(r - r_exp).max() < 1e-9 * max(1, np.abs(r_exp).max())
    assert np.abs(J - J_exp).max() < 1e-9 * max(1, np.abs(J_exp).max())


def backend_marginalize(lib, P, huber=1.0, prior_weight=100.0, x_eval=None):
    w = P["w"]
    obs = _f64(P["obs"])
    n = obs.shape[1]
    poses, inv = _f64(w["poses"]), _f64(w["invdepth"])
    K, L = poses.shape[0], inv.shape[0]
    cap = 6 * K + L + 7
    sizes = np.zeros(2, np.int32)
    rem_ids, rem_index, rem_size = np.zeros(K + L + 2, np.int64), np.zeros(K + L + 2, np.int32), np.zeros(K + L + 2, np.int32)
    n_rem = np.zeros(1, np.int32)
    Hp, bp, J0, e0 = np.zeros(cap * cap), np.zeros(cap), np.zeros(cap * cap), np.zeros(cap)
    res = np.zeros(cap)
    err = C.create_string_buffer(512)
    xe = None if x_eval is None else _f64(x_eval)
    rc = lib.icgh_backend_marginalize(n, _p(obs), _p(_i32(P["ii"])), _p(_i32(P["jj"])), _p(_i32(P["ll"])), K, _p(poses), _p(_f64(w["ext"])), L,
                                      _p(inv), C.c_double(w["td"]), C.c_double(huber), C.c_double(prior_weight), 1, 1, _p(sizes),
                                      _p(rem_ids), _p(rem_index), _p(rem_size), _p(n_rem), _p(Hp), _p(bp), _p(J0), _p(e0), _p(xe),
                                      _p(res) if xe is not None else None, err, 512)
    assert rc == 0, (rc, err.value)
    m, r = int(sizes[0]), int(sizes[1])
    nb = int(n_rem[0])
    return dict(m=m, r=r, ids=rem_ids[:nb], index=rem_index[:nb], size=rem_size[:nb], Hp=Hp[:r * r].reshape(r, r), bp=bp[:r],
                J0=J0[:r * r].reshape(r, r), e0=e0[:r], res=res[:r])


def oracle_marginalized_system(oracle, P, w, out, huber=1.0, prior_weight=100.0, scalar_prior=None):
    """(Hp, bp) of the marginalization scenario of capi.cc (reprojection factors + PosePriorFactor on pose 0 [+ a ScalarPriorFactor
    (landmark, x0, weight)]) at the parameter values `w`, from the ORACLE: orc_reproj per factor, orc_marg's constructEquation and Schur
    complement (factors/marginalization_info.h:170-230), permuted from this file's column layout to the library's retained order
    (`out`: ids / index / size / m of a library run of the same structure)."""
    r_, J_ = oracle.reproj_eval(P["obs"], P["ii"], P["jj"], P["ll"], w["poses"], w["ext"], w["invdepth"], w["td"], huber=huber)
    H, b = oracle.reproj_accumulate_normal(r_, J_, P["ii"], P["jj"], P["ll"], P["col_pose"], P["col_ext"], P["col_lm"], P["col_td"], P["local_size"])
    # PosePriorFactor of capi.cc: residual w*[dp ; 2 vec(dq)], prior translated by +0.01 in x -> r = (-w*0.01, 0, ...)
    Jp = np.zeros((6, 6))
    Jp[:3, :3] = prior_weight * np.eye(3)
    Jp[3:, 3:] = prior_weight * np.eye(3)
    rp = np.zeros(6)
    rp[0] = prior_weight * -0.01
    c0 = P["col_pose"][0]
    H[c0:c0 + 6, c0:c0 + 6] += Jp.T @ Jp
    b[c0:c0 + 6] -= Jp.T @ rp
    if scalar_prior is not None:  # ScalarPriorFactor of capi.cc: r = weight (x - x0)
        l, x0, wgt = scalar_prior
        cl = P["col_lm"][l]
        H[cl, cl] += wgt * wgt
        b[cl] -= wgt * (wgt * (w["invdepth"][l] - x0))
    m = P["m"]
    _, _, Hp_o, bp_o = oracle.marginalize(H, b, m)
    # our retained column of each id
    def our_col(i):
        if i < 100000:
            return P["col_pose"][i]
        if i < 900000:
            return P["col_lm"][i - 100000]
        return P["col_ext"] if i == 900000 else P["col_td"]
    r = out["r"]
    Pm = np.zeros((r, r))  # lib index <- our index
    for i, idx, sz in zip(out["ids"], out["index"], out["size"]):
        ls = 6 if sz == 7 else sz
        for k in range(ls):
            Pm[idx - out["m"] + k, our_col(int(i)) - m + k] = 1.0
    assert np.allclose(Pm.sum(0), 1) and np.allclose(Pm.sum(1), 1)
    return Pm @ Hp_o @ Pm.T, Pm @ bp_o


def batch_window_parameters(P, n_windows, jitter=1e-3):
    """the parameter values icgh_backend_marginalize_batch (capi.cc) gives its n_windows copies of problem P: window 0 as it is, every
    further window with positions moved by jitter * u and inverse depths scaled by 1 + jitter * u, u from the 64-bit LCG of capi.cc"""
    state, mask = [0x9E3779B97F4A7C15], (1 << 64) - 1

    def rnd():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) & mask
        return float((state[0] >> 11) & ((1 << 53) - 1)) / float(1 << 52) - 1.0

    out = []
    for k in range(n_windows):
        w = dict(P["w"])
        w["poses"], w["invdepth"] = np.array(P["w"]["poses"], np.float64), np.array(P["w"]["invdepth"], np.float64)
        if k > 0:
            for p in range(w["poses"].shape[0]):
                for c in range(3):
                    w["poses"][p, c] += jitter * rnd()
            for l in range(len(w["invdepth"])):
                w["invdepth"][l] *= 1.0 + jitter * rnd()
        out.append(w)
    return out


def check_marginalization(lib, oracle):
    P = md.make_problem(n_lm=80, n_kf=6, seed=2)
    w = P["w"]
    out = backend_marginalize(lib, P, huber=1.0, prior_weight=100.0)
    m = P["m"]
    assert out["m"] == m and out["r"] == P["local_size"] - m
    Hp_exp, bp_exp = oracle_marginalized_system(oracle, P, w, out)
    scale = np.abs(Hp_exp).max()
    assert np.abs(out["Hp"] - Hp_exp).max() < 1e-8 * scale
    assert np.abs(out["bp"] - bp_exp).max() < 1e-8 * max(1.0, np.abs(bp_exp).max())
    # linearization identities on the library's own J0/e0
    ev, V = np.linalg.eigh(0.5 * (out["Hp"] + out["Hp"].T))
    Hp_trunc = V @ np.diag(np.where(ev > 1e-8, ev, 0)) @ V.T
    assert np.abs(out["J0"].T @ out["J0"] - Hp_trunc).max() < 1e-6 * scale
    proj = V @ np.diag((ev > 1e-8).astype(float)) @ V.T
    assert np.abs(out["J0"].T @ out["e0"] + proj @ out["bp"]).max() < 1e-6 * max(1.0, np.abs(out["bp"]).max())
    # MarginalizationFactor::Evaluate at a perturbed point == oracle's evaluation with the same J0/e0
    rng = np.random.RandomState(5)
    x0, x = [], []
    for i, sz in zip(out["ids"], out["size"]):
        i = int(i)
        if i < 100000:
            base = w["poses"][i]
            x0.append(base)
            x.append(rd.pose_plus(base, rng.normal(0, 1e-3, 6)))
        elif i < 900000:
            base = np.array([w["invdepth"][i - 100000]])
            x0.append(base)
            x.append(base + rng.normal(0, 1e-4, 1))
        elif i == 900000:
            x0.append(w["ext"])
            x.append(rd.pose_plus(w["ext"], rng.normal(0, 1e-3, 6)))
        else:
            x0.append(np.array([w["td"]]))
            x.append(np.array([w["td"] + 1e-4]))
    out2 = backend_marginalize(lib, P, huber=1.0, prior_weight=100.0, x_eval=np.concatenate(x))
    res_exp, _ = oracle.marg_factor_eval(out["size"], out["index"] - out["m"], np.concatenate(x0), np.concatenate(x), out2["J0"], out2["e0"],
                                         want_jac=False)
    assert np.abs(out2["res"] - res_exp).max() < 1e-9 * max(1.0, np.abs(res_exp).max())


def check_preintegration(lib, oracle):
    for variant in (0, 1):
        lens = [41, 17, 60]
        imus = [pd.make_interval(n, seed=20 + i) for i, n in enumerate(lens)]
        states = [pd.state(p=(i, 1, 0), v=(2, 0.1 * i, 0)) for i in range(len(lens))]
        offsets = np.cumsum([0] + lens).astype(np.int32)
        pres = [oracle.preint_integrate(variant, imus[i], states[i], pd.PARAMS) for i in range(len(lens))]
        evalp = []
        for i in range(len(lens)):
            pose0, mix0 = pd.split(states[i])
            pose1, mix1 = pd.split(pres[i]["cur"])
            pose1 = rd.pose_plus(pose1, np.array([0.01, -0.02, 0.01, 0.001, 0.002, -0.001]))
            evalp.append(np.concatenate([pose0, mix0, pose1, mix1]))
        n = len(lens)
        cur, res, jac = np.zeros((n, 16)), np.zeros((n, 15)), np.zeros((n, 480))
        err = C.create_string_buffer(512)
        imu = _f64(np.concatenate(imus))
        rc = lib.icgh_backend_preint(variant, n, _p(offsets), _p(imu), _p(_f64(np.stack(states))), _p(_f64(pd.PARAMS)),
                                     _p(_f64(np.stack(evalp))), _p(cur), _p(res), _p(jac), err, 512)
        assert rc == 0, (rc, err.value)
        for i in range(n):
            ep = evalp[i]
            r_exp, J_exp = oracle.preint_evaluate(variant, pres[i], [0, 0, pd.PARAMS[5]], pd.PARAMS[6:9], ep[:7], ep[7:16], ep[16:23], ep[23:32])
            assert np.abs(cur[i] - pres[i]["cur"]).max() < 1e-9 * np.abs(pres[i]["cur"]).max()
            J_cat = np.concatenate([J_exp[0].ravel(), J_exp[1].ravel(), J_exp[2].ravel(), J_exp[3].ravel()])
            # whitening by the inverse covariance amplifies libm-level differences of the integration: 1e-6 relative
            assert np.abs(res[i] - r_exp).max() < 1e-6 * max(1.0, np.abs(r_exp).max())
            assert np.abs(jac[i] - J_cat).max() < 1e-6 * max(1.0, np.abs(J_cat).max())


def check_preintegration_golden(lib):
    """icg::Preintegration + PreintegrationFactor (device P1 + host P2) against outputs of the REFERENCE's own code
    (tests/golden/preint_ref_golden.npz): current state 1e-9, whitened residual / Jacobians 1e-6 relative."""
    from test_oracle_vs_reference import preint_golden_cases
    for k, c in preint_golden_cases():
        variant = int(c["variant"])
        offsets = np.array([0, len(c["imu"])], np.int32)
        pose0, mix0 = pd.split(c["s0"])
        pose1, mix1 = pd.split(c["s1"])
        ep = np.concatenate([pose0, mix0, pose1, mix1])
        cur, res, jac = np.zeros((1, 16)), np.zeros((1, 15)), np.zeros((1, 480))
        err = C.create_string_buffer(512)
        rc = lib.icgh_backend_preint(variant, 1, _p(offsets), _p(_f64(c["imu"])), _p(_f64(c["s0"][None, :])), _p(_f64(c["params"])),
                                     _p(_f64(ep[None, :])), _p(cur), _p(res), _p(jac), err, 512)
        assert rc == 0, (rc, err.value)
        assert np.abs(cur[0] - c["cur"]).max() < 1e-9 * np.abs(c["cur"]).max(), k
        assert np.abs(res[0] - c["r"]).max() < 1e-6 * max(1.0, np.abs(c["r"]).max()), k
        assert np.abs(jac[0] - c["J"]).max() < 1e-6 * max(1.0, np.abs(c["J"]).max()), k


# ---- marginalization against the REFERENCE's own pipeline (golden) -------------------------------------------------------
MARG_GOLDEN_ARGS = dict(n_lm=80, n_kf=6, seed=2)


def _marg_perturbation(i, w):
    """deterministic evaluation point per parameter id (independent of the library's block order)"""
    rng = np.random.RandomState(1000 + int(i) % 100003)
    if i < 100000:
        return rd.pose_plus(w["poses"][i], rng.normal(0, 1e-3, 6))
    if i < 900000:
        return np.array([w["invdepth"][i - 100000]]) + rng.normal(0, 1e-4, 1)
    if i == 900000:
        return rd.pose_plus(w["ext"], rng.normal(0, 1e-3, 6))
    return np.array([w["td"] + 1e-4])


def marginalize_canonical(lib):
    """Runs the standard scenario through `lib` and returns quantities that do not depend on the library's block order or
    on the sign / basis choices of its eigen-solver: Hp, bp in id-sorted column order and, for the marginalization factor at
    a fixed perturbed point, the cost |e|^2 and the gradient J0^T e (id-sorted)."""
    P = md.make_problem(**MARG_GOLDEN_ARGS)
    w = P["w"]
    out = backend_marginalize(lib, P, huber=1.0, prior_weight=100.0)
    x = np.concatenate([_marg_perturbation(int(i), w) for i in out["ids"]])
    out2 = backend_marginalize(lib, P, huber=1.0, prior_weight=100.0, x_eval=x)
    order = np.argsort(out["ids"])
    cols = []
    for k in order:
        ls = 6 if out["size"][k] == 7 else int(out["size"][k])
        c0 = int(out["index"][k] - out["m"])
        cols.extend(range(c0, c0 + ls))
    cols = np.array(cols)
    e = out2["res"]
    return dict(m=out["m"], r=out["r"], ids=np.sort(out["ids"]), Hp=out["Hp"][np.ix_(cols, cols)], bp=out["bp"][cols],
                cost=float(e @ e), grad=(out2["J0"].T @ e)[cols])


def check_marginalization_golden(lib, path):
    g = np.load(path)
    c = marginalize_canonical(lib)
    assert c["m"] == int(g["m"]) and c["r"] == int(g["r"]) and np.array_equal(c["ids"], g["ids"])
    scale = np.abs(g["Hp"]).max()
    assert np.abs(c["Hp"] - g["Hp"]).max() < 1e-8 * scale
    assert np.abs(c["bp"] - g["bp"]).max() < 1e-8 * max(1.0, np.abs(g["bp"]).max())
    assert abs(c["cost"] - float(g["cost"])) < 1e-8 * max(1.0, float(g["cost"]))
    assert np.abs(c["grad"] - g["grad"]).max() < 1e-7 * max(1.0, np.abs(g["grad"]).max())


def backend_marginalize_batch(lib, P, n_windows, mode, dense_window=-1, jitter=1e-3, huber=1.0, prior_weight=100.0, host_threads=0, reps=1):
    """icgh_backend_marginalize_batch (capi.cc): the marginalizations of n_windows jittered copies of problem P, mode 0 = one
    MarginalizationBatch, mode 1 = one MarginalizationInfo::marginalization() after the other; reps: the set is marginalized that many times on
    the same batch object (seconds = the fastest repetition, outputs of the last)."""
    w = P["w"]
    obs = _f64(P["obs"])
    n = obs.shape[1]
    poses, inv = _f64(w["poses"]), _f64(w["invdepth"])
    K, L = poses.shape[0], inv.shape[0]
    cap = 6 * K + L + 7
    sizes, counts, seconds = np.zeros(2, np.int32), np.zeros(2, np.int32), np.zeros(1)
    Hp, bp, J0, e0 = (np.zeros(n_windows * cap * cap), np.zeros(n_windows * cap), np.zeros(n_windows * cap * cap), np.zeros(n_windows * cap))
    err = C.create_string_buffer(512)
    rc = lib.icgh_backend_marginalize_batch(mode, n_windows, dense_window, C.c_double(jitter), reps, n, _p(obs), _p(_i32(P["ii"])), _p(_i32(P["jj"])),
                                            _p(_i32(P["ll"])), K, _p(poses), _p(_f64(w["ext"])), L, _p(inv), C.c_double(w["td"]),
                                            C.c_double(huber), C.c_double(prior_weight), host_threads, _p(sizes), _p(Hp), _p(bp), _p(J0), _p(e0),
                                            _p(counts), _p(seconds), err, 512)
    assert rc == 0, (rc, err.value)
    r = int(sizes[1])
    W = n_windows
    return dict(m=int(sizes[0]), r=r, Hp=Hp[:W * r * r].reshape(W, r, r), bp=bp[:W * r].reshape(W, r), J0=J0[:W * r * r].reshape(W, r, r),
                e0=e0[:W * r].reshape(W, r), structured=int(counts[0]), dense=int(counts[1]), seconds=float(seconds[0]))


def check_marginalization_batch(lib, oracle=None, bitwise=False):
    """M2 + M3 for the windows of many streams in one pass (host/marg_batch.h): every window's Schur complement and
    linearization equal what MarginalizationInfo::marginalization() gives the same window on its own (marginalization_info.h:73-101) —
    all windows on the landmark-eliminated path; one window with a host factor on an inverse depth (dense M2 + M3 for that window only);
    the process-wide dense switch; window 0 (no jitter) equals the single-window entry point the reference-code golden is checked on.
    oracle: EVERY window of the batch is additionally checked against the oracle's own assembly + Schur complement of that window's
    (jittered) parameters — not only against the library's per-window path.
    bitwise: batch and per-window path agree bit for bit (the device assembles in a fixed order since round 6; the CPU backend always did)."""
    for (n_lm, n_kf, seed), W in (((80, 6, 2), 5), ((300, 10, 3), 12), ((40, 4, 7), 1)):
        P = md.make_problem(n_lm=n_lm, n_kf=n_kf, seed=seed)
        layout = backend_marginalize(lib, P) if oracle is not None else None
        params = batch_window_parameters(P, W) if oracle is not None else None
        for dense_window in (-1, min(2, W - 1)):
            a = backend_marginalize_batch(lib, P, W, 0, dense_window)
            b = backend_marginalize_batch(lib, P, W, 1, dense_window)
            assert a["m"] == b["m"] and a["r"] == b["r"]
            assert (a["structured"], a["dense"]) == (b["structured"], b["dense"]) == ((W, 0) if dense_window < 0 else (W - 1, 1))
            if bitwise:
                assert np.array_equal(a["Hp"], b["Hp"]) and np.array_equal(a["bp"], b["bp"]), (W, dense_window)
            for k in range(W):
                if oracle is not None:
                    assert layout["r"] == a["r"] and layout["m"] == a["m"]
                    sp = None
                    if k == dense_window:
                        l0 = int(P["ll"][0])
                        sp = (l0, params[k]["invdepth"][l0] * 1.01, 50.0)
                    Hp_exp, bp_exp = oracle_marginalized_system(oracle, P, params[k], layout, scalar_prior=sp)
                    sc = np.abs(Hp_exp).max()
                    for got in (a, b):
                        assert np.abs(got["Hp"][k] - Hp_exp).max() < 1e-8 * sc, (W, dense_window, k, np.abs(got["Hp"][k] - Hp_exp).max() / sc)
                        assert np.abs(got["bp"][k] - bp_exp).max() < 1e-8 * max(1.0, np.abs(bp_exp).max()), (W, dense_window, k)
                scale = np.abs(b["Hp"][k]).max()
                assert np.abs(a["Hp"][k] - b["Hp"][k]).max() < 1e-9 * scale, (k, np.abs(a["Hp"][k] - b["Hp"][k]).max() / scale)
                assert np.abs(a["bp"][k] - b["bp"][k]).max() < 1e-9 * max(1.0, np.abs(b["bp"][k]).max())
                # the prior itself (the eigenbasis of a linearization is free in degenerate subspaces): J0^T J0 and J0^T e0
                assert np.abs(a["J0"][k].T @ a["J0"][k] - b["J0"][k].T @ b["J0"][k]).max() < 1e-7 * scale
                assert np.abs(a["J0"][k].T @ a["e0"][k] - b["J0"][k].T @ b["e0"][k]).max() < 1e-7 * max(1.0, np.abs(b["bp"][k]).max())
            if W > 1:  # the windows are different problems (the jitter moved them)
                assert np.abs(a["Hp"][1] - a["Hp"][0]).max() > 1e-6 * np.abs(a["Hp"][0]).max()
            if dense_window != 0:  # (window 0 as the single-window entry point builds it: no extra host factor)
                single = backend_marginalize(lib, P)
                assert single["r"] == a["r"]
                assert np.abs(a["Hp"][0] - single["Hp"]).max() < 1e-9 * np.abs(single["Hp"]).max()
                assert np.abs(a["bp"][0] - single["bp"]).max() < 1e-9 * max(1.0, np.abs(single["bp"]).max())
    P = md.make_problem(n_lm=80, n_kf=6, seed=2)
    lib.icgh_backend_marginalization_force_dense(1)
    try:
        a = backend_marginalize_batch(lib, P, 4, 0)
        b = backend_marginalize_batch(lib, P, 4, 1)
    finally:
        lib.icgh_backend_marginalization_force_dense(0)
    assert (a["structured"], a["dense"]) == (0, 4) == (b["structured"], b["dense"])
    c = backend_marginalize_batch(lib, P, 4, 0)
    for k in range(4):
        scale = np.abs(b["Hp"][k]).max()
        assert np.abs(a["Hp"][k] - b["Hp"][k]).max() < 1e-9 * scale
        assert np.abs(a["Hp"][k] - c["Hp"][k]).max() < 1e-9 * scale  # dense == landmark-eliminated, as for a window on its own
        assert np.abs(a["bp"][k] - c["bp"][k]).max() < 1e-9 * max(1.0, np.abs(c["bp"][k]).max())
    # a batch object re-used for the next set of windows (clear() keeps the device context and its buffers)
    f = backend_marginalize_batch(lib, P, 4, 0, reps=3)
    assert np.abs(f["Hp"] - c["Hp"]).max() < 1e-9 * np.abs(c["Hp"]).max() and (f["structured"], f["dense"]) == (4, 0)
    # 15-keyframe windows (BASELINE configs[3]: 15 x 6 + 7 = 97 free camera columns = a 76 KB LDS tile; rounds 2-4 capped the tile at 64 KB /
    # 82 columns and sent such windows down the dense path) are batched on gfx950's 160 KiB of LDS: same prior as each window alone
    P15 = md.make_problem(n_lm=120, n_kf=15, seed=9)
    g = backend_marginalize_batch(lib, P15, 3, 0)
    h = backend_marginalize_batch(lib, P15, 3, 1)
    assert (g["structured"], g["dense"]) == (3, 0) and (h["structured"], h["dense"]) == (3, 0)
    for k in range(3):
        scale = np.abs(h["Hp"][k]).max()
        assert np.abs(g["Hp"][k] - h["Hp"][k]).max() < 1e-9 * scale and np.abs(g["bp"][k] - h["bp"][k]).max() < 1e-9 * max(1.0, np.abs(h["bp"][k]).max())
    # host threads: the per-window phases spread over the pool give the same numbers as one thread
    d = backend_marginalize_batch(lib, P, 9, 0, host_threads=1)
    e = backend_marginalize_batch(lib, P, 9, 0, host_threads=4)
    # (bit-equal on the CPU backend; the device assembles with floating-point atomics, whose order differs from launch to launch)
    assert np.abs(d["Hp"] - e["Hp"]).max() < 1e-9 * np.abs(d["Hp"]).max() and np.abs(d["bp"] - e["bp"]).max() < 1e-9 * max(1.0, np.abs(d["bp"]).max())


def check_marginalization_paths(lib):
    """M2 + M3: the landmark-eliminated path (device assembly + elimination of the 1x1 inverse-depth blocks, small host finish) against
    the reference's dense construction + pseudo-inverse (marginalization_info.h:170-230) on the same problems: same Hp, bp, cost at
    a perturbed point; and the conditioning guard — a landmark with (numerically) no information sends the call down the dense path."""
    lib.icgh_backend_marginalization_structured.restype = C.c_int
    for n_lm, n_kf, seed in ((80, 6, 2), (300, 10, 3), (40, 4, 7)):
        P = md.make_problem(n_lm=n_lm, n_kf=n_kf, seed=seed)
        lib.icgh_backend_marginalization_force_dense(0)
        a = backend_marginalize(lib, P)
        assert lib.icgh_backend_marginalization_structured() == 1
        lib.icgh_backend_marginalization_force_dense(1)
        b = backend_marginalize(lib, P)
        assert lib.icgh_backend_marginalization_structured() == 0
        lib.icgh_backend_marginalization_force_dense(0)
        assert a["m"] == b["m"] and a["r"] == b["r"] and np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["index"], b["index"])
        scale = np.abs(b["Hp"]).max()
        assert np.abs(a["Hp"] - b["Hp"]).max() < 1e-9 * scale, np.abs(a["Hp"] - b["Hp"]).max() / scale
        assert np.abs(a["bp"] - b["bp"]).max() < 1e-9 * max(1.0, np.abs(b["bp"]).max())
        # the prior itself (basis independent): J0^T J0 and J0^T e0
        assert np.abs(a["J0"].T @ a["J0"] - b["J0"].T @ b["J0"]).max() < 1e-7 * scale
        assert np.abs(a["J0"].T @ a["e0"] - b["J0"].T @ b["e0"]).max() < 1e-7 * max(1.0, np.abs(b["bp"]).max())
    # guard: one landmark observed with an enormous standard deviation -> h_ll far below the floor -> dense path, same numbers as forced dense
    P = md.make_problem(n_lm=60, n_kf=5, seed=4)
    weak = P["ll"] == P["ll"][0]
    obs = P["obs"].copy()
    obs[14, weak] = 1e9  # observation row 14 = std (pixel error / focal length), see icg_reproj_set_factors
    P2 = dict(P, obs=obs)
    a = backend_marginalize(lib, P2)
    assert lib.icgh_backend_marginalization_structured() == 0, "the conditioning guard did not trigger"
    lib.icgh_backend_marginalization_force_dense(1)
    b = backend_marginalize(lib, P2)
    lib.icgh_backend_marginalization_force_dense(0)
    # (same path twice: equal up to the summation order of the device's FP64 atomics)
    sc = np.abs(b["Hp"]).max()
    assert np.abs(a["Hp"] - b["Hp"]).max() <= 1e-12 * sc and np.abs(a["bp"] - b["bp"]).max() <= 1e-12 * max(1.0, np.abs(b["bp"]).max())
