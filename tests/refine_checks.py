"""Shared body: map -> optimizer -> map on the windows the tracker built (VisualWindow + WindowSolver + WindowCulling)."""
import ctypes as C

import numpy as np

import ins_utils as iu


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def refine(sb, pose_b_c12, prior_weight=50.0, std=1.5, iters1=6, iters2=18, chi2=5.991, max_kf=16):
    n = sb.n
    out7, kf = np.zeros((n, 7)), np.zeros((n, max_kf, 14))
    err = C.create_string_buffer(512)
    rc = sb.lib.icgh_batch_refine_windows(C.c_void_p(sb.h_), _p(np.ascontiguousarray(pose_b_c12, np.float64)), C.c_double(0.0), C.c_double(std),
                                          C.c_double(prior_weight), iters1, iters2, C.c_double(chi2), _p(out7), max_kf, _p(kf), err, 512)
    assert rc == 0, (rc, err.value)
    return out7, kf


def check_refinement(lib_path, n_streams=3, n_frames=30, engine="table"):
    """tracks synthetic streams whose INS priors carry N(0.02 m, 0.1 deg) noise, then optimizes every stream's sliding window
    (reprojection factors from the map + pose priors at the INS poses) and writes the result back: the keyframe poses must move
    TOWARDS the true camera poses (the visual factors average the independent prior noise down), the cost must drop, and the
    culling pass must leave a usable map"""
    import harness as H
    w, h = 640, 480
    cam = H.camera_for(w, h)
    sb = H.StreamBatch(lib_path, n_streams, w, h, cam, max_features=100, window=10, engine=engine)  # icg::Map itself, or a view of the track table
    scene = H.SynthScene(sb.lib, w, h, cam, tex_size=1024, threads=4)
    t0 = 100.0
    for k in range(n_frames):
        frames = [scene.render(k, stream=s) for s in range(n_streams)]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for s in range(n_streams)])
        sb.step([f.ctypes.data for f in frames], w, np.full(n_streams, t0 + k / 20.0), poses)
    lm_before = [sb.stats(s)["landmarks"] for s in range(n_streams)]
    out7, kf = refine(sb, iu.pose_b_c())
    results = []
    for s in range(n_streams):
        nk = int(out7[s][0])
        assert nk >= 5 and out7[s][1] > 200, out7[s]
        assert out7[s][3] < 0.7 * out7[s][2], out7[s]  # cost after < cost before
        e_prior, e_ref = [], []
        for k in range(nk):
            stamp = kf[s][k][0]
            fk = int(round((stamp - t0) * 20.0))
            Rt, tt = scene.pose(fk, stream=s)
            Ri, ti = scene.ins_pose(fk, stream=s)
            tr = kf[s][k][11:14]
            e_prior.append(np.linalg.norm(ti - tt))
            e_ref.append(np.linalg.norm(tr - tt))
        results.append((np.sqrt(np.mean(np.square(e_prior))), np.sqrt(np.mean(np.square(e_ref)))))
        assert sb.stats(s)["landmarks"] >= 0.5 * lm_before[s], (s, sb.stats(s)["landmarks"], lm_before[s])
    rms_prior = np.sqrt(np.mean([r[0] ** 2 for r in results]))
    rms_ref = np.sqrt(np.mean([r[1] ** 2 for r in results]))
    assert rms_ref < 0.95 * rms_prior, (results, out7)  # weakly observed scale limits the gain; it must not get worse
    # the tracker keeps working on the refined map
    for k in range(n_frames, n_frames + 6):
        frames = [scene.render(k, stream=s) for s in range(n_streams)]
        poses = np.stack([H.pose12(*scene.ins_pose(k, stream=s)) for s in range(n_streams)])
        st = sb.step([f.ctypes.data for f in frames], w, np.full(n_streams, t0 + k / 20.0), poses)
        assert all(int(x) == 2 for x in st), st
    final = [sb.dump(s, 0) for s in range(n_streams)]
    sb.close()
    return results, out7, final
