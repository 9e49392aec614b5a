"""GPU parity: detection (F7), fundamental-matrix RANSAC (F6) and triangulation (F8) vs the CPU oracle.
Index-like outputs (inlier masks, corner sets and their order, block ids) must be bit-exact; corner coordinates and
triangulated points are also required bit-exact here because both sides use the same IEEE operation order."""
import numpy as np
import pytest

import synth
from test_oracle_geometry import two_view

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx1280():
    import icgvins
    c = icgvins.Context(1280, 720, n_slots=3, max_batch=3, max_points=8192)
    c.set_camera(synth.CAM_1280)
    yield c
    c.close()


def grid_for(w, h, max_features):
    """Tracking ctor arithmetic, tracking.cc:66-85."""
    lround = lambda v: int(np.floor(v + 0.5))  # C lround/round: half away from zero (positive arguments here)
    bc, br = lround(w / 200.0), lround(h / 200.0)
    bw, bh = w // bc, h // br
    per = lround(max_features / (bc * br))
    md = lround(200.0 / np.sqrt(per * 1.5))
    return [bc, br, bw, bh, md, per]


def test_grid_arithmetic_matches_survey():
    assert grid_for(1280, 720, 300) == [6, 4, 213, 180, 45, 13]
    assert grid_for(640, 480, 100) == [3, 2, 213, 240, 40, 17]
    assert grid_for(1920, 1080, 500) == [10, 5, 192, 216, 52, 10]


@pytest.mark.parametrize("with_mask", [False, True])
def test_detect_matches_oracle_c2(oracle, ctx1280, with_mask):
    w, h = 1280, 720
    imgs = [synth.texture(w, h, seed=80), synth.texture(w, h, seed=81)]
    ctx1280.preprocess([0, 2], imgs)
    grid = grid_for(w, h, 300)
    nblk = grid[0] * grid[1]
    rng = np.random.RandomState(5)
    quotas, masks = [], []
    for j in range(2):
        q = np.full(nblk, grid[5], np.int32)
        if with_mask:
            q -= rng.randint(0, grid[5] + 2, nblk)  # some blocks full (<=0), some partially filled
            masks.append(synth.random_points(120, w, h, 0, seed=90 + j))
        else:
            masks.append(np.zeros((0, 2), np.float32))
        quotas.append(q)
    mask_off = np.cumsum([0] + [len(m) for m in masks]).astype(np.int32)
    out, cnt, blk = ctx1280.detect([0, 2], grid, mask_off, np.concatenate(masks), np.concatenate(quotas), 400)
    for j, img in enumerate(imgs):
        exp_pts, exp_blk = oracle.detect(oracle.clahe(img), grid, masks[j], quotas[j], 400)
        assert cnt[j] == len(exp_pts) and cnt[j] > 50
        assert np.array_equal(blk[j, :cnt[j]], exp_blk)
        assert np.array_equal(out[j, :cnt[j]].view(np.uint32), exp_pts.view(np.uint32))


def test_detect_small_image_and_last_block(oracle):
    import icgvins
    w, h = 640, 480
    c = icgvins.Context(w, h, n_slots=1, max_batch=1, max_points=64)
    try:
        img = synth.texture(w, h, seed=82)
        c.preprocess([0], [img])
        grid = grid_for(w, h, 100)
        q = np.full(6, grid[5], np.int32)
        out, cnt, blk = c.detect([0], grid, [0, 0], np.zeros((0, 2)), q, 200)
        exp_pts, exp_blk = oracle.detect(oracle.clahe(img), grid, np.zeros((0, 2)), q, 200)
        assert cnt[0] == len(exp_pts)
        assert np.array_equal(blk[0, :cnt[0]], exp_blk)
        assert np.array_equal(out[0, :cnt[0]].view(np.uint32), exp_pts.view(np.uint32))
        # all quotas <= 0 -> nothing
        out, cnt, _ = c.detect([0], grid, [0, 0], np.zeros((0, 2)), np.zeros(6, np.int32), 200)
        assert cnt[0] == 0
    finally:
        c.close()


def test_fm_ransac_matches_oracle(oracle, ctx1280):
    sets = []
    for seed, n, frac in ((1, 200, 0.25), (2, 60, 0.4), (3, 15, 0.0), (4, 10, 0.0), (5, 300, 0.6), (6, 120, 0.1)):
        p1, p2, _, _ = two_view(n, seed=seed, outlier_frac=frac, noise=0.3)
        sets.append((p1, p2))
    offsets = np.cumsum([0] + [len(s[0]) for s in sets]).astype(np.int32)
    mask = ctx1280.fm_ransac(offsets, np.concatenate([s[0] for s in sets]), np.concatenate([s[1] for s in sets]))
    for k, (p1, p2) in enumerate(sets):
        got = mask[offsets[k]:offsets[k + 1]]
        if len(p1) < 15:
            assert np.all(got == 1)  # the reference skips RANSAC below 15 points (tracking.cc:547)
            continue
        ok, exp, _, iters = oracle.fm_ransac(p1, p2)
        assert ok == 1
        assert np.array_equal(got, exp), (k, iters, got.sum(), exp.sum())


def test_fm_ransac_degenerate_all_identical(oracle, ctx1280):
    # all correspondences identical: the 7-point system is rank deficient; must not crash and must agree
    p = np.tile(np.array([[100.0, 200.0]], np.float32), (20, 1))
    mask = ctx1280.fm_ransac([0, 20], p, p)
    ok, exp, _, _ = oracle.fm_ransac(p, p)
    assert np.array_equal(mask, exp)


def test_fm_ransac_device_loop_equals_host_loop_and_oracle(oracle, ctx1280):
    """icg_fm_ransac_device (k_fm_ransac_sets: subset draws from cv::RNG, seven-point solves, scoring and the best / niters recurrence of a
    whole RANSAC run inside one workgroup — the kernel the device-resident tracker launches on its segments) against the host-looped entry
    point and the oracle: identical inlier masks for clean and contaminated sets, many rounds (60 % outliers: niters stays in the hundreds),
    sets below 15 points (left untouched), the degenerate all-identical set and a collinear lattice (checkSubset's RNG-consuming redraws);
    the iteration counts come from log / pow values tabulated by the host's libm (fm_denom_table)."""
    sets = []
    for seed, n, frac in ((11, 200, 0.25), (12, 60, 0.4), (13, 15, 0.0), (14, 10, 0.0), (15, 300, 0.6), (16, 120, 0.1), (17, 640, 0.5), (18, 33, 0.7),
                          (19, 64, 0.0), (20, 65, 0.3)):
        p1, p2, _, _ = two_view(n, seed=seed, outlier_frac=frac, noise=0.3)
        sets.append((p1, p2))
    same = np.tile(np.array([[100.0, 200.0]], np.float32), (20, 1))
    sets.append((same, same))
    gx, gy = np.meshgrid(np.arange(6, dtype=np.float32) * 40 + 50, np.arange(5, dtype=np.float32) * 40 + 60)
    lat = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    sets.append((lat, lat + np.float32([3.0, -2.0])))
    offsets = np.cumsum([0] + [len(s[0]) for s in sets]).astype(np.int32)
    a, b = np.concatenate([s[0] for s in sets]), np.concatenate([s[1] for s in sets])
    dev = ctx1280.fm_ransac_device(offsets, a, b)
    host = ctx1280.fm_ransac(offsets, a, b)
    assert np.array_equal(dev, host)
    for k, (p1, p2) in enumerate(sets):
        got = dev[offsets[k]:offsets[k + 1]]
        if len(p1) < 15:
            assert np.all(got == 1)
            continue
        ok, exp, _, iters = oracle.fm_ransac(p1, p2)
        assert np.array_equal(got, exp), (k, iters, got.sum(), exp.sum())


def test_triangulate_matches_oracle(oracle, ctx1280):
    _, _, _, (K, R, t, X) = two_view(300, seed=7)
    T0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    T1 = np.hstack([R, t[:, None]])
    pc0 = X / X[:, 2:]
    c1 = (R @ X.T).T + t
    pc1 = c1 / c1[:, 2:]
    rng = np.random.RandomState(0)
    pc1[:, :2] += rng.normal(0, 1e-3, (300, 2))
    got = ctx1280.triangulate(0, 1, np.stack([T0.ravel(), T1.ravel()]), pc0, pc1)
    exp = np.stack([oracle.triangulate(T0, T1, a, b) for a, b in zip(pc0, pc1)])
    assert np.array_equal(got, exp)
    assert np.median(np.abs(got - X)) < 0.5  # ~1 px of noise at 6-40 m depth


def test_detect_disc_list_mask(oracle):
    """The detection mask is never written as a plane (round 4): every wave of k_min_eig_nms tests its tile against the list of disc
    centres.  Results must equal the CPU restatement (which draws cv::circle into a byte mask) for masks that change every call: centres
    on half-pixel ties (cvRound: ties to even), outside the image, on the image border, clustered inside one tile, more than 64 centres
    (several ballot chunks), and two jobs with different lists in one call (mask_off)."""
    import icgvins
    w, h = 640, 480
    c = icgvins.Context(w, h, n_slots=2, max_batch=2, max_points=256)
    try:
        img0, img1 = synth.texture(w, h, seed=83), synth.texture(w, h, seed=183)
        c.preprocess([0, 1], [img0, img1])
        clahe = [oracle.clahe(img0), oracle.clahe(img1)]
        grid = grid_for(w, h, 100)
        q = np.full(6, grid[5], np.int32)
        rng = np.random.RandomState(5)

        def one_mask(kind):
            if kind == 0:
                return rng.uniform([20, 20], [w - 20, h - 20], (3 + rng.randint(5), 2))
            if kind == 1:  # ties and borders
                return np.array([[100.5, 50.5], [101.5, 51.5], [0.0, 0.0], [w - 1, h - 1], [w - 0.5, 200.5], [320.5, 0.49]])
            if kind == 2:  # outside the image, reaching in (and not)
                return np.array([[-30.0, 100.0], [w + 25.0, 300.0], [200.0, -39.0], [250.0, h + 38.9], [-500.0, -500.0], [w + 41.0, 20.0]])
            if kind == 3:  # a cluster inside one 60 x 16 tile + scattered
                return np.concatenate([rng.uniform([300, 200], [330, 210], (12, 2)), rng.uniform([0, 0], [w, h], (10, 2))])
            return rng.uniform([-10, -10], [w + 10, h + 10], (150, 2))  # three ballot chunks

        for call in range(15):
            m0, m1 = one_mask(call % 5).astype(np.float32), one_mask((call + 2) % 5).astype(np.float32)
            out, cnt, blk = c.detect([0, 1], grid, [0, len(m0), len(m0) + len(m1)], np.concatenate([m0, m1]), np.concatenate([q, q]), 200)
            for job, m in ((0, m0), (1, m1)):
                exp_pts, exp_blk = oracle.detect(clahe[job], grid, m, q, 200)
                assert cnt[job] == len(exp_pts), (call, job)
                assert np.array_equal(blk[job, :cnt[job]], exp_blk), (call, job)
                assert np.array_equal(out[job, :cnt[job]].view(np.uint32), exp_pts.view(np.uint32)), (call, job)
    finally:
        c.close()


def test_wait_modes_give_identical_results(oracle):
    """icg_ctx_set_wait_mode only changes how the host waits (spin vs query+sleep), never the results."""
    import ctypes as C
    import icgvins
    w, h = 640, 480
    c = icgvins.Context(w, h, n_slots=2, max_batch=2, max_points=512)
    try:
        a = synth.texture(w, h, seed=84)
        b = synth.shift_image(a, 2.25, -1.5)
        c.preprocess([0, 1], [a, b])
        pts = synth.random_points(200, w, h, 12, seed=9)
        ref_out, ref_st = c.lk_track_fb([0] * 200, [1] * 200, pts, pts)
        assert c.lib.icg_ctx_set_wait_mode(c.h, 7, 20) == -1
        assert c.lib.icg_ctx_set_wait_mode(c.h, 1, 0) == -1
        for mode, us in ((1, 20), (1, 200), (0, 0), (1, 5)):
            assert c.lib.icg_ctx_set_wait_mode(c.h, mode, us) == 0
            out, st = c.lk_track_fb([0] * 200, [1] * 200, pts, pts)
            assert np.array_equal(st, ref_st)
            assert np.array_equal(out.view(np.uint32), ref_out.view(np.uint32))
    finally:
        c.close()


def test_detect_dense_mask_and_fully_masked_blocks(oracle, ctx1280):
    """a tracked frame: ~250 existing features blank most of the image (k_min_eig_nms leaves tiles whose owned pixels are all masked
    before reading the image), plus one frame whose mask covers EVERYTHING (no corner, like the oracle)"""
    w, h = 1280, 720
    imgs = [synth.texture(w, h, seed=83), synth.texture(w, h, seed=84)]
    ctx1280.preprocess([0, 1], imgs)
    grid = grid_for(w, h, 300)
    nblk = grid[0] * grid[1]
    dense = synth.random_points(250, w, h, 0, seed=93)
    gx, gy = np.meshgrid(np.arange(0, w + 40, 40, dtype=np.float32), np.arange(0, h + 40, 40, dtype=np.float32))
    everything = np.stack([gx.ravel(), gy.ravel()], 1)  # discs of radius 45 on a 40-px lattice cover the plane
    masks = [dense, everything]
    quotas = [np.full(nblk, grid[5], np.int32)] * 2
    mask_off = np.cumsum([0] + [len(m) for m in masks]).astype(np.int32)
    out, cnt, blk = ctx1280.detect([0, 1], grid, mask_off, np.concatenate(masks), np.concatenate(quotas), 400)
    for j, img in enumerate(imgs):
        exp_pts, exp_blk = oracle.detect(oracle.clahe(img), grid, masks[j], quotas[j], 400)
        assert cnt[j] == len(exp_pts)
        assert np.array_equal(blk[j, :cnt[j]], exp_blk)
        assert np.array_equal(out[j, :cnt[j]].view(np.uint32), exp_pts.view(np.uint32))
    assert 0 < cnt[0] < 200 and cnt[1] == 0


def test_detect_row_slivers_between_disc_bands(oracle):
    """k_min_eig_nms streams only the RUNS of rows that hold an unmasked pixel (round 6: one wave per 60 x 64 block; gaps of up to six
    masked rows are streamed through, longer ones are skipped and the rolling windows primed again).  Bands of closely spaced discs leave
    slivers of 1..12 unmasked rows between them — at every phase against the 64-row blocks, against the 4-row load chunks and against the
    ROI edges (the two mirrored product rows of a ROI live in a separate copy of the row step) — and sparse discs leave long runs; every
    pattern must give the oracle's corners, bit for bit, in the oracle's order."""
    import icgvins
    w, h = 640, 480
    c = icgvins.Context(w, h, n_slots=1, max_batch=1, max_points=4096)
    try:
        img = synth.texture(w, h, seed=85)
        c.preprocess([0], [img])
        clahe = oracle.clahe(img)
        grid = grid_for(w, h, 100)
        q = np.full(6, grid[5], np.int32)
        r = grid[4]  # disc radius = min distance (40)
        xs = np.arange(0, w + 8, 8, dtype=np.float32)
        for first, gaps in ((0, (1, 2, 3, 5, 6)), (17, (7, 8, 12, 1, 4)), (38, (3, 3, 9, 2, 6)), (-25, (2, 13, 1, 7, 5)), (60, (4, 6, 7, 1, 1))):
            cys, y = [], float(first)
            for g in gaps:
                cys.append(y)
                y += 2 * r + 1 + g  # a band of discs covers 2 r + 1 rows at its centre columns: g rows stay free before the next band
            m = np.array([(x, cy) for cy in cys for x in xs], np.float32)
            out, cnt, blk = c.detect([0], grid, [0, len(m)], m, q, 300)
            exp_pts, exp_blk = oracle.detect(clahe, grid, m, q, 300)
            assert cnt[0] == len(exp_pts) and cnt[0] > 0, (first, cnt[0], len(exp_pts))
            assert np.array_equal(blk[0, :cnt[0]], exp_blk), first
            assert np.array_equal(out[0, :cnt[0]].view(np.uint32), exp_pts.view(np.uint32)), first
        # sparse discs: long runs, blocks without any masked pixel next to blocks that are fully masked
        rng = np.random.RandomState(11)
        for n in (1, 6, 25):
            m = rng.uniform([0, 0], [w, h], (n, 2)).astype(np.float32)
            out, cnt, blk = c.detect([0], grid, [0, len(m)], m, q, 300)
            exp_pts, exp_blk = oracle.detect(clahe, grid, m, q, 300)
            assert cnt[0] == len(exp_pts) and cnt[0] > 0
            assert np.array_equal(blk[0, :cnt[0]], exp_blk), n
            assert np.array_equal(out[0, :cnt[0]].view(np.uint32), exp_pts.view(np.uint32)), n
    finally:
        c.close()


def test_fm_ransac_check_subset_on_lattice_points(oracle, ctx1280):
    """FMEstimatorCallback::checkSubset on the device path: lattice points have many collinear triples, so subsets are rejected and
    redrawn (the RNG keeps advancing) — the mask must still equal the oracle's; a set on ONE line has no valid subset at all"""
    gx, gy = np.meshgrid(np.arange(8, dtype=np.float32) * 35 + 90, np.arange(6, dtype=np.float32) * 35 + 70)
    p1 = np.stack([gx.ravel(), gy.ravel()], 1)
    rng = np.random.RandomState(3)
    p2 = (p1 * np.float32(1.01) + np.float32([4.0, -2.5])).astype(np.float32)
    p2[::7] += rng.uniform(-30, 30, (len(p2[::7]), 2)).astype(np.float32)  # gross outliers
    line = np.stack([np.arange(24, dtype=np.float32) * 9 + 20, np.arange(24, dtype=np.float32) * 4 + 11], 1)
    sets = [(p1, p2), (line, (line + np.float32(2.0)).astype(np.float32))]
    offsets = np.cumsum([0] + [len(s[0]) for s in sets]).astype(np.int32)
    mask = ctx1280.fm_ransac(offsets, np.concatenate([s[0] for s in sets]), np.concatenate([s[1] for s in sets]))
    ok, exp, _, _ = oracle.fm_ransac(p1, p2)
    assert np.array_equal(mask[:len(p1)], exp)
    assert not np.array_equal(oracle.ransac_subsets(len(p1), 30, p1, p2), oracle.ransac_subsets(len(p1), 30))  # rejections did happen
    ok2, exp2, _, it2 = oracle.fm_ransac(*sets[1])
    assert ok2 == 0 and it2 == 0 and np.array_equal(mask[len(p1):], exp2) and mask[len(p1):].sum() == 0
