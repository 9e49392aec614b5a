"""Known-answer tests pinning the ORACLE's detection / RANSAC / triangulation restatements
(SURVEY.md Appendix B.7-B.10); the reference ships no tests for them."""
import numpy as np

import synth


def two_view(n, seed, outlier_frac=0.0, noise=0.0):
    rng = np.random.RandomState(seed)
    K = np.array([[787.0, 0, 640], [0, 787.0, 360], [0, 0, 1]])
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(6, 40, n)], 1)
    ang = 0.05
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.5, 0.05, 0.1])
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    p1 = x1[:, :2] / x1[:, 2:]
    p2 = x2[:, :2] / x2[:, 2:]
    p2 += rng.normal(0, noise, p2.shape) if noise > 0 else 0
    nout = int(n * outlier_frac)
    out_idx = rng.choice(n, nout, replace=False)
    p2[out_idx] += rng.uniform(15, 60, (nout, 2)) * rng.choice([-1, 1], (nout, 2))
    truth = np.ones(n, bool)
    truth[out_idx] = False
    return p1.astype(np.float32), p2.astype(np.float32), truth, (K, R, t, X)


def test_circle_table_matches_drawing(oracle):
    for r in (1, 5, 40, 45, 52, 62):
        m = np.full((2 * r + 9, 2 * r + 9), 255, np.uint8)
        c = r + 4
        oracle.draw_circle(m, c, c, r, 0)
        hw = oracle.circle_halfwidths(r)
        exp = np.full_like(m, 255)
        for dy in range(-r, r + 1):
            w = hw[abs(dy)]
            if w >= 0:
                exp[c + dy, c - w:c + w + 1] = 0
        assert np.array_equal(m, exp), r
        # roughly a disc, symmetric, radius r on the axes
        assert m[c, c - r] == 0 and m[c, c - r - 1] == 255 and m[c - r, c] == 0 and m[c - r - 1, c] == 255
        assert np.array_equal(m, m[::-1]) and np.array_equal(m, m[:, ::-1]) and np.array_equal(m, m.T)
    # clipping at the image border must not crash or wrap
    m = np.full((30, 30), 255, np.uint8)
    oracle.draw_circle(m, 2, 27, 10, 0)
    assert m[27, 0] == 0 and m[29, 2] == 0 and m[10, 20] == 255


def test_min_eigen_is_high_on_corners_low_on_edges(oracle):
    img = np.zeros((80, 80), np.uint8)
    img[40:, 40:] = 200  # one corner at (40,40), edges along x=40 and y=40
    eig = oracle.min_eigen_map(img, (0, 0, 80, 80))
    assert eig[39:42, 39:42].max() > 50 * max(eig[60, 39:42].max(), 1e-9)  # vertical edge: min eigenvalue ~0
    assert eig[10, 10] == 0
    # ROI evaluation peeks at real pixels outside the ROI: interior of a ROI equals the full-image map
    img2 = synth.texture(200, 150, seed=3)
    full = oracle.min_eigen_map(img2, (0, 0, 200, 150))
    roi = oracle.min_eigen_map(img2, (30, 20, 100, 90))
    assert np.array_equal(roi[1:-1, 1:-1], full[21:109, 31:129])


def test_good_features_respects_quota_distance_mask(oracle):
    img = synth.texture(400, 300, seed=4)
    mask = np.full((300, 400), 255, np.uint8)
    oracle.draw_circle(mask, 200, 150, 60, 0)
    roi = (10, 10, 380, 280)
    pts = oracle.good_features(img, mask, roi, 25, 0.01, 30)
    assert 5 < len(pts) <= 25
    d = np.linalg.norm(pts[:, None] - pts[None], axis=2) + np.eye(len(pts)) * 1e9
    assert d.min() >= 30
    for x, y in pts:
        assert mask[int(y) + roi[1], int(x) + roi[0]] == 255
        assert 1 <= x <= roi[2] - 2 and 1 <= y <= roi[3] - 2
    # strongest first
    eig = oracle.min_eigen_map(img, roi)
    vals = [eig[int(y), int(x)] for x, y in pts]
    assert all(vals[i] >= vals[i + 1] for i in range(len(vals) - 1))
    # every corner is a 3x3 local maximum of the response
    for x, y in pts:
        x, y = int(x), int(y)
        assert eig[y, x] == eig[y - 1:y + 2, x - 1:x + 2].max()


def test_subpix_converges_to_true_corner(oracle):
    # anti-aliased checker corner at a known sub-pixel location
    cx, cy = 50.3, 47.6
    ss = 8
    hi = np.zeros((100 * ss, 100 * ss))
    X, Y = np.meshgrid((np.arange(100 * ss) + 0.5) / ss, (np.arange(100 * ss) + 0.5) / ss)
    hi[((X < cx) & (Y < cy)) | ((X >= cx) & (Y >= cy))] = 200
    img = hi.reshape(100, ss, 100, ss).mean((1, 3)).astype(np.uint8)
    out = oracle.corner_subpix(img, (0, 0, 100, 100), np.array([[50.0, 48.0], [52.0, 46.0]], np.float32))
    # pixel centres are at integer coordinates: continuous corner (cx,cy) sits at (cx-0.5, cy-0.5)
    assert np.abs(out - np.array([cx - 0.5, cy - 0.5])).max() < 0.15
    # a point on a flat area stays where it is (singular system)
    flat = oracle.corner_subpix(img, (0, 0, 100, 100), np.array([[20.0, 20.0]], np.float32))
    assert np.array_equal(flat, np.array([[20.0, 20.0]], np.float32))


def test_detect_block_order_and_offsets(oracle):
    w, h = 640, 480
    img = synth.texture(w, h, seed=5)
    grid = [3, 2, 213, 240, 40, 17]
    quota = [17, 0, 5, 17, 17, 3]
    pts, blk = oracle.detect(img, grid, np.zeros((0, 2)), quota, 200)
    assert len(pts) > 20
    assert np.all(np.diff(blk) >= 0) and 1 not in blk
    for k in range(6):
        sel = pts[blk == k]
        assert len(sel) <= max(quota[k], 0)
        if len(sel):
            c, r = k % 3, k // 3
            assert np.all(sel[:, 0] >= c * 213) and np.all(sel[:, 0] < (c + 1) * 213)
            assert np.all(sel[:, 1] >= r * 240) and np.all(sel[:, 1] < (r + 1) * 240)
    # masking existing features keeps new corners at least ~min_dist away from them
    pts2, _ = oracle.detect(img, grid, pts[:10], [17] * 6, 200)
    d = np.linalg.norm(pts2[:, None] - pts[None, :10], axis=2)
    assert d.min() > 33.0  # disc radius 40 around the rounded point, minus <=5 px sub-pixel refinement


def test_seven_point_models_fit_their_sample(oracle):
    p1, p2, _, _ = two_view(7, seed=1)
    F = oracle.seven_point(p1.astype(np.float64), p2.astype(np.float64))
    assert 1 <= len(F) <= 3
    for Fk in F:
        for a, b in zip(p1, p2):
            x1 = np.array([a[0], a[1], 1.0])
            x2 = np.array([b[0], b[1], 1.0])
            l = Fk @ x1
            assert abs(x2 @ l) / np.hypot(l[0], l[1]) < 1e-6
        assert abs(np.linalg.det(Fk / np.linalg.norm(Fk))) < 1e-10


def test_ransac_rng_stream(oracle):
    idx = oracle.ransac_subsets(300, 50)
    assert idx.min() >= 0 and idx.max() < 300
    assert all(len(set(r)) == 7 for r in idx)
    # cv::RNG((uint64)-1): first draw = (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFF
    st = (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF)
    assert idx[0, 0] == (st & 0xFFFFFFFF) % 300


def _cv_solve_cubic(c):
    """cv::solveCubic (core/src/mathfuncs.cpp, OpenCV 4.x) restated with numpy's libm: the closed forms that define the ORDER"""
    a0, a1, a2, a3 = [float(v) for v in c]
    if a0 == 0:
        if a1 == 0:
            return [] if a2 == 0 else [-a3 / a2]
        d = a2 * a2 - 4 * a1 * a3
        if d < 0:
            return []
        d = np.sqrt(d)
        q1, q2 = (-a2 + d) * 0.5, (a2 + d) * -0.5
        q = q1 if abs(q1) > abs(q2) else q2
        return [q / a1, a3 / q] if d > 0 else [q / a1]
    a1, a2, a3 = a1 / a0, a2 / a0, a3 / a0
    Q = (a1 * a1 - 3 * a2) / 9.0
    R = (2 * a1 ** 3 - 9 * a1 * a2 + 27 * a3) / 54.0
    d = Q ** 3 - R * R
    if d > 0:
        theta = np.arccos(R / np.sqrt(Q ** 3))
        t0, t1, t2 = -2 * np.sqrt(Q), theta / 3.0, a1 / 3.0
        return [t0 * np.cos(t1) - t2, t0 * np.cos(t1 + 2 * np.pi / 3) - t2, t0 * np.cos(t1 + 4 * np.pi / 3) - t2]
    d = np.sqrt(-d)
    e = np.cbrt(d + abs(R))
    if R > 0:
        e = -e
    return [(e + Q / e) - a1 / 3.0]


def test_cubic_roots_in_solvecubic_order(oracle):
    """the transcendental-free solver returns the roots cv::solveCubic returns, IN ITS ORDER (three real roots: smallest, largest,
    middle; quadratic fallback: q/a1 then a3/q) — the order decides which of equally-scoring 7-point models RANSAC keeps"""
    rng = np.random.RandomState(5)
    n3 = n1 = 0
    for _ in range(400):
        c = rng.uniform(-3, 3, 4)
        if abs(c[0]) < 0.05:
            continue
        exp = _cv_solve_cubic(c)
        got = oracle.solve_cubic(c)
        assert len(got) == len(exp)
        assert np.allclose(got, exp, rtol=1e-9, atol=1e-9)
        n3 += len(exp) == 3
        n1 += len(exp) == 1
    assert n3 > 50 and n1 > 50
    # explicit: (x-1)(x-2)(x-5) -> (1, 5, 2)
    assert np.allclose(oracle.solve_cubic([1, -8, 17, -10]), [1, 5, 2], atol=1e-12)
    # quadratic fallback (leading coefficient zero): 2x^2 - 3x - 5 -> q = 5 (the larger-magnitude one of 5, -2): (5/2, -5/5)
    assert np.allclose(oracle.solve_cubic([0, 2, -3, -5]), _cv_solve_cubic([0, 2, -3, -5]), atol=1e-12)
    assert np.allclose(oracle.solve_cubic([0, 2, -3, -5]), [2.5, -1.0], atol=1e-12)


def _have_collinear(pts):
    """calib3d precomp.hpp haveCollinearPoints(m, count): the LAST point against every pair of the earlier ones"""
    pts = np.asarray(pts, np.float32).astype(np.float64)
    i = len(pts) - 1
    for j in range(i):
        dx1, dy1 = pts[j] - pts[i]
        for k in range(j):
            dx2, dy2 = pts[k] - pts[i]
            if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


class _CvRng:
    def __init__(self):
        self.state = 0xFFFFFFFFFFFFFFFF

    def uniform(self, a, b):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return (self.state & 0xFFFFFFFF) % (b - a) + a


def _get_subsets(p1, p2, n_hyp):
    """RANSACPointSetRegistrator::getSubset (ptsetreg.cpp, 4.x) + FMEstimatorCallback::checkSubset, independent restatement"""
    rng, n, out = _CvRng(), len(p1), []
    for _ in range(n_hyp):
        for _attempt in range(10000):
            idx = []
            for i in range(7):
                v = rng.uniform(0, n)
                while v in idx:
                    v = rng.uniform(0, n)
                idx.append(v)
            if not _have_collinear(p1[idx]) and not _have_collinear(p2[idx]):
                break
        else:
            return out
        out.append(idx)
    return out


def test_ransac_check_subset_rejects_collinear_samples(oracle):
    """FMEstimatorCallback::checkSubset: a sample whose 7th point is collinear with two earlier ones (in either image) is redrawn,
    and the redraw consumes RNG state — pinned against an independent restatement on lattice points (many collinear triples)"""
    gx, gy = np.meshgrid(np.arange(6, dtype=np.float32) * 40 + 100, np.arange(5, dtype=np.float32) * 40 + 80)
    p1 = np.stack([gx.ravel(), gy.ravel()], 1)                         # 30 lattice points
    p2 = (p1 + np.float32([3.0, -2.0])).astype(np.float32)             # collinearity is preserved in the second image
    assert oracle.have_collinear_points(p1[[0, 7, 3, 9, 20, 11, 14]])   # (0, 7, 14) lie on the main diagonal; 14 is LAST
    assert not oracle.have_collinear_points(p1[[0, 7, 14, 9, 20, 11, 3]])  # the same triple not ending in the last point: not tested
    exp = _get_subsets(p1, p2, 40)
    got = oracle.ransac_subsets(30, 40, p1, p2)
    assert np.array_equal(got, np.array(exp, np.int32))
    free = oracle.ransac_subsets(30, 40)
    assert not np.array_equal(free, got)                               # rejections happened and shifted the stream
    # generic (non-lattice) points: nothing is rejected, the stream is the unchecked one
    rng = np.random.RandomState(1)
    q1 = rng.uniform(50, 600, (80, 2)).astype(np.float32)
    q2 = (q1 + rng.normal(0, 5, (80, 2))).astype(np.float32)
    assert np.array_equal(oracle.ransac_subsets(80, 40, q1, q2), oracle.ransac_subsets(80, 40))
    # all points on one line: every subset is rejected, findFundamentalMat fails (all-zero mask)
    line = np.stack([np.arange(20, dtype=np.float32) * 7 + 10, np.arange(20, dtype=np.float32) * 3 + 5], 1)
    ok, mask, _, iters = oracle.fm_ransac(line, line + np.float32(1.0))
    assert ok == 0 and mask.sum() == 0 and iters == 0


def test_ransac_flags_outliers(oracle):
    p1, p2, truth, _ = two_view(200, seed=2, outlier_frac=0.25, noise=0.2)
    ok, mask, F, iters = oracle.fm_ransac(p1, p2, 1.5, 0.99)
    assert ok == 1 and iters < 1000
    m = mask.astype(bool)
    assert (m & ~truth).sum() <= 2          # outliers rejected
    assert (m & truth).sum() >= 0.9 * truth.sum()  # inliers kept
    # fewer than 15 points: not run (all-zero mask, rc 0)
    ok2, mask2, _, _ = oracle.fm_ransac(p1[:10], p2[:10])
    assert ok2 == 0 and mask2.sum() == 0
    # clean data terminates after very few hypotheses
    p1c, p2c, _, _ = two_view(100, seed=3)
    ok3, mask3, _, it3 = oracle.fm_ransac(p1c, p2c)
    assert ok3 == 1 and mask3.sum() == 100 and it3 <= 5


def test_triangulation_recovers_point(oracle):
    _, _, _, (K, R, t, X) = two_view(20, seed=4)
    T0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    T1 = np.hstack([R, t[:, None]])
    for Xw in X:
        pc0 = Xw / Xw[2]
        c1 = R @ Xw + t
        pc1 = c1 / c1[2]
        pw = oracle.triangulate(T0, T1, pc0, pc1)
        assert np.abs(pw - Xw).max() < 1e-8 * max(1, np.abs(Xw).max())


def test_undistort_points_vs_numpy_restatement(oracle):
    """cv::undistortPoints(pts, K, D, noArray(), K) (camera.cc:72-74) restated independently in numpy float64: exactly 5
    fixed-point iterations, skew ignored on the input side and applied on the output side, one rounding to float — bit for bit"""
    import synth
    for cam in (synth.CAM_1280, synth.CAM_640, [700.0, 705.0, 631.0, 352.0, 0.8, -0.31, 0.12, 1e-3, -7e-4, -0.02]):
        fx, fy, cx, cy, skew, k1, k2, p1, p2, k3 = [np.float64(v) for v in cam]
        pts = synth.random_points(300, 1280, 720, 0, seed=41)
        u, v = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
        x0, y0 = (u - cx) / fx, (v - cy) / fy
        x, y = x0.copy(), y0.copy()
        for _ in range(5):
            r2 = x * x + y * y
            icd = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            assert np.all(icd > 0)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (x0 - dx) * icd, (y0 - dy) * icd
        exp = np.stack([fx * x + skew * y + cx, fy * y + cy], 1).astype(np.float32)
        got = oracle.undistort(cam, pts)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def _corner_subpix_numpy(img, roi, corner, mask11):
    """cv::cornerSubPix(block, (5,5), (-1,-1), (COUNT+EPS, 20, 0.01)) for ONE corner, written from SURVEY.md App. B.8 (independent of
    oracle/orc_detect.cc): float bilinear 13x13 patch with replicate border at the ROI edge, double accumulation in raster order"""
    rx, ry, rw, rh = roi
    blk = img[ry:ry + rh, rx:rx + rw].astype(np.float32)
    cT = np.float32(corner)
    cI = cT.copy()
    for it in range(20):
        o = cI - np.float32(6)
        io = np.floor(o).astype(int)
        a, b = np.float32(o[0] - io[0]), np.float32(o[1] - io[1])
        w00, w01 = np.float32((1 - a) * (1 - b)), np.float32(a * (1 - b))
        w10, w11 = np.float32((1 - a) * b), np.float32(a * b)
        xs0, xs1 = np.clip(io[0] + np.arange(13), 0, rw - 1), np.clip(io[0] + np.arange(13) + 1, 0, rw - 1)
        ys0, ys1 = np.clip(io[1] + np.arange(13), 0, rh - 1), np.clip(io[1] + np.arange(13) + 1, 0, rh - 1)
        p = (blk[np.ix_(ys0, xs0)] * w00 + blk[np.ix_(ys0, xs1)] * w01 + blk[np.ix_(ys1, xs0)] * w10 + blk[np.ix_(ys1, xs1)] * w11).astype(np.float32)
        sa = sb = sc = b1 = b2 = 0.0
        for i in range(11):
            for j in range(11):
                m = float(mask11[i * 11 + j])
                tgx = float(p[i + 1, j + 2]) - float(p[i + 1, j])
                tgy = float(p[i + 2, j + 1]) - float(p[i, j + 1])
                gxx, gxy, gyy = tgx * tgx * m, tgx * tgy * m, tgy * tgy * m
                px, py = j - 5, i - 5
                sa += gxx
                sb += gxy
                sc += gyy
                b1 += gxx * px + gxy * py
                b2 += gxy * px + gyy * py
        det = sa * sc - sb * sb
        if abs(det) <= np.finfo(np.float64).eps ** 2:
            break
        scale = 1.0 / det
        c2 = np.float32([float(cI[0]) + sc * scale * b1 - sb * scale * b2, float(cI[1]) - sb * scale * b1 + sa * scale * b2])
        err = float(np.float32(np.float32(c2[0] - cI[0]) ** 2 + np.float32(c2[1] - cI[1]) ** 2))
        cI = c2
        if cI[0] < 0 or cI[0] >= rw or cI[1] < 0 or cI[1] >= rh:
            break
        if not err > 1e-4:
            break
    if abs(cI[0] - cT[0]) > 5 or abs(cI[1] - cT[1]) > 5:
        cI = cT
    return cI


def test_corner_subpix_vs_numpy_restatement(oracle):
    import synth
    w, h = 240, 200
    img = synth.texture(w, h, seed=42)
    roi = (20, 15, 180, 160)
    corners = oracle.good_features(img, None, roi, 25, 0.01, 12.0)
    assert len(corners) >= 15
    got = oracle.corner_subpix(img, roi, corners)
    m = oracle.subpix_mask()
    exp = np.stack([_corner_subpix_numpy(img, roi, c, m) for c in corners])
    assert np.abs(got - exp).max() < 2e-5  # (numpy rounds the float32 blend term by term; the oracle's expression is the same up to that)
    assert np.abs(got - corners).max() > 0.05  # the refinement moved the corners
