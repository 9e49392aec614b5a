"""Known-answer / cross-implementation tests pinning the ORACLE's image operators.
The reference has no tests and OpenCV is unavailable, so each operator is checked against an independent
vectorised numpy formulation of the same definition and against analytic properties."""
import numpy as np

import synth


def _reflect(idx, n):
    idx = np.abs(idx)
    return np.where(idx >= n, 2 * (n - 1) - idx, idx)


def test_gray_coefficients(oracle):
    rng = np.random.RandomState(0)
    bgr = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    exp = ((bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 + bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14)
    assert np.array_equal(oracle.bgr2gray(bgr), exp.astype(np.uint8))
    white = np.full((4, 4, 3), 255, np.uint8)
    assert np.all(oracle.bgr2gray(white) == 255)


def test_pyrdown_matches_numpy_separable(oracle):
    img = synth.texture(131, 77, seed=4)
    h, w = img.shape
    dw, dh = (w + 1) // 2, (h + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)
    xs = _reflect(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)
    tmp = (img.astype(np.int64)[:, xs] * k).sum(-1)  # h x dw
    ys = _reflect(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)
    out = (tmp[ys, :] * k[None, :, None]).sum(1)
    exp = ((out + 128) >> 8).astype(np.uint8)
    got = oracle.pyrdown(img)
    assert got.shape == (dh, dw)
    assert np.array_equal(got, exp)
    assert np.all(oracle.pyrdown(np.full((40, 50), 77, np.uint8)) == 77)


def test_scharr_matches_numpy(oracle):
    img = synth.texture(64, 48, seed=5)
    h, w = img.shape
    f = img.astype(np.int64)
    yi = _reflect(np.arange(-1, h + 1), h)
    xi = _reflect(np.arange(-1, w + 1), w)
    p = f[yi][:, xi]  # padded (h+2, w+2)
    ix = 3 * (p[:-2, 2:] - p[:-2, :-2]) + 10 * (p[1:-1, 2:] - p[1:-1, :-2]) + 3 * (p[2:, 2:] - p[2:, :-2])
    iy = 3 * (p[2:, :-2] - p[:-2, :-2]) + 10 * (p[2:, 1:-1] - p[:-2, 1:-1]) + 3 * (p[2:, 2:] - p[:-2, 2:])
    d = oracle.scharr(img)
    assert np.array_equal(d[..., 0], ix)
    assert np.array_equal(d[..., 1], iy)


def test_clahe_properties(oracle):
    img = synth.texture(640, 480, seed=6)
    out, lut = oracle.clahe(img, want_lut=True)
    assert out.shape == img.shape and lut.shape == (441, 256)
    # every tile LUT is a monotone CDF ending at 255 (clipped histogram still sums to the tile area)
    assert np.all(np.diff(lut.astype(int), axis=1) >= 0)
    assert np.all(lut[:, 255] == 255)
    # contrast limited: slope of each LUT is bounded by clip/area*255 (+1 for redistribution and rounding)
    tw, th = 651 // 21, 483 // 21
    clip = max(1, int(3.0 * tw * th / 256))
    max_step = np.diff(lut.astype(int), axis=1).max()
    assert max_step <= np.ceil((clip + tw * th / 256 + 1) * 255.0 / (tw * th)) + 1
    # a constant image maps to one constant value
    c = oracle.clahe(np.full((480, 640), 100, np.uint8))
    assert len(np.unique(c)) == 1
    # deterministic
    assert np.array_equal(out, oracle.clahe(img))


def test_clahe_matches_numpy_restatement(oracle):
    """Independent vectorised restatement of B.2 (different code structure, same definition)."""
    img = synth.texture(160, 120, seed=7)
    h, w = img.shape
    T = 21
    ew, eh = w + (T - w % T), h + (T - h % T)
    tw, th = ew // T, eh // T
    ext = img[_reflect(np.arange(eh), h)][:, _reflect(np.arange(ew), w)]
    area = tw * th
    clip = max(1, int(3.0 * area / 256))
    scale = np.float32(255.0) / np.float32(area)
    luts = np.zeros((T, T, 256), np.uint8)
    for ty in range(T):
        for tx in range(T):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            clipped = np.maximum(hist - clip, 0).sum()
            hist = np.minimum(hist, clip)
            batch, residual = divmod(int(clipped), 256)
            hist += batch
            if residual:
                step = max(256 // residual, 1)
                idx = np.arange(0, 256, step)[:residual]
                hist[idx] += 1
            cdf = np.cumsum(hist).astype(np.float32) * scale
            luts[ty, tx] = np.clip(np.rint(cdf), 0, 255).astype(np.uint8)
    inv_tw, inv_th = np.float32(1.0) / np.float32(tw), np.float32(1.0) / np.float32(th)
    txf = np.arange(w, dtype=np.float32) * inv_tw - np.float32(0.5)
    tyf = np.arange(h, dtype=np.float32) * inv_th - np.float32(0.5)
    tx1 = np.floor(txf).astype(int)
    ty1 = np.floor(tyf).astype(int)
    xa = (txf - tx1.astype(np.float32)).astype(np.float32)
    ya = (tyf - ty1.astype(np.float32)).astype(np.float32)
    xa1, ya1 = np.float32(1) - xa, np.float32(1) - ya
    tx2 = np.minimum(tx1 + 1, T - 1)
    ty2 = np.minimum(ty1 + 1, T - 1)
    tx1 = np.maximum(tx1, 0)
    ty1 = np.maximum(ty1, 0)
    Y, X = np.mgrid[0:h, 0:w]
    v = img
    l11 = luts[ty1[Y], tx1[X], v].astype(np.float32)
    l12 = luts[ty1[Y], tx2[X], v].astype(np.float32)
    l21 = luts[ty2[Y], tx1[X], v].astype(np.float32)
    l22 = luts[ty2[Y], tx2[X], v].astype(np.float32)
    res = (l11 * xa1[X] + l12 * xa[X]) * ya1[Y] + (l21 * xa1[X] + l22 * xa[X]) * ya[Y]
    exp = np.clip(np.rint(res), 0, 255).astype(np.uint8)
    got, lut = oracle.clahe(img, want_lut=True)
    assert np.array_equal(lut.reshape(T, T, 256), luts)
    assert np.array_equal(got, exp)


def test_lk_recovers_known_translation(oracle):
    w, h = 320, 240
    img = synth.texture(w, h, seed=8)
    for dx, dy in ((1.3, -0.7), (4.25, 3.5), (-6.0, 2.0)):
        nxt = synth.shift_image(img, dx, dy)
        pts = synth.random_points(60, w, h, 40, seed=9)
        out, st, err = oracle.lk_track(img, nxt, pts, pts.copy())
        ok = st.astype(bool)
        assert ok.mean() > 0.8
        d = out[ok] - pts[ok]
        assert np.median(np.abs(d[:, 0] - dx)) < 0.1 and np.median(np.abs(d[:, 1] - dy)) < 0.1


def test_lk_fb_culls_border_and_inconsistent(oracle):
    w, h = 320, 240
    img = synth.texture(w, h, seed=10)
    nxt = synth.shift_image(img, 2.0, 1.0)
    pts = synth.random_points(80, w, h, 6, seed=11)
    out, st = oracle.lk_track_fb(img, nxt, pts, pts.copy())
    kept = st.astype(bool)
    assert kept.sum() > 30
    assert np.all(out[kept, 0] >= 5) and np.all(out[kept, 0] <= w - 5) and np.all(out[kept, 1] >= 5) and np.all(out[kept, 1] <= h - 5)
    # an unrelated second image must kill (almost) everything through the forward/backward check
    other = synth.texture(w, h, seed=99)
    _, st2 = oracle.lk_track_fb(img, other, pts, pts.copy())
    assert st2.mean() < 0.2


def test_lk_status_zero_for_flat_patch(oracle):
    img = np.full((120, 160), 128, np.uint8)
    pts = np.array([[80.0, 60.0]], np.float32)
    out, st, _ = oracle.lk_track(img, img, pts, pts.copy())
    assert st[0] == 0  # minEig below threshold at level 0
    assert np.allclose(out, pts)


def test_camera_roundtrip(oracle):
    cam = synth.CAM_1280
    pts = synth.random_points(500, 1280, 720, 5, seed=12)
    und = oracle.undistort(cam, pts)
    back = oracle.distort(cam, und)
    assert np.abs(back - pts).max() < 2e-3
    # principal point is a fixed point
    pp = np.array([[cam[2], cam[3]]], np.float32)
    assert np.allclose(oracle.undistort(cam, pp), pp, atol=1e-4)
