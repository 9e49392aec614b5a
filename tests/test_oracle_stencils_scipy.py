"""Second, independent implementation of the exact-integer stencils the front-end is built on: scipy.ndimage.correlate with
mode="mirror" (= OpenCV BORDER_REFLECT_101: d c b | a b c d | c b a) against the oracle's pyrDown, Scharr, Sobel / min-eigenvalue
response and 3x3 box sums.  OpenCV itself is not available offline; this pins the oracle's border handling,
kernel coefficients and rounding against a library that shares no code with it."""
import numpy as np
import pytest

import synth

ndi = pytest.importorskip("scipy.ndimage")


@pytest.mark.parametrize("shape", [(131, 77), (640, 480), (333, 257), (64, 33)])
def test_pyrdown_vs_scipy(oracle, shape):
    w, h = shape
    img = synth.texture(w, h, seed=11)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = ndi.correlate(img.astype(np.int64), np.outer(k, k), mode="mirror")  # 5x5 Gaussian, reflect-101
    exp = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)                         # even pixels, round-half-up of /256
    got = oracle.pyrdown(img)
    assert got.shape == ((h + 1) // 2, (w + 1) // 2)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("shape", [(64, 48), (321, 123)])
def test_scharr_vs_scipy(oracle, shape):
    w, h = shape
    img = synth.texture(w, h, seed=12).astype(np.int64)
    kx = np.array([[-3, 0, 3], [-10, 0, 10], [-3, 0, 3]], np.int64)
    dx = ndi.correlate(img, kx, mode="mirror")
    dy = ndi.correlate(img, kx.T, mode="mirror")
    got = oracle.scharr(img.astype(np.uint8))
    assert np.array_equal(got[..., 0], dx.astype(np.int16))
    assert np.array_equal(got[..., 1], dy.astype(np.int16))


@pytest.mark.parametrize("shape", [(96, 80), (213, 180)])
def test_min_eigen_response_vs_scipy(oracle, shape):
    """cornerMinEigenVal(blockSize 3, ksize 3) on a whole image: Sobel (exact ints, mirror border) x 1/3060 in float32, products
    in float32, 3x3 un-normalised box sum (mirror border) accumulated in float64, eigenvalue formula in float32 — bit for bit."""
    w, h = shape
    img = synth.texture(w, h, seed=13)
    sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64)
    gx = ndi.correlate(img.astype(np.int64), sx, mode="mirror")
    gy = ndi.correlate(img.astype(np.int64), sx.T, mode="mirror")
    s = np.float32(1.0 / 3060.0)
    dx, dy = gx.astype(np.float32) * s, gy.astype(np.float32) * s
    box = np.ones((3, 3), np.float64)
    # every term is a float32 in [s^2, (1020 s)^2]: the nine-term sums are exact in float64 in any order
    a = ndi.correlate((dx * dx).astype(np.float64), box, mode="mirror")
    b = ndi.correlate((dx * dy).astype(np.float64), box, mode="mirror")
    c = ndi.correlate((dy * dy).astype(np.float64), box, mode="mirror")
    af, bf, cf = a.astype(np.float32) * np.float32(0.5), b.astype(np.float32), c.astype(np.float32) * np.float32(0.5)
    exp = (af + cf) - np.sqrt((af - cf) * (af - cf) + bf * bf)
    got = oracle.min_eigen_map(img, (0, 0, w, h))
    assert got.dtype == np.float32 and exp.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_min_eigen_roi_uses_real_pixels_outside_the_roi(oracle):
    """goodFeaturesToTrack on a block ROI (tracking.cc:632-647): the Sobel of a ROI Mat peeks at the REAL pixels around the ROI
    (OpenCV filters a sub-Mat with its parent's pixels as border), while the box sum reflects at the ROI edge (the covariance maps
    are fresh ROI-sized Mats).  scipy: Sobel on the whole image, crop, then box-sum the crop with mirror."""
    w, h = 200, 150
    img = synth.texture(w, h, seed=14)
    rx, ry, rw, rh = 37, 21, 90, 70
    sx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64)
    gx = ndi.correlate(img.astype(np.int64), sx, mode="mirror")[ry:ry + rh, rx:rx + rw]
    gy = ndi.correlate(img.astype(np.int64), sx.T, mode="mirror")[ry:ry + rh, rx:rx + rw]
    s = np.float32(1.0 / 3060.0)
    dx, dy = gx.astype(np.float32) * s, gy.astype(np.float32) * s
    box = np.ones((3, 3), np.float64)
    a = ndi.correlate((dx * dx).astype(np.float64), box, mode="mirror").astype(np.float32) * np.float32(0.5)
    b = ndi.correlate((dx * dy).astype(np.float64), box, mode="mirror").astype(np.float32)
    c = ndi.correlate((dy * dy).astype(np.float64), box, mode="mirror").astype(np.float32) * np.float32(0.5)
    exp = (a + c) - np.sqrt((a - c) * (a - c) + b * b)
    got = oracle.min_eigen_map(img, (rx, ry, rw, rh))
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_gftt_nms_and_threshold_vs_scipy(oracle):
    """goodFeaturesToTrack's candidate set: response > 0.01 * max, equal to the 3x3 maximum of the thresholded map (grey dilation),
    border pixels excluded — reproduced with scipy.ndimage.grey_dilation; the oracle's first corners must be the strongest
    candidates in (response desc) order."""
    w, h = 160, 120
    img = synth.texture(w, h, seed=15)
    eig = oracle.min_eigen_map(img, (0, 0, w, h))
    th = np.float32(np.float64(eig.max()) * 0.01)
    t = np.where(eig > th, eig, np.float32(0))
    dil = ndi.grey_dilation(t, size=(3, 3), mode="mirror")
    cand = (t != 0) & (t == dil)
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    ys, xs = np.nonzero(cand)
    order = np.lexsort((-(ys * w + xs), -t[ys, xs].astype(np.float64)))  # response desc, then raster address desc
    exp = np.stack([xs[order], ys[order]], 1).astype(np.float32)
    got = oracle.good_features(img, None, (0, 0, w, h), 500, 0.01, 0.0)   # min distance 0: every candidate, in order
    assert len(got) == min(500, len(exp))
    assert np.array_equal(got, exp[:len(got)])
