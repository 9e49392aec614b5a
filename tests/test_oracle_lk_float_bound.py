"""A second, independent implementation of calcOpticalFlowPyrLK (written from SURVEY.md Appendix B.5, in numpy, sharing no code with
oracle/orc_lk.cc) that accumulates the window sums the way OpenCV does — IN FLOAT32 — against the oracle's exact-integer sums.
It bounds the documented deviation (oracle/README.md, "calcOpticalFlowPyrLK"): same control flow (levels, minEig test, epsilon / oscillation
exits, status), positions within 2e-3 px, and a status can only differ where a threshold decision sits inside float rounding.
The accumulation ORDER of real OpenCV is build dependent (SIMD width); two orders are run here (raster, 4-lane strided): the bound has to
hold for both, and they differ from each other by as much as each differs from the oracle."""
import numpy as np
import pytest

import synth

WIN, HALF, MAXLEVEL, MAXCOUNT = 21, 10, 3, 30
F32 = np.float32


def _reflect(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def _pyr(img):
    out = [img]
    k = np.array([1, 4, 6, 4, 1], np.int64)
    for _ in range(MAXLEVEL):
        a = out[-1].astype(np.int64)
        h, w = a.shape
        dw, dh = (w + 1) // 2, (h + 1) // 2
        xs = _reflect(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)
        t = (a[:, xs] * k).sum(-1)
        ys = _reflect(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)
        out.append((((t[ys, :] * k[None, :, None]).sum(1) + 128) >> 8).astype(np.uint8))
    return out


def _scharr(img):
    a = img.astype(np.int64)
    h, w = a.shape
    p = a[_reflect(np.arange(-1, h + 1), h)][:, _reflect(np.arange(-1, w + 1), w)]
    t0 = 3 * (p[:-2] + p[2:]) + 10 * p[1:-1]    # vertical smoothing
    t1 = p[2:] - p[:-2]                          # vertical difference
    dx = t0[:, 2:] - t0[:, :-2]
    dy = 3 * (t1[:, :-2] + t1[:, 2:]) + 10 * t1[:, 1:-1]
    return dx, dy


def _fsum(v, order):
    """float32 accumulation of a 441-vector: 'raster' = one running sum; 'lanes4' = four strided partial sums added at the end"""
    v = v.astype(F32)
    if order == "raster":
        s = F32(0)
        for x in v:
            s = F32(s + x)
        return s
    parts = [F32(0)] * 4
    for k, x in enumerate(v):
        parts[k % 4] = F32(parts[k % 4] + x)
    return F32(F32(parts[0] + parts[1]) + F32(parts[2] + parts[3]))


def _weights(a, b):
    w00 = int(np.rint(F32(F32(F32(1) - a) * F32(F32(1) - b)) * F32(16384)))
    w01 = int(np.rint(F32(a * F32(F32(1) - b)) * F32(16384)))
    w10 = int(np.rint(F32(F32(F32(1) - a) * b) * F32(16384)))
    return w00, w01, w10, 16384 - w00 - w01 - w10


def _patch(img, ix, iy, zero_outside):
    """(WIN+1)x(WIN+1) neighbourhood at integer origin (ix, iy): reflect-101 for images, zero for derivative planes"""
    h, w = img.shape
    ys, xs = iy + np.arange(WIN + 1), ix + np.arange(WIN + 1)
    if zero_outside:
        out = np.zeros((WIN + 1, WIN + 1), np.int64)
        oky, okx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
        out[np.ix_(oky, okx)] = img[np.ix_(ys[oky], xs[okx])]
        return out
    return img[np.ix_(_reflect(ys, h), _reflect(xs, w))].astype(np.int64)


def _blend(p, w, n):
    v = p[:-1, :-1] * w[0] + p[:-1, 1:] * w[1] + p[1:, :-1] * w[2] + p[1:, 1:] * w[3]
    return (v + (1 << (n - 1))) >> n


def lk_float(prev, nxt, pts, guess, order):
    pI, pJ = _pyr(prev), _pyr(nxt)
    der = [_scharr(im) for im in pI]
    out, status = guess.astype(F32).copy(), np.ones(len(pts), np.uint8)
    scale20 = F32(1.0 / (1 << 20))
    for i, pt in enumerate(pts.astype(F32)):
        nextpt = out[i].copy()
        for level in range(MAXLEVEL, -1, -1):
            I, J = pI[level], pJ[level]
            H, W = I.shape
            s = F32(1.0 / (1 << level))
            prevp = pt * s
            nextpt = nextpt * s if level == MAXLEVEL else nextpt * F32(2)
            out[i] = nextpt
            prevp = prevp - F32(HALF)
            ipx, ipy = int(np.floor(prevp[0])), int(np.floor(prevp[1]))
            if ipx < -WIN or ipx >= W or ipy < -WIN or ipy >= H:
                if level == 0:
                    status[i] = 0
                continue
            w = _weights(F32(prevp[0] - F32(ipx)), F32(prevp[1] - F32(ipy)))
            Iw = _blend(_patch(I, ipx, ipy, False), w, 9)
            Ix = _blend(_patch(der[level][0], ipx, ipy, True), w, 14)
            Iy = _blend(_patch(der[level][1], ipx, ipy, True), w, 14)
            A11 = F32(_fsum((Ix * Ix).ravel(), order) * scale20)
            A12 = F32(_fsum((Ix * Iy).ravel(), order) * scale20)
            A22 = F32(_fsum((Iy * Iy).ravel(), order) * scale20)
            D = F32(F32(A11 * A22) - F32(A12 * A12))
            mineig = F32(F32(F32(A22 + A11) - np.sqrt(F32(F32(F32(A11 - A22) * F32(A11 - A22)) + F32(F32(4) * F32(A12 * A12))))) / F32(2 * WIN * WIN))
            if mineig < F32(1e-4) or D < np.finfo(F32).eps:
                if level == 0:
                    status[i] = 0
                continue
            D = F32(F32(1) / D)
            nextpt = nextpt - F32(HALF)
            prevd = np.zeros(2, F32)
            for j in range(MAXCOUNT):
                inx, iny = int(np.floor(nextpt[0])), int(np.floor(nextpt[1]))
                if inx < -WIN or inx >= W or iny < -WIN or iny >= H:
                    if level == 0:
                        status[i] = 0
                    break
                w = _weights(F32(nextpt[0] - F32(inx)), F32(nextpt[1] - F32(iny)))
                diff = _blend(_patch(J, inx, iny, False), w, 9) - Iw
                b1 = F32(_fsum((diff * Ix).ravel(), order) * scale20)
                b2 = F32(_fsum((diff * Iy).ravel(), order) * scale20)
                d = np.array([F32(F32(F32(A12 * b2) - F32(A22 * b1)) * D), F32(F32(F32(A12 * b1) - F32(A11 * b2)) * D)], F32)
                nextpt = nextpt + d
                out[i] = nextpt + F32(HALF)
                if float(d[0]) ** 2 + float(d[1]) ** 2 <= 1e-4:
                    break
                if j > 0 and abs(float(d[0] + prevd[0])) < 0.01 and abs(float(d[1] + prevd[1])) < 0.01:
                    out[i] = out[i] - d * F32(0.5)
                    break
                prevd = d
            nextpt = out[i].copy()
    return out, status


@pytest.mark.parametrize("order", ["raster", "lanes4"])
def test_integer_window_sums_stay_within_float_accumulation_of_opencv(oracle, order):
    w, h = 320, 240
    a = synth.texture(w, h, seed=31)
    b = synth.shift_image(a, 2.6, -1.3)
    ca, cb = oracle.clahe(a), oracle.clahe(b)
    pts = synth.random_points(40, w, h, 14, seed=32)
    pts[-3:] = [[6.0, 7.0], [w - 4.0, h - 6.0], [3.0, h / 2]]  # windows that leave the image
    guess = (pts + np.float32([2.0, -1.0])).astype(np.float32)
    got, st, _ = oracle.lk_track(ca, cb, pts, guess)
    exp, est = lk_float(ca, cb, pts, guess, order)
    assert np.array_equal(st, est), "status differs between exact-integer and float-accumulated window sums"
    ok = st == 1
    assert ok.sum() >= 30
    d = np.abs(got[ok] - exp[ok]).max()
    assert d < 2e-3, d
    # and the tracks are the right ones: the images differ by (2.6, -1.3) px (CLAHE runs on each image separately, so single windows
    # can be off by a pixel; the median is what is asserted)
    inner = ok & (pts[:, 0] > 30) & (pts[:, 0] < w - 30) & (pts[:, 1] > 30) & (pts[:, 1] < h - 30)
    assert np.median(np.abs(got[inner] - pts[inner] - np.float32([2.6, -1.3]))) < 0.25
