"""Deterministic synthetic imagery for tests (integer-seeded; identical bytes wherever it runs)."""
import numpy as np


def texture(w, h, seed=0, blobs=None):
    """Corner-rich u8 texture: 3 octaves of value noise + bright/dark square-ish blobs."""
    rng = np.random.RandomState(seed)
    img = np.zeros((h, w), np.float64)
    for octave, amp in ((64, 70.0), (16, 45.0), (5, 30.0)):
        gh, gw = h // octave + 3, w // octave + 3
        g = rng.rand(gh, gw)
        ys = np.arange(h) / octave
        xs = np.arange(w) / octave
        y0 = ys.astype(int)
        x0 = xs.astype(int)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        img += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    nb = blobs if blobs is not None else (w * h) // 450
    bx = rng.randint(4, w - 12, nb)
    by = rng.randint(4, h - 12, nb)
    bs = rng.randint(3, 8, nb)
    bv = rng.choice([-70.0, 70.0], nb)
    for x, y, s, v in zip(bx, by, bs, bv):
        img[y:y + s, x:x + s] += v
    img = np.clip(img + 20.0, 0, 255)
    return img.astype(np.uint8)


def shift_image(img, dx, dy):
    """Sub-pixel translation by bilinear resampling: out(x,y) = img(x-dx, y-dy)."""
    h, w = img.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = np.clip(xs - dx, 0, w - 1.001)
    sy = np.clip(ys - dy, 0, h - 1.001)
    x0 = sx.astype(int)
    y0 = sy.astype(int)
    fx = sx - x0
    fy = sy - y0
    f = img.astype(np.float64)
    out = (f[y0, x0] * (1 - fx) + f[y0, x0 + 1] * fx) * (1 - fy) + (f[y0 + 1, x0] * (1 - fx) + f[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def random_points(n, w, h, margin, seed):
    rng = np.random.RandomState(seed)
    return np.stack([rng.uniform(margin, w - margin, n), rng.uniform(margin, h - margin, n)], 1).astype(np.float32)


# reference config/gvins.yaml:65,69 intrinsics/distortion, principal point moved to the image centre
CAM_1280 = [787.1611861559479, 787.3928431375225, 640.0, 360.0, 0.0, -0.0917403092279957, 0.08134715036932794,
            0.00017620136958692255, 0.00016737385248865412, 0.0]
CAM_640 = [393.58, 393.70, 320.0, 240.0, 0.0, -0.0917403092279957, 0.08134715036932794, 0.00017620136958692255,
           0.00016737385248865412, 0.0]
