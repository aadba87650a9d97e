"""Host-side image augmentation of the CUB pair dataset (cub/code/data/data.py:57-150; outside the hot path).

The reference composes albumentations transforms (absent from this image, like cv2); this is a numpy / scipy.ndimage restatement of
the two pipelines from the transforms' documented defaults.  UNVERIFIED against albumentations itself: the random draws follow the
documented distributions, not albumentations' generator call order, so pipelines agree in distribution, not sample for sample.

Both pipelines take one or more uint8 HxWx3 images and apply the SAME drawn parameters to all of them (albumentations'
``additional_targets``: data.py:60-62, 119-135) -- that is what keeps ``view1`` / ``view0_target`` photometrically in sync and
``view0`` / ``view0_target`` geometrically in sync (data.py:166-173).

  appearance (data.py:64-101), whole pipeline with p = 0.9:
      OneOf[MedianBlur(3), Blur(3)] p=.5;  3 x OneOf[RandomBrightnessContrast, RGBShift, HueSaturationValue] p=.8;
      ToGray p=.1;  ChannelShuffle p=.3
  shape (data.py:104-117), whole pipeline with p = 0.9:
      HorizontalFlip p=.3;  ShiftScaleRotate(shift .0625, scale .25, rotate 25 deg, replicate border) p=.3;
      OneOf[PiecewiseAffine(4x4 grid, scale .03-.05), ElasticTransform(alpha 1, sigma 50, alpha_affine 50, replicate border)] p=.3
"""
import numpy as np
from scipy import ndimage


# ------------------------------------------------------------------------------------------------------------ photometric
def _box3(img):
    """cv2.blur(ksize 3): 3x3 mean, reflect-101 border, rounded."""
    out = ndimage.uniform_filter(img.astype(np.float32), size=(3, 3, 1), mode="mirror")
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _median3(img):
    return ndimage.median_filter(img, size=(3, 3, 1), mode="nearest")


def _brightness_contrast(img, alpha, beta):
    """RandomBrightnessContrast, brightness_by_max=True: x * alpha + beta * 255 through a 256-entry table."""
    lut = np.clip(np.arange(256, dtype=np.float32) * alpha + beta * 255.0, 0, 255).astype(np.uint8)
    return lut[img]


def _rgb_shift(img, shifts):
    return np.clip(img.astype(np.int16) + np.asarray(shifts, np.int16), 0, 255).astype(np.uint8)


def rgb_to_hsv_u8(img):
    """cv2.COLOR_RGB2HSV for uint8: H in [0, 180), S and V in [0, 255]."""
    x = img.astype(np.float32)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    v = x.max(-1)
    d = v - x.min(-1)
    s = np.where(v > 0, d / np.maximum(v, 1e-12) * 255.0, 0.0)
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) * 60.0
    h = np.where(d > 0, h, 0.0)
    h = np.where(h < 0, h + 360.0, h) / 2.0
    return np.stack([np.rint(h) % 180, np.rint(s), v], -1).astype(np.uint8)


def hsv_to_rgb_u8(hsv):
    h = hsv[..., 0].astype(np.float32) * 2.0 / 60.0
    s = hsv[..., 1].astype(np.float32) / 255.0
    v = hsv[..., 2].astype(np.float32)
    i = np.floor(h).astype(np.int32) % 6
    f = h - np.floor(h)
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([r, g, b], -1)), 0, 255).astype(np.uint8)


def _hue_sat_val(img, dh, ds, dv):
    hsv = rgb_to_hsv_u8(img).astype(np.int16)
    hsv[..., 0] = (hsv[..., 0] + dh) % 180
    hsv[..., 1] = np.clip(hsv[..., 1] + ds, 0, 255)
    hsv[..., 2] = np.clip(hsv[..., 2] + dv, 0, 255)
    return hsv_to_rgb_u8(hsv.astype(np.uint8))


def _to_gray(img):
    g = np.rint(img.astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32))
    return np.repeat(np.clip(g, 0, 255).astype(np.uint8)[..., None], 3, -1)


def _draw_color_op(rng):
    """One of RandomBrightnessContrast / RGBShift / HueSaturationValue (equal weights), with its parameters."""
    k = rng.randint(3)
    if k == 0:
        a, b = 1.0 + rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)
        return lambda im: _brightness_contrast(im, a, b)
    if k == 1:
        sh = [int(round(rng.uniform(-20, 20))) for _ in range(3)]
        return lambda im: _rgb_shift(im, sh)
    dh, ds, dv = int(round(rng.uniform(-20, 20))), int(round(rng.uniform(-30, 30))), int(round(rng.uniform(-20, 20)))
    return lambda im: _hue_sat_val(im, dh, ds, dv)


def appearance_ops(rng, p=0.9):
    """Draw one realisation of the appearance pipeline: a list of uint8 -> uint8 image functions."""
    ops = []
    if rng.rand() >= p:
        return ops
    if rng.rand() < 0.5:
        ops.append(_median3 if rng.randint(2) == 0 else _box3)
    for _ in range(3):
        if rng.rand() < 0.8:
            ops.append(_draw_color_op(rng))
    if rng.rand() < 0.1:
        ops.append(_to_gray)
    if rng.rand() < 0.3:
        perm = rng.permutation(3)
        ops.append(lambda im: np.ascontiguousarray(im[..., perm]))
    return ops


# -------------------------------------------------------------------------------------------------------------- geometric
def _warp(img, ys, xs):
    """Bilinear resampling at (ys, xs) with a replicated border (cv2.BORDER_REPLICATE)."""
    out = np.empty(ys.shape + (img.shape[2],), np.float32)
    for c in range(img.shape[2]):
        out[..., c] = ndimage.map_coordinates(img[..., c].astype(np.float32), [ys, xs], order=1, mode="nearest")
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _affine_grid(h, w, mat):
    """Source coordinates of every output pixel for a 2x3 OUTPUT->SOURCE matrix over (x, y, 1)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return mat[1, 0] * xx + mat[1, 1] * yy + mat[1, 2], mat[0, 0] * xx + mat[0, 1] * yy + mat[0, 2]


def _shift_scale_rotate(rng, h, w):
    angle = np.deg2rad(rng.uniform(-25, 25))
    scale = 1.0 + rng.uniform(-0.25, 0.25)
    dx, dy = rng.uniform(-0.0625, 0.0625) * w, rng.uniform(-0.0625, 0.0625) * h
    cx, cy = w / 2.0, h / 2.0
    c, s = np.cos(angle) * scale, np.sin(angle) * scale
    fwd = np.array([[c, s, (1 - c) * cx - s * cy + dx], [-s, c, s * cx + (1 - c) * cy + dy], [0, 0, 1]], np.float64)
    inv = np.linalg.inv(fwd)[:2].astype(np.float32)       # cv2.warpAffine inverts the forward matrix the same way
    ys, xs = _affine_grid(h, w, inv)
    return lambda im: _warp(im, ys, xs)


def _piecewise_affine(rng, h, w, rows=4, cols=4):
    """imgaug PiecewiseAffine: a rows x cols control grid whose points move by N(0, scale * size), scale ~ U(.03, .05).
    The displacement between control points is interpolated bilinearly here (imgaug triangulates the grid)."""
    scale = rng.uniform(0.03, 0.05)
    jy = rng.normal(0, scale, (rows, cols)).astype(np.float32) * h
    jx = rng.normal(0, scale, (rows, cols)).astype(np.float32) * w
    gy, gx = np.mgrid[0:h, 0:w].astype(np.float32)
    cy, cx = gy * (rows - 1) / max(h - 1, 1), gx * (cols - 1) / max(w - 1, 1)
    dy = ndimage.map_coordinates(jy, [cy, cx], order=1, mode="nearest")
    dx = ndimage.map_coordinates(jx, [cy, cx], order=1, mode="nearest")
    ys, xs = gy + dy, gx + dx
    return lambda im: _warp(im, ys, xs)


def _elastic(rng, h, w, alpha=1.0, sigma=50.0, alpha_affine=50.0):
    """albumentations ElasticTransform: a random affine from three jittered anchor points, then a Gaussian-smoothed random
    displacement field of amplitude alpha."""
    c = np.float32([w, h]) / 2.0
    sq = min(h, w) // 3
    p1 = np.float32([c + sq, [c[0] + sq, c[1] - sq], c - sq])
    p2 = p1 + rng.uniform(-alpha_affine, alpha_affine, p1.shape).astype(np.float32)
    a = np.concatenate([p1, np.ones((3, 1), np.float32)], 1)
    fwd = np.linalg.solve(a.astype(np.float64), p2.astype(np.float64)).T          # 2x3: p2 = fwd @ (p1, 1)
    inv = np.linalg.inv(np.vstack([fwd, [0, 0, 1]]))[:2].astype(np.float32)
    ys, xs = _affine_grid(h, w, inv)
    dx = ndimage.gaussian_filter(rng.rand(h, w).astype(np.float32) * 2 - 1, sigma) * alpha
    dy = ndimage.gaussian_filter(rng.rand(h, w).astype(np.float32) * 2 - 1, sigma) * alpha
    ys, xs = ys + dy, xs + dx
    return lambda im: _warp(im, ys, xs)


def shape_ops(rng, h, w, p=0.9):
    ops = []
    if rng.rand() >= p:
        return ops
    if rng.rand() < 0.3:
        ops.append(lambda im: np.ascontiguousarray(im[:, ::-1]))
    if rng.rand() < 0.3:
        ops.append(_shift_scale_rotate(rng, h, w))
    if rng.rand() < 0.3:
        ops.append(_piecewise_affine(rng, h, w) if rng.randint(2) == 0 else _elastic(rng, h, w))
    return ops


# ------------------------------------------------------------------------------------------------------------ entry points
def _to_u8(image):
    return ((image + 1.0) * 255.0 / 2.0).astype(np.uint8)          # data.py:120-121 (truncating cast)


def _from_u8(image):
    return image.astype(np.float32) * 2.0 / 255.0 - 1.0             # data.py:131-133


def _apply(ops, images):
    out = []
    for im in images:
        u = _to_u8(im)
        for op in ops:
            u = op(u)
        out.append(_from_u8(u))
    return out


def stochastic_appearance_augmentation(rng, *images):
    """data.py:119-135: the same photometric realisation on every image; float [-1, 1] in and out (through uint8)."""
    return _apply(appearance_ops(rng), images)


def stochastic_shape_augmentation(rng, *images):
    """data.py:138-154."""
    h, w = images[0].shape[:2]
    return _apply(shape_ops(rng, h, w), images)
