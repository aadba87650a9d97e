"""Independent NumPy restatements of the non-convolution ops on the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function cites the reference
lines it follows (paths relative to /root/reference; N = cub/code/nn.py,
M = cub/code/SB_model48i/model.py).  These are written with plain loops /
NumPy broadcasting, deliberately *not* sharing code with oracle/ref_model.py
(torch), so the two can cross-check each other.
"""
import math

import numpy as np


# --------------------------------------------------------------------------- util.py
def fill_triangular(x, upper=False):
    """cub/code/util.py:878-995 (clockwise spiral fill).

    Known answer (util.py:894-902): [1..6] -> [[4,0,0],[6,5,0],[3,2,1]].
    """
    x = np.asarray(x)
    m = x.shape[-1]
    n = int(round(math.sqrt(0.25 + 2.0 * m) - 0.5))
    if n * (n + 1) // 2 != m:
        raise ValueError("Input right-most shape ({}) does not correspond to a triangular matrix.".format(m))
    if upper:
        cat = np.concatenate([x, x[..., n:][..., ::-1]], axis=-1)
    else:
        cat = np.concatenate([x[..., n:], x[..., ::-1]], axis=-1)
    mat = cat.reshape(x.shape[:-1] + (n, n))
    return np.triu(mat) if upper else np.tril(mat)


def fill_triangular_index(n):
    """Index map of the lower spiral fill: out[i, j] = x[idx[i, j]] for j <= i (else -1).

    Derived from util.py:981-993: row-major position q = i*n + j of
    concat(x[n:], reverse(x)); q < m-n -> x[n+q], else x[m-1-(q-(m-n))].
    """
    m = n * (n + 1) // 2
    idx = -np.ones((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(i + 1):
            q = i * n + j
            idx[i, j] = n + q if q < m - n else m - 1 - (q - (m - n))
    return idx


# --------------------------------------------------------------------------- nn.py basics
def leaky_relu(x, alpha=0.2):
    """N:755-756 -> tf.nn.leaky_relu default alpha 0.2."""
    return np.where(x > 0, x, alpha * x)


def softmax_lastdim(x):
    """N:58-62 (spatial=False) -> tf.nn.softmax over the last axis."""
    z = x - x.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def hard_max(y, axis=-1):
    """N:134-136: equality-to-max (ties give several ones)."""
    return (y == y.max(axis=axis, keepdims=True)).astype(y.dtype)


def spatial_softmax(x):
    """N:65-71: softmax over H*W per (n, c); x is [N,H,W,C]."""
    n, h, w, c = x.shape
    f = x.transpose(0, 3, 1, 2).reshape(n * c, h * w)
    p = softmax_lastdim(f)
    return p.reshape(n, c, h, w).transpose(0, 2, 3, 1)


def probs_to_mu_sigma(probs):
    """N:1541-1587 with scaling_factor == 1 (M:440): grid is (y, x), linspace(-1,1) inclusive."""
    n, h, w, k = probs.shape
    ys = np.linspace(-1.0, 1.0, h).astype(probs.dtype)
    xs = np.linspace(-1.0, 1.0, w).astype(probs.dtype)
    mu = np.zeros((n, k, 2), probs.dtype)
    second = np.zeros((n, k, 2, 2), probs.dtype)
    for i in range(h):
        for j in range(w):
            g = np.array([ys[i], xs[j]], probs.dtype)
            p = probs[:, i, j, :]  # [n,k]
            mu += p[..., None] * g
            second += p[..., None, None] * np.outer(g, g)
    sigma = second - mu[..., :, None] * mu[..., None, :]
    return mu, sigma


def mu_to_pixel(mu, h):
    """M:441 / M:459: tf.cast(mu*h/2 + h/2, int32) truncates toward zero."""
    return np.trunc(mu * h / 2.0 + h / 2.0).astype(np.int32)


def draw_rect(centers, ph, pw, h, w, dtype=np.float32, order="xy"):
    """tfutils.draw_rect (EXTERNAL, semantics inferred - SURVEY 8a-9, Appendix C).

    centers: int [K,2], read as (x, y) for order "xy" (default: the reading the reference's step-0 patch_loss supports,
    see oracle/ref_model.py draw_rect) or (y, x) for "yx".  Box spans c-ph//2 .. c+ph//2 INCLUSIVE
    (33 px for patch_size 32, matches step-0 patch_loss 15294.75 ~= 128^2-33^2,
    cub/train/log.txt:244), clipped to the image.  Returns [K,h,w].
    """
    k = centers.shape[0]
    out = np.zeros((k, h, w), dtype)
    iy, ix = (1, 0) if order == "xy" else (0, 1)
    for i in range(k):
        cy, cx = int(centers[i, iy]), int(centers[i, ix])
        y0, y1 = max(cy - ph // 2, 0), min(cy + ph // 2, h - 1)
        x0, x1 = max(cx - pw // 2, 0), min(cx + pw // 2, w - 1)
        if y1 >= y0 and x1 >= x0:
            out[i, y0:y1 + 1, x0:x1 + 1] = 1
    return out


def add_coordinates(x):
    """N:2123-2154: append xx (column index / (H-1)) then yy (row index / (W-1)), both *2-1."""
    n, xd, yd, _ = x.shape  # x_dim = shape[1] (rows), y_dim = shape[2] (cols)
    xx = np.tile(np.arange(yd)[None, None, :], (n, xd, 1)).astype(x.dtype) / max(1, xd - 1)
    yy = np.tile(np.arange(xd)[None, :, None], (n, 1, yd)).astype(x.dtype) / max(1, yd - 1)
    xx = xx * 2 - 1
    yy = yy * 2 - 1
    return np.concatenate([x, xx[..., None], yy[..., None]], axis=-1)


def bilinear_up2(x):
    """N:844-846 -> tf.image.resize_images(BILINEAR) of TF-1.14: legacy kernel,
    align_corners=False, NO half-pixel centres: src = dst/2, i1 = min(i0+1, in-1)."""
    n, h, w, c = x.shape
    out = np.zeros((n, 2 * h, 2 * w, c), x.dtype)
    for oy in range(2 * h):
        sy = oy * (h / (2.0 * h))
        y0 = int(math.floor(sy)); y1 = min(y0 + 1, h - 1); wy = sy - y0
        for ox in range(2 * w):
            sx = ox * (w / (2.0 * w))
            x0 = int(math.floor(sx)); x1 = min(x0 + 1, w - 1); wx = sx - x0
            top = x[:, y0, x0] * (1 - wx) + x[:, y0, x1] * wx
            bot = x[:, y1, x0] * (1 - wx) + x[:, y1, x1] * wx
            out[:, oy, ox] = top * (1 - wy) + bot * wy
    return out


def image_gradients(x):
    """tf.image.image_gradients (N:1446): forward differences, last row/col zero."""
    dy = np.zeros_like(x); dx = np.zeros_like(x)
    dy[:, :-1] = x[:, 1:] - x[:, :-1]
    dx[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1]
    return dy, dx


def squared_grad(x):
    """N:1366-1390 (fd_kernel/tf_grad/tf_squared_grad): SAME-padded correlation with
    0.5*[0, .5, -.5] along W and along H -> 0.25*(x[c]-x[c+1]), zero beyond the border."""
    xr = np.concatenate([x[:, :, 1:], np.zeros_like(x[:, :, :1])], axis=2)
    xd = np.concatenate([x[:, 1:], np.zeros_like(x[:, :1])], axis=1)
    gw = 0.25 * (x - xr)
    gh = 0.25 * (x - xd)
    return gw * gw + gh * gh


def mumford_shah(x, alpha, lam):
    """N:1393-1398."""
    g = squared_grad(x)
    r = np.minimum(alpha * g, lam)
    smooth = np.where(alpha * g < lam, r, 0.0)
    contour = np.where(alpha * g >= lam, r, 0.0)
    return r, smooth, contour


def categorical_kl(probs):
    """M:21-25."""
    k = float(probs.shape[-1])
    return float(np.mean(np.sum(probs * np.log(k * probs + 1e-20), axis=-1)))


def kl_improper_gmrf(mean):
    """N:1444-1451."""
    dy, dx = image_gradients(mean)
    e = 0.5 * (dy * dy + dx * dx)
    return float(np.mean(e.sum(axis=(1, 2, 3))))


def full_latent(params, dim):
    """N:1134-1177: mean, L (exp-diag, rows/sqrt(i+1)), log_diag."""
    mean = params[:, :dim]
    L = fill_triangular(params[:, dim:])
    log_diag = np.diagonal(L, axis1=1, axis2=2).copy()
    rw = np.sqrt(np.arange(dim) + 1.0).reshape(1, dim, 1)
    L = L / rw
    for i in range(dim):
        L[:, i, i] = np.exp(log_diag[:, i])
    return mean, L, log_diag


def full_latent_kl(mean, L, log_diag):
    """N:1196-1208."""
    kl = 0.5 * np.sum((L * L).sum(axis=2) - 1.0 + mean * mean - 2.0 * log_diag, axis=1)
    return float(kl.mean())


def tf_hm(P, h, w, stddev):
    """N:1639-1702 (exp=True): integer pixel grid in (x, y) order."""
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    grid = np.stack([xs, ys], 2).astype(np.float32).reshape(1, h, w, 1, 2)
    d = (grid - P[:, None, None]) ** 2
    d = -d / (2 * stddev[:, None, None] ** 2)
    return np.exp(d.sum(4))


def tf_hm3(h, w, mu, L):
    """N:1976-2021: MultivariateNormalTriL(mu, L).prob on the (y, x) linspace(-1,1) grid."""
    b, p, _ = mu.shape
    ys = np.linspace(-1.0, 1.0, h); xs = np.linspace(-1.0, 1.0, w)
    out = np.zeros((b, h, w, p), np.float64)
    for bi in range(b):
        for pi in range(p):
            Lm = L[bi, pi].astype(np.float64)
            det = abs(Lm[0, 0] * Lm[1, 1])
            for i in range(h):
                for j in range(w):
                    d = np.array([ys[i], xs[j]]) - mu[bi, pi]
                    z = np.linalg.solve(Lm, d)
                    out[bi, i, j, pi] = math.exp(-0.5 * float(z @ z)) / (2 * math.pi * det)
    return out


# --------------------------------------------------------------------------- schedules / optimiser
def linear_var(step, start, end, start_value, end_value, clip_min=0.0, clip_max=1.0):
    """N:1064-1073."""
    v = (end_value - start_value) / (end - start) * (float(step) - start) + start_value
    return float(min(max(v, clip_min), clip_max))


def staircase_var(step, start, start_value, step_size, stair_factor, clip_min=0.0, clip_max=1.0):
    """N:1076-1083."""
    v = stair_factor ** ((float(step) - start) // step_size) * start_value
    return float(min(max(v, clip_min), clip_max))


def tf_adam_step(p, g, m, v, t, lr, beta1, beta2, eps=1e-8):
    """tf.train.AdamOptimizer (TF 1.14): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMA; p -= lr_t * m / (sqrt(v) + eps)  (eps OUTSIDE the corrected root)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v
