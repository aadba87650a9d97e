"""CPU restatement of the in-graph thin-plate-spline augmentation (TEST INFRASTRUCTURE).

The reference calls ``eddata.utils.tps`` (cub/code/SB_model48i/model.py:282-311), an un-vendored dependency whose
yaml block is marked "adapted from https://github.com/CompVis/unsupervised-disentangling"
(train_cub_subset_tps.yaml:188).  Neither package is available offline, so this follows the PUBLISHED algorithm of that
code base (Lorenz et al., CVPR 2019: random scale / rotation / offset of a jittered control-point set plus random TPS
vectors, TPS solved as in the TF "TPS-STN" recipe, bilinear sampling with clamped indices): **parity unpinned**, every
constant below is an inference.  All randomness enters as explicit U(0,1) draws so the HIP path can be compared exactly.
"""
import math

import torch

# control points of unsupervised-disentangling's tps_parameters (x, y) in [-1, 1]^2  (UNVERIFIED)
CONTROL_POINTS = [(-.5, -.5), (.5, -.5), (-.5, .5), (.5, .5), (.2, -.2), (-.2, .2), (.2, .2), (-.2, -.2), (0., 0.)]
N_UNIFORMS = 2 * len(CONTROL_POINTS) * 2 + 2 + 2 + 2 + 1     # coord jitter, vectors, offset, offset_2, t_scal, rotation


def uniforms_to_params(u, scal, tps_scal, rot_scal, off_scal, scal_var, augm_scal=1.0):
    """u [n, N_UNIFORMS] in [0,1) -> dict(coord, vector, offset, offset_2, t_scal, rot_mat)  (tps_parameters)."""
    n, K = u.shape[0], len(CONTROL_POINTS)
    rng = lambda t, lo, hi: lo + (hi - lo) * t
    base = torch.tensor(CONTROL_POINTS, dtype=u.dtype).view(1, K, 2)
    i = 0
    coord = base + rng(u[:, i:i + 2 * K].reshape(n, K, 2), -0.2, 0.2); i += 2 * K
    vector = rng(u[:, i:i + 2 * K].reshape(n, K, 2), -tps_scal, tps_scal); i += 2 * K
    offset = rng(u[:, i:i + 2].reshape(n, 1, 2), -off_scal, off_scal); i += 2
    offset_2 = rng(u[:, i:i + 2].reshape(n, 1, 2), -off_scal, off_scal); i += 2
    t_scal = rng(u[:, i:i + 2], scal * (1.0 - scal_var), scal * (1.0 + scal_var)) * augm_scal; i += 2
    rot = rng(u[:, i:i + 1], -rot_scal, rot_scal)
    rot_mat = torch.stack([torch.cos(rot), -torch.sin(rot), torch.sin(rot), torch.cos(rot)], dim=-1).reshape(n, 2, 2)
    return {"coord": coord, "vector": vector, "offset": offset, "offset_2": offset_2, "t_scal": t_scal, "rot_mat": rot_mat}


def make_input_tps_param(p):
    """Scale about `offset`, rotate about `offset_2`; returns (coord, t_vector) with coord + t_vector = the targets."""
    scaled = p["t_scal"].unsqueeze(1) * (p["coord"] + p["vector"] - p["offset"]) + p["offset"]
    t_vector = torch.einsum("blk,bck->bcl", p["rot_mat"], scaled - p["offset_2"]) + p["offset_2"] - p["coord"]
    return p["coord"], t_vector


def solve_system(coord, vector):
    """T [n, 2, K+3]: f(x) = T @ [1, x, y, phi(|x - c_1|), ..., phi(|x - c_K|)], phi(d2) = d2 log(d2 + 1e-6),
    with f(c_i) = c_i + v_i and the usual affine side conditions."""
    n, K, _ = coord.shape
    ones = torch.ones(n, K, 1, dtype=coord.dtype)
    p = torch.cat([ones, coord], dim=2)                                   # [n,K,3]
    d2 = ((p.unsqueeze(2) - p.unsqueeze(1)) ** 2).sum(dim=3)              # [n,K,K]
    r = d2 * torch.log(d2 + 1e-6)
    W0 = torch.cat([p, r], dim=2)                                         # [n,K,K+3]
    W1 = torch.cat([torch.zeros(n, 3, 3, dtype=coord.dtype), p.transpose(1, 2)], dim=2)
    W = torch.cat([W0, W1], dim=1)                                        # [n,K+3,K+3]
    tp = torch.cat([coord + vector, torch.zeros(n, 3, 2, dtype=coord.dtype)], dim=1)
    T = torch.linalg.solve(W, tp)                                         # [n,K+3,2]
    return T.transpose(1, 2).contiguous()


def sample_positions(T, coord, h, w):
    """Source coordinates (x_s, y_s) in [-1,1] for every output pixel: [n, h, w] each."""
    n, K, _ = coord.shape
    xs = torch.linspace(-1.0, 1.0, w, dtype=T.dtype).view(1, 1, w).expand(n, h, w)
    ys = torch.linspace(-1.0, 1.0, h, dtype=T.dtype).view(1, h, 1).expand(n, h, w)
    d2 = (xs.unsqueeze(-1) - coord[:, :, 0].view(n, 1, 1, K)) ** 2 + (ys.unsqueeze(-1) - coord[:, :, 1].view(n, 1, 1, K)) ** 2
    r = d2 * torch.log(d2 + 1e-6)
    feats = torch.cat([torch.ones(n, h, w, 1, dtype=T.dtype), xs.unsqueeze(-1), ys.unsqueeze(-1), r], dim=-1)   # [n,h,w,K+3]
    out = torch.einsum("nck,nhwk->nchw", T, feats)
    return out[:, 0], out[:, 1]


def interpolate(img, x_s, y_s):
    """Bilinear sampling of the classic TF spatial-transformer `_interpolate`: pixel = (coord + 1) * size / 2, corner
    indices clamped to the image, weights from the clamped corners."""
    n, h, w, c = img.shape
    x = (x_s + 1.0) * w / 2.0
    y = (y_s + 1.0) * h / 2.0
    x0 = torch.floor(x); y0 = torch.floor(y)
    x1 = x0 + 1; y1 = y0 + 1
    x0c, x1c = x0.clamp(0, w - 1), x1.clamp(0, w - 1)
    y0c, y1c = y0.clamp(0, h - 1), y1.clamp(0, h - 1)
    idx = lambda yy, xx: img.reshape(n, h * w, c).gather(1, (yy.long() * w + xx.long()).reshape(n, h * w, 1).expand(n, h * w, c)).reshape(n, h, w, c)
    wa = ((x1c - x) * (y1c - y)).unsqueeze(-1); wb = ((x1c - x) * (y - y0c)).unsqueeze(-1)
    wc = ((x - x0c) * (y1c - y)).unsqueeze(-1); wd = ((x - x0c) * (y - y0c)).unsqueeze(-1)
    return wa * idx(y0c, x0c) + wb * idx(y1c, x0c) + wc * idx(y0c, x1c) + wd * idx(y1c, x1c)


def thin_plate_spline(img, coord, vector):
    T = solve_system(coord, vector)
    x_s, y_s = sample_positions(T, coord, img.shape[1], img.shape[2])
    return interpolate(img, x_s, y_s)


def make_tps(views, u, tps_parameters):
    """cub model.py:282-311: views 0 and 1 get independent transforms (u [2B, N_UNIFORMS]); the target gets view0's."""
    v0, v1, vt = views
    B = v0.shape[0]
    coord, vector = make_input_tps_param(uniforms_to_params(u.to(v0.dtype), **tps_parameters))
    a = thin_plate_spline(torch.cat([v0, v1], 0), coord, vector)
    t = thin_plate_spline(vt, coord[:B], vector[:B])
    return a[:B], a[B:], t
