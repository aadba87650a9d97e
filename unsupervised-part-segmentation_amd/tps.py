"""In-graph thin-plate-spline augmentation of the CUB model (cub/code/SB_model48i/model.py:282-311).

The reference imports it from the un-vendored ``eddata.utils.tps`` (the yaml block says "adapted from
https://github.com/CompVis/unsupervised-disentangling", train_cub_subset_tps.yaml:188); the three entry points keep
their names and argument meaning -- ``tps_parameters``, ``make_input_tps_param``, ``ThinPlateSpline`` -- but the
arithmetic is re-derived from the published algorithm (Lorenz et al., CVPR 2019): UNVERIFIED, parity unpinned.
The parameter draw and the (K+3)x(K+3) TPS solve are tiny device-side torch ops; the warp itself is the HIP kernel
``ups_tps_warp``.
"""
import torch

from . import lib as L

# control points (x, y) in [-1, 1]^2 of unsupervised-disentangling's tps_parameters (UNVERIFIED)
CONTROL_POINTS = [(-.5, -.5), (.5, -.5), (-.5, .5), (.5, .5), (.2, -.2), (-.2, .2), (.2, .2), (-.2, -.2), (0., 0.)]
N_UNIFORMS = 2 * len(CONTROL_POINTS) * 2 + 2 + 2 + 2 + 1


def tps_parameters(batch_size, scal, tps_scal, rot_scal, off_scal, scal_var, augm_scal=1.0, uniforms=None, generator=None,
                   device=None):
    """Random scale / rotation / offsets + jittered control points and TPS vectors for ``batch_size`` samples.
    ``uniforms`` [batch_size, N_UNIFORMS] in [0,1) may be passed explicitly (tests); otherwise drawn on ``device``."""
    K = len(CONTROL_POINTS)
    if uniforms is None:
        uniforms = torch.rand(batch_size, N_UNIFORMS, generator=generator, device=device, dtype=torch.float32)
    u = uniforms.to(torch.float32)
    n = u.shape[0]
    rng = lambda t, lo, hi: lo + (hi - lo) * t
    base = torch.tensor(CONTROL_POINTS, dtype=torch.float32, device=u.device).view(1, K, 2)
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
    scaled = p["t_scal"].unsqueeze(1) * (p["coord"] + p["vector"] - p["offset"]) + p["offset"]
    t_vector = torch.einsum("blk,bck->bcl", p["rot_mat"], scaled - p["offset_2"]) + p["offset_2"] - p["coord"]
    return p["coord"], t_vector


def solve_system(coord, vector):
    """T [n, 2, K+3] with f(c_i) = c_i + v_i (fp64 solve on the device, result in fp32)."""
    c = coord.double()
    n, K, _ = c.shape
    p = torch.cat([torch.ones(n, K, 1, dtype=c.dtype, device=c.device), c], dim=2)
    d2 = ((p.unsqueeze(2) - p.unsqueeze(1)) ** 2).sum(dim=3)
    r = d2 * torch.log(d2 + 1e-6)
    W = torch.cat([torch.cat([p, r], dim=2),
                   torch.cat([torch.zeros(n, 3, 3, dtype=c.dtype, device=c.device), p.transpose(1, 2)], dim=2)], dim=1)
    tp = torch.cat([c + vector.double(), torch.zeros(n, 3, 2, dtype=c.dtype, device=c.device)], dim=1)
    return torch.linalg.solve(W, tp).transpose(1, 2).contiguous().float()


def ThinPlateSpline(U, coord, vector, out_size=None, n_c=None):
    """U [n,H,W,C] fp32 -> (warped images, None); out_size / n_c are implied by U (kept for signature compatibility)."""
    U = U.contiguous().float()
    n, h, w, c = U.shape
    T = solve_system(coord, vector)
    cd = coord.contiguous().float()
    out = torch.empty_like(U)
    L.call("ups_tps_warp", L.ptr(U), L.ptr(T), L.ptr(cd), L.ptr(out), n, h, w, c, cd.shape[1], L.stream())
    return out, None


def make_tps(views, tps_params, uniforms=None, generator=None):
    """model.py:282-311: views 0 and 1 get independent transforms, the target (if any) gets view0's."""
    B = views[0].shape[0]
    batch = torch.cat(views[:2], dim=0)
    pd = tps_parameters(2 * B, uniforms=uniforms, generator=generator, device=batch.device, **tps_params)
    coord, vector = make_input_tps_param(pd)
    t_images, _ = ThinPlateSpline(batch, coord, vector)
    out = [t_images[:B], t_images[B:]]
    if len(views) > 2:
        t, _ = ThinPlateSpline(views[2], coord[:B], vector[:B])
        out.append(t)
    return out
