"""torch-CPU restatement of the reference training graph (fp32 or fp64).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED except
fill_triangular.  Citations: M = cub/code/SB_model48i/model.py,
N = cub/code/nn.py, Y = cub/code/SB_model48i/train_cub_subset_tps.yaml
(all under /root/reference).  TF-1.14 op semantics follow SURVEY.md Appendix A.

Everything is functional: parameters are an ordered dict  name -> tensor  with the
reference's variable names (``encoder_0/conv2d_3/V`` is HWIO, ``.../b`` is [Cout],
N:40-46, N:644-652); all random draws are explicit inputs (``noise`` dict), so the
HIP path can be compared on identical numbers.  Activations are NHWC.
"""
import math
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

SUBMODULES = ("encoder_0", "encoder_1", "decoder_delta", "decoder_visualize",
              "mi0_discriminator", "mi1_discriminator", "mi_estimator")
DSIZE = 512  # M:10

VGG_WIDTHS = (64, 128, 256, 512, 512)
VGG_DEPTHS = (2, 2, 4, 4, 2)          # convs per block up to block5_conv2 (Keras VGG19, no top)
VGG_BGR_MEAN = (103.939, 116.779, 123.68)


# ----------------------------------------------------------------------------- parameters
def param_rng(seed, name):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def uniform_init(seed, name, shape, bound):
    """N:634-652: V and b ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in))."""
    u = torch.rand(shape, generator=param_rng(seed, name), dtype=torch.float64)
    return ((2.0 * u - 1.0) * bound).to(torch.float32)


class Scope(object):
    """One ``nn.model_arg_scope`` (N:17-23): fresh layer counters per template call,
    variables shared through ``make_template`` (N:26-33) by name."""

    def __init__(self, params, prefix, activation, coords, seed=None):
        self.params, self.prefix, self.activation, self.coords = params, prefix, activation, coords
        self.seed = seed            # not None -> create missing variables
        self.counter = 0

    # N:2123-2154
    @staticmethod
    def add_coordinates(x):
        n, xd, yd, _ = x.shape
        col = torch.arange(yd, dtype=x.dtype) / max(1, xd - 1) * 2 - 1
        row = torch.arange(xd, dtype=x.dtype) / max(1, yd - 1) * 2 - 1
        xx = col.view(1, 1, yd, 1).expand(n, xd, yd, 1)
        yy = row.view(1, xd, 1, 1).expand(n, xd, yd, 1)
        return torch.cat([x, xx, yy], dim=-1)

    # N:617-711
    def conv2d(self, x, num_filters, k=3, stride=1):
        if self.coords:
            x = self.add_coordinates(x)
        name = "{}/conv2d_{}".format(self.prefix, self.counter)
        self.counter += 1
        cin = x.shape[-1]
        if name + "/V" not in self.params:
            assert self.seed is not None, "missing variable " + name
            bound = math.sqrt(1.0 / (cin * k * k))
            self.params[name + "/V"] = uniform_init(self.seed, name + "/V", (k, k, cin, num_filters), bound)
            self.params[name + "/b"] = uniform_init(self.seed, name + "/b", (num_filters,), bound)
        V = self.params[name + "/V"].to(x.dtype)
        b = self.params[name + "/b"].to(x.dtype)
        return conv2d_same(x, V, b, stride)

    def nin(self, x, n):              # N:811-813
        return self.conv2d(x, n, k=1)

    def downsample(self, x, n):       # N:816-817
        return self.conv2d(x, n, k=3, stride=2)

    def activate(self, x):            # N:747-758
        if self.activation is None:
            return x
        if self.activation == "leaky_relu":
            return F.leaky_relu(x, 0.2)
        if self.activation == "relu":
            return F.relu(x)
        if self.activation == "elu":
            return F.elu(x)
        raise NotImplementedError(self.activation)

    def residual_block(self, x, skipin=None, conv=None):   # N:1042-1056 (dropout keep_prob = 1)
        conv = conv or self.conv2d
        c = x.shape[-1]
        residual = x
        if skipin is not None:
            skipin = self.nin(self.activate(skipin), c)
            residual = torch.cat([residual, skipin], dim=-1)
        residual = self.activate(residual)
        residual = conv(residual, c)
        return x + residual

    @staticmethod
    def upsample_linear(x):           # N:834-847: num_units is ignored for method "linear"
        return bilinear_up2(x)

    def upsample(self, x, num_units, method="subpixel"):
        """N:820-849.  "subpixel": conv2d to 4 * num_units (a variable of this scope, CoordConv included) + tf.depth_to_space(2):
        out[b, 2h+i, 2w+j, c] = y[b, h, w, (2i + j) * C + c]; "nearest_neighbor": every pixel repeated 2x2; "linear" ignores
        num_units.  ("conv_transposed" -- weight-normalised deconv2d, N:938-1039 -- is not restated.)"""
        if method == "linear":
            return bilinear_up2(x)
        if method == "nearest_neighbor":
            return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        if method == "subpixel":
            y = self.conv2d(x, 4 * num_units)
            n, h, w, _ = y.shape
            y = y.reshape(n, h, w, 2, 2, num_units).permute(0, 1, 3, 2, 4, 5)
            return y.reshape(n, 2 * h, 2 * w, num_units)
        raise NotImplementedError(method)


def conv2d_same(x, V, b, stride=1):
    """tf.nn.conv2d(x, V, [1,s,s,1], 'SAME') + b, NHWC/HWIO (Appendix A.1)."""
    kh, kw = V.shape[0], V.shape[1]
    h, w = x.shape[1], x.shape[2]
    oh, ow = -(-h // stride), -(-w // stride)
    ph = max((oh - 1) * stride + kh - h, 0)
    pw = max((ow - 1) * stride + kw - w, 0)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = F.conv2d(xn, V.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def bilinear_up2(x):
    """Legacy TF-1 bilinear x2 (Appendix A.2): even outputs copy, odd outputs average
    neighbours, last row/col clamps.  x is NHWC."""
    def up(t, dim):
        n = t.shape[dim]
        nxt = torch.cat([t.narrow(dim, 1, n - 1), t.narrow(dim, n - 1, 1)], dim=dim)
        odd = 0.5 * (t + nxt)
        st = torch.stack([t, odd], dim=dim + 1)
        shp = list(t.shape); shp[dim] = 2 * n
        return st.reshape(shp)
    return up(up(x, 1), 2)


# ----------------------------------------------------------------------------- network builders
def encoder_model(sc, x, out_size, config, extra_resnets):
    """M:38-54."""
    h = sc.conv2d(x, config[0])
    h = sc.residual_block(h)
    for nf in config[1:]:
        h = sc.downsample(h, nf)
        h = sc.residual_block(h)
    for _ in range(extra_resnets):
        h = sc.residual_block(h)
    h = sc.activate(h)
    h = h.mean(dim=(1, 2), keepdim=True)
    return sc.nin(h, out_size)


def single_decoder_model(sc, h, n_out, config, upsample_config):
    """M:134-156 (all shipped upsample methods are "linear")."""
    if isinstance(upsample_config, str):
        upsample_config = [upsample_config] * (len(config) - 1)
    assert len(upsample_config) == len(config) - 1
    h = sc.nin(h, 4 * 4 * config[-1])
    h = h.reshape(-1, 4, 4, config[-1])
    h = sc.conv2d(h, config[-1])
    h = sc.residual_block(h)
    for nf, u in zip(config[-2::-1], upsample_config[-1::-1]):
        h = sc.residual_block(h)
        h = sc.upsample(h, nf, u)
    h = sc.residual_block(h)
    return sc.conv2d(h, n_out)


def hourglass_model(sc, x, config, extra_resnets, n_out=3, upsample_method="linear"):
    """M:80-131 with alpha = pi = None (M:91-92)."""
    hs = []
    h = sc.conv2d(x, config[0])
    h = sc.residual_block(h)
    for nf in config[1:]:
        h = sc.downsample(h, nf)
        h = sc.residual_block(h)
        hs.append(h)
    for _ in range(extra_resnets):
        h = sc.residual_block(h)
    for i, nf in enumerate(config[-2::-1]):
        h = sc.residual_block(h, skipin=hs[-(i + 1)])
        h = sc.upsample(h, nf, upsample_method)
    h = sc.residual_block(h)
    return sc.conv2d(h, n_out)


def discriminator_model(sc, pair):
    """M:159-173."""
    outs = []
    for z in pair:
        h = sc.nin(z, DSIZE)
        for _ in range(4):
            h = sc.residual_block(h, conv=sc.nin)
        h = sc.activate(h)
        h = sc.nin(h, DSIZE)
        outs.append(h)
    return (outs[0] * outs[1]).sum(dim=(1, 2, 3)).unsqueeze(-1)


class Nets(object):
    """The seven templates of M:349-380 sharing one parameter dict."""

    def __init__(self, config, params=None, seed=None):
        self.config = config
        self.params = OrderedDict() if params is None else params
        self.seed = seed

    def _scope(self, name, kw):
        return Scope(self.params, name, kw.get("activation", "relu"), kw.get("coords", False), self.seed)

    def e_pi(self, x):
        kw = self.config["encoder0"]
        z = self.config.get("z0_size", 256)
        return encoder_model(self._scope("encoder_0", kw), x, z + z * (z + 1) // 2, kw["config"], kw["extra_resnets"])

    def e_alpha(self, x):
        kw = self.config["encoder1"]
        return encoder_model(self._scope("encoder_1", kw), x, self.config.get("local_app_size", 64),
                             kw["config"], kw["extra_resnets"])

    def dv(self, z):
        kw = self.config["dv"]
        return single_decoder_model(self._scope("decoder_visualize", kw), z, self.config["n_parts"],
                                    kw["config"], kw.get("upsample_config", "subpixel"))

    def dd(self, x):
        kw = self.config["final_hour"]
        return hourglass_model(self._scope("decoder_delta", kw), x, kw["config"], kw["extra_resnets"],
                               upsample_method=kw.get("upsample_method", "subpixel"))

    def critic(self, name, pair):
        return discriminator_model(self._scope(name, self.config["discriminator"]), pair)

    def dsingle(self, name, z):
        """d_single / d_alpha / d_pi of deepfashion/code/SB_model48c/model.py:320-323 (n_out = 3, DF:132-154)."""
        kw = self.config["d_single"]
        return single_decoder_model(self._scope(name, kw), z, 3, kw["config"], kw.get("upsample_config", "subpixel"))


def is_48c(config):
    """The DeepFashion model variant (deepfashion/code/SB_model48c/model.py): selected by the yaml's
    ``model: ...SB_model48c.model.TrainModel`` or an explicit ``variant: sb48c`` key."""
    return config.get("variant") == "sb48c" or "SB_model48c" in str(config.get("model", ""))


def init_params(config, seed=0):
    """Create every trainable variable by running the templates once on zeros."""
    nets = Nets(config, seed=seed)
    b, s = 1, config["spatial_size"]
    z, a, p = config.get("z0_size", 256), config.get("local_app_size", 64), config["n_parts"]
    with torch.no_grad():
        v = torch.zeros(b, s, s, 3)
        nets.e_pi(v); nets.e_alpha(v)
        nets.dv(torch.zeros(b, 1, 1, z))
        nets.dd(torch.zeros(b, s, s, a + p))
        for name in ("mi0_discriminator", "mi1_discriminator", "mi_estimator"):
            nets.critic(name, (torch.zeros(b, 1, 1, z), torch.zeros(b, 1, 1, a)))
        if is_48c(config):
            nets.dsingle("d_single", torch.zeros(b, 1, 1, z + a))
            nets.dsingle("d_alpha", torch.zeros(b, 1, 1, a))
            nets.dsingle("d_pi", torch.zeros(b, 1, 1, z))
    return nets.params


# ----------------------------------------------------------------------------- distributions
def fill_triangular(x):
    """cub/code/util.py:981-993 (lower): concat(x[n:], reverse(x)).reshape(n,n), lower band."""
    m = x.shape[-1]
    n = int(round(math.sqrt(0.25 + 2.0 * m) - 0.5))
    assert n * (n + 1) // 2 == m
    cat = torch.cat([x[..., n:], torch.flip(x, dims=[-1])], dim=-1)
    return torch.tril(cat.reshape(x.shape[:-1] + (n, n)))


class FullLatent(object):
    """N:1134-1208."""

    def __init__(self, parameters, dim):
        p = parameters.reshape(parameters.shape[0], -1)
        self.dim = dim
        self.mean = p[:, :dim]
        L = fill_triangular(p[:, dim:])
        self.log_diag = torch.diagonal(L, dim1=1, dim2=2)
        rw = torch.sqrt(torch.arange(dim, dtype=p.dtype) + 1.0).view(1, dim, 1)
        L = L / rw
        eye = torch.eye(dim, dtype=p.dtype)
        self.L = L * (1 - eye) + torch.diag_embed(torch.exp(self.log_diag))

    def sample(self, eps, noise_level=1.0, stochastic=True):
        if not stochastic:
            out = self.mean
        else:
            out = self.mean + torch.matmul(self.L, (noise_level * eps).unsqueeze(-1)).squeeze(-1)
        return out.unsqueeze(1).unsqueeze(1)

    def kl(self):
        kl = 0.5 * ((self.L ** 2).sum(dim=2) - 1.0 + self.mean ** 2 - 2.0 * self.log_diag).sum(dim=1)
        return kl.mean()


# ----------------------------------------------------------------------------- part path
def ste(y_hard, y):
    """N:154-168."""
    return (y_hard - y).detach() + y


def hard_max(y):
    """N:134-136."""
    return (y == y.max(dim=-1, keepdim=True).values).to(y.dtype)


def spatial_softmax(x):
    """N:65-71."""
    n, h, w, c = x.shape
    f = x.permute(0, 3, 1, 2).reshape(n * c, h * w)
    return torch.softmax(f, dim=-1).reshape(n, c, h, w).permute(0, 2, 3, 1)


def probs_to_mu_sigma(probs):
    """N:1541-1587, scaling_factor = 1."""
    n, h, w, k = probs.shape
    ys = torch.linspace(-1.0, 1.0, h, dtype=probs.dtype)
    xs = torch.linspace(-1.0, 1.0, w, dtype=probs.dtype)
    mesh = torch.stack([ys.view(h, 1).expand(h, w), xs.view(1, w).expand(h, w)], dim=-1)
    mu = torch.einsum("ijl,aijk->akl", mesh, probs)
    mesh2 = torch.einsum("ijm,ijn->ijmn", mesh, mesh)
    sigma = torch.einsum("ijmn,aijk->akmn", mesh2, probs) - torch.einsum("akm,akn->akmn", mu, mu)
    return mu, sigma


RECT_ORDER = "xy"


def draw_rect(centers, ph, pw, h, w, dtype, order=RECT_ORDER):
    """tfutils.draw_rect -- EXTERNAL, inferred: inclusive c-ph//2..c+ph//2.

    ``order``: how the external helper reads the two columns of ``centers``.  The call sites (M:441-442, 459-460) pass
    ``mu`` from probs_to_mu_sigma, whose columns are (y, x) (N:1570-1576).  "xy" (default): the helper takes
    (x, y) -- i.e. the reference draws each rectangle at the TRANSPOSED location of its part's centroid; "yx": it takes
    (y, x).  Pinned statistically by the reference's step-0 log: run through this graph at P=25, B=8, 128x128, random init,
    patch_loss is 15301 +- 10 under "xy" and 15227 +- 4 under "yx" (rectangle on top of its own part -> 6 % more of the
    part's mass inside); the reference logged 15294.75 (cub/train/log.txt:244; tests/test_oracle.py)."""
    ys = torch.arange(h).view(1, h, 1)
    xs = torch.arange(w).view(1, 1, w)
    iy, ix = (1, 0) if order == "xy" else (0, 1)
    cy = centers[:, iy].view(-1, 1, 1)
    cx = centers[:, ix].view(-1, 1, 1)
    inside = ((ys - cy).abs() <= ph // 2) & ((xs - cx).abs() <= pw // 2)
    return inside.to(dtype)


def patch_mask(sample_hard, gamma, patch_size, order=RECT_ORDER):
    """M:437-445 / M:456-463: spatial softmax of gamma*hard -> mu -> int px -> rectangle [N,H,W,P].
    Returns the rectangles and their centres as (row, column) of the box actually drawn (see draw_rect)."""
    n, h, w, p = sample_hard.shape
    corrected = spatial_softmax(sample_hard * gamma)
    mu, _ = probs_to_mu_sigma(corrected)
    px = torch.trunc(mu.reshape(n * p, 2) * h / 2.0 + h / 2.0).to(torch.int64).detach()
    rect = draw_rect(px, patch_size, patch_size, h, w, sample_hard.dtype, order)
    rc = px.flip(-1) if order == "xy" else px
    return rect.reshape(n, p, h, w).permute(0, 2, 3, 1), rc.reshape(n, p, 2)


# ----------------------------------------------------------------------------- forward graph  (M:313-521)
def forward(params, config, views, noise, lon=1.0, dtype=torch.float32, seed=None):
    """views: dict view0/view1/view0_target [B,S,S,3]; noise: eps_pi0 [7,B,Z], eps_pi1 [B,Z],
    eps_l0/eps_l1 [B,S,S,P] (+ tps_u [2B, tps.N_UNIFORMS] uniforms when use_tps is set)."""
    nets = Nets(config, params, seed)
    o = {}
    df = is_48c(config)
    v0, v1 = views["view0"].to(dtype), views["view1"].to(dtype)
    vt = v0 if df else views["view0_target"].to(dtype)      # DF:253,669: two inputs, the target is view0 itself
    if config.get("use_tps", False):                         # M:334-337, 282-311 (oracle/tps.py: UNVERIFIED semantics)
        from . import tps as TPS
        v0, v1, vt2 = TPS.make_tps((v0, v1, vt), noise["tps_u"].to(dtype), config["tps_parameters"])
        vt = v0 if df else vt2
        o["tps_view0"], o["tps_view1"], o["tps_view0_target"] = v0, v1, vt
    B = v0.shape[0]
    Z = config.get("z0_size", 256)
    gamma = config.get("gamma", 3.0)
    patch = config.get("patch_size", 32)
    test_mode = config.get("test_mode", False)
    stochastic_e0 = not test_mode
    stochastic_l = config.get("stochastic_l", not test_mode)

    # pose (M:382-392)
    d0 = FullLatent(nets.e_pi(v0), Z)
    d1 = FullLatent(nets.e_pi(v1), Z)
    o["z00"], o["z01"] = d0, d1
    # appearance (M:394-397)
    alpha_v0 = nets.e_alpha(v0)
    alpha_v1 = nets.e_alpha(v1)
    z1_indep = torch.flip(alpha_v0, dims=[0])
    # masks (M:399-430)
    eps0 = noise["eps_pi0"].to(dtype)
    pi0 = d0.sample(eps0[0], stochastic=stochastic_e0)
    pi1 = d1.sample(noise["eps_pi1"].to(dtype), stochastic=stochastic_e0)
    o["pi_sample_v0"], o["pi_sample_v1"] = pi0, pi1
    l0_mean = nets.dv(pi0)
    l1_mean = nets.dv(pi1)
    l0 = l0_mean + noise["eps_l0"].to(dtype) if stochastic_l else l0_mean
    l1 = l1_mean + noise["eps_l1"].to(dtype) if stochastic_l else l1_mean
    o["l0_mean"], o["l1_mean"], o["l0"], o["l1"] = l0_mean, l1_mean, l0, l1
    m0 = torch.softmax(l0, dim=-1)
    m1 = torch.softmax(l1, dim=-1)
    o["m0"], o["m1"] = m0, m1
    hard0 = ste(hard_max(m0), m0)                       # M:434-436
    hard1 = ste(hard_max(m1), m1)                       # M:453-455
    o["hard0"], o["hard1"] = hard0, hard1
    if df:      # DF:402-413: no rectangles
        o["rect0"] = o["rect1"] = o["px0"] = o["px1"] = None
    else:
        order = config.get("rect_order", RECT_ORDER)
        o["rect0"], o["px0"] = patch_mask(hard0, gamma, patch, order)   # M:437-445
        o["rect1"], o["px1"] = patch_mask(hard1, gamma, patch, order)   # M:456-463
    o["out_parts_soft"] = torch.softmax(l0_mean, dim=-1)     # M:469
    o["out_parts_hard"] = torch.argmax(o["out_parts_soft"], dim=3)   # M:470
    o["m0_sample_argmax"] = torch.argmax(m0, dim=3)          # M:447

    # part-wise appearance (M:478-480; N:81-113: part-major batch p*B+b)
    parts = v1.unsqueeze(3) * hard1.unsqueeze(4)             # [B,H,W,P,3]  M:176-187
    Bq, H, W, P, C = parts.shape
    xp = parts.permute(3, 0, 1, 2, 4).reshape(P * Bq, H, W, C)
    yp = nets.e_alpha(xp)                                    # [P*B,1,1,A]
    feat = yp.reshape(P, Bq, 1, 1, -1).permute(1, 2, 3, 0, 4).reshape(Bq, P, -1)
    o["local_app_features1"] = feat
    # unpool (M:225-249, 482-484)
    inj = torch.einsum("bhwp,bpf->bhwf", hard0, feat)
    inj = torch.cat([inj, hard0], dim=3)
    o["dd_input"] = inj
    o["generated"] = nets.dd(inj)                            # M:485
    o["target"] = vt
    o["crop_yx"] = noise.get("crop_yx")                      # window corner of perceptual mode "resize256_crop224"

    # critics (M:502-521)
    def smp(i, level=1.0):
        return d0.sample(eps0[i], noise_level=level, stochastic=stochastic_e0)
    o["logit_joint0"] = nets.critic("mi0_discriminator", (smp(1, lon), alpha_v1))
    o["logit_marginal0"] = nets.critic("mi0_discriminator", (smp(2, lon), z1_indep))
    o["logit_joint1"] = nets.critic("mi1_discriminator", (smp(3), alpha_v1))
    o["logit_marginal1"] = nets.critic("mi1_discriminator", (smp(4), z1_indep))
    o["mi_logit_joint"] = nets.critic("mi_estimator", (smp(5), alpha_v1))
    o["mi_logit_marginal"] = nets.critic("mi_estimator", (smp(6), z1_indep))
    if df:
        # DF:491-505: three single decoders on batch item 0, inputs under stop_gradient (two more draws of z_00)
        zj = torch.cat([smp(7), alpha_v1], dim=3)[:1]
        o["global_generated"] = nets.dsingle("d_single", zj.detach())
        o["alpha_generated"] = nets.dsingle("d_alpha", alpha_v1[:1].detach())
        o["pi_generated"] = nets.dsingle("d_pi", smp(8)[:1].detach())
        o["x0"], o["x1"] = v0[:1], v1[:1]
    return o


# ----------------------------------------------------------------------------- perceptual loss (EXTERNAL)
def vgg_params(seed=7, widths=VGG_WIDTHS, depths=VGG_DEPTHS):
    """Stand-in for the Keras VGG19 ImageNet weights (not obtainable here): He-normal, zero bias."""
    p = OrderedDict()
    cin = 3
    for bi, (wd, dp) in enumerate(zip(widths, depths)):
        for ci in range(dp):
            name = "vgg19/block{}_conv{}".format(bi + 1, ci + 1)
            g = param_rng(seed, name)
            std = math.sqrt(2.0 / (9 * cin))
            p[name + "/V"] = (torch.randn((3, 3, cin, wd), generator=g, dtype=torch.float64) * std).float()
            p[name + "/b"] = torch.zeros(wd)
            cin = wd
    return p


def vgg_features(vp, x, depths=VGG_DEPTHS):
    """UNVERIFIED restatement of edflow VGG19Features.extract_features: [-1,1] RGB -> 0..255,
    BGR, minus ImageNet mean; features = input_1 and block{1..5}_conv2 (post-ReLU)."""
    x = (x + 1.0) * 127.5
    x = torch.flip(x, dims=[-1]) - torch.tensor(VGG_BGR_MEAN, dtype=x.dtype)
    feats = [x]
    h = x
    for bi, dp in enumerate(depths):
        if bi > 0:
            h = F.max_pool2d(h.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        for ci in range(dp):
            name = "vgg19/block{}_conv{}".format(bi + 1, ci + 1)
            h = F.relu(conv2d_same(h, vp[name + "/V"].to(x.dtype), vp[name + "/b"].to(x.dtype)))
            if ci == 1:
                feats.append(h)
    return feats


def perceptual_loss(vp, target, generated, mode="native", depths=VGG_DEPTHS, crop=None):
    """sum_l mean|f_l(target) - f_l(generated)| (feature weights 1, gram weight 0: M:608).
    mode: the three readings of edflow's `VGG19Features(original_scale=True)` (M:610-612; edflow source absent, UNVERIFIED):
    "native" (images as they are), "resize256" (bilinear to 256x256), "resize256_crop224" (bilinear to 256x256, then ONE random
    224x224 window for the whole batch and for both images -- tf.random_crop of the concatenated tensor; the window's corner
    `crop` = (y, x) in [0, 32] is an explicit noise input, like every other random draw of the step)."""
    if mode in ("resize256", "resize256_crop224"):
        assert target.shape[1] in (128, 256), "resize256 is restated for 128^2 (legacy bilinear x2) and 256^2 (identity) inputs"
        if target.shape[1] == 128:
            target, generated = bilinear_up2(target), bilinear_up2(generated)
        if mode == "resize256_crop224":
            oy, ox = int(crop[0]), int(crop[1])
            assert 0 <= oy <= 32 and 0 <= ox <= 32
            target, generated = target[:, oy:oy + 224, ox:ox + 224], generated[:, oy:oy + 224, ox:ox + 224]
    elif mode != "native":
        raise ValueError(mode)
    ft = vgg_features(vp, target, depths)
    fg = vgg_features(vp, generated, depths)
    return sum((a - b).abs().mean() for a, b in zip(ft, fg))


# ----------------------------------------------------------------------------- schedules
def make_var(step, spec):
    """edflow make_var; formulas mirrored in-tree at N:1064-1083."""
    o = dict(spec["options"])
    cmin, cmax = o.pop("clip_min", 0.0), o.pop("clip_max", 1.0)
    if spec["var_type"] == "linear":
        v = (o["end_value"] - o["start_value"]) / (o["end"] - o["start"]) * (float(step) - o["start"]) + o["start_value"]
    elif spec["var_type"] == "staircase":
        v = o["stair_factor"] ** ((float(step) - o["start"]) // o["step_size"]) * o["start_value"]
    else:
        raise ValueError(spec["var_type"])
    return float(min(max(v, cmin), cmax))


def make_linear_var(step, start, end, start_value, end_value, clip_min=0.0, clip_max=1.0):
    v = (end_value - start_value) / (end - start) * (float(step) - start) + start_value
    return float(min(max(v, clip_min), clip_max))


# ----------------------------------------------------------------------------- losses  (M:604-932)
def initial_state(config):
    """Non-trainable scalars: lon (M:503), loa/lor (M:890,921), 7 EMAs (M:829-834,861-866)."""
    mi = config["MI"]
    return {"lon": 1.0, "loa": float(mi.get("loa_init", 0.0)), "lor": float(mi.get("lor_init", 7.5)),
            "avg_acc0": 0.5, "avg_acc1": 0.5, "avg_acc_error": 0.0, "avg_loss_dis0": 1.0,
            "avg_loss_dis1": 1.0, "avg_mim": 0.0, "avg_independent_mim": 0.0}


def squared_grad(x):
    """N:1366-1390."""
    xr = torch.cat([x[:, :, 1:], torch.zeros_like(x[:, :, :1])], dim=2)
    xd = torch.cat([x[:, 1:], torch.zeros_like(x[:, :1])], dim=1)
    return (0.25 * (x - xr)) ** 2 + (0.25 * (x - xd)) ** 2


def losses(o, config, state, step, vp, perceptual_mode="native", vgg_depths=VGG_DEPTHS):
    """Returns (losses per optimizer key, log scalars, new state).  State used inside the
    losses is the PRE-update value (Appendix A.15)."""
    log = OrderedDict()
    dt = o["l0"].dtype
    H = o["l0"].shape[1]
    dim = H * H * 3                                           # M:613
    df = is_48c(config)
    rec = perceptual_loss(vp, o["target"], o["generated"], perceptual_mode, vgg_depths, crop=o.get("crop_yx"))
    auto_rec_loss = 1e-3 * 0.5 * dim * rec                    # M:614-619
    log["perceptual"] = rec
    if df:      # DF:672-684
        global_rec = 1e-3 * 0.5 * dim * perceptual_loss(vp, o["x0"], o["global_generated"], perceptual_mode, vgg_depths)
        alpha_rec = 1e-3 * 0.5 * dim * perceptual_loss(vp, o["x1"], o["alpha_generated"], perceptual_mode, vgg_depths)
        pi_rec = 1e-3 * 0.5 * dim * perceptual_loss(vp, o["x0"], o["pi_generated"], perceptual_mode, vgg_depths)

    w_gmrf = make_var(step, config["prior_gmrf_weight"])
    w_ms = make_var(step, config["prior_mumford_sha_weight"])
    w_kl = make_linear_var(step, **config["kl_weight"])
    log["mumford_sha_lambda"] = make_var(step, config["mumford_sha_lambda"])
    log["mumford_sha_alpha"] = make_var(step, config["mumford_sha_alpha"])

    # M:652-657, N:1444-1451
    lm = o["l0_mean"]
    dy = lm[:, 1:] - lm[:, :-1]
    dx = lm[:, :, 1:] - lm[:, :, :-1]
    prior_gmrf = 0.5 * ((dy ** 2).sum(dim=(1, 2, 3)) + (dx ** 2).sum(dim=(1, 2, 3))).mean()
    log["prior_gmrf"] = prior_gmrf; log["prior_gmrf_weight"] = w_gmrf
    log["prior_gmrf_weighted"] = w_gmrf * prior_gmrf

    # M:659-665, M:21-25
    P = o["m0"].shape[-1]
    mask0_kl = sum((m * torch.log(P * m + 1e-20)).sum(dim=-1).mean() for m in (o["m0"], o["m1"]))
    log["mask0_kl_weight"] = w_kl; log["mask0_kl"] = mask0_kl; log["mask0_kl_weighted"] = w_kl * mask0_kl

    # M:667-681 (softmax_cross_entropy_with_logits_v2 back-propagates into labels too)
    log_probs = o["l0"]
    p_labels = torch.softmax(log_probs, dim=-1)
    ef = "cross_entropy" if df else config.get("entropy_func", "cross_entropy")    # DF:744-746: always the STE labels
    if ef == "cross_entropy":
        labels = ste(hard_max(p_labels), p_labels)
    elif ef == "entropy":
        labels = p_labels
    else:
        raise ValueError("unkown entropy_func")
    weakly = (-(labels * torch.log_softmax(log_probs, dim=-1)).sum(dim=-1)).mean()

    # M:683-719
    gamma = config.get("gamma", 3.0)
    if df:      # DF:750-776: no gamma, no rectangle, renormalised maps, squared variances
        c1 = spatial_softmax(torch.softmax(o["l1"], dim=-1))
        c1 = c1 / c1.sum(dim=(1, 2), keepdim=True)
        _, sigma = probs_to_mu_sigma(c1)
        variances = (sigma[:, :, 0, 0] ** 2 + sigma[:, :, 1, 1] ** 2).sum(dim=1).mean()
        for i in range(sigma.shape[1]):
            log["sigma1_{:02d}".format(i)] = sigma[0, i, 0, 0]
            log["sigma2_{:02d}".format(i)] = sigma[0, i, 1, 1]
    else:
        c1 = spatial_softmax(torch.softmax(o["l1"], dim=-1) * gamma) * (1 - o["rect1"]).detach()
        _, sigma = probs_to_mu_sigma(c1)
        variances = (sigma[:, :, 0, 0] + sigma[:, :, 1, 1]).sum(dim=1).mean()
    w_var = make_var(step, config["variance_weight"])
    log["variance_loss_weighted"] = w_var * variances; log["variance_loss"] = variances
    log["variance_weight"] = w_var
    w_weak = make_var(step, config["weakly_superv_loss_weight_p"])
    log["weakly_superv_loss_weight_p"] = w_weak; log["weakly_superv_loss_p"] = weakly
    log["weakly_superv_loss_p_weighted"] = weakly * w_weak

    L = OrderedDict()
    L["encoder_0"] = auto_rec_loss; L["encoder_1"] = auto_rec_loss; L["decoder_delta"] = auto_rec_loss
    extra = OrderedDict()       # loss_k - auto_rec_loss, built explicitly (merged-gradient scheme)

    if df:
        # DF:719-722, 1112-1114, deepfashion/code/nn.py:1388-1391,1451-1455: Mumford-Shah on the noise-free LOGITS,
        # alpha / lambda from the yaml schedules, summed (not squared) per image
        ms_alpha, ms_lambda = log["mumford_sha_alpha"], log["mumford_sha_lambda"]
        prior_ms = torch.clamp(ms_alpha * squared_grad(o["l0_mean"]), max=ms_lambda).sum(dim=(1, 2, 3)).mean()
        log["prior_mumford_sha"] = prior_ms; log["prior_mumford_sha_weight"] = w_ms
        log["prior_mumford_sha_weighted"] = prior_ms * w_ms
        L["encoder_0"] = auto_rec_loss + global_rec           # DF:809-814 (the extra term has no path to these keys:
        L["encoder_1"] = auto_rec_loss + global_rec           #  d_single's input is under stop_gradient)
        L["d_single"], L["d_alpha"], L["d_pi"] = global_rec, alpha_rec, pi_rec
        if not config.get("pretrain", False):                # DF:830-838
            extra["decoder_visualize"] = (w_gmrf * prior_gmrf + prior_ms * w_ms + w_kl * mask0_kl + weakly * w_weak
                                          + w_var * variances)
            L["decoder_visualize"] = auto_rec_loss + extra["decoder_visualize"]
        else:
            L["decoder_visualize"] = auto_rec_loss
    else:
        # M:744-769 (alpha = 1, lambda = 1e-2 hard-coded at the call site)
        g = squared_grad(o["m0"])
        r = torch.clamp(g, max=1.0e-2)
        smooth = torch.where(g < 1.0e-2, r, torch.zeros_like(r))
        contour = torch.where(g >= 1.0e-2, r, torch.zeros_like(r))
        sq = lambda t: (t.sum(dim=(1, 2)) ** 2).sum(dim=1).mean()
        p_mumford_sha = w_ms * sq(r)
        area_cost = 1.0e-12 * sq(o["m0"])
        # M:771-783
        patch_loss = (o["hard0"] * (1 - o["rect0"]).detach()).sum(dim=(1, 2, 3)).mean()
        w_patch = make_var(step, config["patch_loss_weight"])
        log["patch_loss"] = patch_loss; log["patch_loss_weight"] = w_patch
        log["patch_loss_weighted"] = patch_loss * w_patch

        if not config.get("pretrain", False):                    # M:785-797
            extra["decoder_visualize"] = (w_gmrf * prior_gmrf + w_kl * mask0_kl + weakly * w_weak
                                          + w_var * variances + p_mumford_sha + area_cost + patch_loss * w_patch)
            L["decoder_visualize"] = auto_rec_loss + extra["decoder_visualize"]
        else:
            L["decoder_visualize"] = auto_rec_loss

    # M:800-834
    sp = F.softplus
    loss_dis0 = 0.5 * (sp(-o["logit_joint0"]).mean() + sp(o["logit_marginal0"]).mean())
    loss_dis1 = 0.5 * (sp(-o["logit_joint1"]).mean() + sp(o["logit_marginal1"]).mean())
    L["mi0_discriminator"] = loss_dis0
    L["mi1_discriminator"] = loss_dis1
    L["mi_estimator"] = 0.5 * (sp(-o["mi_logit_joint"]).mean() + sp(o["mi_logit_marginal"]).mean())

    def acc(lj, lmarg):
        return float(((lj > 0).sum() + (lmarg < 0).sum()).item()) / (2 * lj.shape[0])
    dis0_acc = acc(o["logit_joint0"], o["logit_marginal0"])
    dis1_acc = acc(o["logit_joint1"], o["logit_marginal1"])
    est_acc = acc(o["mi_logit_joint"], o["mi_logit_marginal"])

    new = dict(state)
    ema = lambda k, v: 0.99 * state[k] + (1.0 - 0.99) * float(v.detach() if torch.is_tensor(v) else v)     # M:28-35
    new["avg_acc0"] = ema("avg_acc0", dis0_acc)
    new["avg_acc1"] = ema("avg_acc1", dis1_acc)
    new["avg_acc_error"] = ema("avg_acc_error", dis1_acc - dis0_acc)
    new["avg_loss_dis0"] = ema("avg_loss_dis0", loss_dis0)
    new["avg_loss_dis1"] = ema("avg_loss_dis1", loss_dis1)

    mi = config["MI"]
    MI_TARGET, MI_SLACK = mi.get("mi_target", 0.125), mi.get("mi_slack", 0.05)
    mim = (o["logit_joint0"]).mean()                          # logit_constraint(real=False), M:855
    ind_mim = (o["logit_joint1"]).mean()                      # M:856-858
    new["avg_mim"] = ema("avg_mim", mim)
    new["avg_independent_mim"] = ema("avg_independent_mim", ind_mim)
    avg_mim = max(0.0, state["avg_mim"]); avg_ind = max(0.0, state["avg_independent_mim"])
    loo = min(max((avg_ind - avg_mim) / (avg_ind + 1e-6), 0.0), 1.0)   # M:869-873
    log["avg_acc_error"] = state["avg_acc_error"]; log["avg_mim"] = avg_mim
    log["avg_independent_mim"] = avg_ind; log["loo"] = loo
    log["lon_gain"] = -loo + 0.025; log["model_lon"] = state["lon"]   # LON_ADAPTIVE False (M:842)

    if config.get("adversarial_regularization", True):       # M:886-909
        loa = state["loa"]; loa_lr = mi.get("loa_lr", 4.0)
        loa_gain = mim - (1.0 - MI_SLACK) * MI_TARGET
        if mi.get("loa_adaptive", True):
            new["loa"] = max(0.0, loa + loa_lr * float(loa_gain))
            active = 1.0 if loa_lr * float(loa_gain) >= -loa else 0.0
            adv = active * (loa * loa_gain + loa_lr / 2.0 * loa_gain ** 2)
        else:
            adv = loa * loa_gain
        L["encoder_0"] = L["encoder_0"] + adv
        extra["encoder_0"] = adv
        log["adversarial_weight"] = loa; log["adversarial_constraint"] = mim
        log["adversarial_weighted_loss"] = adv; log["loa"] = loa; log["loa_gain"] = loa_gain

    if config.get("variational_regularization", True):       # M:913-930
        assert not config.get("test_mode", False)
        bottleneck = o["z00"].kl()
        lor = state["lor"]
        lor_gain = ind_mim - MI_TARGET
        if mi.get("lor_adaptive", True):
            new["lor"] = min(max(lor + mi.get("lor_lr", 0.05) * float(lor_gain), mi.get("lor_min", 1.0)),
                             mi.get("lor_max", 7.5))
        beta_0 = config.get("beta_0", 1.0)
        bw = beta_0 * math.exp(lor) * bottleneck
        L["encoder_0"] = L["encoder_0"] + bw
        extra["encoder_0"] = extra["encoder_0"] + bw if "encoder_0" in extra else bw
        log["bottleneck_weight"] = lor; log["bottleneck_loss"] = bottleneck
        log["bottleneck_weighted_loss"] = bw; log["lor"] = lor
        log["explor"] = beta_0 * math.exp(lor); log["lor_gain"] = lor_gain

    for k in L:
        log["loss_" + k] = L[k]
    log["dis0_accuracy"] = dis0_acc; log["dis1_accuracy"] = dis1_acc; log["est_accuracy"] = est_acc
    log["avg_dis0_accuracy"] = state["avg_acc0"]; log["avg_dis1_accuracy"] = state["avg_acc1"]
    log["avg_loss_dis0"] = state["avg_loss_dis0"]; log["avg_loss_dis1"] = state["avg_loss_dis1"]
    log["mi_constraint"] = mim; log["independent_mi_constraint"] = ind_mim
    if not df:
        log["zr_mumford_sha"] = p_mumford_sha
        log["z_mumford_sha_smoothness_cost"] = sq(smooth); log["z_mumford_sha_contour_cost"] = sq(contour)
        log["z_area_cost"] = area_cost; log["prior_mumford_sha_weight"] = w_ms
    # pieces reused by the merged-gradient scheme
    log["_auto_rec_loss"] = auto_rec_loss
    log["_extra"] = extra
    for k in config.get("fix_weights", []):                   # M:1062-1067
        L.pop(k, None)
    return L, log, new


# ----------------------------------------------------------------------------- per-key gradients + TF Adam
def key_params(params, key):
    """edflow: var_list = [v for v in model.variables if key in v.name]."""
    return [n for n in params if key in n]


def gradients(params, config, views, noise, state, step, vp, dtype=torch.float32,
              perceptual_mode="native", vgg_depths=VGG_DEPTHS, scheme="per_key"):
    """d loss_k / d params_k for every optimizer key (M:739-742, 786-815).

    scheme "per_key": one autograd.grad per key (the literal TF semantics).
    scheme "merged" : identical mathematics with shared backward passes (rec loss walked
    once), used for the CPU baseline timing.
    """
    leaf = OrderedDict((n, p.detach().to(dtype).requires_grad_(True)) for n, p in params.items())
    o = forward(leaf, config, views, noise, lon=state["lon"], dtype=dtype)
    L, log, new_state = losses(o, config, state, step, vp, perceptual_mode, vgg_depths)
    grads = OrderedDict()
    if scheme == "per_key":
        keys = list(L.keys())
        for i, k in enumerate(keys):
            names = key_params(leaf, k)
            gs = torch.autograd.grad(L[k], [leaf[n] for n in names], retain_graph=(i + 1 < len(keys)),
                                     allow_unused=True)
            for n, gg in zip(names, gs):
                grads[n] = torch.zeros_like(leaf[n]) if gg is None else gg
    else:
        rec = log["_auto_rec_loss"]
        rec_keys = [k for k in ("encoder_0", "encoder_1", "decoder_delta", "decoder_visualize") if k in L]
        names = [n for k in rec_keys for n in key_params(leaf, k)]
        gs = torch.autograd.grad(rec, [leaf[n] for n in names], retain_graph=True, allow_unused=True)
        for n, gg in zip(names, gs):
            grads[n] = torch.zeros_like(leaf[n]) if gg is None else gg
        for k in L:
            names = key_params(leaf, k)
            extra = log["_extra"].get(k) if k in rec_keys else L[k]
            if not (torch.is_tensor(extra) and extra.requires_grad):
                for n in names:
                    grads.setdefault(n, torch.zeros_like(leaf[n]))
                continue
            gs = torch.autograd.grad(extra, [leaf[n] for n in names], retain_graph=True, allow_unused=True)
            for n, gg in zip(names, gs):
                base = grads.get(n, torch.zeros_like(leaf[n]))
                grads[n] = base if gg is None else base + gg
    return o, L, log, new_state, grads


def learning_rate(config, step):
    """edflow TFBaseTrainer: linear decay lr -> 0 between lr_decay_begin and lr_decay_end (Y:24-26)."""
    lr = config.get("lr", 1e-4)
    b = config.get("lr_decay_begin", 1000); e = config.get("lr_decay_end", 1001)
    return make_linear_var(step, b, e, lr, 0.0, 0.0, lr)


def init_adam(params):
    return {"t": 0, "m": OrderedDict((n, torch.zeros_like(p)) for n, p in params.items()),
            "v": OrderedDict((n, torch.zeros_like(p)) for n, p in params.items())}


def train_step(params, adam, config, views, noise, state, step, vp, dtype=torch.float32,
               perceptual_mode="native", vgg_depths=VGG_DEPTHS, scheme="per_key",
               beta1=0.5, beta2=0.9, eps=1e-8):
    """One reference training step.  TF Adam (Appendix A.12); betas = edflow defaults (UNVERIFIED)."""
    o, L, log, new_state, grads = gradients(params, config, views, noise, state, step, vp, dtype,
                                            perceptual_mode, vgg_depths, scheme)
    lr = learning_rate(config, step)
    adam["t"] += 1
    t = adam["t"]
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    new_params = OrderedDict()
    trainable = set(n for k in L for n in key_params(params, k))
    with torch.no_grad():
        for n, p in params.items():
            if n not in trainable:
                new_params[n] = p
                continue
            g = grads[n].to(p.dtype)
            adam["m"][n] = beta1 * adam["m"][n] + (1 - beta1) * g
            adam["v"][n] = beta2 * adam["v"][n] + (1 - beta2) * g * g
            new_params[n] = p - lr_t * adam["m"][n] / (adam["v"][n].sqrt() + eps)
    return new_params, adam, new_state, o, L, log, grads


# ----------------------------------------------------------------------------- synthetic inputs
def synthetic_views(config, seed=1234, smooth=True, batch=None):
    B = batch or config["batch_size"]; S = config["spatial_size"]
    g = torch.Generator(); g.manual_seed(seed)
    out = {}
    for k in ("view0", "view1", "view0_target"):
        if smooth:
            x = torch.randn(B, 3, max(S // 8, 2), max(S // 8, 2), generator=g)
            x = torch.tanh(1.5 * F.interpolate(x, size=(S, S), mode="bilinear", align_corners=True))
            out[k] = x.permute(0, 2, 3, 1).contiguous()
        else:
            out[k] = torch.rand(B, S, S, 3, generator=g) * 2 - 1
    return out


def synthetic_noise(config, seed=4321, batch=None):
    B = batch or config["batch_size"]; S = config["spatial_size"]
    Z = config.get("z0_size", 256); P = config["n_parts"]
    g = torch.Generator(); g.manual_seed(seed)
    out = {"eps_pi0": torch.randn(9 if is_48c(config) else 7, B, Z, generator=g), "eps_pi1": torch.randn(B, Z, generator=g),
           "eps_l0": torch.randn(B, S, S, P, generator=g), "eps_l1": torch.randn(B, S, S, P, generator=g)}
    if config.get("use_tps", False):
        from . import tps as TPS
        out["tps_u"] = torch.rand(2 * B, TPS.N_UNIFORMS, generator=g)
    out["crop_yx"] = torch.randint(0, 33, (2,), generator=g)      # drawn last: the earlier draws keep their values
    return out
