"""Sub-network builders of the part-discovery model on top of the HIP operators.

Same structure, variable names and creation order as the reference builders
(cub/code/SB_model48i/model.py: encoder_model 38-54, hourglass_model 80-131, single_decoder_model
134-156, discriminator_model 159-173; arg-scope / naming / weight sharing cub/code/nn.py:17-46), but
every ``activate`` is folded into the load of the consuming convolution and CoordConv channels are an
affine epilogue, so no activated / coordinate-augmented tensor is ever written to HBM.
"""
import math
import os
import zlib
from collections import OrderedDict

import torch

from . import lib as L
from . import switches as SW
from . import ops
from .ops import ConvLayer, round8

SUBMODULES = ("encoder_0", "encoder_1", "decoder_delta", "decoder_visualize",
              "mi0_discriminator", "mi1_discriminator", "mi_estimator")
EXTRA_48C = ("d_single", "d_alpha", "d_pi")      # deepfashion/code/SB_model48c/model.py:320-323, 809-815
DSIZE = 512                     # model.py:10


def is_48c(config):
    """DeepFashion model variant (deepfashion/code/SB_model48c/model.py), selected by the yaml's ``model:`` path or an
    explicit ``variant: sb48c`` key."""
    return config.get("variant") == "sb48c" or "SB_model48c" in str(config.get("model", ""))


def submodules(config):
    return SUBMODULES + (EXTRA_48C if is_48c(config) else ())
VGG_WIDTHS = (64, 128, 256, 512, 512)
VGG_DEPTHS = (2, 2, 4, 4, 2)


def _rng(seed, name):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def uniform_init(seed, name, shape, bound):
    """nn.py:634-652: V and b ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); deterministic per variable name."""
    u = torch.rand(shape, generator=_rng(seed, name), dtype=torch.float64)
    return ((2.0 * u - 1.0) * bound).to(torch.float32)


class ParamBank(object):
    """All trainable variables, grouped per optimizer key into flat fp32 buffers
    (params / grads / Adam m / Adam v) so that the optimizer and the DP all-reduce touch one
    contiguous range per sub-network.  ``specs``: name -> (shape, bound)."""

    def __init__(self, specs, device, seed=0, keys=SUBMODULES):
        self.specs = specs
        self.device = device
        self.groups = OrderedDict()
        self.params, self.grads = OrderedDict(), OrderedDict()
        self.adam_m, self.adam_v = OrderedDict(), OrderedDict()      # per-variable views of the flat Adam slots
        for key in keys:
            names = [n for n in specs if key in n]      # edflow: var_list = [v for v in variables if key in v.name]
            total = sum(int(torch.Size(specs[n][0]).numel()) for n in names)
            flat = {k: torch.zeros(total, dtype=torch.float32, device=device) for k in ("p", "g", "m", "v")}
            off = 0
            for n in names:
                shape = specs[n][0]
                cnt = int(torch.Size(shape).numel())
                self.params[n] = flat["p"][off:off + cnt].view(shape).requires_grad_(True)
                self.grads[n] = flat["g"][off:off + cnt].view(shape)
                self.adam_m[n] = flat["m"][off:off + cnt].view(shape)
                self.adam_v[n] = flat["v"][off:off + cnt].view(shape)
                off += cnt
            self.groups[key] = {"names": names, "flat": flat, "t": 0}
        self.initialize(seed)

    @torch.no_grad()
    def initialize(self, seed):
        for n, (shape, bound) in self.specs.items():
            self.params[n].copy_(uniform_init(seed, n, shape, bound))
        ops.WeightVersion.value += 1

    @torch.no_grad()
    def load(self, state):
        """state: name -> tensor (the oracle / checkpoint format)."""
        for n, t in state.items():
            if n in self.params:
                self.params[n].copy_(t.to(torch.float32))
        ops.WeightVersion.value += 1

    def state(self):
        return OrderedDict((n, p.detach().cpu().clone()) for n, p in self.params.items())


class Act(object):
    """Activation handle: tensor (None in the shape-only dry run) + logical channel count."""
    __slots__ = ("t", "n", "h", "w", "c", "mask", "f8", "fmt", "post", "bits")

    def __init__(self, t, n, h, w, c, mask=None, f8=None, fmt=None, post=False, bits=None):
        # bits: the sign bytes of `t` its producer wrote (ops.SignBits; [n,h,w,ld/8] uint8 or None)
        self.bits = bits
        # fmt = L.F16: `t` holds fp16 in a bf16 container (ops.py module docstring); None: what t.dtype says
        # post: `t` holds act(x) of its scope's activation instead of x (post-activation storage, ops.ConvLayer.in_post)
        self.fmt, self.post = fmt, post
        # mask = (hard, hard_bits, view_f32): `t` is the unmasked view [B,h,w,8] and the handle stands for the n = P*B
        # part images view[b] * hard[b,:,:,p] that the first convolution forms while it loads (ops.conv)
        # f8 = the fp8 copy of act(t) its producer wrote (ops.Fp8: {"t", "act", "slot"}) or None
        self.t, self.n, self.h, self.w, self.c, self.mask, self.f8 = t, n, h, w, c, mask, f8


class Scope(object):
    """One nn.model_arg_scope: fresh counter per template call, variables shared by name."""

    def __init__(self, owner, prefix, activation, coords, fmt=None):
        self.owner, self.prefix, self.coords = owner, prefix, coords
        self.act = L.ACT[activation]
        self.counter = 0
        self.fmt = fmt               # L.F16: every tensor of this scope's forward pass is fp16 (the mask decoder)

    def _layer(self, cin, cout, k, stride, act_in, in_post=False, out_act=L.ACT_NONE):
        name = "{}/conv2d_{}".format(self.prefix, self.counter)
        self.counter += 1
        own = self.owner
        if own.dry:
            cin_v = cin + (2 if self.coords else 0)
            bound = math.sqrt(1.0 / (cin_v * k * k))
            own.specs[name + "/V"] = ((k, k, cin_v, cout), bound)
            own.specs[name + "/b"] = ((cout,), bound)
            return None
        key = (name, act_in, in_post, out_act)
        lay = own.layers.get(key)
        if lay is None:
            lay = ConvLayer(name, own.bank.params[name + "/V"], own.bank.params[name + "/b"], k, stride, self.coords, act_in)
            lay.grad_V, lay.grad_b = own.bank.grads[name + "/V"], own.bank.grads[name + "/b"]
            lay.registry = own.prep
            lay.f16 = self.fmt == L.F16
            lay.in_post, lay.out_act = in_post, out_act
            own.layers[key] = lay
        return lay

    def post_ok(self):
        """Post-activation storage is used for leaky-ReLU scopes (the residual must be recoverable from act(x)), unless switched
        off (`post_activation_storage: False`, A/B runs).  In fp8 mode a producer's fp8 copy of such a tensor is the plain
        quantisation of what it stores (ops.conv_forward)."""
        return self.act == L.ACT_LRELU and self.owner.post_storage

    def conv2d(self, x, cout, k=3, stride=1, act_in=L.ACT_NONE, res=None, res_self=False, out_f32=False, post=False):
        """post: store the output as act(out) -- its only convolution consumer is a residual block / an activated nin of THIS
        scope (the builders below say so); the consumer then runs with the activation already applied."""
        post = bool(post) and self.post_ok() and not out_f32
        in_post = bool(x.post)
        if act_in == L.ACT_ELU:         # materialised: elu is not of the max(x, slope * x) family the kernels fuse into their loads
            if res_self:                # x + conv(elu(x)): the residual is the UN-activated input
                res, res_self = x, False
            if x.t is not None:
                x = Act(ops.EluFn.apply(x.t, self.fmt), x.n, x.h, x.w, x.c, mask=None, fmt=x.fmt)
            act_in = L.ACT_NONE
        if in_post and act_in != self.act:
            raise L.UpsError("{}: a post-activation tensor feeds a convolution that wants {}".format(self.prefix, act_in))
        lay = self._layer(x.c, cout, k, stride, act_in, in_post, self.act if post else L.ACT_NONE)
        ho, wo = ops.same_geometry(x.h, k, stride)[0], ops.same_geometry(x.w, k, stride)[0]
        if lay is None:
            return Act(None, x.n, ho, wo, cout, fmt=None if out_f32 else self.fmt, post=post)
        if ops.Fp8.enabled:     # hand the input's fp8 copy in, ask for one of the output (consumed with this scope's activation)
            ops.Fp8.next_in, ops.Fp8.next_out_act, ops.Fp8.last_out = x.f8, (self.act if self.act != L.ACT_ELU else None), None
        assert x.fmt == self.fmt and (res is None or res.fmt == self.fmt), "tensor format does not match the scope's"
        # a post-activation output feeds an activated convolution of this scope: its producer writes the sign bytes that
        # convolution's input gradient needs (ops.SignBits), and this convolution hands its own input's on
        ops.SignBits.want, ops.SignBits.last = bool(post), None
        t = ops.conv(x.t, lay, res=None if res is None else res.t, res_self=res_self, out_f32=out_f32, mask=x.mask, fmt=self.fmt,
                     res_post=bool(res is not None and res.post), x_bits=x.bits if (act_in != L.ACT_NONE and x.mask is None) else None)
        f8 = ops.Fp8.last_out if ops.Fp8.enabled else None
        ops.Fp8.last_out = None
        return Act(t, x.n, ho, wo, cout, f8=f8, fmt=None if out_f32 else self.fmt, post=post, bits=ops.SignBits.take())

    def nin(self, x, cout, **kw):
        return self.conv2d(x, cout, k=1, **kw)

    def downsample(self, x, cout, post=False):
        return self.conv2d(x, cout, k=3, stride=2, post=post)

    def residual_block(self, x, skipin=None, k=3, post=False):
        """x + conv(act(x [++ nin(act(skip))]))  (nn.py:1042-1056, dropout keep_prob = 1).  post: see conv2d."""
        if skipin is None:
            return self.conv2d(x, x.c, k=k, act_in=self.act, res_self=True, post=post)
        s = self.nin(skipin, x.c, act_in=self.act, post=x.post)        # the concatenated operand is in ONE storage form
        if x.t is None:
            cat = Act(None, x.n, x.h, x.w, 2 * x.c, fmt=self.fmt, post=x.post)
        else:
            assert x.c % 8 == 0 and s.post == x.post
            cat = Act(torch.cat([x.t, s.t], dim=-1), x.n, x.h, x.w, 2 * x.c, fmt=self.fmt, post=x.post)
        return self.conv2d(cat, x.c, act_in=self.act, res=x, post=post)

    def upsample_linear(self, x, post=False):
        post = bool(post) and self.post_ok()
        if x.post:
            raise L.UpsError("{}: bilinear up-sampling of a post-activation tensor".format(self.prefix))
        if x.t is None:
            return Act(None, x.n, 2 * x.h, 2 * x.w, x.c, fmt=self.fmt, post=post)
        if not ops.Fp8.enabled or self.act == L.ACT_ELU:
            ops.SignBits.want, ops.SignBits.last = bool(post), None
            t = ops.BilinearFn.apply(x.t, None, 0, 0.2, self.fmt, self.act if post else 0)
            return Act(t, x.n, 2 * x.h, 2 * x.w, x.c, fmt=self.fmt, post=post, bits=ops.SignBits.take())
        # fp8: the up-sampled tensor feeds this scope's next convolution -- hand it an e4m3 copy of act(y) (and, backwards, the
        # convolution below an e5m2 copy of the gradient); per-call-site scale slots live with the model.  A scope whose FORWARD
        # stays fp16 (the mask decoder: its logits decide the masks, fp8 operands cost IoU) keeps the fp16 / post-activation
        # forward of the bf16 mode and only hands the gradient copy on.
        sites = self.owner.__dict__.setdefault("f8_sites", {})
        site = sites.setdefault("{}/upsample@{}".format(self.prefix, self.counter), {})
        if self.fmt == L.F16:
            ops.SignBits.want, ops.SignBits.last = bool(post), None
            t = ops.BilinearFn.apply(x.t, site, 0, 0.2, self.fmt, self.act if post else 0)
            return Act(t, x.n, 2 * x.h, 2 * x.w, x.c, fmt=self.fmt, post=post, bits=ops.SignBits.take())
        ops.Fp8.last_out = None
        t = ops.BilinearFn.apply(x.t, site, self.act, 0.2)
        f8, ops.Fp8.last_out = ops.Fp8.last_out, None
        return Act(t, x.n, 2 * x.h, 2 * x.w, x.c, f8=f8)

    def upsample(self, x, num_units, method="subpixel", post=False):
        """nn.upsample (nn.py:820-849): "linear" (ignores num_units), "subpixel" (conv2d to 4 * num_units + depth_to_space),
        "nearest_neighbor".  ("conv_transposed" -- the weight-normalised deconv2d of nn.py:938-1039 -- is not built.)"""
        if method == "linear":
            return self.upsample_linear(x, post=post)
        if x.post:
            raise L.UpsError("{}: up-sampling of a post-activation tensor".format(self.prefix))
        if method == "subpixel":
            y = self.conv2d(x, 4 * num_units)            # a variable of this scope (CoordConv included), no activation
            if y.t is None:
                return Act(None, x.n, 2 * x.h, 2 * x.w, num_units, fmt=self.fmt)
            return Act(ops.DepthToSpaceFn.apply(y.t, num_units, self.fmt), x.n, 2 * x.h, 2 * x.w, num_units, fmt=self.fmt)
        if method == "nearest_neighbor":
            if x.t is None:
                return Act(None, x.n, 2 * x.h, 2 * x.w, x.c, fmt=self.fmt)
            return Act(ops.Nearest2xFn.apply(x.t, self.fmt), x.n, 2 * x.h, 2 * x.w, x.c, fmt=self.fmt)
        raise NotImplementedError("upsample method '{}' (linear | subpixel | nearest_neighbor)".format(method))

    def act_mean(self, x):
        assert self.fmt is None, "act_mean has no fp16 form"
        if x.t is None:
            return Act(None, x.n, 1, 1, x.c)
        return Act(ops.ActMeanFn.apply(x.t, self.act, 0.2, bool(x.post)), x.n, 1, 1, x.c)


def encoder_model(sc, x, out_size, config, extra_resnets, out_f32=False):
    """model.py:38-54.  (`post=True`: the tensor's only convolution consumer is an activated one of this scope -- a residual
    block or the activate + mean at the end -- so it is stored post-activation, Scope.conv2d.)"""
    h = sc.conv2d(x, config[0], post=True)
    last = len(config) == 1
    h = sc.residual_block(h, post=last)                       # feeds a downsample (plain) unless it is the last level
    for li, nf in enumerate(config[1:]):
        h = sc.downsample(h, nf, post=True)
        h = sc.residual_block(h, post=li == len(config) - 2)
    for _ in range(extra_resnets):
        h = sc.residual_block(h, post=True)
    h = sc.act_mean(h)
    return sc.nin(h, out_size, out_f32=out_f32)


def single_decoder_model(sc, z, n_out, config, upsample_config, out_f32=True):
    """model.py:134-156 (upsample 'linear' ignores num_units: nn.py:834-847)."""
    if isinstance(upsample_config, str):
        upsample_config = [upsample_config] * (len(config) - 1)
    assert len(upsample_config) == len(config) - 1
    c = config[-1]
    h = sc.nin(z, 4 * 4 * c)
    if h.t is not None:
        assert h.t.shape[-1] == 16 * c
        h = Act(h.t.view(h.n, 4, 4, c), h.n, 4, 4, c, fmt=h.fmt)
    else:
        h = Act(None, h.n, 4, 4, c, fmt=h.fmt)
    h = sc.conv2d(h, c, post=True)
    h = sc.residual_block(h, post=True)                       # next: a residual block in either case
    for nf, u in zip(config[-2::-1], upsample_config[-1::-1]):
        h = sc.residual_block(h)                              # next: the up-sampling (plain input)
        h = sc.upsample(h, nf, u, post=True)
    h = sc.residual_block(h)                                  # next: the plain output convolution
    return sc.conv2d(h, n_out, out_f32=out_f32)


def hourglass_model(sc, x, config, extra_resnets, n_out=3, upsample_method="subpixel"):
    """model.py:80-131 with alpha = pi = None (model.py:91-92)."""
    hs = []
    h = sc.conv2d(x, config[0], post=True)
    # (a block's output that also feeds a downsample keeps the plain form; the skip nin then activates on load)
    h = sc.residual_block(h, post=len(config) == 1)
    for li, nf in enumerate(config[1:]):
        h = sc.downsample(h, nf, post=True)
        h = sc.residual_block(h, post=li == len(config) - 2)
        hs.append(h)
    for _ in range(extra_resnets):
        h = sc.residual_block(h, post=True)
    for i, nf in enumerate(config[-2::-1]):
        h = sc.residual_block(h, skipin=hs[-(i + 1)])
        h = sc.upsample(h, nf, upsample_method, post=True)
    h = sc.residual_block(h)
    return sc.conv2d(h, n_out)


def discriminator_towers(sc, pair):
    """model.py:159-173 up to the two 512-d embeddings (the dot product is done by the caller)."""
    outs = []
    for z in pair:
        h = sc.nin(z, DSIZE, post=True)
        for _ in range(4):
            h = sc.residual_block(h, k=1, post=True)
        h = sc.nin(h, DSIZE, act_in=sc.act)
        outs.append(h)
    return outs


class Nets(object):
    """The seven templates of model.py:349-380 bound to one ParamBank."""

    def __init__(self, config, device, seed=0):
        self.config = config
        self.dry, self.specs, self.layers, self.bank = True, OrderedDict(), {}, None
        self.prep = ops.PrepRegistry()
        # `mask_decoder_dtype: fp16` (default with precision bf16): decoder_visualize's FORWARD tensors and forward weights are
        # fp16 -- its logits decide the part masks, and bf16's 2^-9 per operand / per store leaves 0.7 % logit error after
        # its ~15 layers (mask IoU 0.93-0.97 on confident masks; fp16: 0.09 %, IoU >= 0.998; tests/bf16_emulation_study.py).
        # Gradients, the encoders, the image decoder and the perceptual trunk stay bf16.
        # precision fp8 (round 4): the same -- fp8 operands (2^-4) in the mask decoder's FORWARD cost part-mask IoU 0.98 / 0.90 where
        # north_star asks 0.99, so that forward stays fp16 and the mode spends its fp8 MFMAs on the decoder's two input-gradient
        # passes, on decoder_delta and wherever else an operand arrives as a copy (`mask_decoder_dtype: same` = the round-3 form).
        prec = str(config.get("precision", "bf16")).lower()
        sixteen = ("bf16", "bfloat16", "fp8", "f8", "e4m3")
        md = str(config.get("mask_decoder_dtype", "fp16" if prec in sixteen else "same")).lower()
        if md in ("fp16", "f16", "half") and prec not in sixteen:
            raise ValueError("mask_decoder_dtype: fp16 needs precision: bf16 or fp8")
        self.scope_fmt = {"decoder_visualize": L.F16} if md in ("fp16", "f16", "half") else {}
        # `post_activation_storage` (default on): tensors whose only convolution consumer activates them are stored as act(x)
        # (Scope.conv2d); False restores activation-on-load everywhere (A/B runs, debugging)
        self.post_storage = bool(config.get("post_activation_storage", SW.flag("UPS_POST_ACT")))
        S = config["spatial_size"]
        Z, A, P = config.get("z0_size", 256), config.get("local_app_size", 64), config["n_parts"]
        img = Act(None, 1, S, S, 3)
        self.e_pi(img); self.e_alpha(img)
        self.dv(Act(None, 1, 1, 1, Z, fmt=self.scope_fmt.get("decoder_visualize")))
        self.dd(Act(None, 1, S, S, A + P))
        for name in ("mi0_discriminator", "mi1_discriminator", "mi_estimator"):
            self.critic(name, (Act(None, 1, 1, 1, Z), Act(None, 1, 1, 1, A)))
        if is_48c(config):
            self.dsingle("d_single", Act(None, 1, 1, 1, Z + A))
            self.dsingle("d_alpha", Act(None, 1, 1, 1, A))
            self.dsingle("d_pi", Act(None, 1, 1, 1, Z))
        self.dry = False
        self.bank = ParamBank(self.specs, device, seed, keys=submodules(config))

    def _scope(self, name, kw):
        return Scope(self, name, kw.get("activation", "relu"), kw.get("coords", False), fmt=self.scope_fmt.get(name))

    def e_pi(self, x):
        kw = self.config["encoder0"]
        z = self.config.get("z0_size", 256)
        return encoder_model(self._scope("encoder_0", kw), x, z + z * (z + 1) // 2, kw["config"], kw["extra_resnets"],
                             out_f32=True)

    def e_alpha(self, x):
        kw = self.config["encoder1"]
        return encoder_model(self._scope("encoder_1", kw), x, self.config.get("local_app_size", 64), kw["config"],
                             kw["extra_resnets"])

    def dv(self, z):
        kw = self.config["dv"]
        return single_decoder_model(self._scope("decoder_visualize", kw), z, self.config["n_parts"], kw["config"],
                                    kw.get("upsample_config", "subpixel"))

    def dd(self, x):
        kw = self.config["final_hour"]
        return hourglass_model(self._scope("decoder_delta", kw), x, kw["config"], kw["extra_resnets"],
                               upsample_method=kw.get("upsample_method", "subpixel"))

    def critic(self, name, pair):
        return discriminator_towers(self._scope(name, self.config["discriminator"]), pair)

    def critic_layers(self, name, widths):
        """The ConvLayer objects discriminator_towers would run for inputs of `widths` channels, [[6 layers] per tower] -- the same
        objects (variables, converted copies, gradient views) under the same keys, for the grouped launches of ops.TowersFn.  None
        when the scope does not use the post-activation storage form those launches are written for."""
        sc = self._scope(name, self.config["discriminator"])
        if not sc.post_ok() or sc.coords:
            return None
        towers = []
        for c in widths:
            tw = [sc._layer(c, DSIZE, 1, 1, L.ACT_NONE, False, sc.act)]
            for _ in range(4):
                tw.append(sc._layer(DSIZE, DSIZE, 1, 1, sc.act, True, sc.act))
            tw.append(sc._layer(DSIZE, DSIZE, 1, 1, sc.act, True, L.ACT_NONE))
            towers.append(tw)
        return towers

    def dsingle(self, name, z):
        """d_single / d_alpha / d_pi (SB_model48c:320-323): single_decoder_model with n_out = 3; the image stays in the
        activation dtype because it feeds the perceptual trunk."""
        kw = self.config["d_single"]
        return single_decoder_model(self._scope(name, kw), z, 3, kw["config"], kw.get("upsample_config", "subpixel"),
                                    out_f32=False)


def read_vgg_weights(path):
    """`vgg_weights` file -> {"vgg19/block{b}_conv{c}/V": HWIO kernel, ".../b": bias}.  Accepts .npz or torch files keyed
    either that way or the Keras way ("block1_conv1/kernel", "block1_conv1/bias", with or without a ":0" suffix, or
    "block1_conv1_W" / "block1_conv1_b" as in the keras-applications h5 dumps)."""
    import numpy as np
    raw = dict(np.load(path)) if str(path).endswith(".npz") else torch.load(path, map_location="cpu")
    out = {}
    for k, v in raw.items():
        t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).float()
        k = k[:-2] if k.endswith(":0") else k
        k = k.split("vgg19/")[-1]
        for suf, tgt in (("/kernel", "/V"), ("_W", "/V"), ("/V", "/V"), ("/bias", "/b"), ("_b", "/b"), ("/b", "/b")):
            if k.endswith(suf):
                out["vgg19/" + k[:-len(suf)] + tgt] = t
                break
    return out


# --------------------------------------------------------------------------- perceptual trunk (EXTERNAL, stand-in weights)
class VggTrunk(object):
    """Keras-VGG19 topology up to block5_conv2 with frozen weights (edflow VGG19Features, UNVERIFIED;
    the ImageNet weights are not obtainable offline, so He-normal stand-ins are generated per name --
    ``load`` accepts real Keras kernels in HWIO)."""

    def __init__(self, device, seed=7, widths=VGG_WIDTHS, depths=VGG_DEPTHS, post_storage=True, fp8=True):
        self.depths = depths
        self.layers = []
        self.fp8 = bool(fp8)            # precision fp8: blocks 2-4 on e4m3 operands handed from producer to consumer
        self.f8_sites = {}              # (block, input shape) -> scale slot state of that block's max-pool copy
        cin = 3
        for bi, (wd, dp) in enumerate(zip(widths, depths)):
            blk = []
            for ci in range(dp):
                name = "vgg19/block{}_conv{}".format(bi + 1, ci + 1)
                std = math.sqrt(2.0 / (9 * cin))
                V = (torch.randn((3, 3, cin, wd), generator=_rng(seed, name), dtype=torch.float64) * std).float().to(device)
                b = torch.zeros(wd, device=device)
                first = bi == 0 and ci == 0
                lay = ConvLayer(name, V, b, 3, 1, False, L.ACT_NONE if first else L.ACT_RELU)
                lay.frozen = True
                if post_storage:     # every feature map is stored as relu(y): the next convolution stages it untouched (LDS-DMA
                    lay.out_act = L.ACT_RELU          # patch); the 2x2 max-pool commutes with ReLU, the L1 terms re-apply it
                    lay.in_post = not first           # (idempotent), act' is read off the sign either way
                blk.append(lay)
                cin = wd
            self.layers.append(blk)

    def load(self, state):
        missing = [lay.name for blk in self.layers for lay in blk if lay.name + "/V" not in state or lay.name + "/b" not in state]
        if missing:
            raise KeyError("vgg_weights lacks {}".format(missing))
        for blk in self.layers:
            for lay in blk:
                if tuple(state[lay.name + "/V"].shape) != tuple(lay.V.shape):
                    raise ValueError("{}: kernel {} does not fit {} (HWIO expected)".format(
                        lay.name, tuple(state[lay.name + "/V"].shape), tuple(lay.V.shape)))
                lay.V.copy_(state[lay.name + "/V"]); lay.b.copy_(state[lay.name + "/b"])
                for ent in lay._cache.values():
                    ent["version"] = -1

    def features(self, x_img, act_dtype):
        """x_img [n,H,W,>=3] in [-1,1] -> list of (pre-activation feature, logical channels, act for L1)."""
        x = ops.VggPreFn.apply(x_img, act_dtype)
        feats = [(x, 3, L.ACT_NONE)]
        h = x
        # fp8 mode (round 5): the 2x2 max-pool in front of a block writes the e4m3 copy of its output, every convolution of the block
        # that feeds another convolution writes one of its own, so blocks 2-4 run on fp8 operands (block 1 is thin, block 5's 8x8
        # maps are packed four to a tile: both stay bf16).  VGG_FP8 = False: the trunk stays bf16 (A/B runs; `vgg_fp8` config key)
        f8 = ops.Fp8.enabled and ops.Fp8.PRODUCER and self.fp8
        h8 = None
        hb = None       # sign bytes of h from the convolution that produced it (ops.SignBits): the next one's input gradient reads them
        for bi, blk in enumerate(self.layers):
            if bi > 0:
                if f8:
                    site = self.f8_sites.setdefault((bi, tuple(h.shape)), {})
                    ops.Fp8.last_out = None
                    h = ops.MaxPoolFn.apply(h, site, L.ACT_RELU)     # max-pool commutes with ReLU: pool the pre-activations
                    h8, ops.Fp8.last_out = ops.Fp8.last_out, None
                else:
                    h = ops.MaxPoolFn.apply(h)     # max-pool commutes with ReLU: pool the pre-activations
                hb = None
            for ci, lay in enumerate(blk):
                if f8:
                    ops.Fp8.next_in, ops.Fp8.next_out_act, ops.Fp8.last_out = h8, (L.ACT_RELU if ci + 1 < len(blk) else None), None
                ops.SignBits.want = bool(torch.is_grad_enabled() and lay.out_act and ci + 1 < len(blk))
                ops.SignBits.last = None
                h = ops.ConvFn.apply(h, lay.V, lay.b, None, lay, 0, False, None, None, None, None, None, False,
                                     hb if lay.in_post else None)
                hb = ops.SignBits.take()
                if f8:
                    h8, ops.Fp8.last_out = ops.Fp8.last_out, None
                if ci == 1:
                    feats.append((h, lay.co, L.ACT_RELU))
        return feats

    def loss(self, target_img, generated_img, act_dtype, target_features=None):
        """sum_l mean |f_l(target) - f_l(generated)| (feature weights 1, gram weight 0: model.py:608).
        target_features: f(target) computed earlier (the target is a data input: the trainer evaluates it on a side stream)."""
        if target_features is None:
            with torch.no_grad():
                target_features = self.features(target_img, act_dtype)
        ft = target_features
        fg = self.features(generated_img, act_dtype)
        total = None
        for (a, c, act), (b, _, _) in zip(ft, fg):
            term = ops.L1MeanFn.apply(a, b, c, act)
            total = term if total is None else total + term
        return total
