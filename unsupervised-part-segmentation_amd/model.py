"""``TrainModel`` / ``Trainer``: the edflow-facing surface of the reference
(cub/code/SB_model48i/model.py: TrainModel 251-521, Trainer 570-1068) on the MI355X-native path.

The reference builds a static TF graph and lets ``session.run`` execute it; here ``Trainer.train_step``
plays the role of one ``session.run(train_op)``: forward, the seven per-sub-network losses, the
per-key gradients (each sub-network is updated with the gradient of ITS OWN loss only, model.py:739-742,
786-815), TF-style Adam and the Lagrangian / EMA state updates -- all on device, no host sync.

Tape layout (cuts where the per-key semantics need different upstream gradients):
  A  encoder_0 -> latent parameters            (gets d rec, d adversarial, d bottleneck)
  B  decoder_visualize: z -> logits            (weights: d(rec + priors); input z: d rec only)
  C  hard masks -> parts -> encoder_1 -> unpool -> decoder_delta -> perceptual loss
  D  the three critics
The part path between B and C (soft-max, hard max, moments, rectangles, priors) is not taped at all:
its forward and its fused backward are single HIP kernels.
"""
import ctypes as C
import contextlib
import math
import os
from collections import OrderedDict
from collections.abc import Mapping

import torch

from . import lib as L
from . import switches as SW
from . import nets as N
from . import ops
from . import dist as D
from . import tps as TPS
from .nets import Act
from .schedules import make_var, make_linear_var


PERCEPTUAL_INPUTS = ("native", "resize256", "resize256_crop224")
LATE_JOIN = SW.flag("UPS_LATE_JOIN")      # A/B switch: single rank joins the weight-gradient stream only before Adam
# A/B switch (off: measured neutral, 2 032 / 2 034 against 2 017 / 2 057 img/s): enqueue the mask decoder's forward before the critics
STATE_KERNEL = SW.flag("UPS_STATE_KERNEL")          # A/B switch: the state update as one launch (ups_state_update)
STATE_KEYS = ("avg_acc0", "avg_acc1", "avg_acc_error", "avg_loss_dis0", "avg_loss_dis1", "avg_mim", "avg_independent_mim", "loa", "lor")
CRITIC_STREAMS = SW.flag("UPS_CRITIC_STREAMS")      # A/B switch: the three critics on three side streams
EARLY_ALPHA = SW.flag("UPS_EARLY_ALPHA")      # A/B switch: appearance code on "aux" beside the pose encoder
JOIN_TIMING = SW.flag("UPS_JOIN_TIMING")
# data parallel: bucket all-reduces are enqueued from the weight-gradient stream's position instead of after a join of the launching
# stream with it (UPS_DP_SIDE_LAUNCH=0: the round-4 form, A/B runs)
DP_SIDE_LAUNCH = SW.flag("UPS_DP_SIDE_LAUNCH")
EARLY_ADAM = SW.flag("UPS_EARLY_ADAM")    # A/B switch: ... and queues each key's Adam behind its weight gradients


def _scalar(v, device):
    return torch.tensor(float(v), dtype=torch.float32, device=device)


class TrainModel(object):
    """Mirror of model.py:251-280: ``inputs``, ``outputs``, ``variables``, ``n_parts``."""

    def __init__(self, config, device=None, seed=None):
        self.config = config
        if not torch.cuda.is_available():
            raise L.UpsError("TrainModel needs a HIP device: the product path has no CPU fallback")
        L.load()
        self.device = torch.device(device if device is not None else "cuda:{}".format(torch.cuda.current_device()))
        self.pretty = config.get("use_pretty", False)
        self.n_parts = config.get("n_parts")
        self.use_tps = config.get("use_tps", False)             # model.py:334-337: in-graph TPS augmentation (tps.py, UNVERIFIED)
        self.tps_parameters = dict(config.get("tps_parameters") or {}) if self.use_tps else None
        prec = str(config.get("precision", "bf16")).lower()
        self.act_dtype = torch.float32 if prec in ("fp32", "f32", "float32") else torch.bfloat16
        # precision: fp8 (BASELINE config #5) = bf16 tensors, forward and input gradient of the wide 3x3 / stride-1 convolutions
        # with e4m3 / e5m2 MFMA operands and fp32 accumulation (ops.Fp8); weight gradients stay on the bf16 kernels
        # the fp8 state (scale slots, hand-off, switches, counters) is an object of THIS model; `fp8_copy_only: False` lets every
        # eligible layer convert its operand in the kernel (one-step runs: no producer has a delayed scale yet)
        self.fp8 = ops.Fp8.activate(ops.Fp8State(prec in ("fp8", "f8", "e4m3"), config.get("fp8_copy_only")))
        self.patch_size = config.get("patch_size", 32)
        self.df = N.is_48c(config)          # DeepFashion SB_model48c variant (two inputs, no rectangles, extra decoders)
        self.nets = N.Nets(config, self.device, seed if seed is not None else config.get("seed", 0))
        self.bank = self.nets.bank
        self._last = {}

    # model.py:260-263
    @property
    def inputs(self):
        B, S = self.config["batch_size"], self.config["spatial_size"]
        names = ("view0", "view1") if self.df else ("view0", "view1", "view0_target")     # SB_model48c:253
        return {k: (B, S, S, 3) for k in names}

    @property
    def variables(self):
        return self.bank.params

    # model.py:265-280 -- evaluated for the most recent batch given to ``forward``
    @property
    def outputs(self):
        out = dict(self._last)
        if self.use_tps:
            out.update(getattr(self, "_tps", {}))        # model.py:272-279
        return out

    def to_act(self, x_f32, fmt=None, out=None):
        """fp32 [n,H,W,c] -> activation dtype, 8-padded channels (fmt = L.F16: fp16 in a bf16 container, ops.py).
        out: a contiguous destination of that shape (e.g. one half of a batch that two calls fill: no torch.cat)."""
        x_f32 = x_f32.contiguous()
        if out is None:
            out = torch.empty(x_f32.shape[:-1] + (ops.round8(x_f32.shape[-1]),), dtype=self.act_dtype, device=x_f32.device)
        assert out.is_contiguous() and out.shape[:-1] == x_f32.shape[:-1] and out.shape[-1] == ops.round8(x_f32.shape[-1])
        rows = x_f32.numel() // x_f32.shape[-1]
        L.call("ups_pad_convert", L.ptr(x_f32), x_f32.shape[-1], L.ptr(out), L.dt(out) if fmt is None else fmt, out.shape[-1], rows,
               L.stream())
        return out

    def latent_act(self, z_f32):
        """A latent sample [n,Z] as the mask decoder's input handle (in the decoder's tensor format)."""
        n, Z = z_f32.shape
        fmt = self.nets.scope_fmt.get("decoder_visualize")
        return Act(self.to_act(z_f32.view(n, 1, 1, Z), fmt), n, 1, 1, Z, fmt=fmt)

    def part_images(self, view_act, view_f32, hard, hard_bits):
        """The P*B part images view1[b] * hard1[b,:,:,p] in part-major order (mask_parts + apply_partwise, model.py:176-187,
        nn.py:97-103) as an activation handle for encoder_1.  Where the fused form applies (`fuse_mask_parts`, default on; bf16,
        16-aligned images, P <= 32) the [P*B,S,S,8] tensor is never built: the first convolution masks the view while it loads
        and its input gradient is reduced to d/d hard in the epilogue."""
        B, S, _, P = hard.shape
        if (hard_bits is not None and self.config.get("fuse_mask_parts", True)
                and ops.masked_conv_eligible(self.act_dtype, S, P)):
            return Act(view_act.contiguous(), P * B, S, S, 3, mask=(hard, hard_bits, view_f32))
        return Act(ops.MaskPartsFn.apply(view_f32, hard, self.act_dtype), P * B, S, S, 3)

    @torch.no_grad()
    def forward(self, batch, noise=None):
        """Inference graph (test_mode semantics when ``noise`` is None): fills ``outputs``."""
        cfg = self.config
        ops.Fp8.activate(self.fp8)
        v0 = batch["view0"].to(self.device, torch.float32)
        v1 = batch["view1"].to(self.device, torch.float32)
        B, S = v0.shape[0], v0.shape[1]
        Z, A, P = cfg.get("z0_size", 256), cfg.get("local_app_size", 64), self.n_parts
        img01 = self.to_act(torch.cat([v0, v1], 0))
        pe = self.nets.e_pi(Act(img01, 2 * B, S, S, 3)).t.view(2 * B, -1)
        if noise is None:
            z = pe[:, :Z].contiguous()
        else:
            s0, _ = ops.latent_fwd(pe[:B].contiguous(), noise["eps_pi0"][:1].to(self.device), [1.0], False)
            s1, _ = ops.latent_fwd(pe[B:].contiguous(), noise["eps_pi1"][None].to(self.device), [1.0], False)
            z = torch.cat([s0[0], s1[0]], 0)
        lm = self.nets.dv(self.latent_act(z)).t
        eps = None if noise is None else torch.cat([noise["eps_l0"], noise["eps_l1"]], 0).to(self.device)
        _, m, hard, _, bits = ops.part_softmax(lm, eps, want_bits=P <= 32)
        _, soft, _, amax = ops.part_softmax(lm[:B].contiguous(), None, want_hard=False, want_argmax=True)
        yp = self.nets.e_alpha(self.part_images(img01[B:], v1.contiguous(), hard[B:].contiguous(),
                                                None if bits is None else bits[B:].contiguous())).t
        feat = yp.float().view(P, B, A).permute(1, 0, 2).contiguous()
        inj = ops.UnpoolFn.apply(hard[:B].contiguous(), feat, self.act_dtype)
        gen = self.nets.dd(Act(inj, B, S, S, A + P)).t
        self._last = {"generated": gen[..., :3].float(), "m0_sample": m[:B], "out_parts_hard": amax,
                      "out_parts_soft": soft, "view0_mask00_rgb": mask2rgb(m[:B])}
        return self._last


def mask_colors(n_parts):
    """nn.py:2118-2120 (inferno colour table); needs matplotlib, falls back to a grey ramp."""
    try:
        from matplotlib import pyplot as plt
        import numpy as np
        return torch.tensor(plt.cm.inferno(np.linspace(0, 1, n_parts))[:, :3], dtype=torch.float32)
    except Exception:  # pragma: no cover
        return torch.linspace(0, 1, n_parts).view(-1, 1).repeat(1, 3)


def mask2rgb(mask):
    """nn.py:2067-2089: one-hot(argmax) x colours in [-1,1] (visualisation output only)."""
    P = mask.shape[3]
    col = ((mask_colors(P) - 0.5) * 2).to(mask.device)
    return col[mask.argmax(dim=3)]


class _Step(object):
    """The tensors and constants of one training step, handed from segment to segment (Trainer._step_begin ... _log_thunk)."""


class _LazyLosses(Mapping):
    """What train_step returns: the per-key losses OF THAT STEP, computed when first read and cached (the object keeps the step's
    own thunk, so reading it after later steps still yields this step's values).  A read-only Mapping (dict(x), json via
    dict(x), isinstance(x, Mapping) work)."""

    def __init__(self, trainer):
        self._t, self._thunk = trainer, trainer._lazy_logs
        # (HIP-graph mode computes the values with the step, into buffers every replay overwrites: nothing to defer)
        self._vals = trainer._losses if self._thunk is None else None

    def _d(self):
        if self._vals is None:
            t = self._t
            if t._lazy_logs is self._thunk:
                self._vals = t.losses                   # still the trainer's latest step: materialise there (log_ops share it)
            elif t._done_thunk is self._thunk:
                self._vals = t._losses                  # ... which somebody has already done
            else:
                self._vals = self._thunk()[0]           # later steps have replaced the trainer's view: evaluate this step's own
            self._thunk = None
        return self._vals

    def __getitem__(self, k):
        return self._d()[k]

    def __iter__(self):
        return iter(self._d())

    def __len__(self):
        return len(self._d())


class Trainer(object):
    """Mirror of model.py:570-1068 plus the pieces edflow's TFBaseTrainer supplied (session loop,
    one Adam per loss key over the variables whose name contains the key, logging cadence)."""

    def __init__(self, config, root=None, model=None, **kwargs):
        self.config, self.root, self.model = config, root, model
        self.device = model.device
        self.logger = kwargs.get("logger")
        self.global_step = 0
        self._log_ops, self.img_ops, self.update_ops = OrderedDict(), OrderedDict(), []
        self.world_size = kwargs.get("world_size", 1)
        self.rank = kwargs.get("rank", 0)
        self.process_group = kwargs.get("process_group")
        self.beta1 = config.get("beta1", 0.5)        # edflow TFBaseTrainer defaults (UNVERIFIED)
        self.beta2 = config.get("beta2", 0.9)
        # Global optimizer knobs beyond the yaml (all optimizers alike; defaults = tf.train.AdamOptimizer's epsilon, no warm-up, no
        # clipping).  They exist so that hypotheses about edflow's TFBaseTrainer (source absent) are config keys, not patches:
        # tools/pin_log.py sweeps them against the reference's training log.
        self.adam_eps = float(config.get("adam_eps", 1e-8))
        self.lr_warmup_steps = int(config.get("lr_warmup_steps", 0))       # linear ramp of lr over the first N global steps
        self.grad_clip_norm = float(config.get("grad_clip_norm", 0.0))     # per optimizer key, global L2 norm (0 = off)
        # `probe`: DIAGNOSTIC hooks (tools/pin_log.py, the conditional-pin test) -- never set by a shipped config:
        #   lr_scale: {optimizer key: factor}  that key alone steps with factor * lr
        #   rec_scale: s                        decoder_visualize's gradient sees priors + s * (reconstruction term), M:786-797
        self.probe = dict(config.get("probe") or {})
        unknown = set(self.probe) - {"lr_scale", "rec_scale"}
        if unknown:
            raise ValueError("probe: unknown hook(s) {}".format(sorted(unknown)))
        self.perceptual_input = config.get("perceptual_input", "native")
        if float(config.get("gram_weight", 0.0)) != 0.0:        # model.py:608: default 0.0 in every shipped yaml
            raise NotImplementedError("gram_weight != 0 (Gram-matrix terms of edflow's VGG19Features) is not on the shipped path")
        if config.get("use_pretty", False) or config.get("add_pretty", False):
            raise NotImplementedError("the 'pretty' image discriminator (model.py:190-212) is not used by the shipped configs")
        vw = config.get("vgg_widths", N.VGG_WIDTHS)
        self.vgg = N.VggTrunk(self.device, seed=config.get("vgg_seed", 7), widths=tuple(vw),
                              post_storage=model.nets.post_storage,
                              # (measured, round 5: the trunk on fp8 copies -- 20 more launches per step on e4m3 operands -- runs the
                              # step 1.4 % SLOWER: the pools' and the convolutions' copy emission costs more than blocks 2-4 gain at
                              # 64 images; off by default, `vgg_fp8: True` / UPS_VGG_FP8=1 for measurements)
                              fp8=bool(config.get("vgg_fp8", SW.flag("UPS_VGG_FP8"))))
        # `vgg_weights`: npz / torch file with the Keras VGG19 ImageNet kernels in HWIO (edflow downloads them at run time;
        # they are not obtainable offline).  Without it the perceptual loss runs on seeded He-normal stand-ins: fine for
        # timing and parity, NOT for training a model that should match the reference's part quality -- say so loudly.
        if config.get("vgg_weights"):
            self.vgg.load(N.read_vgg_weights(config["vgg_weights"]))
            self.vgg_pretrained = True
        else:
            self.vgg_pretrained = False
            msg = ("perceptual trunk runs on seeded stand-in weights (no `vgg_weights` in the config): losses are not the "
                   "reference's ImageNet-VGG19 perceptual loss")
            if self.logger:
                self.logger.warning(msg)
            elif root is not None and self.rank == 0:
                import sys
                sys.stderr.write("[WARNING] " + msg + "\n")
        mi = config["MI"]
        d = self.device
        # non-trainable state (model.py:503, 829-834, 861-866, 890, 921) -- device scalars
        self.state = {"lon": _scalar(1.0, d), "loa": _scalar(mi.get("loa_init", 0.0), d),
                      "lor": _scalar(mi.get("lor_init", 7.5), d),
                      "avg_acc0": _scalar(0.5, d), "avg_acc1": _scalar(0.5, d), "avg_acc_error": _scalar(0.0, d),
                      "avg_loss_dis0": _scalar(1.0, d), "avg_loss_dis1": _scalar(1.0, d),
                      "avg_mim": _scalar(0.0, d), "avg_independent_mim": _scalar(0.0, d)}
        self._gen = torch.Generator(device=d)
        self._gen.manual_seed(D.shard_seed(config.get("noise_seed", 4321), kwargs.get("rank", 0)))      # TPS uniforms, crop window
        self._noise = ops.NoiseStream(D.shard_seed(config.get("noise_seed", 4321), kwargs.get("rank", 0)))  # the sampling noise
        self._lazy_logs, self._done_thunk = None, None
        self.switches_report = SW.report()       # the environment switches that differ from their defaults (switches.py), logged once
        if self.logger:
            self.logger.info(self.switches_report)
        self._adam_done, self._step_graph_lr = set(), None
        self._adam_stepped = set()      # keys whose Adam step of the RUNNING training step has been enqueued (cleared when the step ends)
        # `stream_plan` (full | compact | auto): how the step's logical streams map onto HIP streams (ops.Streams.set_plan).  auto =
        # full on one rank, compact under data parallelism, where the collectives' stream needs a hardware queue of its own
        # (measured with a stand-in for the collectives on one GPU: tools/probes/stream_dp.py, profiles/round5_stream_dp.txt)
        plan = str(config.get("stream_plan", SW.value("UPS_STREAM_PLAN"))).lower()
        if plan == "auto":
            plan = "compact" if (self.world_size > 1 or D.FORCE_COLLECTIVES) else "full"
        self.stream_plan = plan
        ops.Streams.set_plan(plan)
        self._reduce_marks = []         # (key, bytes) of the bucket all-reduces of the running step, in launch order
        self._poisoned = None           # set when a step failed half-way through its optimizer updates (_after_failed_step)
        self._losses = OrderedDict((k, None) for k in self.loss_keys())
        self._early, self._early_hooked = {}, False
        self._graph_enabled = bool(config.get("hip_graph", SW.flag("UPS_GRAPH")))
        self._g = None
        self._cap = None                # set while the step is being captured into HIP graphs (_capture_step)

    # ------------------------------------------------------------------ losses / log scalars, materialised on first use
    def _set_logs(self, losses, log):
        self._losses, self._log_ops, self._lazy_logs = losses, log, None

    def _materialize(self):
        if self._lazy_logs is not None:
            thunk, self._lazy_logs = self._lazy_logs, None
            self._losses, self._log_ops = thunk()
            self._done_thunk = thunk

    @property
    def losses(self):
        self._materialize()
        return self._losses

    @property
    def log_ops(self):
        self._materialize()
        return self._log_ops

    @log_ops.setter
    def log_ops(self, value):
        self._log_ops = value

    # ------------------------------------------------------------------ edflow hook surface
    def loss_keys(self):
        keys = list(N.submodules(self.config))
        for k in self.config.get("fix_weights", []):      # model.py:1062-1067
            if k in keys:
                keys.remove(k)
        return keys

    def make_loss_ops(self):
        """model.py:604: returns {optimizer key: loss}.  In this eager design the values are the
        device scalars of the most recent step (None before the first step)."""
        return self.losses

    def get_restore_variables(self):
        """model.py:571-590: name-substring exclude list."""
        vs = list(self.model.variables.keys())
        default_exclude = ["pretty_discriminator", "beta1_power_7", "beta2_power_7", "phase_weight", "grad_weight",
                           "phase_gamma"]
        for name in self.config.get("restore_exclude", default_exclude):
            vs = [v for v in vs if name not in v]
        return vs

    def set_global_step(self, step):
        self.global_step = int(step)

    def initialize(self, checkpoint_path=None):
        """model.py:592-602: lazy restore, missing variables ignored, step parsed from the file name."""
        if checkpoint_path is None:
            return
        from . import tfckpt
        if tfckpt.is_bundle(checkpoint_path):
            return self._initialize_from_tf(checkpoint_path)
        ck = torch.load(checkpoint_path, map_location="cpu")
        keep = set(self.get_restore_variables())
        self.model.bank.load({n: t for n, t in ck["params"].items() if n in keep})
        for key, grp in self.model.bank.groups.items():
            st = ck.get("adam", {}).get(key)
            if st is not None and st["m"].numel() == grp["flat"]["m"].numel():
                grp["flat"]["m"].copy_(st["m"]); grp["flat"]["v"].copy_(st["v"]); grp["t"] = st["t"]
        for k, v in ck.get("state", {}).items():
            if k in self.state:
                self.state[k].fill_(float(v))
        if "noise_offset" in ck:
            self._noise.offset = int(ck["noise_offset"])
        if "gen_state" in ck:
            # The generator is seeded PER RANK (shard_seed) and only rank 0 writes checkpoints: its state is restored on the rank and
            # world size that wrote it only.  Every other rank (and any restore under a different world size) re-seeds with its own
            # shard seed advanced by the restored step, so that the ranks keep drawing DIFFERENT TPS uniforms / crop windows
            # (round-5 advisor: all ranks used to resume with rank 0's stream).
            same = int(ck.get("rank", 0)) == self.rank and int(ck.get("world_size", 1)) == self.world_size
            restored = False
            if same:
                try:
                    self._gen.set_state(ck["gen_state"])
                    restored = True
                except Exception:       # a state written on another device type / torch build
                    pass
            if not restored:
                self._gen.manual_seed(D.shard_seed(self.config.get("noise_seed", 4321), self.rank) + 1000003 * (1 + int(ck.get("global_step", 0))))
        base = os.path.basename(checkpoint_path)
        digits = "".join(ch for ch in base.split("-")[-1] if ch.isdigit())
        # the stored step is authoritative (it agrees with the restored Adam t); the file name is the fallback (model.py:597-601)
        self.set_global_step(int(ck["global_step"]) if "global_step" in ck else (int(digits) if digits else 0))
        if self.logger:
            self.logger.info("Lazily restored from {}".format(checkpoint_path))

    TF_UNNAMED_ORDER = ("lon", "avg_acc0", "avg_acc1", "avg_acc_error", "avg_loss_dis0", "avg_loss_dis1", "avg_mim",
                        "avg_independent_mim", "loa", "lor")

    def tf_unnamed_order(self):
        """The state scalars the REFERENCE graph creates for this config, in creation order (TensorFlow names unnamed variables
        `Variable`, `Variable_1`, ... by it): `loa` exists only under adversarial_regularization (model.py:886-890), `lor` only
        under variational_regularization (model.py:913-921); with one of them off the later names shift down."""
        skip = set()
        if not self.config.get("adversarial_regularization", True):
            skip.add("loa")
        if not self.config.get("variational_regularization", True):
            skip.add("lor")
        return tuple(k for k in self.TF_UNNAMED_ORDER if k not in skip)

    def _initialize_from_tf(self, prefix):
        """A TensorFlow-1.x checkpoint of the reference (``model.ckpt-<step>.index`` + ``.data-*``): variables are matched by
        name exactly as slim.assign_from_checkpoint(ignore_missing_vars=True) does (model.py:597-601), Adam slots
        (``<var>/Adam``, ``<var>/Adam_1``) are taken when present, the step comes from the file name."""
        from . import tfckpt
        import numpy as np
        bundle = tfckpt.read_bundle(prefix)
        params, m, v, _other = tfckpt.to_trainer_state(bundle)
        keep = set(self.get_restore_variables())
        bank = self.model.bank
        loaded = {n: torch.from_numpy(np.ascontiguousarray(a)) for n, a in params.items()
                  if n in keep and n in bank.params and tuple(a.shape) == tuple(bank.params[n].shape)}
        bank.load(loaded)
        with torch.no_grad():
            for n in loaded:
                if n in m and n in v:
                    bank.adam_m[n].copy_(torch.from_numpy(np.ascontiguousarray(m[n])))
                    bank.adam_v[n].copy_(torch.from_numpy(np.ascontiguousarray(v[n])))
        digits = "".join(ch for ch in os.path.basename(prefix).split("-")[-1] if ch.isdigit())
        self.set_global_step(int(digits) if digits else 0)
        for key, grp in bank.groups.items():
            if any(n in loaded and n in m for n in grp["names"]):
                grp["t"] = self.global_step        # one Adam step per global step and key (beta powers are not stored per name)
        if self.logger:
            self.logger.info("Lazily restored {} of {} variables from the TensorFlow checkpoint {}".format(
                len(loaded), len(bank.params), prefix))
        self.restored_from_tf = sorted(loaded)
        # what slim.assign_from_checkpoint would ALSO have restored but cannot be matched here: the reference's unnamed
        # non-trainable scalars (Lagrangian multipliers, EMAs) and the optimizers' beta powers.  Say so instead of silently
        # restarting them (the multipliers / EMAs start from their yaml initial values, Adam's step count from the file name).
        # The reference's graph creates ten unnamed scalar tf.Variables (eight / nine with a regulariser off, tf_unnamed_order),
        # which TensorFlow names by creation order:
        # `Variable` = lon (define_graph, model.py:503), `Variable_1..5` = the EMAs of make_loss_ops in source order
        # (model.py:829-834: avg_acc0, avg_acc1, avg_acc_error, avg_loss_dis0, avg_loss_dis1), `_6`, `_7` = the EMAs of the two MI
        # constraints (model.py:862, 865), `_8` = loa (890), `_9` = lor (921).  A bundle that holds exactly these ten scalars is
        # mapped in that order (UNVERIFIED against a real checkpoint: the shipped ones are Git-LFS stubs); anything else is
        # reported below instead of guessed.
        order = self.tf_unnamed_order()
        unnamed = ["Variable"] + ["Variable_{}".format(i) for i in range(1, len(order))]
        found = [n for n in _other if n == "Variable" or n.startswith("Variable_")]
        self.state_from_tf = {}
        if sorted(found) == sorted(unnamed) and all(np.asarray(_other[n]).size == 1 for n in unnamed):
            for key, n in zip(order, unnamed):
                if key in self.state:
                    self.state[key].fill_(float(np.asarray(_other[n]).reshape(-1)[0]))
                    self.state_from_tf[key] = n
                _other.pop(n)
            if self.logger:
                self.logger.info("Lagrangian state restored from the checkpoint's unnamed variables by creation order: {}".format(
                    ", ".join("{}<-{}".format(k, v) for k, v in self.state_from_tf.items())))
        self.not_restored_from_tf = sorted(n for n in _other if n != "global_step")
        if self.not_restored_from_tf:
            msg = ("TensorFlow checkpoint {}: {} non-trainable / optimizer scalars are NOT restored (unnamed in the reference's graph: "
                   "{} ...): lon / loa / lor and the EMAs restart from their initial values, Adam's bias correction from step {}"
                   .format(prefix, len(self.not_restored_from_tf), ", ".join(self.not_restored_from_tf[:4]), self.global_step))
            if self.logger:
                self.logger.warning(msg)
            else:
                import sys
                sys.stderr.write("[WARNING] " + msg + "\n")

    def export_tf_checkpoint(self, prefix):
        """Write the trainable variables (+ Adam slots) as a TensorFlow tensor bundle the reference's graph can restore."""
        from . import tfckpt
        import numpy as np
        bank = self.model.bank
        tensors = {}
        for n, p in bank.params.items():
            tensors[n] = p.detach().cpu().numpy()
            tensors[n + "/Adam"] = bank.adam_m[n].detach().cpu().numpy()
            tensors[n + "/Adam_1"] = bank.adam_v[n].detach().cpu().numpy()
        tensors["global_step"] = np.asarray(self.global_step, dtype=np.int64)
        # the Lagrangian state under the names TensorFlow gives the reference's ten unnamed scalar variables (creation order:
        # see _initialize_from_tf)
        for i, key in enumerate(self.tf_unnamed_order()):
            if key in self.state:
                tensors["Variable" if i == 0 else "Variable_{}".format(i)] = np.asarray(float(self.state[key]), dtype=np.float32)
        tfckpt.write_bundle(prefix, tensors)

    def save_checkpoint(self, path):
        if self._poisoned:
            raise RuntimeError("refusing to write a checkpoint: " + self._poisoned)
        bank = self.model.bank
        torch.save({"params": bank.state(), "global_step": self.global_step,
                    "adam": {k: {"m": g["flat"]["m"].cpu(), "v": g["flat"]["v"].cpu(), "t": g["t"]}
                             for k, g in bank.groups.items()},
                    "state": {k: float(v) for k, v in self.state.items()},
                    # a resumed run continues the sampling-noise stream instead of replaying steps 0..N's draws
                    "noise_offset": self._noise.offset, "gen_state": self._gen.get_state().cpu(),
                    "rank": self.rank, "world_size": self.world_size}, path)

    # ------------------------------------------------------------------ helpers
    def draw_noise(self, B):
        cfg = self.config
        S, P, Z = cfg["spatial_size"], self.model.n_parts, cfg.get("z0_size", 256)
        r = lambda *s: self._noise.randn(*s, device=self.device)      # the library's own Philox kernel (ops.NoiseStream)
        # (the mask noise of both views as ONE [2B,S,S,P] tensor: what part_softmax takes -- two tensors would be concatenated,
        # 168 MB of copies at B = 64; explicit noise may still come as eps_l0 / eps_l1, the fixtures' form)
        out = {"eps_pi0": r(9 if self.model.df else 7, B, Z), "eps_pi1": r(B, Z), "eps_l": r(2 * B, S, S, P)}
        if self.perceptual_input == "resize256_crop224":     # corner of the step's 224x224 window of the 256x256 images
            out["crop_yx"] = torch.randint(0, 33, (2,), generator=self._gen, device=self.device, dtype=torch.int32)
        return out

    def learning_rate(self):
        cfg = self.config
        lr = cfg.get("lr", 1e-4)
        lr = make_linear_var(self.global_step, cfg.get("lr_decay_begin", 1000), cfg.get("lr_decay_end", 1001), lr, 0.0,
                             0.0, lr)
        if self.lr_warmup_steps > 0:
            lr = lr * min(1.0, (self.global_step + 1.0) / self.lr_warmup_steps)
        return lr

    def _prior(self, view, n, S, P, l, lm, m, hard, px, per_np, sums, w, g_hard=None, dl=None, bwd=False, dl_rec=None):
        d = L.PriorDesc()
        d.n, d.h, d.w, d.P, d.view = n, S, S, P, view
        d.half_h = d.half_w = self.model.patch_size // 2
        if self.model.df:        # SB_model48c:719-776: CE labels always, no gamma, alpha / lambda from the yaml schedules
            d.variant, d.entropy_ce, d.gamma = 1, 1, 1.0
            d.ms_alpha = make_var(self.global_step, self.config["mumford_sha_alpha"])
            d.ms_lambda = make_var(self.global_step, self.config["mumford_sha_lambda"])
            d.w_ms_logits = w.get("msl", 0.0)
        else:
            d.variant = 0
            ef = self.config.get("entropy_func", "cross_entropy")
            if ef not in ("cross_entropy", "entropy"):
                raise ValueError("unkown entropy_func")               # model.py:680 (spelling as in the reference)
            d.entropy_ce = int(ef == "cross_entropy")
            d.gamma = float(self.config.get("gamma", 3.0))
            d.ms_alpha, d.ms_lambda = 1.0, 1.0e-2                  # hard-coded at model.py:744-746
        d.w_kl, d.w_entropy, d.w_ms, d.w_area = w["kl"], w["entropy"], w["ms"], w["area"]
        d.w_patch, d.w_gmrf, d.w_var = w["patch"], w["gmrf"], w["var"]
        g = lambda t: t.data_ptr() if t is not None else None
        d.l, d.l_mean, d.m, d.hard, d.px = g(l), g(lm), g(m), g(hard), g(px)
        d.per_np, d.sums, d.g_hard, d.dl = g(per_np), g(sums), g(g_hard), g(dl)
        d.dl_rec = g(dl_rec)
        L.call("ups_prior_bwd" if bwd else "ups_prior_fwd", C.byref(d), L.stream())

    # ------------------------------------------------------------------ one session.run(train_op)
    def train_step(self, batch, noise=None):
        """One session.run(train_op).  With ``hip_graph: True`` the whole step -- ~1 000 kernel launches on three streams -- is
        captured once into HIP graphs and replayed (one graph on a single GPU; under data parallelism a sequence of graphs cut
        at the collectives); see ``_graph_step`` / ``_capture_step``."""
        ops.Fp8.activate(self.model.fp8)
        if self._poisoned:
            raise RuntimeError("this trainer's state is inconsistent: " + self._poisoned + " -- restore a checkpoint (Trainer.initialize)")
        try:
            return self._train_step(batch, noise)
        except BaseException as e:
            self._after_failed_step(e)
            raise

    def _after_failed_step(self, exc):
        """A step that raises after some optimizer keys have already taken their Adam step on the weight-gradient stream
        (EARLY_ADAM) leaves those keys at step t + 1 and the others at t, with the converted weight copies stale.  Make the
        device state coherent (side streams joined, copies re-converted from whatever the masters now hold) and refuse further
        steps / checkpoints: the run has to restart from its last checkpoint."""
        done = sorted(self._adam_done | self._adam_stepped)     # early (side-stream) steps AND a partly run final Adam loop
        self._adam_done, self._adam_stepped = set(), set()
        ops.Streams.master_busy.clear()
        if not done:
            return
        try:
            ops.Streams.join(self.device, names=("wgrad", "aux", "pre"))
            ops.WeightVersion.value += 1
            self.model.nets.prep.refresh()
            if ops.Fp8.enabled:
                ops.Fp8.after_step()
        except Exception:
            pass
        self._poisoned = ("a training step failed ({}: {}) after the optimizer keys {} had been updated while the others had not"
                          .format(type(exc).__name__, exc, done))

    def _train_step(self, batch, noise=None):
        if self._graph_enabled and not self.model.use_tps:
            # one device scalar carries Adam's bias-corrected step size: usable only while every trained key is at the same
            # Adam step (not after restoring a checkpoint whose keys were trained for different numbers of steps)
            ts = set(self.model.bank.groups[k]["t"] for k in self.loss_keys())
            if len(ts) <= 1:
                return self._graph_step(batch, noise)
        return self._step_impl(batch, noise)

    # ------------------------------------------------------------------ HIP-graph replay of the step
    def _schedule_signature(self):
        """Everything the captured launches bake in by value: the step's schedule constants."""
        cfg, step = self.config, self.global_step
        keys = ("prior_gmrf_weight", "prior_mumford_sha_weight", "variance_weight", "weakly_superv_loss_weight_p",
                "patch_loss_weight", "mumford_sha_alpha", "mumford_sha_lambda")
        sig = tuple(make_var(step, cfg[k]) for k in keys if k in cfg)
        return sig + (make_linear_var(step, **cfg["kl_weight"]), bool(cfg.get("pretrain", False)))

    def _graph_step(self, batch, noise):
        """Static input / noise buffers + a device scalar for Adam's step size; the graph is (re)captured after two eager
        steps and whenever a schedule constant changes (staircases move every few thousand steps).  The small-batch
        configs of the reference (batch 8: ~2 400 launches for 24 ms of GPU work) are launch-bound without it."""
        dev = self.device
        B, S = batch["view0"].shape[0], batch["view0"].shape[1]
        if self._g is not None and tuple(next(iter(self._g["in"].values())).shape[:2]) != (B, S):
            self._g = None                       # the capture bakes in shapes and workspace pointers: start over
        if self._g is None:
            self._g = {"in": {k: torch.empty((B, S, S, 3), dtype=torch.float32, device=dev) for k in self.model.inputs},
                       "noise": {k: torch.empty_like(v) for k, v in self.draw_noise(B).items()},
                       "lr": torch.zeros(1, dtype=torch.float32, device=dev), "graph": None, "sig": None, "eager": 0}
        g = self._g
        for k, buf in g["in"].items():
            buf.copy_(batch[k], non_blocking=True)
        for k, buf in g["noise"].items():
            if noise is None:
                if k == "crop_yx":
                    buf.random_(0, 33, generator=self._gen)
                else:
                    self._noise.fill(buf)
            elif k == "eps_l" and "eps_l" not in noise:          # explicit noise in the fixtures' two-tensor form
                buf[:B].copy_(noise["eps_l0"], non_blocking=True)
                buf[B:].copy_(noise["eps_l1"], non_blocking=True)
            elif k == "crop_yx" and "crop_yx" not in noise:      # explicit noise without a window corner: draw it as usual
                buf.random_(0, 33, generator=self._gen)
            else:
                buf.copy_(noise[k], non_blocking=True)
        t = self.model.bank.groups[self.loss_keys()[0]]["t"] + 1
        lr = self.learning_rate()
        g["lr"].fill_(lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t))
        sig = self._schedule_signature() + (lr > 0,)
        # warm-up: kernel attributes, side streams, allocator pools; fp8: the copy hand-off settles over four steps (first maxima,
        # first copies, unread producers going quiet) and the captured graph freezes whatever it sees
        if g["eager"] < (5 if ops.Fp8.enabled else 2):
            g["eager"] += 1
            out = self._step_impl(g["in"], g["noise"], graph_lr=g["lr"])
            self._after_graph_step()
            return out
        if g["graph"] is None or g["sig"] != sig:
            g["graph"], g["sig"] = self._capture_step(g), sig   # (capturing does not execute: the replay below runs the step)
        self._replay_step(g["graph"])
        self._after_graph_step()
        return _LazyLosses(self)

    # Data parallel: a collective cannot sit inside a captured region on every backend (gloo reduces on the host), and the
    # buckets should start their all-reduce as early as in the eager step.  The step is therefore captured as a SEQUENCE of
    # graphs, cut wherever the eager step talks to the other ranks: after the backward segment of each group of optimizer keys
    # (bucket all-reduce, asynchronous, overlapping the next segments) and before the Lagrangian / EMA update (the averaged
    # batch-mean scalars).  One python pass records all segments (the tape objects simply live on between them; the graphs
    # share one memory pool and are always replayed in capture order); at replay the collectives run eagerly in between.
    def _capture_step(self, g):
        dev = self.device
        torch.cuda.synchronize(dev)
        cap = {"graphs": [], "bounds": [], "pool": torch.cuda.graph_pool_handle(), "stream": torch.cuda.Stream(dev), "cur": None}
        cap["stream"].wait_stream(torch.cuda.current_stream(dev))
        self._cap = cap
        try:
            with torch.cuda.stream(cap["stream"]):
                self._segment_begin()
                self._step_impl(g["in"], g["noise"], graph_lr=g["lr"])
                self._segment_end()
        except BaseException:
            # leave the stream usable for the eager trainer: end the open capture (its graph is discarded), rejoin the side
            # streams, and stop trying to capture
            try:
                with torch.cuda.stream(cap["stream"]):
                    if cap["cur"] is not None:
                        ops.Streams.join(dev, names=("wgrad", "aux1", "aux2", "aux", "pre"))
                        cap["cur"].capture_end()
            except Exception:
                pass
            self._graph_enabled, self._g = False, None
            raise
        finally:
            self._cap = None
        torch.cuda.current_stream(dev).wait_stream(cap["stream"])
        torch.cuda.synchronize(dev)
        return cap

    def _segment_begin(self):
        cap = self._cap
        cap["cur"] = torch.cuda.CUDAGraph()
        cap["cur"].capture_begin(pool=cap["pool"])

    def _segment_end(self):
        cap = self._cap
        ops.Streams.join(self.device, names=("wgrad", "aux"))      # every forked stream rejoins before the capture ends
        cap["cur"].capture_end()
        cap["graphs"].append(cap["cur"])
        cap["cur"] = None

    def _boundary(self, kind, payload):
        """Called from inside the step while it is being captured: close the running segment, note what has to happen between
        it and the next one (kind "grads": all-reduce these keys' buckets; "scalars": average this tensor), open the next."""
        self._segment_end()
        self._cap["bounds"].append((kind, payload))
        self._segment_begin()

    def _replay_step(self, cap):
        bank = self.model.bank
        last_grads = max([i for i, (kind, _) in enumerate(cap["bounds"]) if kind == "grads"], default=-1)
        handles = []
        for i, graph in enumerate(cap["graphs"]):
            graph.replay()
            if i < len(cap["bounds"]):
                kind, payload = cap["bounds"][i]
                if kind == "grads":
                    for k in payload:
                        handles.append(D.allreduce_bucket(bank.groups[k]["flat"]["g"], self.world_size, self.process_group))
                    if i == last_grads:
                        D.wait_all(handles)             # the next segment is the optimizer
                else:
                    D.average_scalars(payload, self.world_size, self.process_group)

    def _after_graph_step(self):
        # (WeightVersion is not bumped: the step's own batched weight_prep has already refreshed every converted copy)
        for k in self.loss_keys():               # keys under `fix_weights` are not stepped (as in eager mode)
            self.model.bank.groups[k]["t"] += 1
        self.global_step += 1

    # ------------------------------------------------------------------ the step, segment by segment.  `c` (a _Step) carries the
    # step's tensors from one segment to the next; the order of the calls in _step_impl IS the order of the launches.
    def _step_begin(self, batch, noise):
        """Inputs (+ in-graph TPS, model.py:334-337), noise, schedule constants; starts the perceptual features of the TARGET on
        the "pre" stream (they depend on the data only and fill the bubbles of the small layers at the network ends)."""
        cfg, model, dev = self.config, self.model, self.device
        c = _Step()
        c.keys, c.step, c.df, c.T = self.loss_keys(), self.global_step, model.df, model.act_dtype
        df = c.df
        v0 = batch["view0"].to(dev, torch.float32).contiguous()
        v1 = batch["view1"].to(dev, torch.float32).contiguous()
        vt = v0 if df else batch["view0_target"].to(dev, torch.float32).contiguous()    # SB_model48c:669: target = view0
        if model.use_tps:           # model.py:334-337, 282-311
            tu = None if noise is None or "tps_u" not in noise else noise["tps_u"].to(dev, torch.float32)
            aug = TPS.make_tps([v0, v1] if df else [v0, v1, vt], model.tps_parameters, uniforms=tu, generator=self._gen)
            v0, v1 = aug[0], aug[1]
            vt = v0 if df else aug[2]
            model._tps = {"tps_view0": v0, "tps_view1": v1, "tps_view0_target": vt}
        c.v0, c.v1, c.vt = v0, v1, vt
        c.B, c.S = v0.shape[0], v0.shape[1]
        c.Z, c.A, c.P = cfg.get("z0_size", 256), cfg.get("local_app_size", 64), model.n_parts
        c.gamma = float(cfg.get("gamma", 3.0))
        c.half = model.patch_size // 2
        c.main_stream = torch.cuda.current_stream(dev)
        # `perceptual_input` (the three readings of edflow VGG19Features(original_scale=True), model.py:610-612, UNVERIFIED):
        # native | resize256 (128x128 inputs are up-sampled x2 with the legacy bilinear kernel, 256x256 inputs already have that
        # size) | resize256_crop224 (then ONE random 224x224 window for the whole batch, target and reconstruction alike;
        # its corner is the explicit noise input `crop_yx`, int32 [2] on the device)
        c.pmode = self.perceptual_input
        if c.pmode not in PERCEPTUAL_INPUTS:
            raise NotImplementedError("perceptual_input: {} ({})".format(c.pmode, " | ".join(PERCEPTUAL_INPUTS)))
        c.resize2x = False
        if c.pmode != "native":
            if c.S not in (128, 256):
                raise NotImplementedError("perceptual_input: {} is restated for 128x128 and 256x256 inputs only (got {})".format(c.pmode, c.S))
            c.resize2x = c.S == 128
        if noise is None:
            noise = self.draw_noise(c.B)
        c.crop_yx = None
        if c.pmode == "resize256_crop224":
            if df:
                raise NotImplementedError("perceptual_input: resize256_crop224 is restated for the SB_model48i variants only")
            if "crop_yx" not in noise:          # explicit (fixture-style) noise without a window corner: draw it as usual
                noise = dict(noise, crop_yx=torch.randint(0, 33, (2,), generator=self._gen, device=dev, dtype=torch.int32))
            c.crop_yx = noise["crop_yx"].to(dev, torch.int32).contiguous()
        c.ft_pre, c.ft_ready = None, None
        if ops.Streams.enabled:
            pre = ops.Streams.get("pre", dev)
            # (Letting this block start before the previous step's tail -- it depends on the batch alone -- measured neutral twice,
            # UPS_PRE_FREE in round 5: docs/design/negative_results.md; retired in round 6.)
            pre.wait_stream(c.main_stream)
            with torch.cuda.stream(pre), torch.no_grad():
                tgt_pre = vt if c.pmode == "native" else self._perceptual_view(c, model.to_act(vt))
                c.ft_pre = self.vgg.features(tgt_pre, c.T)
            c.ft_ready = pre.record_event()
        c.noise = {k: v.to(dev, torch.float32).contiguous() for k, v in noise.items() if k != "crop_yx"}
        c.st = self.state
        # ---- schedule constants of this step (model.py:621-646, 709-726, 774-779)
        step = c.step
        c.w_gmrf = make_var(step, cfg["prior_gmrf_weight"])
        c.w_ms = make_var(step, cfg["prior_mumford_sha_weight"])
        c.w_kl = make_linear_var(step, **cfg["kl_weight"])
        c.w_var = make_var(step, cfg["variance_weight"])
        c.w_weak = make_var(step, cfg["weakly_superv_loss_weight_p"])
        c.w_patch = 0.0 if df else make_var(step, cfg["patch_loss_weight"])
        c.pretrain = bool(cfg.get("pretrain", False))
        mi = cfg["MI"]
        c.mi, c.MI_TARGET, c.MI_SLACK = mi, mi.get("mi_target", 0.125), mi.get("mi_slack", 0.05)
        c.beta_0 = cfg.get("beta_0", 1.0)
        c.var_reg = cfg.get("variational_regularization", True)
        return c

    def _perceptual_view(self, c, x_act):
        """What the perceptual trunk sees of an image in the activation layout [n,S,S,8] (`perceptual_input`)."""
        if c.resize2x:
            x_act = ops.BilinearFn.apply(x_act)
        if c.crop_yx is not None:
            x_act = ops.CropFn.apply(x_act, c.crop_yx, 224, 224)
        return x_act

    def _fwd_pose(self, c):
        """A forward: pose encoder + full-covariance latent (model.py:382-409); nine / seven draws of z_0, one of z_1."""
        model, nets = self.model, self.model.nets
        B, S = c.B, c.S
        c.img01 = torch.empty((2 * B, S, S, 8), dtype=c.T, device=self.device)     # both views, each converted into its half
        model.to_act(c.v0, out=c.img01[:B])
        model.to_act(c.v1, out=c.img01[B:])
        if EARLY_ALPHA:
            self._alpha(c)
        c.pe = nets.e_pi(Act(c.img01, 2 * B, S, S, 3)).t                  # fp32 [2B,1,1,NP], taped
        c.pe2 = c.pe.detach().view(2 * B, -1)
        c.pe_v0, c.pe_v1 = c.pe2[:B].contiguous(), c.pe2[B:].contiguous()
        lon = 1.0                                                         # LON_ADAPTIVE = False (model.py:842): lon stays 1
        c.levels0 = [1.0, lon, lon, 1.0, 1.0, 1.0, 1.0]                   # draw order model.py:406,506,509,512,514,518,520
        if c.df:
            c.levels0 += [1.0, 1.0]                                       # SB_model48c:492,502: two more draws of z_00
        c.samples0, c.kl_rows = ops.latent_fwd(c.pe_v0, c.noise["eps_pi0"], c.levels0, True)
        c.samples1, _ = ops.latent_fwd(c.pe_v1, c.noise["eps_pi1"][None], [1.0], False)

    def _alpha(self, c):
        """Appearance code of the whole views: it feeds the critics only -> forward only (model.py:394-397).  It depends on the
        images alone, so it is enqueued on the "aux" stream BEFORE the pose encoder goes to the launching stream: the two encoders
        run side by side, and the critics can start the moment the latent samples exist."""
        nets, B, S = self.model.nets, c.B, c.S

        def run():
            with torch.no_grad():
                c.alpha = nets.e_alpha(Act(c.img01, 2 * B, S, S, 3)).t            # [2B,1,1,A]
                c.alpha_in = torch.cat([c.alpha[B:], torch.flip(c.alpha[:B], dims=[0])], 0).contiguous()
        if ops.Streams.enabled:
            aux = ops.Streams.get("aux", self.device)
            aux.wait_stream(c.main_stream)
            with torch.cuda.stream(aux):
                run()
        else:
            run()

    def _critics(self, c, forked_at=None):
        """D: the three critics (model.py:502-521, 800-866), their own gradients, the adversarial gradient d adv / d z_joint0 for
        encoder_0 (model.py:886-909) and, for SB_model48c, the three single-sample decoders.  They depend on the latent samples
        and on the appearance code of the whole views only, and the main path needs them again at the encoder_0 backward: the
        whole block (72 tiny GEMMs that cannot fill the chip) runs on the "aux" stream beside segments B and C."""
        cfg, model, nets, bank = self.config, self.model, self.model.nets, self.model.bank
        B, S, Z, A, keys, st, mi = c.B, c.S, c.Z, c.A, c.keys, c.st, c.mi
        if not EARLY_ALPHA:
            with torch.no_grad():
                c.alpha = nets.e_alpha(Act(c.img01, 2 * B, S, S, 3)).t            # [2B,1,1,A]
                c.alpha_in = torch.cat([c.alpha[B:], torch.flip(c.alpha[:B], dims=[0])], 0).contiguous()
        alpha, alpha_in = c.alpha, c.alpha_in
        crit = {}
        c.adv, c.g_adv = None, None
        names = ("mi0_discriminator", "mi1_discriminator", "mi_estimator")

        def one(ci, name):
            pi_in = model.to_act(torch.cat([c.samples0[1 + 2 * ci], c.samples0[2 + 2 * ci]], 0).view(2 * B, 1, 1, Z))
            pi_in.requires_grad_(name == "mi0_discriminator")
            h_pi, h_al = nets.critic(name, (Act(pi_in, 2 * B, 1, 1, Z), Act(alpha_in, 2 * B, 1, 1, A)))
            # dot product of the two embeddings, logistic losses, accuracy and the mean joint logit: one launch (+ one backward)
            loss, acc, mean_joint = ops.CriticHeadFn.apply(h_pi.t, h_al.t, B, N.DSIZE)
            crit[name] = (loss, mean_joint, acc, pi_in)
            if ci == 0:
                c.mim = mean_joint                                            # logit_constraint(real=False), model.py:855
                if cfg.get("adversarial_regularization", True):               # model.py:886-909
                    loa, loa_lr = st["loa"], mi.get("loa_lr", 4.0)
                    loa_gain = c.mim - (1.0 - c.MI_SLACK) * c.MI_TARGET
                    if mi.get("loa_adaptive", True):
                        active = (loa_lr * loa_gain.detach() >= -loa).float()
                        c.adv = active * (loa * loa_gain + loa_lr / 2.0 * loa_gain ** 2)
                    else:
                        c.adv = loa * loa_gain
                    if "encoder_0" in keys:
                        with ops.skip_wgrad():
                            c.g_adv = torch.autograd.grad([c.adv], [pi_in], retain_graph=True)[0].float().view(2 * B, Z)[:B]
            elif ci == 1:
                c.ind_mim = mean_joint
            if name in keys:
                torch.autograd.grad([loss], [bank.params[n] for n in bank.groups[name]["names"]])

        # the three critics are independent chains of ~80 launches of a few blocks each; behind a main stream whose kernels fill
        # every CU each of those launches waits for a slot, so the chains run side by side on three streams (all joined into "aux")
        cur = torch.cuda.current_stream(self.device)
        # (not in a captured step: replayed from a HIP graph the three branches cost more than they save -- 1 887 against 1 919 img/s
        # with one critic stream, eager 1 979 -- so a capture keeps the one-stream form)
        multi = ops.Streams.enabled and ops.Streams.on_aux(self.device) and CRITIC_STREAMS and self._step_graph_lr is None
        # (the two extra critic streams stay created even when the critics run as grouped launches: which streams share a hardware
        # queue depends on the creation order, and the order without them sits in the slower cluster -- docs/design/negative_results.md)
        sides = [cur] + ([ops.Streams.get("aux{}".format(i), self.device) for i in (1, 2)] if multi else [cur, cur])
        for sd in sides[1:]:
            if sd is not cur:
                if forked_at is not None:          # (the launching stream has moved on to the mask decoder: fork where the samples exist)
                    sd.wait_event(forked_at)
                else:
                    sd.wait_stream(c.main_stream)  # forked from the launching stream (a HIP-graph capture wants first-level forks) ...
                sd.wait_stream(cur)                # ... and behind the appearance code
        if not self._critics_grouped(c, names, crit, alpha_in):
            for ci, name in enumerate(names):
                with torch.cuda.stream(sides[ci]):
                    one(ci, name)
        if c.df:
            # SB_model48c:491-505, 672-684, 812-814: three single-sample decoders on batch item 0 (inputs under
            # stop_gradient), each with its own perceptual loss and optimizer key; one per side stream, behind that stream's critic
            scale = 1e-3 * 0.5 * (S * S * 3)

            def a1():
                return alpha[B:B + 1, ..., :A].float().reshape(1, A)
            ins = {"d_single": (lambda: torch.cat([c.samples0[7][:1], a1()], 1), c.v0[:1], Z + A),
                   "d_alpha": (a1, c.v1[:1], A), "d_pi": (lambda: c.samples0[8][:1], c.v0[:1], Z)}
            for i, (name, (zin, tgt, cz)) in enumerate(ins.items()):
                with torch.cuda.stream(sides[i]):
                    g_img = nets.dsingle(name, Act(model.to_act(zin().view(1, 1, 1, cz)), 1, 1, 1, cz)).t
                    lss = scale * self.vgg.loss(tgt.contiguous(), g_img, c.T)
                    crit[name] = (lss, g_img)
                    if name in keys:
                        torch.autograd.grad([lss], [bank.params[n] for n in bank.groups[name]["names"]])
        # ("aux1" / "aux2" are joined wherever "aux" is: ops.Streams.join)
        c.crit = crit
        c.loss_dis0, _, c.acc0, _ = crit["mi0_discriminator"]
        c.loss_dis1, _, c.acc1, _ = crit["mi1_discriminator"]
        c.loss_est, _, c.acc_est, _ = crit["mi_estimator"]

    def _critics_grouped(self, c, names, crit, alpha_in):
        """The three critics through ops.TowersFn (round 5): six towers of six 1x1 layers as 6 forward launches, 5 + 1 input-gradient
        launches per backward call and ONE launch for all 72 weight / bias gradients, on the current ("aux") stream -- against ~210
        launches of a few blocks each on three streams.  The same graph as `one` above: the adversarial term differentiates critic
        0's pi tower alone (skip_wgrad, retain_graph), the critics' own losses are differentiated together (their variables are
        disjoint, so the sum's gradient is each loss's).  Returns False when the towers do not have the form the grouped launches
        take (fp32 / non-leaky / materialised storage): the caller then runs the generic path."""
        cfg, model, nets, bank = self.config, self.model, self.model.nets, self.model.bank
        B, Z, A, keys, st, mi = c.B, c.Z, c.A, c.keys, c.st, c.mi
        if not ops.TOWERS:
            return False
        towers = []
        for name in names:
            tw = nets.critic_layers(name, (Z, A))
            if tw is None:
                return False
            towers += tw
        pi_ins = [model.to_act(torch.cat([c.samples0[1 + 2 * ci], c.samples0[2 + 2 * ci]], 0).view(2 * B, 1, 1, Z)) for ci in range(3)]
        xs = [x for ci in range(3) for x in (pi_ins[ci], alpha_in)]
        if not ops.towers_eligible(towers, xs) or Z % 128 or pi_ins[0].shape[-1] != Z:      # (the adversarial term's input gradient is
            return False                                                                    # written 128 channels per block)
        pi_ins[0].requires_grad_(True)
        params = [t for tw in towers for lay in tw for t in (lay.V, lay.b)]
        hs = ops.TowersFn.apply(towers, *(xs + params))
        losses = []
        for ci, name in enumerate(names):
            loss, acc, mean_joint = ops.CriticHeadFn.apply(hs[2 * ci], hs[2 * ci + 1], B, N.DSIZE)
            crit[name] = (loss, mean_joint, acc, pi_ins[ci])
            if ci == 0:
                c.mim = mean_joint                                            # logit_constraint(real=False), model.py:855
                if cfg.get("adversarial_regularization", True):               # model.py:886-909
                    loa, loa_lr = st["loa"], mi.get("loa_lr", 4.0)
                    loa_gain = c.mim - (1.0 - c.MI_SLACK) * c.MI_TARGET
                    if mi.get("loa_adaptive", True):
                        active = (loa_lr * loa_gain.detach() >= -loa).float()
                        c.adv = active * (loa * loa_gain + loa_lr / 2.0 * loa_gain ** 2)
                    else:
                        c.adv = loa * loa_gain
                    if "encoder_0" in keys:
                        with ops.skip_wgrad():
                            c.g_adv = torch.autograd.grad([c.adv], [pi_ins[0]], retain_graph=True)[0].float().view(2 * B, Z)[:B]
            elif ci == 1:
                c.ind_mim = mean_joint
            if name in keys:
                losses.append((name, loss))
        if losses:
            torch.autograd.grad([l for _, l in losses], [bank.params[n] for name, _ in losses for n in bank.groups[name]["names"]])
        return True

    def _fwd_masks(self, c):
        """B forward: mask decoder z -> logits (model.py:411-412), then the un-taped part path (model.py:414-473): l = mean + eps,
        soft-max, hard max, the moments of gamma * hard and the rectangle centres."""
        cfg, model, nets = self.config, self.model, self.model.nets
        B, S, P = c.B, c.S, c.P
        z_act = model.latent_act(torch.cat([c.samples0[0], c.samples1[0]], 0))
        c.z_leaf = z_act.t.requires_grad_(True)
        c.l_mean = nets.dv(z_act).t                                       # fp32 [2B,S,S,P], taped
        c.lm = c.l_mean.detach()
        # model.py:420-421 / nn.py:1427-1433: l = mean + eps unless `stochastic_l: False` (default: not test_mode)
        stochastic_l = cfg.get("stochastic_l", not cfg.get("test_mode", False))
        eps_l = None
        if stochastic_l:
            eps_l = c.noise["eps_l"] if "eps_l" in c.noise else torch.cat([c.noise["eps_l0"], c.noise["eps_l1"]], 0)
        # (the moments of gamma * hard -- the input of the rectangle centres, model.py:437-440 -- come out of the same pass)
        if c.df:
            c.l, c.m, c.hard, _, c.hbits = ops.part_softmax(c.lm, eps_l, want_bits=P <= 32)
            hstats = None
        else:
            c.l, c.m, c.hard, _, c.hbits, hstats = ops.part_softmax(c.lm, eps_l, want_bits=P <= 32, moments_gamma=c.gamma)
        # [2B,P,2] rectangle centres (stop-gradient); SB_model48c has no rectangles
        if c.df:
            c.px = None
        else:
            if hstats is None:
                hstats = ops.spatial_moments(c.hard, c.gamma)
            c.px = ops.moments_to_px(hstats, S, cfg.get("rect_order", "xy"))
        c.hard0 = c.hard[:B].detach().requires_grad_(True)
        c.hard1 = c.hard[B:].detach().requires_grad_(True)

    def _fwd_reconstruction(self, c):
        """C forward: part-wise appearance -> unpool -> image decoder -> perceptual loss (model.py:478-485, 607-619)."""
        model, nets = self.model, self.model.nets
        B, S, A, P, T = c.B, c.S, c.A, c.P, c.T
        parts = model.part_images(c.img01[B:], c.v1, c.hard1, None if c.hbits is None else c.hbits[B:].contiguous())
        yp = nets.e_alpha(parts).t                                         # [P*B,1,1,A]
        c.feat = yp.float().view(P, B, A).permute(1, 0, 2).contiguous()    # [B,P,A]
        inj = ops.UnpoolFn.apply(c.hard0, c.feat, T)
        c.gen = nets.dd(Act(inj, B, S, S, A + P)).t                        # [B,S,S,8]
        gen_in = self._perceptual_view(c, c.gen)
        if c.ft_pre is not None:
            c.main_stream.wait_event(c.ft_ready)
            c.rec = self.vgg.loss(None, gen_in, T, target_features=c.ft_pre)
        else:
            tgt_in = c.vt if c.pmode == "native" else self._perceptual_view(c, model.to_act(c.vt))
            c.rec = self.vgg.loss(tgt_in, gen_in, T)
        c.auto_rec = (1e-3 * 0.5 * (S * S * 3)) * c.rec                    # model.py:613-619

    def _bwd_reconstruction(self, c):
        """C backward: encoder_1 / decoder_delta weight gradients (they land in the flat buckets) and d rec / d hard masks."""
        bank = self.model.bank
        rec_keys = [k for k in ("encoder_1", "decoder_delta") if k in c.keys]
        rec_params = [bank.params[n] for k in rec_keys for n in bank.groups[k]["names"]]
        gr = torch.autograd.grad([c.auto_rec], [c.hard0, c.hard1] + rec_params)
        c.g_hard0, c.g_hard1 = gr[0].contiguous(), gr[1].contiguous()
        return self._launch_reduce(rec_keys)

    def _priors(self, c):
        """Mask priors: fused forward sums + fused analytic backward (model.py:652-797).  One backward launch per view emits both
        d(rec + priors)/dl (what the decoder_visualize key sees) and d(rec)/dl (what encoder_0 sees)."""
        dev = self.device
        B, S, P, df = c.B, c.S, c.P, c.df
        nfl = L.load().ups_prior_sums_floats(B, P)
        c.sums0 = torch.empty(nfl, dtype=torch.float32, device=dev)
        c.sums1 = torch.empty(nfl, dtype=torch.float32, device=dev)
        per_np0 = torch.empty((B, P, 8), dtype=torch.float32, device=dev)
        l0, l1, m0, m1 = c.l[:B], c.l[B:], c.m[:B], c.m[B:]
        px0, px1 = (None, None) if df else (c.px[:B].contiguous(), c.px[B:].contiguous())
        wz = {"kl": 0.0, "entropy": 0.0, "ms": 0.0, "area": 0.0, "patch": 0.0, "gmrf": 0.0, "var": 0.0, "msl": 0.0}
        if c.pretrain:
            wp = dict(wz)
        elif df:     # SB_model48c:830-838
            wp = {"kl": c.w_kl, "entropy": c.w_weak, "ms": 0.0, "area": 0.0, "patch": 0.0, "gmrf": c.w_gmrf, "var": c.w_var, "msl": c.w_ms}
        else:
            wp = {"kl": c.w_kl, "entropy": c.w_weak, "ms": c.w_ms, "area": 1.0e-12, "patch": c.w_patch, "gmrf": c.w_gmrf, "var": c.w_var,
                  "msl": 0.0}
        self._prior(0, B, S, P, l0, c.lm[:B], m0, c.hard[:B], px0, per_np0, c.sums0, wp)
        # view 1: variance moments (model.py:683-707; SB_model48c:750-756: no gamma, no rectangle) and, from the same pass over the
        # map, its categorical KL (sums1[0]) -- the separate forward launch of view 1 is gone
        c.stats_v = ops.spatial_moments(m1.contiguous(), 1.0 if df else c.gamma, rect_px=px1, half=c.half, kl_sums=c.sums1)
        c.dl_tot = torch.empty_like(c.lm)
        c.dl_rec = torch.empty_like(c.lm)
        self._prior(0, B, S, P, l0, c.lm[:B], m0, c.hard[:B], px0, per_np0, c.sums0, wp, c.g_hard0, c.dl_tot[:B], bwd=True,
                    dl_rec=c.dl_rec[:B])
        self._prior(1, B, S, P, l1, None, m1, None, px1, c.stats_v, c.sums1, wp, c.g_hard1, c.dl_tot[B:], bwd=True, dl_rec=c.dl_rec[B:])
        rs = self.probe.get("rec_scale")
        if rs is not None and float(rs) != 1.0:     # diagnostic hook: how strongly the mask decoder's update follows the reconstruction term
            c.dl_tot.copy_((c.dl_tot - c.dl_rec) + float(rs) * c.dl_rec)

    def _bwd_mask_decoder(self, c):
        """B backward: the weights see rec + priors, the latent sees rec only (one extra input-gradient pass)."""
        bank = self.model.bank
        if "decoder_visualize" in c.keys:
            dv_params = [bank.params[n] for n in bank.groups["decoder_visualize"]["names"]]
            torch.autograd.grad([c.l_mean], dv_params, grad_outputs=[c.dl_tot], retain_graph=True)
        pending = self._launch_reduce([k for k in ("decoder_visualize",) if k in c.keys])
        with ops.skip_wgrad():
            c.gz = torch.autograd.grad([c.l_mean], [c.z_leaf], grad_outputs=[c.dl_rec])[0].float().view(2 * c.B, c.Z)
        return pending

    def _bwd_pose(self, c):
        """A backward (model.py:739, 909, 930): d rec / d z (from B), the adversarial gradient on the joint sample of mi0 and the
        bottleneck beta_0 * exp(lor) * KL."""
        dev, bank = self.device, self.model.bank
        B, Z = c.B, c.Z
        if c.var_reg:
            assert not self.config.get("test_mode", False)
            explor = torch.exp(c.st["lor"])
        if "encoder_0" not in c.keys:
            return
        g_s0 = torch.zeros((len(c.levels0), B, Z), dtype=torch.float32, device=dev)
        g_s0[0] = c.gz[:B]
        if c.g_adv is not None:
            g_s0[1] = c.g_adv
        gp0 = ops.latent_bwd(c.pe_v0, c.noise["eps_pi0"], c.levels0, g_s0, explor.reshape(1) if c.var_reg else None,
                             c.beta_0 / B if c.var_reg else 0.0)
        gp1 = ops.latent_bwd(c.pe_v1, c.noise["eps_pi1"][None], [1.0], c.gz[B:].contiguous()[None], None, 0.0)
        g_pe = torch.cat([gp0, gp1], 0).view_as(c.pe)
        e0_params = [bank.params[n] for n in bank.groups["encoder_0"]["names"]]
        if (self.world_size > 1 or D.FORCE_COLLECTIVES) and not self._early_hooked:      # layers exist once the first forward has run
            self._hook_early_reduce()
            self._early_hooked = True
        torch.autograd.grad([c.pe], e0_params, grad_outputs=[g_pe])

    def _next_state(self, c):
        """update_ops (model.py:28-35, 829-834, 861-866, 890-909, 921-930): EMAs and the two multipliers from the batch-mean
        scalars -- averaged over the ranks first, so that every replica holds the same state.  Appendix A.15: the losses of this
        step used the pre-update state."""
        cfg, st, mi = self.config, c.st, c.mi
        stats = torch.stack([c.mim.detach(), c.ind_mim.detach(), c.acc0, c.acc1, c.loss_dis0.detach(), c.loss_dis1.detach()])
        if getattr(self, "_cap", None) is not None and self.world_size > 1:
            self._boundary("scalars", stats)             # (the tensor lives in the graphs' pool: same address at every replay)
        else:
            D.average_scalars(stats, self.world_size, self.process_group)
        new = dict(st)
        up_loa = bool(cfg.get("adversarial_regularization", True) and mi.get("loa_adaptive", True))
        up_lor = bool(c.var_reg and mi.get("lor_adaptive", True))
        if STATE_KERNEL and stats.is_cuda and all(st[k].dtype == torch.float32 and st[k].device == stats.device for k in STATE_KEYS):
            # one launch instead of ~30 scalar ones (they were the last thing a step enqueued, behind a drained GPU)
            old = torch.stack([st[k].reshape(()) for k in STATE_KEYS])
            vec = torch.empty_like(old)
            L.call("ups_state_update", L.ptr(stats), L.ptr(old), L.ptr(vec), 0.99, 1.0 - 0.99, int(up_loa), mi.get("loa_lr", 4.0),
                   (1.0 - c.MI_SLACK) * c.MI_TARGET, int(up_lor), mi.get("lor_lr", 0.05), c.MI_TARGET, mi.get("lor_min", 1.0),
                   mi.get("lor_max", 7.5), L.stream())
            for i, k in enumerate(STATE_KEYS):
                if k not in ("loa", "lor") or (k == "loa" and up_loa) or (k == "lor" and up_lor):
                    new[k] = vec[i]
            return new
        g_mim, g_ind, g_acc0, g_acc1, g_l0, g_l1 = stats.unbind(0)
        ema = lambda old, val: 0.99 * old + (1.0 - 0.99) * val            # model.py:28-35
        new["avg_acc0"] = ema(st["avg_acc0"], g_acc0); new["avg_acc1"] = ema(st["avg_acc1"], g_acc1)
        new["avg_acc_error"] = ema(st["avg_acc_error"], g_acc1 - g_acc0)
        new["avg_loss_dis0"] = ema(st["avg_loss_dis0"], g_l0); new["avg_loss_dis1"] = ema(st["avg_loss_dis1"], g_l1)
        new["avg_mim"] = ema(st["avg_mim"], g_mim); new["avg_independent_mim"] = ema(st["avg_independent_mim"], g_ind)
        if cfg.get("adversarial_regularization", True) and mi.get("loa_adaptive", True):
            new["loa"] = torch.clamp(st["loa"] + mi.get("loa_lr", 4.0) * (g_mim - (1.0 - c.MI_SLACK) * c.MI_TARGET), min=0.0)
        if c.var_reg and mi.get("lor_adaptive", True):
            new["lor"] = torch.clamp(st["lor"] + mi.get("lor_lr", 0.05) * (g_ind - c.MI_TARGET), mi.get("lor_min", 1.0),
                                     mi.get("lor_max", 7.5))
        return new

    def _log_thunk(self, c):
        """Losses per key + log ops (model.py:648-966; same names as the reference).  REPORTING only: ~150 scalar launches that no
        gradient depends on (the per-key gradients were taken from the pieces above).  In the eager trainer the returned closure is
        evaluated when somebody reads `losses` / `log_ops` (log steps, tests) and keeps only detached scalars and the small
        reduction buffers alive; inside a captured HIP graph it runs with the step."""
        cfg, st = self.config, c.st
        B, S, P, df, keys, step = c.B, c.S, c.P, c.df, c.keys, c.step
        w_gmrf, w_ms, w_kl, w_var, w_weak, w_patch, pretrain = c.w_gmrf, c.w_ms, c.w_kl, c.w_var, c.w_weak, c.w_patch, c.pretrain
        beta_0, var_reg, MI_TARGET, MI_SLACK = c.beta_0, c.var_reg, c.MI_TARGET, c.MI_SLACK
        sums0, sums1, stats_v, kl_rows = c.sums0, c.sums1, c.stats_v, c.kl_rows
        auto_rec, rec = c.auto_rec.detach(), c.rec.detach()
        adv = c.adv.detach() if c.adv is not None else None
        loss_dis0, loss_dis1, loss_est = c.loss_dis0.detach(), c.loss_dis1.detach(), c.loss_est.detach()
        acc0, acc1, acc_est = c.acc0, c.acc1, c.acc_est
        mim, ind_mim = c.mim.detach(), c.ind_mim.detach()
        crit_d = {k: c.crit[k][0].detach() for k in N.EXTRA_48C} if df else {}
        lr_now = self.learning_rate()          # (of THIS step: the counters have moved on when the logs are read)
        log = OrderedDict()

        def build_logs():
            bottleneck = kl_rows.sum(dim=1).mean()                            # nn.py:1196-1208
            bw = beta_0 * torch.exp(st["lor"]) * bottleneck if var_reg else None
            npx = float(B * S * S)
            prior_gmrf = sums0[3] / B
            mask0_kl = (sums0[0] + sums1[0]) / npx
            weakly = sums0[1] / npx
            patch_loss = sums0[2] / B
            p_ms = w_ms * sums0[4] / B
            area_cost = 1.0e-12 * sums0[5] / B
            Zs = stats_v[..., 1]
            if df:       # SB_model48c:757-776: squared diagonal variances of the (already normalised) maps
                s00 = stats_v[..., 6] / Zs - (stats_v[..., 3] / Zs) ** 2
                s11 = (stats_v[..., 5] - stats_v[..., 6]) / Zs - (stats_v[..., 4] / Zs) ** 2
                variances = (s00 ** 2 + s11 ** 2).sum(dim=1).mean()
                prior_ms = sums0[2] / B                                   # variant 1: the patch slot holds sum min(alpha g, lambda)
                prior_total = w_gmrf * prior_gmrf + prior_ms * w_ms + w_kl * mask0_kl + weakly * w_weak + w_var * variances
            else:
                variances = (stats_v[..., 5] / Zs - (stats_v[..., 3] / Zs) ** 2 - (stats_v[..., 4] / Zs) ** 2).sum(dim=1).mean()
                prior_total = (w_gmrf * prior_gmrf + w_kl * mask0_kl + weakly * w_weak + w_var * variances + p_ms + area_cost
                               + patch_loss * w_patch)

            Ls = OrderedDict()
            Ls["encoder_0"] = auto_rec + (adv if adv is not None else 0.0) + (bw if bw is not None else 0.0)
            Ls["encoder_1"] = auto_rec
            Ls["decoder_delta"] = auto_rec
            Ls["decoder_visualize"] = auto_rec if pretrain else auto_rec + prior_total
            Ls["mi0_discriminator"], Ls["mi1_discriminator"], Ls["mi_estimator"] = loss_dis0, loss_dis1, loss_est
            if df:       # SB_model48c:809-815 (the global term has no gradient path to the encoders: stop_gradient inputs)
                Ls["encoder_0"] = Ls["encoder_0"] + crit_d["d_single"]
                Ls["encoder_1"] = Ls["encoder_1"] + crit_d["d_single"]
                for k in N.EXTRA_48C:
                    Ls[k] = crit_d[k]
            losses_d = OrderedDict((k, Ls[k].detach()) for k in keys)
            avg_mim = torch.clamp(st["avg_mim"], min=0.0); avg_ind = torch.clamp(st["avg_independent_mim"], min=0.0)
            loo = torch.clamp((avg_ind - avg_mim) / (avg_ind + 1e-6), 0.0, 1.0)
            log.update({"prior_gmrf": prior_gmrf, "prior_gmrf_weight": w_gmrf, "prior_gmrf_weighted": w_gmrf * prior_gmrf,
                        "mask0_kl_weight": w_kl, "mask0_kl": mask0_kl, "mask0_kl_weighted": w_kl * mask0_kl,
                        "variance_loss_weighted": w_var * variances, "variance_loss": variances, "variance_weight": w_var,
                        "weakly_superv_loss_weight_p": w_weak, "weakly_superv_loss_p": weakly,
                        "weakly_superv_loss_p_weighted": weakly * w_weak,
                        "patch_loss": patch_loss, "patch_loss_weight": w_patch, "patch_loss_weighted": patch_loss * w_patch,
                        "mumford_sha_lambda": make_var(step, cfg["mumford_sha_lambda"]),
                        "mumford_sha_alpha": make_var(step, cfg["mumford_sha_alpha"]),
                        "avg_acc_error": st["avg_acc_error"], "avg_mim": avg_mim, "avg_independent_mim": avg_ind,
                        "loo": loo, "lon_gain": -loo + 0.025, "model_lon": st["lon"]})
            if df:
                for k in ("patch_loss", "patch_loss_weight", "patch_loss_weighted"):
                    log.pop(k)
            for k in Ls:
                log["loss_" + k] = Ls[k].detach()
            log.update({"dis0_accuracy": acc0, "dis1_accuracy": acc1, "avg_dis0_accuracy": st["avg_acc0"],
                        "avg_dis1_accuracy": st["avg_acc1"], "avg_loss_dis0": st["avg_loss_dis0"],
                        "avg_loss_dis1": st["avg_loss_dis1"], "est_accuracy": acc_est,
                        "mi_constraint": mim, "independent_mi_constraint": ind_mim})
            if adv is not None:
                log.update({"adversarial_weight": st["loa"], "adversarial_constraint": mim,
                            "adversarial_weighted_loss": adv, "loa": st["loa"],
                            "loa_gain": mim - (1.0 - MI_SLACK) * MI_TARGET})
            if bw is not None:
                log.update({"bottleneck_weight": st["lor"], "bottleneck_loss": bottleneck, "bottleneck_weighted_loss": bw,
                            "lor": st["lor"], "explor": beta_0 * torch.exp(st["lor"]), "lor_gain": ind_mim - MI_TARGET})
            if df:
                log.update({"prior_mumford_sha": prior_ms, "prior_mumford_sha_weight": w_ms, "prior_mumford_sha_weighted": prior_ms * w_ms,
                            "perceptual": rec, "lr": lr_now})
                for i in range(P):
                    log["sigma1_{:02d}".format(i)] = s00[0, i]
                    log["sigma2_{:02d}".format(i)] = s11[0, i]
            else:
                log.update({"zr_mumford_sha": p_ms, "z_mumford_sha_smoothness_cost": sums0[6] / B,
                            "z_mumford_sha_contour_cost": sums0[7] / B, "z_area_cost": area_cost,
                            "prior_mumford_sha_weight": w_ms, "perceptual": rec, "lr": lr_now})
            return losses_d, log
        return build_logs

    def _step_impl(self, batch, noise=None, graph_lr=None):
        """One training step on the launching stream (+ the "pre", "aux" and "wgrad" side streams): forward A -> D (aux) -> B ->
        C, backward C -> priors -> B -> A, each optimizer key's bucket reduced as soon as its segment is complete, Adam, state."""
        dev = self.device
        self._step_graph_lr, self._adam_done, self._adam_stepped = graph_lr, set(), set()
        c = self._step_begin(batch, noise)
        self._fwd_pose(c)
        # (enqueueing the mask decoder's forward pass BEFORE the critics' block measured neutral twice, UPS_CRITICS_LATE in round 5:
        # retired in round 6)
        if ops.Streams.enabled:
            aux = ops.Streams.get("aux", dev)
            aux.wait_stream(c.main_stream)
            with torch.cuda.stream(aux):
                self._critics(c)
        else:
            self._critics(c)
        self._fwd_masks(c)
        self._fwd_reconstruction(c)
        pending = self._bwd_reconstruction(c)
        self._priors(c)
        pending += self._bwd_mask_decoder(c)
        # the critics' block (aux stream) must be complete from here on: g_adv, the critic losses and their gradients
        if JOIN_TIMING and graph_lr is None:       # debug: how long the launching stream sits at this join (tools/probes/join_wait.py)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            ops.Streams.join(dev, names=("aux",))
            ev[1].record()
            self._join_events = (getattr(self, "_join_events", []) + [ev])[-256:]       # (bounded: a debug aid for short runs)
        else:
            ops.Streams.join(dev, names=("aux",))
        pending += self._launch_reduce([k for k in ("mi0_discriminator", "mi1_discriminator", "mi_estimator") + N.EXTRA_48C
                                        if k in c.keys])
        self._bwd_pose(c)
        # ---- gradient all-reduce (data parallel) + TF Adam per key
        pending += self._launch_reduce([k for k in ("encoder_0",) if k in c.keys])
        self._finish_step(c.keys, pending, graph_lr)
        new = self._next_state(c)
        build_logs = self._log_thunk(c)
        st = c.st
        if graph_lr is not None:        # graph mode: state lives in fixed device scalars, python counters advance outside
            losses_now, log = build_logs()
            # the logged state is the PRE-update value in both modes: snapshot before the in-place update below
            for k, v in list(log.items()):
                if torch.is_tensor(v) and any(v is sv for sv in st.values()):
                    log[k] = v.clone()
            self._set_logs(losses_now, log)
            for k in st:
                if new[k] is not st[k]:
                    st[k].copy_(new[k])
        else:
            self._lazy_logs = build_logs        # (the closure holds the PRE-update state dict `st`: self.state is replaced, not modified)
            self.state = new
            self.global_step += 1
        self._debug = {"l_mean": c.lm, "l": c.l, "m": c.m, "hard": c.hard, "px": c.px, "generated": c.gen.detach(),
                       "feat": c.feat.detach(), "dl_tot": c.dl_tot, "dl_rec": c.dl_rec, "g_hard0": c.g_hard0, "g_hard1": c.g_hard1,
                       "pe": c.pe2}
        return _LazyLosses(self)

    def _hook_early_reduce(self):
        """The 1x1 head of encoder_0 (258 x 33152 weights = 34 of the key's 54.6 MB) is the FIRST weight gradient of the
        last backward segment: its slice of the flat bucket starts its all-reduce as soon as it has been enqueued, so
        only the remaining 20 MB follow the end of the backward pass."""
        grp = self.model.bank.groups["encoder_0"]
        head = max((n for n in grp["names"] if n.endswith("/V")), key=lambda n: int(n.split("conv2d_")[1].split("/")[0]))
        prefix = head[:-2]
        off = 0
        for n in grp["names"]:
            if n.startswith(prefix + "/"):
                break
            off += self.model.bank.params[n].numel()
        tail = sum(self.model.bank.params[n].numel() for n in grp["names"] if n.startswith(prefix + "/"))
        assert off + tail == grp["flat"]["g"].numel(), "the head's variables must close the flat bucket"
        trainer = self

        def launch():
            if "encoder_0" in trainer._early or getattr(trainer, "_cap", None) is not None:
                return                      # (graph capture: the whole bucket is reduced at the segment boundary)
            if ops.Streams.enabled and DP_SIDE_LAUNCH:       # behind the head's weight gradient on ITS stream (see _launch_reduce)
                side = ops.Streams.get("wgrad", trainer.device)
                w2 = ops.Streams._pool.get(("wgrad2", torch.device(trainer.device).index))
                if w2 is not None:
                    side.wait_stream(w2)
                with torch.cuda.stream(side):
                    h = D.allreduce_bucket(grp["flat"]["g"][off:], trainer.world_size, trainer.process_group)
            else:
                ops.Streams.join(trainer.device, names=("wgrad",))
                h = D.allreduce_bucket(grp["flat"]["g"][off:], trainer.world_size, trainer.process_group)
            trainer._early["encoder_0"] = (off, h)

        for key, lay in self.model.nets.layers.items():
            if key[0] == prefix:
                lay.after_wgrad = launch

    def _launch_reduce(self, key_list):
        """Called when the backward segment of these optimizer keys is complete: their weight gradients (side stream)
        are joined and each key's flat gradient bucket starts its RCCL all-reduce (sum; 1/world is folded into Adam),
        overlapping the backward segments that are still to run.  Returns the work handles."""
        if not key_list:
            return []
        if getattr(self, "_cap", None) is not None:      # graph capture under data parallelism: a segment boundary
            if self.world_size > 1 or D.FORCE_COLLECTIVES:
                self._boundary("grads", list(key_list))
            return []
        if self.world_size == 1 and not D.FORCE_COLLECTIVES and LATE_JOIN:
            # a single rank has nothing to reduce: the launching stream need not wait for the weight-gradient stream here (it
            # would idle whenever that stream lags); both meet before the end of the step (_finish_step).  The tensors the side
            # stream reads stay referenced until then (ops.Streams.keep).  The keys' Adam updates are queued right BEHIND their
            # weight gradients on that stream (EARLY_ADAM): the fp32 master weights are not read again this step -- every
            # convolution works on the converted copies, refreshed once all keys have stepped -- so the 0.9 GB optimizer stream
            # runs in the shadow of the remaining backward pass instead of on an otherwise empty chip at the end.
            if EARLY_ADAM and ops.Streams.enabled and self._step_graph_lr is None:
                side = ops.Streams.get("wgrad", self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))     # (critics: their gradients were taken on "aux", joined by now)
                side.wait_stream(ops.Streams.get("wgrad2", self.device))     # (the CoordConv rows of these keys' weight gradients)
                with torch.cuda.stream(side):
                    self._adam(key_list, None)
                    ev = side.record_event()
                # a converted-weight cache entry created later in this step (a new (dtype, size) instance, the depth-to-space or
                # fp8 copies) reads the fp32 master: it must see the finished update, not race with it (ops.ConvLayer._wait_master)
                for k in key_list:
                    ops.Streams.master_busy[k] = ev
                self._adam_done.update(key_list)
            return []
        bank = self.model.bank
        handles = []
        # The all-reduce of a bucket has to wait for the segment's weight gradients -- they run on the "wgrad" side stream -- but
        # the LAUNCHING stream does not: the collective is enqueued from the side stream's position (torch's process group orders
        # its own stream behind the stream that is current at the call), so the backward pass goes on while the bucket is reduced.
        # (Until round 5 the launching stream joined the side stream here: under data parallelism it idled at every segment
        # boundary for as long as the weight-gradient queue lagged -- the 2 % a single rank gains from LATE_JOIN.)
        if ops.Streams.enabled and DP_SIDE_LAUNCH:
            side = ops.Streams.get("wgrad", self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))      # (critics: their gradients were taken on "aux", joined by now)
            w2 = ops.Streams._pool.get(("wgrad2", torch.device(self.device).index))
            if w2 is not None:
                side.wait_stream(w2)                                       # the CoordConv rows of these keys' weight gradients
            ctx = torch.cuda.stream(side)
        else:
            ops.Streams.join(self.device, names=("wgrad",))
            ctx = contextlib.nullcontext()
        with ctx:
            for k in key_list:
                g = bank.groups[k]["flat"]["g"]
                early = self._early.pop(k, None)
                if early is not None:            # the tail slice is already in flight (see _hook_early_reduce)
                    handles.append(early[1])
                    self._reduce_marks.append((k + "[head]", (g.numel() - early[0]) * 4))      # one mark per handle (dp_wait_ms)
                    g = g[:early[0]]
                handles.append(D.allreduce_bucket(g, self.world_size, self.process_group))
                self._reduce_marks.append((k, g.numel() * 4))
        return handles

    def _adam(self, keys, graph_lr):
        """One fused TF-Adam launch per key on the current stream (Appendix A.12); advances the keys' step counters (eager mode)."""
        bank = self.model.bank
        lr = self.learning_rate()
        lr_scale = self.probe.get("lr_scale") or {}
        if graph_lr is not None and (lr_scale or self.grad_clip_norm > 0):
            raise NotImplementedError("probe.lr_scale / grad_clip_norm are eager-mode options (hip_graph: False)")
        for k in keys:
            grp = bank.groups[k]
            f = grp["flat"]
            if graph_lr is not None:
                lr_t = graph_lr
            else:
                grp["t"] += 1
                self._adam_stepped.add(k)         # (for _after_failed_step: this key's counter and weights have moved)
                t = grp["t"]
                lr_t = lr * lr_scale.get(k, 1.0) * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
                if self.grad_clip_norm > 0:         # tf.clip_by_global_norm over the key's variables (device side, no sync)
                    gn = torch.linalg.vector_norm(f["g"]) / self.world_size
                    f["g"].mul_(torch.clamp(self.grad_clip_norm / gn.clamp_min(1e-30), max=1.0))
            ops.adam_step(f["p"], f["g"], f["m"], f["v"], lr_t, self.beta1, self.beta2, self.adam_eps, 1.0 / self.world_size)

    def _finish_step(self, keys, handles, graph_lr=None):
        """Wait for the buckets, then one fused Adam launch per key (tf.train.AdamOptimizer semantics, Appendix A.12).
        graph_lr: device scalar holding lr_t (HIP-graph mode; the python step counters then advance outside)."""
        if JOIN_TIMING and graph_lr is None:       # debug: the tail of the weight-gradient stream that nothing hides (join_wait.py)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            ops.Streams.join(self.device)
            ev[1].record()
            self._tail_events = (getattr(self, "_tail_events", []) + [ev])[-256:]
        else:
            timed = bool(handles) and graph_lr is None and any(h is not None for h in handles) and not torch.cuda.is_current_stream_capturing()
            if timed:       # data parallel: what the launching stream WAITS at the end of the backward pass, split into its parts
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(handles) + 2)]
                evs[0].record()
            ops.Streams.join(self.device)
            if timed:
                evs[1].record()
                for i, h in enumerate(handles):
                    if h is not None:
                        h.wait()
                    evs[2 + i].record()
                self._dp_wait_events = (getattr(self, "_dp_wait_events", []) + [(evs, list(self._reduce_marks))])[-64:]
                handles = []
        self._reduce_marks = []
        D.wait_all(handles)
        self._adam([k for k in keys if k not in self._adam_done], graph_lr)
        self._adam_done, self._adam_stepped = set(), set()
        ops.Streams.master_busy.clear()         # (the join above ordered this stream behind every early Adam)
        ops.Streams.epoch += 1                  # lazy weight conversions of this step are ordered before everything that follows
        if graph_lr is None:
            ops.WeightVersion.value += 1
        self.model.nets.prep.refresh()          # one launch: every layer's converted weights + CoordConv tables
        if ops.Fp8.enabled:
            ops.Fp8.after_step()

    def dp_wait_ms(self):
        """Data parallel: mean time per step the launching stream waited at the end of the backward pass -- for the weight-gradient
        streams to drain (`side_streams`), then, bucket by bucket in launch order, for each gradient all-reduce that had not
        finished by then (`<key>` with its bytes; 0 when the collective was fully hidden behind the backward pass).  Synchronises."""
        rec = getattr(self, "_dp_wait_events", [])
        if not rec:
            return None
        torch.cuda.synchronize(self.device)
        out, n = OrderedDict(), 0
        for evs, marks in rec:
            n += 1
            out["side_streams"] = out.get("side_streams", 0.0) + evs[0].elapsed_time(evs[1])
            for i in range(len(evs) - 2):
                key = "{}:{}B".format(*marks[i]) if i < len(marks) else "bucket{}".format(i)
                out[key] = out.get(key, 0.0) + evs[1 + i].elapsed_time(evs[2 + i])
        return OrderedDict((k, round(v / n, 4)) for k, v in out.items())

    # ------------------------------------------------------------------ edflow iterate(): log cadence of LoggingHook
    def fetch_logs(self):
        out = OrderedDict()
        for k, v in self.log_ops.items():
            out[k] = float(v) if torch.is_tensor(v) else float(v)
        return out

    def iterate(self, batch_iterator, num_steps=None, log_fn=print):
        """edflow's session loop with its hooks.  LoggingHook cadence as cub/train/log.txt:221-860 shows it (an IntervalHook whose
        interval doubles after every trigger up to ``log_freq``): global steps 0, 2, 4, 8, ..., 128, then every ``log_freq``
        (250, 500, ...); the keys are printed in alphabetical order with ``global_step`` among them (log.txt:204-263), followed
        by the ``project root`` line."""
        cfg = self.config
        num_steps = num_steps if num_steps is not None else cfg.get("num_steps", 1000000)
        log_freq, ckpt_freq = cfg.get("log_freq", 250), cfg.get("ckpt_freq", 10000)
        interval = 1
        log_fn("[INFO] [Trainer]: " + self.switches_report)
        for batch in batch_iterator:
            if self.global_step >= num_steps:
                break
            s = self.global_step
            self.train_step(batch)
            if s % interval == 0:
                interval = min(2 * interval, max(1, log_freq))
                logs = self.fetch_logs()
                logs["global_step"] = s
                for k in sorted(logs):
                    log_fn("[INFO] [LoggingHook]: {}: {}".format(k, logs[k]))
                if self.root:
                    log_fn("[INFO] [LoggingHook]: project root: {}".format(os.path.join(self.root, "train")))
                # failure detection on log steps only (the step itself never synchronises with the host).  Losses are per rank:
                # the decision is made collective, so that every rank of a data-parallel run leaves together instead of the
                # healthy ones waiting in the next all-reduce until the RCCL timeout
                bad = [k for k, v in logs.items() if k.startswith("loss_") and not math.isfinite(v)]
                flag = torch.tensor([1.0 if bad else 0.0], dtype=torch.float32, device=self.device)
                D.sum_flag(flag, self.world_size, self.process_group)
                if float(flag) > 0:
                    raise FloatingPointError("non-finite {} at global step {} (on {} rank(s))".format(
                        ", ".join(bad) if bad else "loss on another rank", s, int(float(flag))))
            # edflow CheckpointHook: the file is named after the global step the restored run continues FROM
            if ckpt_freq and self.global_step % ckpt_freq == 0:
                self._checkpoint()
        self._checkpoint()                # final state at loop exit

    def _checkpoint(self):
        """model.ckpt-<step> of THIS process's current state: written to a temporary name and renamed, so a file of that name
        left by an earlier (or diverged) run in the same project root is replaced, never kept."""
        if not self.root or self.rank != 0:
            return
        path = os.path.join(self.root, "train", "checkpoints", "model.ckpt-{}".format(self.global_step))
        if getattr(self, "_last_ckpt", None) == (path, self.global_step):
            return                        # this very state was just written (loop exit right after a periodic checkpoint)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp-{}".format(os.getpid())
        self.save_checkpoint(tmp)
        os.replace(tmp, path)
        self._last_ckpt = (path, self.global_step)
