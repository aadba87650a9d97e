"""Host-side operators over the C ABI: tap geometry for TF 'SAME' convolutions and the autograd
wrappers (forward / dgrad / wgrad) that let PyTorch own the tape while every FLOP runs in
libupsparts_hip.so.  Reference semantics: cub/code/nn.py (conv2d 617-711, residual_block 1042-1056,
upsample 834-847), SURVEY.md Appendix A.

Activations are NHWC tensors whose last dimension is the PHYSICAL channel count (multiple of 8);
weights stay fp32 in the TF variable layout HWIO and are re-laid-out / converted once per optimizer step.

fp16 forward tensors (``fmt = L.F16``, the mask decoder: 10 mantissa bits instead of bf16's 7 at the same MFMA rate -- what
north_star's part-mask IoU >= 0.99 needs, tests/bf16_emulation_study.py) live in torch.bfloat16 CONTAINERS: the autograd engine casts
every returned gradient to the dtype of the forward tensor, and the gradients of these layers are bf16 (range).  The format
travels beside the tensor (nets.Act.fmt, ConvFn's ``fmt`` argument); nothing but the HIP kernels reads those bits.
"""
import ctypes as C
import os

import torch

from . import lib as L
from . import switches as SW


def round8(c):
    return (c + 7) // 8 * 8


def same_geometry(size, k, stride):
    """TF 'SAME': out = ceil(in/stride); pad_before = pad_total // 2 (Appendix A.1)."""
    out = -(-size // stride)
    pad_total = max((out - 1) * stride + k - size, 0)
    return out, pad_total // 2


class _Workspace(object):
    """Grow-only device scratch, one buffer per (device, stream): launches on different HIP streams may overlap."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes, device):
        key = (device, L.raw_stream(device))
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


WORKSPACE = _Workspace()
COLSUM_WS = _Workspace()


class Streams(object):
    """Side HIP streams of the training step.  ``wgrad``: every layer's weight gradient is enqueued there while the
    input gradient continues on the launching stream (the two are independent; small layers that cannot fill 256 CUs
    then overlap).  ``aux``: the critics + the appearance code of the whole views, which the main path only needs
    again at the encoder_0 backward.  UPS_NO_OVERLAP=1 keeps everything on one stream (A/B runs, debugging)."""
    enabled = not SW.flag("UPS_NO_OVERLAP")
    _pool = {}
    _raw = {}       # (name, device index) -> raw hipStream_t
    epoch = 0          # bumped by the trainer at the end of every step (all side streams joined): scopes ConvLayer._mark_ready
    master_busy = {}   # optimizer key -> event on the "wgrad" stream behind that key's early Adam launch (model.Trainer._launch_reduce)
    _alive = {}     # device index -> tensors the "wgrad" stream still reads.  Holding references until the next join keeps
                    # their memory out of the allocator without record_stream (whose deferred frees made the caching
                    # allocator reserve ~9x the live set: 84 GB at B = 64); once the launching stream has waited for the
                    # side stream, dropping them is safe -- every later use of that memory is ordered behind the join.

    @classmethod
    def keep(cls, device, *tensors):
        cls._alive.setdefault(torch.device(device).index, []).extend(tensors)

    # Stream plan.  "full": every logical stream is a HIP stream of its own (seven with the launching stream) -- the fastest form on
    # ONE rank.  "compact": the logical streams fold onto three -- launching stream, "wgrad" (weight gradients
    # and their CoordConv rows), "aux" (target features, appearance code, the three critics) -- which leaves the fourth of the
    # four hardware queues (GPU_MAX_HW_QUEUES) to the collectives' stream under data parallelism: more than four ACTIVE queues are
    # time-sliced by the hardware scheduler (+30 % on the step).  The trainer picks the plan (`stream_plan`, model.Trainer).
    COMPACT_ALIAS = {"pre": "aux", "aux1": "aux", "aux2": "aux", "wgrad2": "wgrad"}
    alias = {}

    @classmethod
    def set_plan(cls, plan):
        if plan not in ("full", "compact"):
            raise ValueError("stream plan '{}' (full | compact)".format(plan))
        cls.alias = dict(cls.COMPACT_ALIAS) if plan == "compact" else {}

    @classmethod
    def get(cls, name, device):
        name = cls.alias.get(name, name)
        key = (name, torch.device(device).index)
        st = cls._pool.get(key)
        if st is None:
            st = torch.cuda.Stream(device=device)
            cls._pool[key] = st
            cls._raw[key] = st.cuda_stream
        return st

    _pads = []

    @classmethod
    def on_aux(cls, device):
        cur, idx = L.raw_stream(device), torch.device(device).index
        return any(cls._raw.get((n, idx)) == cur for n in ("aux", "aux1", "aux2"))      # (aux1 / aux2: critics two and three)

    @classmethod
    def join(cls, device, names=("wgrad", "aux")):
        """The current stream waits for everything enqueued on the side streams so far."""
        if not cls.enabled:
            return
        cur = torch.cuda.current_stream(device)
        if "wgrad" in names:
            names = tuple(names) + ("wgrad2",)          # the CoordConv rows of the weight gradients (conv_wgrad) belong to it
        if "aux" in names:
            names = tuple(names) + ("aux1", "aux2")         # critics two and three (Trainer._critics)
        seen = set()
        for n in names:
            n = cls.alias.get(n, n)
            if n in seen:
                continue
            seen.add(n)
            st = cls._pool.get((n, torch.device(device).index))
            if st is not None and st != cur:
                cur.wait_stream(st)
        if "wgrad" in names:
            cls._alive.pop(torch.device(device).index, None)


class KernelTimer(object):
    """bench.py hook: HIP events (recorded on the launch stream) around every launch of the patch kernel for one named
    convolution -- its forward and its input-gradient launches: the same kernel on the same problem size -- so the roofline
    figure is that kernel's own duration, measured live."""
    layer, enabled, events, flops = None, False, [], 0.0
    kinds = []        # "fwd" / "dgrad" per timed launch

    @classmethod
    def active(cls):
        # (events recorded while a HIP graph is being captured become graph nodes and cannot be read back: replayed steps are not timed)
        return cls.enabled and not torch.cuda.is_current_stream_capturing()

    @classmethod
    def mean_ms(cls, kind=None):
        ev = [e for e, k in zip(cls.events, cls.kinds) if kind is None or k == kind]
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))


class WeightVersion(object):
    """Bumped by the optimizer; ConvLayer caches of converted weights key on it."""
    value = 0


class SignBits(object):
    """Bit-packed activation signs (ups_conv_desc.sign_out / dact_bits, round 5).  The input gradient of a convolution needs one
    bit of every element of the layer's forward input -- the sign, for act' -- and re-read the whole 16-bit tensor for it.  The
    PRODUCER of such a tensor (a convolution's epilogue, the x2 bilinear kernel) now also writes [n,h,w,c/8] sign bytes, the handle
    carries them (nets.Act.bits) and the consuming convolution's backward passes them to ups_conv_igemm next to `dact`.
    Hand-off like Fp8.last_out: `want` is set by the caller that will keep the bits, `last` by the producer."""
    ENABLED = SW.flag("UPS_SIGN_BITS")
    want = False
    last = None
    stats = None        # a dict when a probe wants to know which input gradients ran without bits (tools/probes/sign_bits_coverage.py)

    @classmethod
    def take(cls):
        b, cls.last, cls.want = cls.last, None, False
        return b


class PrepRegistry(object):
    """Converted-weight buffers of the layers of a model, refreshed by ONE launch after the optimizer step
    (ups_weight_prep_batch) instead of two small launches per layer and step."""

    def __init__(self):
        self.entries = []          # (layer, ent, dtype_code, hi, wi)
        self.tables = {}           # dtype_code -> (items_dev, prefix_dev, n, total_blocks)
        self.extra = []            # per-layer conversions outside the batched launch (depth-to-space dgrad weights)
        self.dirty = True

    def register(self, layer, ent, dtype_code, hi, wi):
        self.entries.append((layer, ent, dtype_code, hi, wi))
        self.dirty = True

    def _build(self):
        self.tables = {}
        lib = L.load()
        for dcode in sorted(set(e[2] for e in self.entries)):
            ents = [e for e in self.entries if e[2] == dcode]
            arr = (L.PrepItem * len(ents))()
            prefix = [0]
            for i, (lay, ent, _, hi, wi) in enumerate(ents):
                it = arr[i]
                it.src = lay.V.data_ptr()
                it.w_fwd = ent["w_fwd"].data_ptr() if ent["w_fwd"] is not None else None
                it.w_dgrad = ent["w_dgrad"].data_ptr() if ent["w_dgrad"] is not None else None
                it.ctab = ent["ctab"].data_ptr() if ent["ctab"] is not None else None
                it.ntaps, it.cin_v, it.ci_log, it.co = lay.k * lay.k, lay.cin_v, lay.ci_log, lay.co
                it.ci_pad, it.dgrad_rows, it.dgrad_k = round8(lay.ci_log), lay.ci_log, round8(lay.co)
                it.kh = it.kw = lay.k
                it.in_sy = it.in_sx = lay.stride
                dy, dx, _ = lay.fwd_taps(hi, wi)
                for r in range(3):
                    it.dy[r] = dy[r * lay.k] if r < lay.k else 0
                    it.dx[r] = dx[r] if r < lay.k else 0
                it.ax, it.ay = 2.0 / max(1, hi - 1), 2.0 / max(1, wi - 1)
                prefix.append(prefix[-1] + lib.ups_prep_item_blocks(C.byref(it), dcode))
            dev = ents[0][0].V.device
            items_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            prefix_dev = torch.tensor(prefix, dtype=torch.int64, device=dev)
            self.tables[dcode] = (items_dev, prefix_dev, len(ents), prefix[-1])
        self.dirty = False

    def refresh(self):
        """Re-convert every registered layer from the current fp32 master weights."""
        if not self.entries:
            return
        if self.dirty:
            self._build()
        for dcode, (items_dev, prefix_dev, n, total) in self.tables.items():
            L.call("ups_weight_prep_batch", L.ptr(items_dev), L.ptr(prefix_dev), n, total, dcode, L.stream())
        for (_lay, ent, _d, _h, _w) in self.entries:
            ent["version"] = WeightVersion.value
            ent["ready"] = None
        for fn in self.extra:
            fn()


class Fp8State(object):
    """fp8 forward of the wide 3x3 / stride-1 convolutions (BASELINE config #5: e4m3 MFMA operands, fp32 accumulate, bf16
    tensors; ``precision: fp8`` in the config).  Weights are scaled per output channel when they are converted; activations
    use delayed per-tensor scaling: every launch records max |act(x)| (kernel epilogue, 64 atomic slots per layer) and
    ``update()`` -- once per step, three tiny launches for all layers -- turns it into the next step's scale
    448 * MARGIN / amax.  A layer's first launch scales from the tensor at hand.

    One instance per model (``TrainModel.fp8``): scale slots, layer list, hand-off state, switches and counters.  The operators
    reach the state of the model that is running through the module-level ``Fp8`` proxy; ``TrainModel`` / ``Trainer`` activate
    their own instance before they touch a layer, so two models alive in one process (a trainer plus an evaluation model, a
    test sweep) never share slots.  Layers cache slot numbers of THEIR model's state (``generation`` guards against a layer
    being driven under a foreign state)."""
    MARGIN = 0.5            # headroom for the step-to-step growth of amax (e4m3 max normal = 448)
    MAX_LAYERS = 512
    E5M2_MAX = 57344.0
    _generations = 0

    def __init__(self, enabled=False, copy_only=None):
        Fp8State._generations += 1
        self.generation = Fp8State._generations
        self.enabled = bool(enabled)
        self.amax = None            # [MAX_LAYERS, 64] fp32
        self.scale = None           # [MAX_LAYERS] fp32
        self.fmax = None            # [MAX_LAYERS] fp32: largest normal of the slot's format
        self.count = 0
        self.layers = []            # trainable layers with e4m3 weights (re-converted by after_step)
        self.GRAD = True            # input gradients of the fp8 layers on e5m2 operands (False: bf16 kernels)
        # weight gradients of the wide 3x3 layers on e4m3 x e5m2 operands (conv_wgrad3x3_f8.hip) wherever the gradient arrives with
        # its producer's e5m2 copy; UPS_F8_WGRAD=0: bf16 weight gradients (A/B runs)
        self.WGRAD = SW.flag("UPS_F8_WGRAD")
        # COPY_ONLY: a layer takes the fp8 kernels only when its operand arrives quantised (a copy written by the producing
        # bilinear / convolution kernel); layers whose operand would have to be converted inside the kernel (24 staging
        # registers, one block per CU: slower than bf16) stay on the bf16 kernels.  None: follows PRODUCER.
        # (config key `fp8_copy_only`)
        self.COPY_ONLY = copy_only
        # fp8 copies handed from layer to layer: the producing convolution's epilogue writes e4m3(act(out) * scale) next to its
        # bf16 output (scale = the delayed scale of that tensor), the consuming convolution stages those bytes without any
        # conversion.  nets.Scope passes the handle along: next_in / next_out_act are set right before ops.conv, last_out is
        # read right after.  UPS_F8_PRODUCER=0 switches the hand-off off (every eligible layer then converts its bf16 operand
        # inside the kernel).
        self.PRODUCER = SW.flag("UPS_F8_PRODUCER")
        self.next_in = None         # {"t": uint8 tensor, "act": UPS_ACT_*, "slot": scale slot} of the coming call's input
        self.next_out_act = None    # activation-on-load of the consumer of the coming call's output (None: no copy wanted)
        self.last_out = None        # the copy the last call wrote (same dict), or None
        self.steps = 0              # update() calls so far (a tensor's copy starts one step after its first maximum was recorded)
        self.stats = {"fwd_f8": 0, "fwd_copy_in": 0, "fwd_copy_out": 0, "dgrad_f8": 0, "dgrad_copy_in": 0, "dgrad_copy_out": 0,
                      "wgrad_f8": 0}
        self.grad_side = {}         # data_ptr of a gradient tensor -> (weakref to it, its e5m2 copy): dgrad epilogue -> next dgrad

    def slot(self, device):
        if self.amax is None or self.amax.device != device:
            self.amax = torch.zeros((self.MAX_LAYERS, 64), dtype=torch.float32, device=device)
            self.scale = torch.ones((self.MAX_LAYERS,), dtype=torch.float32, device=device)
            self.fmax = torch.full((self.MAX_LAYERS,), 448.0, dtype=torch.float32, device=device)   # e4m3; gradient slots: e5m2
            self.count = 0
            self.layers = []
        i = self.count
        self.count += 1
        if i >= self.MAX_LAYERS:
            raise L.UpsError("more than {} fp8 layers".format(self.MAX_LAYERS))
        return i

    def update(self):
        """Next step's activation scales from this step's maxima (layers that did not run keep theirs)."""
        if self.amax is None or self.count == 0:
            return
        n = self.count
        m = self.amax[:n].amax(dim=1)
        self.scale[:n] = torch.where(m > 0, (self.fmax[:n] * self.MARGIN) / m.clamp_min(1e-30), self.scale[:n])
        self.amax[:n].zero_()
        self.steps += 1

    def after_step(self):
        """After the optimizer step: new activation scales, e4m3 copies of the updated weights (one launch per layer)."""
        self.update()
        self.grad_side.clear()          # copies nobody read this step are released (they pinned a gradient-sized tensor each)
        for lay in self.layers:
            for key, tr in (("f8", 0), ("f8g", 1)):
                ent = lay._cache.get(key)
                if ent is not None:
                    L.call("ups_weight_prep_f8", L.ptr(lay.V), lay.k * lay.k, lay.cin_v, lay.ci_log, lay.co, tr,
                           L.ptr(ent["w"]), L.ptr(ent["deq"]), L.stream())
                    ent["version"] = WeightVersion.value

    @staticmethod
    def wanted(site):
        """A producer keeps writing its copy only while somebody reads it: after two unread copies the site goes quiet."""
        return not (site.get("emitted", 0) >= 2 and site.get("used", 0) == 0)

    @staticmethod
    def mark_used(copy):
        site = copy.get("site")
        if site is not None:
            site["used"] = site.get("used", 0) + 1

    def copy_only(self):
        return self.PRODUCER if self.COPY_ONLY is None else self.COPY_ONLY

    @staticmethod
    def usable(src, layer, x, ldi):
        """src (a producer's fp8 copy handle or None) is this layer's input, quantised with this layer's activation-on-load."""
        return (src is not None and src.get("t") is not None and src["act"] == layer.act_in and tuple(src["t"].shape) == tuple(x.shape)
                and layer.co > 32 and ldi % 16 == 0)

    def register_grad_copy(self, t, copy):
        import weakref
        if len(self.grad_side) > 256:
            self.grad_side.clear()
        # t._version: the autograd engine's InputBuffer accumulates other branches INTO a buffered gradient in place when it
        # holds the last reference (a weakref does not count as one) -- the same object then arrives holding g1 + g2 while the
        # copy still holds quantised g1; every in-place add bumps the version counter
        self.grad_side[t.data_ptr()] = (weakref.ref(t), tuple(t.shape), copy, t._version)

    def grad_copy(self, g):
        """The e5m2 copy of exactly this tensor object, unmodified since the copy was written, or None."""
        ent = self.grad_side.pop(g.data_ptr(), None)
        if ent is None or ent[0]() is not g or ent[1] != tuple(g.shape) or ent[3] != g._version:
            return None
        return ent[2]

    def layer_entry(self, layer, key):
        """A layer's cached fp8 entry (slot numbers, converted weights) -- only if it was made under THIS state."""
        ent = layer._cache.get(key)
        if ent is not None and ent.get("gen") != self.generation:
            raise L.UpsError("{}: fp8 entry '{}' belongs to another model's Fp8State (generation {} != {}): a layer is being "
                             "driven while a different model's state is active".format(layer.name, key, ent.get("gen"), self.generation))
        return ent

    def eligible_grad(self, layer, g, x):
        """Input gradient of a stride-1 3x3 layer on fp8 operands: K = output channels of the forward."""
        return (self.enabled and self.GRAD and g.dtype == torch.bfloat16 and layer.k == 3 and layer.stride == 1
                and x.shape[1] % 16 == 0 and x.shape[2] % 16 == 0 and round8(layer.co) % 64 == 0 and layer.ci_log >= 64
                and g.shape[-1] >= round8(layer.co))

    def eligible_wgrad(self, layer, g, x, mask):
        """Weight gradient on the block-scaled fp8 MFMA (mirrors eligible8 of conv_wgrad3x3_f8.hip): K = pixels."""
        return (self.enabled and self.WGRAD and mask is None and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
                and layer.k == 3 and layer.stride == 1 and x.shape[1] % 16 == 0 and x.shape[2] % 16 == 0
                and round8(layer.ci_log) % 64 == 0 and layer.co % 128 == 0 and g.shape[-1] % 16 == 0)

    def eligible(self, layer, x):
        return (self.enabled and x.dtype == torch.bfloat16 and layer.k == 3 and layer.stride == 1
                and x.shape[1] % 16 == 0 and x.shape[2] % 16 == 0 and round8(layer.ci_log) % 64 == 0 and layer.co >= 64)


class _Fp8Proxy(object):
    """``ops.Fp8``: attribute access goes to the ACTIVE Fp8State (the running model's; a disabled default otherwise)."""

    def __init__(self):
        object.__setattr__(self, "_cur", Fp8State(False))

    def activate(self, state):
        object.__setattr__(self, "_cur", state)
        return state

    @property
    def current(self):
        return object.__getattribute__(self, "_cur")

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_cur"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_cur"), name, value)


Fp8 = _Fp8Proxy()


class fp8_scope(object):
    """``with ops.fp8_scope(enabled=True, copy_only=False) as F:`` -- a fresh Fp8State active inside the block (kernel-level tests,
    tools), the previous one restored afterwards."""

    def __init__(self, enabled=True, copy_only=None):
        self.state = Fp8State(enabled, copy_only)

    def __enter__(self):
        self.prev = Fp8.current
        return Fp8.activate(self.state)

    def __exit__(self, *a):
        Fp8.activate(self.prev)


class ConvLayer(object):
    """One conv2d variable pair (V [kh,kw,Cin(+2),Cout] HWIO fp32, b [Cout]) + its launch geometry."""

    def __init__(self, name, V, b, k, stride, coords, act_in, slope=0.2):
        self.name, self.V, self.b = name, V, b
        self.k, self.stride, self.coords = k, stride, coords
        self.act_in = L.ACT[act_in] if not isinstance(act_in, int) else act_in
        self.slope = slope
        self.cin_v = V.shape[2]
        self.ci_log = self.cin_v - (2 if coords else 0)
        self.co = V.shape[3]
        self.grad_V = None          # optional preallocated views into a flat gradient buffer
        self.grad_b = None
        self.frozen = False         # frozen weights (perceptual trunk): converted copies survive optimizer steps
        self.after_wgrad = None     # optional callback run right after this layer's weight gradient has been enqueued
        self.registry = None        # PrepRegistry of the owning model (batched refresh) or None (lazy per-layer prep)
        self.f16 = False            # forward tensors of this layer are fp16 (nets.Scope.fmt)
        # post-activation storage (ups_conv_desc.out_act / res_act): in_post = the input tensor already holds act_in(x) -- the
        # forward and the weight gradient stage it as it is (LDS-DMA patch), the input gradient still reads act' off its sign;
        # out_act = the output is stored as out_act(y) because its consumer would apply that activation on load
        self.in_post = False
        self.out_act = L.ACT_NONE
        self._cache = {}

    def _mark_ready(self, ent):
        """A lazy conversion was just enqueued on the current stream: remember where, so that a consumer on ANOTHER stream of the same
        step (the appearance encoder first runs on "aux", the frozen trunk on "pre", both again on the launching stream) waits for
        it instead of racing with it.  The batched refresh at the end of a step runs on the launching stream, from which every
        side stream forks afterwards: it clears the mark."""
        dev = self.V.device
        if torch.cuda.is_current_stream_capturing():       # (captures start after eager steps: nothing converts lazily in them)
            ent["ready"] = None
            return
        ent["ready"] = (L.raw_stream(dev), torch.cuda.current_stream(dev).record_event(), set(), Streams.epoch)

    def _wait_ready(self, ent):
        rd = ent.get("ready")
        if rd is not None:
            if rd[3] != Streams.epoch:          # an earlier step's conversion: every stream has been joined and re-forked since
                ent["ready"] = None
                return
            cur = L.raw_stream(self.V.device)
            if cur != rd[0] and cur not in rd[2] and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(self.V.device).wait_event(rd[1])
                rd[2].add(cur)

    def _wait_master(self):
        """Before reading the fp32 master weights on this stream: wait for an Adam launch of this layer's optimizer key that is
        still in flight on the weight-gradient stream (the trainer's early per-key Adam; keys match by substring, as edflow's
        variable lists do)."""
        if Streams.master_busy:
            cur = torch.cuda.current_stream(self.V.device)
            for key, ev in Streams.master_busy.items():
                if key in self.name:
                    cur.wait_event(ev)

    # ---- converted weights (refreshed when the optimizer has stepped)
    def prepared(self, dtype_code, hi, wi, need_dgrad):
        key = (dtype_code, hi, wi)
        ent = self._cache.get(key)
        dev = self.V.device
        td = L.torch_dtype(dtype_code)
        ntaps = self.k * self.k
        ci_pad = round8(self.ci_log)
        bk = 16 if dtype_code == L.F32 else 32          # blocked-K layout [tap][k-chunk][row][64 B]
        if ent is None:                                 # persistent buffers (pointers stay valid for the batched refresh)
            # a layer with fp16 forward tensors: forward weights as fp16, input-gradient weights as bf16 (one copy each)
            want_f = not (self.f16 and dtype_code == L.BF16)
            want_d = dtype_code != L.F16
            ent = {"version": -1,
                   "w_fwd": torch.empty((ntaps, -(-ci_pad // bk), self.co, bk), dtype=td, device=dev) if want_f else None,
                   "w_dgrad": torch.empty((ntaps, -(-round8(self.co) // bk), self.ci_log, bk), dtype=td, device=dev) if want_d else None,
                   "ctab": torch.empty((64, 3, self.co), dtype=torch.float32, device=dev) if (self.coords and want_f) else None}
            self._cache[key] = ent
            if self.registry is not None and not self.frozen:
                self.registry.register(self, ent, dtype_code, hi, wi)
        if ent["version"] != WeightVersion.value and not (self.frozen and ent["version"] >= 0):
            self._wait_master()
            L.call("ups_weight_prep", L.ptr(self.V), ntaps, self.cin_v, self.ci_log, self.co, dtype_code,
                   L.ptr(ent["w_fwd"]), ci_pad, L.ptr(ent["w_dgrad"]), self.ci_log, round8(self.co), L.stream())
            if ent["ctab"] is not None:
                dy, dx, _ = self.fwd_taps(hi, wi)
                ax, ay = 2.0 / max(1, hi - 1), 2.0 / max(1, wi - 1)     # nn.py:2145-2148 (xx / (H-1), yy / (W-1))
                L.call("ups_coord_table", L.ptr(self.V), self.k, self.k, self.ci_log, self.co,
                       (C.c_int32 * 9)(*dy), (C.c_int32 * 9)(*dx), self.stride, self.stride, ax, ay,
                       L.ptr(ent["ctab"]), L.stream())
            ent["version"] = WeightVersion.value
            self._mark_ready(ent)
        else:
            self._wait_ready(ent)
        return ent

    def d2s_channels(self, x):
        """C of the depth-to-space input gradient (ups_conv_desc.d2s) when this layer / tensor can take it, else 0."""
        c = self.ci_log
        ok = (D2S_DGRAD and self.k == 3 and self.stride == 2 and x.dtype == torch.bfloat16 and c >= 8 and (c & (c - 1)) == 0
              and x.shape[-1] == c and x.shape[1] % 32 == 0 and x.shape[2] % 32 == 0)
        return c if ok else 0

    def prepared_d2s(self, hi, wi):
        """Weights of the one-launch input gradient of a 3x3 / stride-2 layer (ups_weight_prep_d2s)."""
        ent = self._cache.get("d2s")
        if ent is None:
            kc = -(-self.co // 32)
            ent = {"version": -1, "w": torch.empty((9, kc, 4 * self.ci_log, 32), dtype=torch.bfloat16, device=self.V.device)}
            _, pby = same_geometry(hi, 3, 2)
            _, pbx = same_geometry(wi, 3, 2)

            def prep():
                self._wait_master()
                L.call("ups_weight_prep_d2s", L.ptr(self.V), self.cin_v, self.ci_log, self.co, pby, pbx, self.ci_log,
                       L.ptr(ent["w"]), L.stream())
                ent["version"] = WeightVersion.value
            ent["prep"] = prep
            self._cache["d2s"] = ent
            if self.registry is not None and not self.frozen:
                self.registry.extra.append(prep)
        if ent["version"] != WeightVersion.value and not (self.frozen and ent["version"] >= 0):
            ent["prep"]()
            self._mark_ready(ent)
        else:
            self._wait_ready(ent)
        return ent

    def prepared_f8_grad(self, g):
        """e4m3 weights of the input-gradient GEMM (rows = input channels, scaled per row) + the gradient tensor's e5m2 scale slot."""
        ent = Fp8.layer_entry(self, "f8g")
        dev = self.V.device
        if ent is None:
            kc = -(-self.co // 64)
            ent = {"version": -1, "slot": Fp8.slot(dev), "primed": False, "gen": Fp8.generation,
                   "w": torch.empty((self.k * self.k, kc, self.ci_log, 64), dtype=torch.uint8, device=dev),
                   "deq": torch.empty((self.ci_log,), dtype=torch.float32, device=dev)}
            self._cache["f8g"] = ent
            Fp8.fmax[ent["slot"]] = Fp8.E5M2_MAX
            if not self.frozen and self not in Fp8.layers:
                Fp8.layers.append(self)
        if ent["version"] != WeightVersion.value and not (self.frozen and ent["version"] >= 0):
            self._wait_master()
            L.call("ups_weight_prep_f8", L.ptr(self.V), self.k * self.k, self.cin_v, self.ci_log, self.co, 1,
                   L.ptr(ent["w"]), L.ptr(ent["deq"]), L.stream())
            ent["version"] = WeightVersion.value
        if not ent["primed"] and g is not None:
            m = g[..., :self.co].abs().amax().float()
            Fp8.scale[ent["slot"]] = torch.where(m > 0, (Fp8.E5M2_MAX * Fp8.MARGIN) / m.clamp_min(1e-30), torch.ones_like(m))
            ent["primed"] = True
        return ent

    def prepared_f8_wgrad(self, x, fmt):
        """Scale slot of the forward input as the fp8 weight gradient quantises it (delayed scaling: the kernel records
        max |act(x)|, ops.Fp8.update turns it into the next step's scale; the first launch scales from the tensor at hand)."""
        ent = Fp8.layer_entry(self, "f8w")
        if ent is None:
            ent = self._cache["f8w"] = {"slot": Fp8.slot(self.V.device), "primed": False, "gen": Fp8.generation}
        if not ent["primed"]:
            xv = x.view(torch.float16) if fmt == L.F16 else x       # (fp16 forward tensors live in bf16 containers)
            m = xv[..., :round8(self.ci_log)].abs().amax().float()
            Fp8.scale[ent["slot"]] = torch.where(m > 0, (448.0 * Fp8.MARGIN) / m.clamp_min(1e-30), torch.ones_like(m))
            ent["primed"] = True
        return ent

    def prepared_f8(self, x):
        """e4m3 weights + per-channel dequantisation factors + this layer's activation-scale slot."""
        ent = Fp8.layer_entry(self, "f8")
        dev = self.V.device
        if ent is None:
            kc = -(-self.ci_log // 64)
            ent = {"version": -1, "slot": Fp8.slot(dev), "primed": False, "gen": Fp8.generation,
                   "w": torch.empty((self.k * self.k, kc, self.co, 64), dtype=torch.uint8, device=dev),
                   "deq": torch.empty((self.co,), dtype=torch.float32, device=dev)}
            self._cache["f8"] = ent
            if not self.frozen and self not in Fp8.layers:
                Fp8.layers.append(self)
        if ent["version"] != WeightVersion.value and not (self.frozen and ent["version"] >= 0):
            self._wait_master()
            L.call("ups_weight_prep_f8", L.ptr(self.V), self.k * self.k, self.cin_v, self.ci_log, self.co, 0,
                   L.ptr(ent["w"]), L.ptr(ent["deq"]), L.stream())
            ent["version"] = WeightVersion.value
        if not ent["primed"] and x is not None:       # first launch of the layer: scale from the tensor at hand (|act(x)| <= |x|)
            m = x[..., :self.ci_log].abs().amax().float()
            Fp8.scale[ent["slot"]] = torch.where(m > 0, (448.0 * Fp8.MARGIN) / m.clamp_min(1e-30), torch.ones_like(m))
            ent["primed"] = True
        return ent

    def out_hw(self, hi, wi):
        return same_geometry(hi, self.k, self.stride)[0], same_geometry(wi, self.k, self.stride)[0]

    def fwd_taps(self, hi, wi):
        _, pby = same_geometry(hi, self.k, self.stride)
        _, pbx = same_geometry(wi, self.k, self.stride)
        dy, dx, tw = [0] * 9, [0] * 9, [0] * 9
        for r in range(self.k):
            for s in range(self.k):
                t = r * self.k + s
                dy[t], dx[t], tw[t] = r - pby, s - pbx, t
        return dy, dx, tw


def _fill_taps(desc, dy, dx, tw, ntaps):
    for t in range(9):
        desc.tap_dy[t] = dy[t] if t < ntaps else 0
        desc.tap_dx[t] = dx[t] if t < ntaps else 0
        desc.tap_w[t] = tw[t] if t < ntaps else 0
    desc.ntaps = ntaps


SPLITK_WS_BYTES = 48 << 20
D2S_DGRAD = not SW.flag("UPS_NO_D2S")      # one-launch input gradient of the stride-2 layers (A/B switch)


def _attach_ws(d, device):
    """Scratch of the split-K path of ups_conv_igemm (per stream; grown once to SPLITK_WS_BYTES)."""
    ws = WORKSPACE.get(SPLITK_WS_BYTES, device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()


def conv_forward(x, layer, res=None, out_f32=False, ldo=None, co_fill=None, mask=None, fmt=None, res_post=False):
    """out = conv(act(x) (+coords), V) + b (+ res);  x [n,hi,wi,ldi].
    fmt = L.F16: x, res and (unless out_f32) out hold fp16 in bf16 containers (module docstring).
    mask = (hard_bits [B,hi,wi] int32, P): x is the UNMASKED view [B,hi,wi,ldi] and the convolution runs on the P*B part images
    x[b] * hard[b,:,:,p] (part-major) without materialising them (model.py:176-187, nn.py:81-113)."""
    n, hi, wi, ldi = x.shape
    if mask is not None:
        n = n * mask[1]
    dcode = L.dt(x) if fmt is None else fmt
    assert dcode != L.F16 or (x.dtype == torch.bfloat16 and mask is None and layer.f16)
    ho, wo = layer.out_hw(hi, wi)
    ent = layer.prepared(dcode, hi, wi, need_dgrad=False)
    ldo = ldo if ldo is not None else (layer.co if out_f32 else round8(layer.co))
    co_fill = co_fill if co_fill is not None else ldo
    out = torch.empty((n, ho, wo, ldo), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    d = L.ConvDesc()
    f8_out = None
    d.dtype = dcode
    d.n, d.hi, d.wi, d.ci, d.ldi = n, hi, wi, round8(layer.ci_log), ldi
    d.ho, d.wo, d.co, d.co_fill, d.ldo = ho, wo, layer.co, co_fill, ldo
    d.out_h, d.out_w, d.out_sy, d.out_sx, d.out_oy, d.out_ox = ho, wo, 1, 1, 0, 0
    d.in_sy = d.in_sx = layer.stride
    dy, dx, tw = layer.fwd_taps(hi, wi)
    _fill_taps(d, dy, dx, tw, layer.k * layer.k)
    d.kh = d.kw = layer.k
    d.act_in, d.act_slope, d.out_f32, d.dact_kind = (L.ACT_NONE if layer.in_post else layer.act_in), layer.slope, int(out_f32), 0
    d.out_act = layer.out_act
    d.res_act = layer.act_in if (res is not None and res_post) else L.ACT_NONE       # residual stored as act(x): inverted in the epilogue
    assert not (d.res_act and layer.act_in != L.ACT_LRELU), "only a leaky-ReLU residual can be stored post-activation"
    d.ldr = res.shape[-1] if res is not None else 0
    d.ldd = 0
    d.in_, d.w, d.out = x.data_ptr(), ent["w_fwd"].data_ptr(), out.data_ptr()
    d.bias = layer.b.data_ptr()
    d.coord_tab = ent["ctab"].data_ptr() if layer.coords else None
    d.res = res.data_ptr() if res is not None else None
    d.dact = None
    if mask is not None:
        d.mask_bits, d.mask_batch = mask[0].data_ptr(), x.shape[0]
    elif dcode != L.F16 and Fp8.eligible(layer, x) and (not Fp8.copy_only() or Fp8.usable(Fp8.next_in, layer, x, ldi)):
        src, want_act = Fp8.next_in, Fp8.next_out_act
        if Fp8.usable(src, layer, x, ldi):
            f8 = layer.prepared_f8(None)               # the producer quantised act(x) with its tensor's scale
            Fp8.stats["fwd_copy_in"] += 1
            Fp8.mark_used(src)
            d.in_f8 = src["t"].data_ptr()
            d.f8_scale = Fp8.scale[src["slot"]:].data_ptr()
        else:
            f8 = layer.prepared_f8(x)
            d.f8_scale = Fp8.scale[f8["slot"]:].data_ptr()
            d.f8_amax = Fp8.amax[f8["slot"]].data_ptr()
        d.w = f8["w"].data_ptr()
        d.f8_deq = f8["deq"].data_ptr()
        Fp8.stats["fwd_f8"] += 1
        if Fp8.PRODUCER and want_act is not None and not out_f32 and ldo % 64 == 0 and co_fill == ldo:
            eo = Fp8.layer_entry(layer, "f8o")
            if eo is None:
                eo = layer._cache["f8o"] = {"slot": Fp8.slot(x.device), "born": Fp8.steps, "gen": Fp8.generation}
            if Fp8.wanted(eo):
                d.out_f8_amax = Fp8.amax[eo["slot"]].data_ptr()
                # a post-activation output already holds want_act(out): its copy is the quantisation of the stored value
                assert not layer.out_act or layer.out_act == want_act
                d.out_f8_act = L.ACT_NONE if layer.out_act else want_act
                if Fp8.steps > eo["born"]:                 # its delayed scale exists
                    t8 = torch.empty(out.shape, dtype=torch.uint8, device=x.device)
                    d.out_f8, d.out_f8_scale = t8.data_ptr(), Fp8.scale[eo["slot"]:].data_ptr()
                    f8_out = {"t": t8, "act": want_act, "slot": eo["slot"], "site": eo}
                    eo["emitted"] = eo.get("emitted", 0) + 1
                    Fp8.stats["fwd_copy_out"] += 1
    Fp8.next_in = Fp8.next_out_act = None
    Fp8.last_out = f8_out
    SignBits.last = None
    if SignBits.want and SignBits.ENABLED and not out_f32 and out.dtype == torch.bfloat16 and ldo % 8 == 0:
        SignBits.last = torch.empty((n, ho, wo, ldo // 8), dtype=torch.uint8, device=x.device)
        d.sign_out = SignBits.last.data_ptr()
    SignBits.want = False
    _attach_ws(d, x.device)
    assert round8(layer.ci_log) <= ldi, (layer.name, layer.ci_log, ldi)
    if KernelTimer.layer == layer.name and KernelTimer.active():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("ups_conv_igemm", C.byref(d), L.stream())
        e1.record()
        KernelTimer.events.append((e0, e1))
        KernelTimer.kinds.append("fwd")
        KernelTimer.flops = 2.0 * n * ho * wo * layer.k * layer.k * layer.cin_v * layer.co
    else:
        L.call("ups_conv_igemm", C.byref(d), L.stream())
    if SignBits.last is not None and not L.load().ups_conv_sign_out_written():
        SignBits.last = None            # the launch went to a kernel that does not write them (best effort: upsparts_hip.h)
    return out


def conv_dgrad(g, x, layer, res=None, mask_view=None, n_parts=0, f8_src="pop", x_bits=None):
    """gx = act'(x) * conv^T(g) (+ res);  g [n,ho,wo,ldg] in the activation dtype, x the forward input.
    mask_view (fp32 [B,hi,wi,3], with n_parts): the forward was the part-masked convolution; returns d loss / d hard
    [B,hi,wi,P] = sum_c gx[p*B+b,...,c] * view[b,...,c] straight from the kernel's epilogue (gx is never written).
    f8_src: the e5m2 copy of g its producer registered (ops.Fp8.grad_copy), None, or "pop" = look it up here.
    x_bits: the sign bytes of x its producer wrote ([n,hi,wi,ldi/8] uint8, ops.SignBits): read instead of x for act'."""
    n, hi, wi, ldi = x.shape
    if x_bits is not None and (tuple(x_bits.shape) != (n, hi, wi, ldi // 8) or layer.act_in == L.ACT_NONE):
        x_bits = None
    if SignBits.stats is not None and layer.act_in != L.ACT_NONE and mask_view is None:
        key = (layer.name, n, hi, wi, ldi, layer.k, layer.stride, x_bits is not None)
        SignBits.stats[key] = SignBits.stats.get(key, 0) + 1
    dcode = L.dt(x)
    ho, wo = layer.out_hw(hi, wi)
    ent = layer.prepared(dcode, hi, wi, need_dgrad=True)
    g_hard = None
    if mask_view is not None:
        assert layer.stride == 1 and layer.act_in == L.ACT_NONE and res is None
        g_hard = torch.empty((n, hi, wi, n_parts), dtype=torch.float32, device=x.device)
        gx = None
        n_img = n * n_parts
    else:
        gx = torch.empty_like(x)
        n_img = n
    k, st = layer.k, layer.stride
    _, pby = same_geometry(hi, k, st)
    _, pbx = same_geometry(wi, k, st)
    cd2s = layer.d2s_channels(x) if mask_view is None else 0
    if cd2s:
        # one stride-1 3x3 convolution over the gradient lattice with 4 C channels (parity class, channel), written
        # depth-to-space: g is read once and whole output rows are stored (the per-class launches below store every other pixel)
        d = L.ConvDesc()
        d.dtype = dcode
        d.n, d.hi, d.wi, d.ci, d.ldi = n, ho, wo, round8(layer.co), g.shape[-1]
        d.ho, d.wo, d.co, d.co_fill, d.ldo = ho, wo, 4 * cd2s, 4 * cd2s, ldi
        d.out_h, d.out_w, d.out_sy, d.out_sx, d.out_oy, d.out_ox = hi, wi, 1, 1, 0, 0
        d.in_sy = d.in_sx = 1
        _fill_taps(d, [t // 3 - 1 for t in range(9)], [t % 3 - 1 for t in range(9)], list(range(9)), 9)
        d.kh = d.kw = 3
        d.act_in, d.act_slope, d.out_f32 = 0, layer.slope, 0
        d.dact_kind = layer.act_in
        d.ldr = res.shape[-1] if res is not None else 0
        d.ldd = ldi
        d.in_, d.w, d.out = g.data_ptr(), layer.prepared_d2s(hi, wi)["w"].data_ptr(), gx.data_ptr()
        d.bias, d.coord_tab = None, None
        d.res = res.data_ptr() if res is not None else None
        d.dact = x.data_ptr() if layer.act_in != L.ACT_NONE else None
        d.dact_bits = x_bits.data_ptr() if (x_bits is not None and d.dact) else None
        d.d2s = cd2s
        _attach_ws(d, x.device)
        L.call("ups_conv_igemm", C.byref(d), L.stream())
        return gx
    classes = [(0, 0)] if st == 1 else [(py, px) for py in range(st) for px in range(st)]
    for (py, px) in classes:
        lat_h = (hi - py + st - 1) // st
        lat_w = (wi - px + st - 1) // st
        if lat_h <= 0 or lat_w <= 0:
            continue
        dy, dx, tw = [], [], []
        for r in range(k):
            if (py + pby - r) % st:
                continue
            for s in range(k):
                if (px + pbx - s) % st:
                    continue
                dy.append((py + pby - r) // st); dx.append((px + pbx - s) // st); tw.append(r * k + s)
        d = L.ConvDesc()
        d.dtype = dcode
        d.n, d.hi, d.wi, d.ci, d.ldi = n_img, ho, wo, round8(layer.co), g.shape[-1]
        d.ho, d.wo, d.co, d.co_fill, d.ldo = lat_h, lat_w, layer.ci_log, ldi, ldi
        d.out_h, d.out_w, d.out_sy, d.out_sx, d.out_oy, d.out_ox = hi, wi, st, st, py, px
        d.in_sy = d.in_sx = 1
        if not dy:      # no tap reaches this parity class: gradient is zero there (still apply res)
            dy, dx, tw = [0], [0], [0]
            raise L.UpsError("empty tap class is not expected for k=3/k=1 'SAME' convolutions")
        _fill_taps(d, dy + [0] * 9, dx + [0] * 9, tw + [0] * 9, len(dy))
        d.kh, d.kw = 1, len(dy)
        d.act_in, d.act_slope, d.out_f32 = 0, layer.slope, 0
        d.dact_kind = layer.act_in
        d.ldr = res.shape[-1] if res is not None else 0
        d.ldd = ldi
        d.in_, d.w, d.out = g.data_ptr(), ent["w_dgrad"].data_ptr(), gx.data_ptr() if gx is not None else None
        d.bias, d.coord_tab = None, None
        d.res = res.data_ptr() if res is not None else None
        d.dact = x.data_ptr() if layer.act_in != L.ACT_NONE else None
        d.dact_bits = x_bits.data_ptr() if (x_bits is not None and d.dact) else None
        if mask_view is not None:
            d.mask_grad, d.mask_view, d.mask_batch = g_hard.data_ptr(), mask_view.data_ptr(), n
        elif st == 1 and Fp8.enabled and Fp8.GRAD and g.dtype == torch.bfloat16:
            src = Fp8.grad_copy(g) if isinstance(f8_src, str) else f8_src
            use_f8 = Fp8.eligible_grad(layer, g, x)
            src_ok = use_f8 and src is not None and layer.ci_log > 32 and g.shape[-1] % 16 == 0
            emit = False
            if use_f8 and (src_ok or not Fp8.copy_only()):
                if src_ok:
                    f8 = layer.prepared_f8_grad(None)      # the producer wrote e5m2(g * scale) in its epilogue
                    Fp8.stats["dgrad_copy_in"] += 1
                    Fp8.mark_used(src)
                    d.in_f8 = src["t"].data_ptr()
                    d.f8_scale = Fp8.scale[src["slot"]:].data_ptr()
                else:
                    f8 = layer.prepared_f8_grad(g)
                    d.f8_scale = Fp8.scale[f8["slot"]:].data_ptr()
                    d.f8_amax = Fp8.amax[f8["slot"]].data_ptr()
                d.w = f8["w"].data_ptr()
                d.f8_deq = f8["deq"].data_ptr()
                d.f8_e5m2 = 1
                Fp8.stats["dgrad_f8"] += 1
                emit = True
            elif layer.k == 3 and round8(layer.co) <= 32 and hi % 16 == 0 and wi % 16 == 0:
                emit = True     # a bf16 launch off the 128-wide two-blocks-per-CU instance (the P-channel head): it can write the copy
            if emit and Fp8.PRODUCER and gx is not None and ldi % 64 == 0 and layer.ci_log == ldi:
                eo = Fp8.layer_entry(layer, "f8go")
                if eo is None:
                    eo = layer._cache["f8go"] = {"slot": Fp8.slot(x.device), "born": Fp8.steps, "gen": Fp8.generation}
                    Fp8.fmax[eo["slot"]] = Fp8.E5M2_MAX
                if Fp8.wanted(eo):
                    d.out_f8_amax = Fp8.amax[eo["slot"]].data_ptr()
                    d.out_f8_act, d.out_f8_e5m2 = L.ACT_NONE, 1
                    if Fp8.steps > eo["born"]:
                        t8 = torch.empty(gx.shape, dtype=torch.uint8, device=x.device)
                        d.out_f8, d.out_f8_scale = t8.data_ptr(), Fp8.scale[eo["slot"]:].data_ptr()
                        Fp8.register_grad_copy(gx, {"t": t8, "slot": eo["slot"], "site": eo})
                        eo["emitted"] = eo.get("emitted", 0) + 1
                        Fp8.stats["dgrad_copy_out"] += 1
        _attach_ws(d, x.device)
        assert round8(layer.co) <= g.shape[-1]
        if KernelTimer.layer == layer.name and KernelTimer.active() and st == 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.call("ups_conv_igemm", C.byref(d), L.stream())
            e1.record()
            KernelTimer.events.append((e0, e1))
            KernelTimer.kinds.append("dgrad")
        else:
            L.call("ups_conv_igemm", C.byref(d), L.stream())
    return gx if mask_view is None else g_hard


COORD_STREAM = SW.flag("UPS_COORD_STREAM")     # A/B switch: CoordConv rows of the weight gradients on their own stream


def conv_wgrad(g, x, layer, mask=None, fmt=None, launch_stream=None, f8_src=None):
    """(dV [kh,kw,cin_v,co] fp32, db [co] fp32).  mask = (hard_bits, P): x is the unmasked view of a part-masked convolution.
    fmt = L.F16: x holds fp16 (converted to bf16, the gradient's type, while it is staged).
    launch_stream: the stream whose position marks "g and x are ready" when the call itself runs on the weight-gradient side stream:
    the CoordConv rows (batch sum of g + two small kernels: they write rows of dV the main kernel does not touch) then go to a
    second side stream, beside the layer's main weight-gradient kernel instead of behind it -- the weight-gradient stream is a
    serial chain that ends the step (tools/timeline.py), and these ~40 small launches were 1 ms of it.
    f8_src: the e5m2 copy of g (the handle ops.Fp8.grad_copy returns): the wide 3x3 / stride-1 layers then run on the fp8 kernel
    (ups_wgrad_desc.dout_f8; x is quantised while it is staged, with this layer's delayed scale)."""
    n, hi, wi, ldi = x.shape
    if mask is not None:
        n = n * mask[1]
    dcode = L.dt(x)
    ho, wo = layer.out_hw(hi, wi)
    dev = x.device
    gV = layer.grad_V if layer.grad_V is not None else torch.empty_like(layer.V)
    gb = layer.grad_b if layer.grad_b is not None else torch.empty_like(layer.b)
    d = L.WgradDesc()
    d.dtype = dcode
    d.n, d.hi, d.wi, d.ci, d.ldi = n, hi, wi, round8(layer.ci_log), ldi
    d.ci_log, d.cin_v = layer.ci_log, layer.cin_v
    d.ho, d.wo, d.co, d.ldo = ho, wo, layer.co, g.shape[-1]
    d.in_sy = d.in_sx = layer.stride
    dy, dx, tw = layer.fwd_taps(hi, wi)
    _fill_taps(d, dy, dx, tw, layer.k * layer.k)
    d.act_in, d.act_slope = (L.ACT_NONE if layer.in_post else layer.act_in), layer.slope
    sk, wsb = C.c_int32(0), C.c_size_t(0)
    d.in_, d.dout, d.grad, d.grad_bias = x.data_ptr(), g.data_ptr(), gV.data_ptr(), gb.data_ptr()
    if mask is not None:
        d.mask_bits, d.mask_batch = mask[0].data_ptr(), x.shape[0]
    d.in_f16 = int(fmt == L.F16)
    if f8_src is not None and f8_src.get("t") is not None and Fp8.eligible_wgrad(layer, g, x, mask) \
            and tuple(f8_src["t"].shape) == tuple(g.shape):
        ew = layer.prepared_f8_wgrad(x, fmt)
        d.dout_f8 = f8_src["t"].data_ptr()
        d.dout_f8_scale = Fp8.scale[f8_src["slot"]:].data_ptr()
        d.in_f8_scale = Fp8.scale[ew["slot"]:].data_ptr()
        d.in_f8_amax = Fp8.amax[ew["slot"]].data_ptr()
        Fp8.stats["wgrad_f8"] += 1
        Fp8.mark_used(f8_src)
    L.call("ups_conv_wgrad_plan", C.byref(d), C.byref(sk), C.byref(wsb))
    ws = WORKSPACE.get(wsb.value, dev)
    d.splitk, d.workspace = sk.value, ws.data_ptr()
    L.call("ups_conv_wgrad", C.byref(d), L.stream())        # dV (main channels) + db from the same dout tiles
    if layer.coords:
        def coord_rows():
            gsum = torch.empty((ho * wo, layer.co), dtype=torch.float32, device=dev)
            L.call("ups_batch_sum", L.ptr(g), dcode, n, ho * wo, layer.co, g.shape[-1], L.ptr(gsum), L.stream())
            ax, ay = 2.0 / max(1, hi - 1), 2.0 / max(1, wi - 1)
            scratch = COLSUM_WS.get((layer.k + 1) * 2 * wo * layer.co * 4, dev)
            L.call("ups_coord_wgrad", L.ptr(gsum), hi, wi, ho, wo, layer.co, layer.k, layer.k,
                   (C.c_int32 * 9)(*dy), (C.c_int32 * 9)(*dx), layer.stride, layer.stride, ax, ay,
                   layer.ci_log, L.ptr(gV), None, L.ptr(scratch), L.stream())
        if launch_stream is not None and COORD_STREAM and Streams.enabled:
            s2 = Streams.get("wgrad2", dev)
            s2.wait_stream(launch_stream)
            with torch.cuda.stream(s2):
                coord_rows()
        else:
            coord_rows()
    return gV, gb


def to_act_dtype(g, like_dtype, co):
    """fp32 gradient [.., c] of an fp32-output conv -> activation dtype with 8-padded channels."""
    ld = round8(co)
    if g.dtype == like_dtype and g.shape[-1] == ld:
        return g.contiguous()
    g = g.contiguous()
    assert g.dtype == torch.float32
    out = torch.empty(g.shape[:-1] + (ld,), dtype=like_dtype, device=g.device)
    rows = g.numel() // g.shape[-1]
    L.call("ups_pad_convert", L.ptr(g), g.shape[-1], L.ptr(out), L.dt(out), ld, rows, L.stream())
    return out


class GradMode(object):
    """ctx.needs_input_grad is fixed when the tape is recorded, not per autograd.grad call; a backward pass
    that only wants input gradients (e.g. d rec / d z through the mask decoder) sets skip_wgrad."""
    skip_wgrad = False


class skip_wgrad(object):
    def __enter__(self):
        self.prev, GradMode.skip_wgrad = GradMode.skip_wgrad, True

    def __exit__(self, *a):
        GradMode.skip_wgrad = self.prev


class ConvFn(torch.autograd.Function):
    """res_mode 0: plain; 1: out = res + conv(x); 2: out = x + conv(act(x)) (residual_block, nn.py:1042-1056)."""

    @staticmethod
    def forward(ctx, x, V, b, res, layer, res_mode, out_f32, ldo, hard=None, hard_bits=None, view_f32=None, fmt=None, res_post=False,
                x_bits=None):
        """hard / hard_bits / view_f32 given: the part-masked convolution (x = the unmasked view in the activation dtype,
        the P*B part images are formed in the kernel's load); the gradient w.r.t. `hard` comes out of the dgrad epilogue."""
        x = x.contiguous()
        r = x if res_mode == 2 else (res.contiguous() if res_mode == 1 else None)
        ctx.mask = None if hard is None else (hard_bits, hard.shape[-1])
        # res_mode 2: the residual is the input itself -- stored post-activation exactly when the layer's input is
        out = conv_forward(x, layer, res=r, out_f32=out_f32, ldo=ldo, mask=ctx.mask, fmt=fmt,
                           res_post=layer.in_post if res_mode == 2 else bool(res_post))
        ctx.save_for_backward(x, view_f32)
        ctx.layer, ctx.res_mode, ctx.fmt = layer, res_mode, fmt
        ctx.x_bits = x_bits             # sign bytes of x from its producer (SignBits): the input gradient reads them instead of x
        return out

    @staticmethod
    def backward(ctx, g):
        x, view_f32 = ctx.saved_tensors
        layer = ctx.layer
        g = to_act_dtype(g, x.dtype, layer.co)
        gx = gV = gb = gres = g_hard = None
        offloaded = False
        # the e5m2 copy of g its producer wrote (fp8 mode): looked up ONCE, read by the weight gradient and by the input gradient
        f8_src = Fp8.grad_copy(g) if (Fp8.enabled and Fp8.GRAD and ctx.mask is None and layer.stride == 1
                                      and g.dtype == torch.bfloat16) else None
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not GradMode.skip_wgrad:
            if Streams.enabled and layer.grad_V is not None and not Streams.on_aux(x.device):
                offloaded = True
                # weight gradient on the side stream (it lands in the layer's view of the flat gradient bucket, which
                # nothing reads before Streams.join); g and x must outlive the side stream's reads
                cur = torch.cuda.current_stream(x.device)
                side = Streams.get("wgrad", x.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    gV, gb = conv_wgrad(g, x, layer, mask=ctx.mask, fmt=ctx.fmt, launch_stream=cur, f8_src=f8_src)
                Streams.keep(x.device, g, x)      # alive until the launching stream has joined the side stream(s)
                if f8_src is not None:
                    Streams.keep(x.device, f8_src["t"])
            else:
                gV, gb = conv_wgrad(g, x, layer, mask=ctx.mask, fmt=ctx.fmt, f8_src=f8_src)
            if layer.after_wgrad is not None:
                layer.after_wgrad()
        if ctx.mask is not None:
            if ctx.needs_input_grad[8]:
                g_hard = conv_dgrad(g, x, layer, mask_view=view_f32, n_parts=ctx.mask[1])
        elif ctx.needs_input_grad[0]:
            gx = conv_dgrad(g, x, layer, res=g if ctx.res_mode == 2 else None, f8_src=f8_src if layer.stride == 1 else "pop",
                            x_bits=ctx.x_bits)
        if ctx.res_mode == 1 and ctx.needs_input_grad[3]:
            # the autograd engine may accumulate other branches into the returned tensor IN PLACE; the side stream
            # is still reading g, so hand out a copy in that case
            gres = g.clone() if offloaded else g
        return gx, gV, gb, gres, None, None, None, None, g_hard, None, None, None, None, None


def conv(x, layer, res=None, res_self=False, out_f32=False, ldo=None, mask=None, fmt=None, res_post=False, x_bits=None):
    """mask = (hard [B,H,W,P] fp32 autograd leaf, hard_bits [B,H,W] int32, view_f32 [B,H,W,3]): part-masked convolution.
    res_post: `res` is stored post-activation (ups_conv_desc.res_act).  x_bits: the sign bytes of x (SignBits)."""
    mode = 2 if res_self else (1 if res is not None else 0)
    if mask is None:
        return ConvFn.apply(x, layer.V, layer.b, res, layer, mode, out_f32, ldo, None, None, None, fmt, res_post, x_bits)
    assert mode == 0
    return ConvFn.apply(x, layer.V, layer.b, None, layer, 0, out_f32, ldo, mask[0], mask[1], mask[2].contiguous(), None, False)


def masked_conv_eligible(dtype, size, n_parts):
    """The fused form runs on the bf16 3x3 / stride-1 patch kernels: 16-aligned images, at most 32 parts."""
    return dtype == torch.bfloat16 and size % 16 == 0 and size >= 16 and n_parts <= 32


class BilinearFn(torch.autograd.Function):
    """site: per-call-site state (a dict owned by the Scope) when the up-sampling feeds fp8 convolutions: its forward then also
    writes the e4m3 copy of act(y), its backward the e5m2 copy of the gradient it returns (ops.Fp8 hand-off)."""

    @staticmethod
    def forward(ctx, x, site=None, act=0, slope=0.2, fmt=None, out_act=0):
        """out_act: the result is stored as out_act(y) (post-activation storage for a consuming residual block; gradients stay
        with respect to y, so the backward is unchanged).  `site` with an fp16 / post-activation forward (the mask decoder in fp8
        mode: its forward stays fp16, round 4): only the BACKWARD hands an e5m2 copy of the gradient on."""
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
        f8_site = site is not None and Fp8.enabled and Fp8.PRODUCER and x.dtype == torch.bfloat16 and c % 64 == 0 and (2 * h) % 16 == 0
        ctx.site, ctx.shape = (site if f8_site else None), (n, h, w, c)
        if out_act:
            SignBits.last = None
            if SignBits.want and SignBits.ENABLED and x.dtype == torch.bfloat16 and c % 8 == 0:
                SignBits.last = torch.empty((n, 2 * h, 2 * w, c // 8), dtype=torch.uint8, device=x.device)
                L.call("ups_bilinear2x_fwd_bits", L.ptr(x), L.ptr(y), L.dt(x) if fmt is None else fmt, n, h, w, c, out_act, slope,
                       L.ptr(SignBits.last), L.stream())
            else:
                L.call("ups_bilinear2x_fwd_act", L.ptr(x), L.ptr(y), L.dt(x) if fmt is None else fmt, n, h, w, c, out_act, slope, L.stream())
            SignBits.want = False
            return y
        f8 = f8_site and fmt != L.F16
        if f8 and "fwd" not in site:
            site["fwd"] = {"slot": Fp8.slot(x.device), "born": Fp8.steps}
        f8 = f8 and Fp8.wanted(site["fwd"])
        if f8:
            so = site["fwd"]
            t8 = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if Fp8.steps > so["born"] else None
            if t8 is not None:
                so["emitted"] = so.get("emitted", 0) + 1
            L.call("ups_bilinear2x_fwd_f8", L.ptr(x), L.ptr(y), n, h, w, c, L.ptr(t8) if t8 is not None else None,
                   L.ptr(Fp8.scale[so["slot"]:]), L.ptr(Fp8.amax[so["slot"]]), act, slope, 0, L.stream())
            Fp8.last_out = {"t": t8, "act": act, "slot": so["slot"], "site": so} if t8 is not None else None
        else:
            L.call("ups_bilinear2x_fwd", L.ptr(x), L.ptr(y), L.dt(x) if fmt is None else fmt, n, h, w, c, L.stream())
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        g = g.contiguous()
        gx = torch.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        site = ctx.site
        so = None
        if site is not None and Fp8.GRAD and g.dtype == torch.bfloat16 and h % 16 == 0:
            so = site.get("bwd")
            if so is None:
                so = site["bwd"] = {"slot": Fp8.slot(g.device), "born": Fp8.steps}
                Fp8.fmax[so["slot"]] = Fp8.E5M2_MAX
            if not Fp8.wanted(so):
                so = None
        if so is not None:
            t8 = torch.empty(gx.shape, dtype=torch.uint8, device=g.device) if Fp8.steps > so["born"] else None
            if t8 is not None:
                so["emitted"] = so.get("emitted", 0) + 1
            L.call("ups_bilinear2x_bwd_f8", L.ptr(g), L.ptr(gx), n, h, w, c, L.ptr(t8) if t8 is not None else None,
                   L.ptr(Fp8.scale[so["slot"]:]), L.ptr(Fp8.amax[so["slot"]]), 1, L.stream())
            if t8 is not None:
                Fp8.register_grad_copy(gx, {"t": t8, "slot": so["slot"], "site": so})
        else:
            L.call("ups_bilinear2x_bwd", L.ptr(g), L.ptr(gx), L.dt(g), n, h, w, c, L.stream())
        return gx, None, None, None, None, None


class DepthToSpaceFn(torch.autograd.Function):
    """tf.depth_to_space(x, 2) of the "subpixel" up-sampling (nn.py:824-827): [n,h,w,ld(4C)] -> [n,2h,2w,round8(C)]."""

    @staticmethod
    def forward(ctx, x, C, fmt=None):
        x = x.contiguous()
        n, h, w, ldx = x.shape
        y = torch.empty((n, 2 * h, 2 * w, round8(C)), dtype=x.dtype, device=x.device)
        L.call("ups_depth_to_space", L.ptr(x), L.ptr(y), L.dt(x) if fmt is None else fmt, n, h, w, C, ldx, y.shape[-1], 0, L.stream())
        ctx.dims = (n, h, w, C, ldx)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, C, ldx = ctx.dims
        g = g.contiguous()
        gx = torch.empty((n, h, w, ldx), dtype=g.dtype, device=g.device)
        L.call("ups_depth_to_space", L.ptr(g), L.ptr(gx), L.dt(g), n, h, w, C, ldx, g.shape[-1], 1, L.stream())
        return gx, None, None


class Nearest2xFn(torch.autograd.Function):
    """tf.image.resize_images(NEAREST_NEIGHBOR) to twice the size (nn.py:828-833): every pixel repeated 2x2."""

    @staticmethod
    def forward(ctx, x, fmt=None):
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
        L.call("ups_nearest2x", L.ptr(x), L.ptr(y), L.dt(x) if fmt is None else fmt, n, h, w, c, 0, L.stream())
        ctx.dims = (n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.dims
        g = g.contiguous()
        gx = torch.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        L.call("ups_nearest2x", L.ptr(g), L.ptr(gx), L.dt(g), n, h, w, c, 1, L.stream())
        return gx, None


class CropFn(torch.autograd.Function):
    """The ho x wo window of an NHWC tensor at the corner held in `yx` (int32 [2] ON THE DEVICE: no host sync, valid inside a
    captured HIP graph) -- `perceptual_input: resize256_crop224` (Trainer)."""

    @staticmethod
    def forward(ctx, x, yx, ho, wo):
        x = x.contiguous()
        n, h, w, c = x.shape
        assert yx.dtype == torch.int32 and yx.numel() == 2 and yx.is_cuda
        y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
        L.call("ups_crop_fwd", L.ptr(x), L.ptr(y), L.dt(x), n, h, w, c, ho, wo, L.ptr(yx), L.stream())
        ctx.save_for_backward(yx)
        ctx.shape = (n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, g):
        (yx,) = ctx.saved_tensors
        n, h, w, c = ctx.shape
        g = g.contiguous()
        gx = torch.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        L.call("ups_crop_bwd", L.ptr(g), L.ptr(gx), L.dt(g), n, h, w, c, g.shape[1], g.shape[2], L.ptr(yx), L.stream())
        return gx, None, None, None


class ActMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope, post=False):
        """post: x already holds act(x) (post-activation storage): plain mean forward, act' from its sign backward."""
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty((n, 1, 1, c), dtype=x.dtype, device=x.device)
        L.call("ups_act_mean_fwd", L.ptr(x), L.ptr(y), L.dt(x), n, h * w, c, L.ACT_NONE if post else act, slope, L.stream())
        ctx.save_for_backward(x)
        ctx.act, ctx.slope = act, slope
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x)
        L.call("ups_act_mean_bwd", L.ptr(x), L.ptr(g), L.ptr(gx), L.dt(x), n, h * w, c, ctx.act, ctx.slope, L.stream())
        return gx, None, None, None


class EluFn(torch.autograd.Function):
    """activate(x, "elu") (nn.py:747-758): the one activation that is materialised -- the convolution kernels fuse only the
    max(x, slope * x) family into their loads; a scope with `activation: elu` (no shipped yaml) runs its convolutions on the
    activated tensor.  fmt = L.F16: fp16 bits in a bf16 container."""

    @staticmethod
    def forward(ctx, x, fmt=None):
        x = x.contiguous()
        y = torch.empty_like(x)
        ctx.dcode = L.dt(x) if fmt is None else fmt
        L.call("ups_elu_fwd", L.ptr(x), L.ptr(y), ctx.dcode, x.numel(), L.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(x)
        # (gradients of fp16 forward tensors are bf16: the derivative is taken in the gradient's type from the fp16 input)
        if ctx.dcode == L.F16:
            xb = x.view(torch.float16).to(torch.bfloat16)
            L.call("ups_elu_bwd", L.ptr(xb), L.ptr(g), L.ptr(gx), L.BF16, x.numel(), L.stream())
        else:
            L.call("ups_elu_bwd", L.ptr(x), L.ptr(g), L.ptr(gx), ctx.dcode, x.numel(), L.stream())
        return gx, None


class MaxPoolFn(torch.autograd.Function):
    """site (fp8 mode): per-call-site state when the pooled tensor feeds an fp8 convolution -- the forward then also writes the
    e4m3 copy of act(y) (ops.Fp8 hand-off, as BilinearFn); `act` = the activation-on-load of that consumer."""

    @staticmethod
    def forward(ctx, x, site=None, act=0):
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
        f8 = (site is not None and Fp8.enabled and Fp8.PRODUCER and x.dtype == torch.bfloat16 and c % 64 == 0
              and (h // 2) % 16 == 0 and (w // 2) % 16 == 0)
        if f8:
            so = site.get("fwd")
            if so is None:
                so = site["fwd"] = {"slot": Fp8.slot(x.device), "born": Fp8.steps}
            f8 = Fp8.wanted(so)
        if f8:
            t8 = torch.empty(y.shape, dtype=torch.uint8, device=x.device) if Fp8.steps > so["born"] else None
            if t8 is not None:
                so["emitted"] = so.get("emitted", 0) + 1
            L.call("ups_maxpool2_fwd_f8", L.ptr(x), L.ptr(y), n, h, w, c, L.ptr(t8) if t8 is not None else None,
                   L.ptr(Fp8.scale[so["slot"]:]), L.ptr(Fp8.amax[so["slot"]]), act, 0.2, L.stream())
            Fp8.last_out = {"t": t8, "act": act, "slot": so["slot"], "site": so} if t8 is not None else None
        else:
            L.call("ups_maxpool2_fwd", L.ptr(x), L.ptr(y), L.dt(x), n, h, w, c, L.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        n, h, w, c = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x)
        L.call("ups_maxpool2_bwd", L.ptr(x), L.ptr(g), L.ptr(gx), L.dt(x), n, h, w, c, L.stream())
        return gx, None, None


class VggPreFn(torch.autograd.Function):
    """[-1,1] RGB -> BGR*255 - ImageNet mean, 8-channel padded (edflow VGG19Features, UNVERIFIED)."""

    @staticmethod
    def forward(ctx, x, act_dtype):
        x = x.contiguous()
        pixels = x.numel() // x.shape[-1]
        y = torch.empty(x.shape[:-1] + (8,), dtype=act_dtype, device=x.device)
        L.call("ups_vgg_preprocess_fwd", L.ptr(x), int(x.dtype == torch.float32), x.shape[-1], L.ptr(y), L.dt(y),
               pixels, L.stream())
        ctx.in_shape, ctx.in_dtype = x.shape, x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        assert ctx.in_dtype == g.dtype, "gradient flows back to an activation-dtype image"
        gx = torch.empty(ctx.in_shape, dtype=g.dtype, device=g.device)
        L.call("ups_vgg_preprocess_bwd", L.ptr(g), L.ptr(gx), L.dt(g), ctx.in_shape[-1],
               g.numel() // 8, L.stream())
        return gx, None


L1_BLOCKS = 1024


class L1MeanFn(torch.autograd.Function):
    """mean |act(a) - act(b)| over the logical channels; gradient w.r.t. b only (a is the target)."""

    @staticmethod
    def forward(ctx, a, b, c_log, act):
        a, b = a.contiguous(), b.contiguous()
        rows, ld = b.numel() // b.shape[-1], b.shape[-1]
        partial = torch.empty(L1_BLOCKS, dtype=torch.float32, device=b.device)
        out = torch.empty((), dtype=torch.float32, device=b.device)
        L.call("ups_l1_fwd", L.ptr(a), L.ptr(b), L.dt(b), rows, c_log, ld, act, L.ptr(partial), L1_BLOCKS, L.stream())
        L.call("ups_sum_scale", L.ptr(partial), L1_BLOCKS, 1.0 / (rows * c_log), L.ptr(out), 0, L.stream())
        ctx.save_for_backward(a, b)
        ctx.c_log, ctx.act = c_log, act
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        rows, ld = b.numel() // b.shape[-1], b.shape[-1]
        gb = torch.empty_like(b)
        g = g.contiguous().float()
        L.call("ups_l1_bwd", L.ptr(a), L.ptr(b), L.ptr(gb), L.dt(b), rows, ctx.c_log, ld, ctx.act, L.ptr(g),
               1.0 / (rows * ctx.c_log), L.stream())
        return None, gb, None, None


class CriticHeadFn(torch.autograd.Function):
    """Head of a separable MI critic (model.py:159-173 last line, 524-536, 821-826, 855): (h_pi [2B,..,K], h_al [2B,..,K]) ->
    (loss, accuracy, mean joint logit) as three device scalars, rows [0,B) = joint pairs, [B,2B) = marginal pairs.  One launch
    forward, one backward (ups_critic_head_*); the gradient arguments are device scalars -- no host synchronisation."""

    @staticmethod
    def forward(ctx, h_pi, h_al, B, K):
        h_pi, h_al = h_pi.contiguous(), h_al.contiguous()
        ld = h_pi.shape[-1]
        assert h_pi.numel() == 2 * B * ld and h_al.shape == h_pi.shape
        logits = torch.empty(2 * B, dtype=torch.float32, device=h_pi.device)
        out = torch.empty(4, dtype=torch.float32, device=h_pi.device)
        L.call("ups_critic_head_fwd", L.ptr(h_pi), L.ptr(h_al), L.dt(h_pi), B, K, ld, L.ptr(logits), L.ptr(out), L.stream())
        ctx.save_for_backward(h_pi, h_al, logits)
        ctx.B, ctx.K = B, K
        loss, acc, mim = out[0], out[1], out[2]
        ctx.mark_non_differentiable(acc)
        return loss, acc, mim

    @staticmethod
    def backward(ctx, g_loss, g_acc, g_mim):
        h_pi, h_al, logits = ctx.saved_tensors
        ld = h_pi.shape[-1]
        gl = g_loss.contiguous().float() if g_loss is not None else None
        gm = g_mim.contiguous().float() if g_mim is not None else None
        gp = torch.empty_like(h_pi) if ctx.needs_input_grad[0] else None
        ga = torch.empty_like(h_al) if ctx.needs_input_grad[1] else None
        if gp is not None or ga is not None:
            L.call("ups_critic_head_bwd", L.ptr(h_pi), L.ptr(h_al), L.ptr(logits), L.ptr(gl), L.ptr(gm), L.dt(h_pi), ctx.B, ctx.K, ld,
                   L.ptr(gp), L.ptr(ga), L.stream())
        return gp, ga, None, None


TOWERS = SW.flag("UPS_TOWERS")      # A/B switch: the critics' towers as grouped launches (ups_towers_*)


def towers_eligible(towers, xs):
    """towers: [[ConvLayer] * L] * T (nets.Nets.critic_layers), xs: their inputs [M, 1, 1, ld].  The grouped launches take bf16 rows,
    the leaky-ReLU post-activation storage form and widths of 32 k / 128 n; anything else keeps the generic convolution path."""
    if not TOWERS or not towers or len(towers) > 8 or not (2 <= len(towers[0]) <= 6):
        return False
    Ln = len(towers[0])
    for tw, x in zip(towers, xs):
        if len(tw) != Ln or x.dtype != torch.bfloat16 or x.shape[1:3] != (1, 1):
            return False
        for l, lay in enumerate(tw):
            if lay.k != 1 or lay.stride != 1 or lay.coords or lay.f16 or lay.ci_log % 32 or lay.co % 128:
                return False
            want_in = (L.ACT_NONE, False) if l == 0 else (L.ACT_LRELU, True)
            if (lay.act_in, lay.in_post) != want_in or lay.out_act != (L.ACT_LRELU if l < Ln - 1 else L.ACT_NONE):
                return False
            if l > 0 and (lay.ci_log != tw[l - 1].co or (l < Ln - 1 and lay.ci_log != lay.co)):
                return False
        # the checks ups_towers_fwd / _bwd make on the first layer's input (csrc/critic.hip: ld0 >= k, ld0 % 8 == 0, 16-byte aligned
        # rows): a view that fails them keeps the generic path instead of raising UpsError in the middle of a step
        if x.shape[-1] < tw[0].ci_log or x.shape[-1] % 8 or (x.is_contiguous() and x.data_ptr() % 16):
            return False
    for l in range(Ln):                     # one launch per layer index: every tower must have the same width there
        if len(set(tw[l].co for tw in towers)) != 1:
            return False
    return True


class TowersFn(torch.autograd.Function):
    """T towers of nin -> residual_block(k = 1) x (L - 2) -> nin (discriminator_model, model.py:159-173) as grouped launches: the same
    layer of every tower in one launch, every weight / bias gradient in one (ups_towers_fwd / _bwd).  apply(towers, x_0 .. x_{T-1},
    V, b of every layer tower-major) -> the T embeddings [M, 1, 1, n].  A backward call takes the towers whose output gradient it is
    given (the adversarial term differentiates critic 0's pi tower alone, under skip_wgrad)."""

    @staticmethod
    def forward(ctx, towers, *tensors):
        T, Ln = len(towers), len(towers[0])
        xs = [t.contiguous() for t in tensors[:T]]
        M, dev = xs[0].shape[0], xs[0].device
        lay_arr = (L.TowerLayer * (T * Ln))()
        for t, tw in enumerate(towers):
            for l, lay in enumerate(tw):
                ent = lay.prepared(L.BF16, 1, 1, need_dgrad=True)
                e = lay_arr[t * Ln + l]
                e.w_fwd, e.w_dgrad, e.bias = ent["w_fwd"].data_ptr(), ent["w_dgrad"].data_ptr(), lay.b.data_ptr()
                e.grad_w = lay.grad_V.data_ptr() if lay.grad_V is not None else None
                e.grad_b = lay.grad_b.data_ptr() if lay.grad_b is not None else None
                e.k, e.n = lay.ci_log, lay.co
        acts = [[torch.empty((M, 1, 1, lay.co), dtype=torch.bfloat16, device=dev) for lay in tw] for tw in towers]
        x0 = (C.c_void_p * T)(*[x.data_ptr() for x in xs])
        ld0 = (C.c_int32 * T)(*[x.shape[-1] for x in xs])
        ap = (C.c_void_p * (T * Ln))(*[a.data_ptr() for tw in acts for a in tw])
        slope = towers[0][0].slope
        L.call("ups_towers_fwd", lay_arr, T, Ln, x0, ld0, ap, M, slope, L.stream())
        outs = tuple(tw[-1] for tw in acts)
        ctx.save_for_backward(*(xs + list(outs)))            # (the pointer arrays below stay valid while these are alive)
        ctx.inner = [tw[:-1] for tw in acts]
        ctx.towers, ctx.arrays, ctx.M, ctx.slope = towers, (lay_arr, x0, ld0, ap), M, slope
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gs):
        towers, M = ctx.towers, ctx.M
        T, Ln = len(towers), len(towers[0])
        xs = list(ctx.saved_tensors[:T])
        lay_arr, x0, ld0, ap = ctx.arrays
        dev = xs[0].device
        gout = [None if g is None else to_act_dtype(g, torch.bfloat16, towers[t][-1].co) for t, g in enumerate(gs)]
        want_w = not GradMode.skip_wgrad
        gx = [torch.empty_like(xs[t]) if (gout[t] is not None and ctx.needs_input_grad[1 + t] and towers[t][0].ci_log % 128 == 0
                                          and xs[t].shape[-1] == towers[t][0].ci_log) else None for t in range(T)]
        if any(gout[t] is not None and ctx.needs_input_grad[1 + t] and gx[t] is None for t in range(T)):
            raise L.UpsError("TowersFn: the input gradient of a tower whose first layer is not 128 k wide")
        ws = [[torch.empty((M, lay.co), dtype=torch.bfloat16, device=dev) if (gout[t] is not None and l < Ln - 1) else None
               for l, lay in enumerate(tw)] for t, tw in enumerate(towers)]
        gp = (C.c_void_p * T)(*[None if g is None else g.data_ptr() for g in gout])
        wp = (C.c_void_p * (T * Ln))(*[None if w is None else w.data_ptr() for tw in ws for w in tw])
        gxp = (C.c_void_p * T)(*[None if g is None else g.data_ptr() for g in gx])
        ldg = (C.c_int32 * T)(*[x.shape[-1] for x in xs])
        L.call("ups_towers_bwd", lay_arr, T, Ln, x0, ld0, ap, gp, wp, gxp, ldg, int(want_w), M, ctx.slope, L.stream())
        grads = [None] + gx
        for t, tw in enumerate(towers):
            for lay in tw:
                if want_w and gout[t] is not None and lay.grad_V is not None:
                    grads += [lay.grad_V, lay.grad_b]
                    if lay.after_wgrad is not None:
                        lay.after_wgrad()
                else:
                    grads += [None, None]
        return tuple(grads)


class MaskPartsFn(torch.autograd.Function):
    """mask_parts + part-major transpose (model.py:176-187, nn.py:97-103): -> [P*B,H,W,8]."""

    @staticmethod
    def forward(ctx, view, hard, act_dtype):
        B, H, W, P = hard.shape
        out = torch.empty((P * B, H, W, 8), dtype=act_dtype, device=hard.device)
        L.call("ups_mask_parts_fwd", L.ptr(view), L.ptr(hard), L.ptr(out), L.dt(out), B, H * W, P, L.stream())
        ctx.save_for_backward(view)
        ctx.shape = (B, H, W, P)
        return out

    @staticmethod
    def backward(ctx, g):
        (view,) = ctx.saved_tensors
        B, H, W, P = ctx.shape
        g = g.contiguous()
        gh = torch.empty((B, H, W, P), dtype=torch.float32, device=g.device)
        L.call("ups_mask_parts_bwd", L.ptr(view), L.ptr(g), L.ptr(gh), L.dt(g), B, H * W, P, L.stream())
        return None, gh, None


class UnpoolFn(torch.autograd.Function):
    """unpool_features + concat with the hard mask (model.py:225-249, 482-484): -> [B,H,W,round8(F+P)]."""

    @staticmethod
    def forward(ctx, hard, feat, act_dtype):
        B, H, W, P = hard.shape
        F = feat.shape[-1]
        ldo = round8(F + P)
        feat = feat.contiguous()
        out = torch.empty((B, H, W, ldo), dtype=act_dtype, device=hard.device)
        L.call("ups_unpool_fwd", L.ptr(hard), L.ptr(feat), L.ptr(out), L.dt(out), B, H * W, P, F, ldo, L.stream())
        ctx.save_for_backward(hard, feat)
        return out

    @staticmethod
    def backward(ctx, g):
        hard, feat = ctx.saved_tensors
        B, H, W, P = hard.shape
        F = feat.shape[-1]
        g = g.contiguous()
        gh = torch.empty_like(hard)
        nfl = L.load().ups_unpool_bwd_floats(B, P, F)
        gf = torch.empty(nfl, dtype=torch.float32, device=g.device)
        L.call("ups_unpool_bwd", L.ptr(hard), L.ptr(feat), L.ptr(g), L.ptr(gh), L.ptr(gf), L.dt(g), B, H * W, P, F,
               g.shape[-1], L.stream())
        return gh, gf[:B * P * F].view(B, P, F), None


# --------------------------------------------------------------------------- raw (non-autograd) part-path / latent calls
def part_softmax(mean, eps=None, want_hard=True, want_argmax=False, want_bits=None, moments_gamma=None):
    """-> (l, m, hard, argmax), or (l, m, hard, argmax, hard_bits) when `want_bits` is given (True / False):
    hard_bits [..] int32 = the hard mask as a bit set per pixel (P <= 32; None when not wanted).
    moments_gamma (mean [n,h,w,P]): additionally returns, as the last element, the spatial soft-max moments of
    gamma * hard -- what spatial_moments(hard, gamma) computes -- from the same pass (None when the shape does not allow it)."""
    mean = mean.contiguous()
    pixels, P = mean.numel() // mean.shape[-1], mean.shape[-1]
    l = torch.empty_like(mean) if eps is not None else mean
    m = torch.empty_like(mean)
    hard = torch.empty_like(mean) if want_hard else None
    am = torch.empty(mean.shape[:-1], dtype=torch.int64, device=mean.device) if want_argmax else None
    bits = torch.empty(mean.shape[:-1], dtype=torch.int32, device=mean.device) if want_bits else None
    stats = None
    if moments_gamma is not None and mean.dim() == 4 and P <= 32 and want_hard:
        n, h, w, _ = mean.shape
        # the fused form needs whole pixel tiles that do not straddle two images: the library says what its tile is
        if (h * w) % L.load().ups_part_softmax_moments_tile(P) == 0:
            nint = L.load().ups_part_softmax_moments_ints(pixels, P)
            stats = torch.empty((n, P, 8), dtype=torch.float32, device=mean.device)
            scratch = torch.empty(nint, dtype=torch.int32, device=mean.device)
            L.call("ups_part_softmax_moments_fwd", L.ptr(mean), L.ptr(eps.contiguous()) if eps is not None else None,
                   L.ptr(l) if eps is not None else None, L.ptr(m), L.ptr(hard), L.ptr(am), L.ptr(bits), n, h, w, P,
                   float(moments_gamma), L.ptr(stats), L.ptr(scratch), L.stream())
    if stats is None:
        L.call("ups_part_softmax_fwd", L.ptr(mean), L.ptr(eps.contiguous()) if eps is not None else None,
               L.ptr(l) if eps is not None else None, L.ptr(m), L.ptr(hard), L.ptr(am), L.ptr(bits), pixels, P, L.stream())
    out = (l, m, hard, am) if want_bits is None else (l, m, hard, am, bits)
    return out + (stats,) if moments_gamma is not None else out


def spatial_moments(x, gamma, rect_px=None, half=0, kl_sums=None):
    """kl_sums (fp32 [>= 16] device buffer): the same pass also writes sum x * log(P x + 1e-20) -- the categorical KL of the map,
    view 1's other prior term -- to kl_sums[0] (ups_spatial_moments_kl)."""
    n, h, w, P = x.shape
    nfl = L.load().ups_spatial_moments_floats(n, P)
    buf = torch.empty(nfl, dtype=torch.float32, device=x.device)
    if kl_sums is not None:
        L.call("ups_spatial_moments_kl", L.ptr(x), n, h, w, P, float(gamma), L.ptr(rect_px), half, half, L.ptr(buf), L.ptr(kl_sums),
               L.stream())
    else:
        L.call("ups_spatial_moments", L.ptr(x), n, h, w, P, float(gamma), L.ptr(rect_px), half, half, L.ptr(buf), L.stream())
    return buf[:n * P * 8].view(n, P, 8)


def moments_to_px(stats, h, order="xy"):
    """(row, column) centres of the rectangles; order "xy": tfutils.draw_rect reads the (y, x) pair as (x, y)."""
    n, P, _ = stats.shape
    px = torch.empty((n, P, 2), dtype=torch.int32, device=stats.device)
    L.call("ups_moments_to_px", L.ptr(stats), n * P, h, int(order == "xy"), L.ptr(px), L.stream())
    return px


def draw_rect(px, h, w, half):
    n, P, _ = px.shape
    out = torch.empty((n, h, w, P), dtype=torch.float32, device=px.device)
    L.call("ups_draw_rect", L.ptr(px), n, h, w, P, half, half, L.ptr(out), L.stream())
    return out


def latent_fwd(params, eps, levels, want_kl):
    """params [B,NP] fp32, eps [S,B,Z] -> samples [S,B,Z], kl_rows [B,Z] or None."""
    S, B, Z = eps.shape
    samples = torch.empty((S, B, Z), dtype=torch.float32, device=params.device)
    kl = torch.empty((B, Z), dtype=torch.float32, device=params.device) if want_kl else None
    L.call("ups_latent_fwd", L.ptr(params), L.ptr(eps.contiguous()), (C.c_float * S)(*levels), S, B, Z,
           L.ptr(samples), L.ptr(kl), L.stream())
    return samples, kl


def latent_bwd(params, eps, levels, g_samples, g_kl_dev, g_kl_scale):
    S, B, Z = eps.shape
    gp = torch.empty_like(params)
    L.call("ups_latent_bwd", L.ptr(params), L.ptr(eps.contiguous()), (C.c_float * S)(*levels),
           L.ptr(g_samples.contiguous()), L.ptr(g_kl_dev), float(g_kl_scale), S, B, Z, L.ptr(gp), L.stream())
    return gp


class NoiseStream(object):
    """Standard-normal noise from the library's own Philox4x32-10 kernel (ups_randn): a (seed, offset) counter stream -- the
    values are a pure function of the seed and of how many values were drawn before, on any launch geometry."""

    def __init__(self, seed):
        self.seed, self.offset = int(seed) & ((1 << 64) - 1), 0

    def fill(self, out):
        assert out.dtype == torch.float32 and out.is_contiguous()
        n = out.numel()
        L.call("ups_randn", L.ptr(out), n, self.seed, self.offset, L.stream())
        self.offset += (n + 3) // 4
        return out

    def randn(self, *shape, device=None):
        return self.fill(torch.empty(*shape, dtype=torch.float32, device=device))


def adam_step(p, g, m, v, lr_t, beta1, beta2, eps, grad_scale=1.0):
    """lr_t: python float, or a 1-element fp32 device tensor (HIP-graph mode: the value is read on the device)."""
    if torch.is_tensor(lr_t):
        L.call("ups_adam_dev", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(lr_t), float(beta1), float(beta2),
               float(eps), float(grad_scale), L.stream())
    else:
        L.call("ups_adam", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), float(lr_t), float(beta1), float(beta2),
               float(eps), float(grad_scale), L.stream())


def gauss_hm(pts, stddev, h, w):
    B, K, _ = pts.shape
    out = torch.empty((B, h, w, K), dtype=torch.float32, device=pts.device)
    L.call("ups_gauss_hm", L.ptr(pts.contiguous()), L.ptr(stddev.contiguous()), L.ptr(out), B, h, w, K, L.stream())
    return out


def gauss_hm3(mu, Lt, h, w):
    B, K, _ = mu.shape
    out = torch.empty((B, h, w, K), dtype=torch.float32, device=mu.device)
    L.call("ups_gauss_hm3", L.ptr(mu.contiguous()), L.ptr(Lt.contiguous()), L.ptr(out), B, h, w, K, L.stream())
    return out
