"""Runs one convolution a few times: target for rocprofv3 --pmc (tools/pmc_conv.sh).
Usage: python tools/one_conv.py [fwd|dgrad|wgrad] [case of tools/bench_conv.py, default dv_rb128] [plain]
The launch is issued the way the model issues it: residual block (res = in), for a 3x3 layer with an activation the input in
post-activation storage and -- the mask decoder's layers (dv_*) -- fp16 forward tensors; `plain`: bf16, activation-on-load;
`bits` (dgrad): act' from the producer's sign bytes (ups_conv_desc.dact_bits) instead of the forward input;
`f8`: the fp8 mode's launch -- the operand arrives as a producer's e4m3 (forward) / e5m2 (input gradient) copy; `wgrad ... f8`: the
fp8 weight gradient (conv_wgrad3x3_f8.hip): the gradient's e5m2 copy + the forward tensor quantised while it is staged."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops
from bench_conv import CASES
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
case = sys.argv[2] if len(sys.argv) > 2 else "dv_rb128"
name, n, h, cin, cout, k, stride, coords, act = [c for c in CASES if c[0] == case][0]
g = torch.Generator().manual_seed(1)
cin_v = cin + (2 if coords else 0)
V = (torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(k * k * cin_v)).to(dev)
b = torch.randn(cout, generator=g).to(dev)
from upsparts_amd import lib
lay = ops.ConvLayer("x/conv2d_0", V, b, k, stride, coords, act)
plain = "plain" in sys.argv[3:]
bits = "bits" in sys.argv[3:]         # forward: the launch also writes the sign bytes; input gradient: it reads them instead of x
f8 = "f8" in sys.argv[3:]
# (the fp8 WEIGHT gradient of a mask-decoder layer reads the fp16 forward tensor as the model hands it over)
fmt = lib.F16 if (not plain and (not f8 or mode == "wgrad") and case.startswith("dv_") and k == 3) else None
lay.f16 = fmt == lib.F16
if not plain and act == "leaky_relu":
    lay.in_post, lay.out_act = True, lib.ACT_LRELU
x = torch.randn(n, h, h, ops.round8(cin), device=dev).to(torch.bfloat16)
if fmt == lib.F16:
    x = x.to(torch.float16).view(torch.bfloat16)
res = x if (cin == cout and stride == 1 and act is not None) else None
F = ops.Fp8
if f8:
    F = ops.Fp8.activate(ops.Fp8State(True))
    sx, sg = F.slot(dev), F.slot(dev)
    F.scale[sx] = 448.0 * F.MARGIN / float(x.float().abs().max()); F.scale[sg] = 57344.0 * F.MARGIN / 6.0
    xf = x.view(torch.float16).float() if fmt == lib.F16 else x.float()
    x8 = (xf * F.scale[sx]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    if fmt != lib.F16:
        F.next_in = {"t": x8, "slot": sx, "act": lay.act_in, "site": None}
y = ops.conv_forward(x, lay, res=res, fmt=fmt, res_post=lay.in_post)
gy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
if f8:
    g8 = (gy.float() * F.scale[sg]).clamp(-57344, 57344).to(torch.float8_e5m2).view(torch.uint8)
xbits = None
if bits:
    pos = (x.view(torch.int16) > 0).view(n, h, h, -1, 8).to(torch.uint8)
    xbits = (pos * (2 ** torch.arange(8, device=dev, dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()
for _ in range(3):
    if mode == "fwd":
        if f8:
            F.next_in = {"t": x8, "slot": sx, "act": lay.act_in, "site": None}
        ops.SignBits.want, ops.SignBits.last = bool(bits), None
        ops.conv_forward(x, lay, res=res, fmt=fmt, res_post=lay.in_post)
        if bits:
            assert ops.SignBits.take() is not None, "the launch did not write the sign bytes"
    elif mode == "dgrad":
        if f8:
            F.register_grad_copy(gy, {"t": g8, "slot": sg, "site": None})
        ops.conv_dgrad(gy, x, lay, res=gy if res is not None else None, x_bits=xbits)
    elif mode == "wgrad":
        ops.conv_wgrad(gy, x, lay, fmt=fmt, f8_src={"t": g8, "slot": sg, "site": None} if f8 else None)
torch.cuda.synchronize()
if f8:
    print(F.stats)
