"""Is the dominant convolution clock- (power-) bound?  The same launch on random operands, on an all-zero activation tensor and on
all-zero activations AND weights: the instruction stream is identical, only the switching activity of the datapaths differs.
Usage (GPU box): python tools/probes/zero_data.py [case]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops, lib
from bench_conv import CASES, timeit
dev = torch.device("cuda:0")
case = sys.argv[1] if len(sys.argv) > 1 else "dv_rb128"
name, n, h, cin, cout, k, stride, coords, act = [c for c in CASES if c[0] == case][0]
g = torch.Generator().manual_seed(1)
cin_v = cin + (2 if coords else 0)
for wz in (False, True):
    V = (torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(k * k * cin_v)).to(dev) * (0.0 if wz else 1.0)
    b = torch.randn(cout, generator=g).to(dev)
    lay = ops.ConvLayer("x%d/conv2d_0" % wz, V, b, k, stride, coords, act)
    lay.f16 = True
    lay.in_post, lay.out_act = True, lib.ACT_LRELU
    for xz in (False, True):
        x = torch.randn(n, h, h, cin, device=dev) * (0.0 if xz else 1.0)
        x = x.to(torch.float16).view(torch.bfloat16)
        gy = (torch.randn(n, h, h, cout, device=dev) * (0.0 if xz else 1.0)).to(torch.bfloat16)
        tf = timeit(lambda: ops.conv_forward(x, lay, res=x, fmt=lib.F16, res_post=True), 10)
        td = timeit(lambda: ops.conv_dgrad(gy, x, lay, res=gy), 10)
        print("%s  weights %-6s activations %-6s  fwd %.3f ms  dgrad %.3f ms" % (case, "zero" if wz else "random", "zero" if xz else "random", tf, td))

# fp8 mode's input gradient (block-scaled K = 128 MFMA on an e5m2 copy of the gradient): the same question
F = ops.Fp8.activate(ops.Fp8State(True))
V = (torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(k * k * cin_v)).to(dev)
b = torch.randn(cout, generator=g).to(dev)
for wz in (False, True):
    lay = ops.ConvLayer("y%d/conv2d_0" % wz, V * (0.0 if wz else 1.0), b, k, stride, coords, act)
    lay.in_post, lay.out_act = True, lib.ACT_LRELU
    x = torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16)
    for xz in (False, True):
        gy = (torch.randn(n, h, h, cout, device=dev) * (0.0 if xz else 1.0)).to(torch.bfloat16)
        sg = F.slot(dev)
        F.scale[sg] = 57344.0 * F.MARGIN / 6.0
        g8 = (gy.float() * F.scale[sg]).clamp(-57344, 57344).to(torch.float8_e5m2).view(torch.uint8)

        def dgrad():
            F.register_grad_copy(gy, {"t": g8, "slot": sg, "site": None})
            return ops.conv_dgrad(gy, x, lay, res=gy)
        td = timeit(dgrad, 10)
        print("%s fp8  weights %-6s gradient %-6s  dgrad %.3f ms   %s" % (case, "zero" if wz else "random", "zero" if xz else "random", td, dict(F.stats)))
