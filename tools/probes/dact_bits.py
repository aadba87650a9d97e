"""What bit-packed activation signs buy the patch kernel's input gradient (round 5, ups_conv_desc.dact_bits): the sign bytes of act'
from a pre-packed [n,h,w,c/8] byte tensor instead of re-reading the forward input (2 B per element for one bit).  The first
version of this probe, against a debug hook, is what motivated the feature (2.23 -> 2.04 ms on dv_rb128, 0.68 -> 0.56 on dv_rb64).
Usage: python tools/probes/dact_bits.py [case of tools/bench_conv.py]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import upsparts_amd  # noqa
from upsparts_amd import ops, lib
from bench_conv import CASES
dev = torch.device("cuda:0")
case = sys.argv[1] if len(sys.argv) > 1 else "dv_rb128"
name, n, h, cin, cout, k, stride, coords, act = [c for c in CASES if c[0] == case][0]
g = torch.Generator().manual_seed(1)
cin_v = cin + (2 if coords else 0)
V = (torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(k * k * cin_v)).to(dev)
b = torch.randn(cout, generator=g).to(dev)
lay = ops.ConvLayer("x/conv2d_0", V, b, k, stride, coords, act)
lay.f16 = True
lay.in_post, lay.out_act = True, lib.ACT_LRELU
x = torch.randn(n, h, h, ops.round8(cin), device=dev).to(torch.float16)
xb = x.view(torch.bfloat16)
y = ops.conv_forward(xb, lay, res=xb, fmt=lib.F16, res_post=True)
gy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
# bit e of byte c / 8 = (x[8 (c / 8) + e] > 0)
pos = (x > 0).view(n, h, h, -1, 8).to(torch.uint8)
bits = (pos * (2 ** torch.arange(8, device=dev, dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()
def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it


fn = lambda: ops.conv_dgrad(gy, xb, lay, res=gy)
fb = lambda: ops.conv_dgrad(gy, xb, lay, res=gy, x_bits=bits)
ref, out = fn().clone(), fb().clone()
print("{} input gradient, act' from the forward input: {:.3f} ms".format(case, timeit(fn)))
print("{} input gradient, act' from packed sign bits:  {:.3f} ms   identical: {}".format(case, timeit(fb), torch.equal(out, ref)))
