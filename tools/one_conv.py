"""Runs one convolution a few times: target for rocprofv3 --pmc (tools/pmc_conv.sh).
Usage: python tools/one_conv.py [fwd|dgrad|wgrad] [case of tools/bench_conv.py, default dv_rb128 with the residual add]"""
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
lay = ops.ConvLayer("x/conv2d_0", V, b, k, stride, coords, act)
x = torch.randn(n, h, h, ops.round8(cin), device=dev).to(torch.bfloat16)
res = x if (cin == cout and stride == 1) else None
y = ops.conv_forward(x, lay, res=res)
gy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
for _ in range(3):
    if mode == "fwd":
        ops.conv_forward(x, lay, res=res)
    elif mode == "dgrad":
        ops.conv_dgrad(gy, x, lay)
    elif mode == "wgrad":
        ops.conv_wgrad(gy, x, lay)
torch.cuda.synchronize()
