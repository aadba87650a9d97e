"""Sustained-load check of the dominant conv (dv res-block at 128x128): burst vs 2 s of back-to-back launches,
with and without the residual epilogue."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops
dev = torch.device("cuda:0")
n, h, c = 128, 128, 256
g = torch.Generator().manual_seed(1)
V = (torch.randn(3, 3, c + 2, c, generator=g) / math.sqrt(9 * c)).to(dev)
b = torch.randn(c, generator=g).to(dev)
lay = ops.ConvLayer("x/conv2d_0", V, b, 3, 1, True, "leaky_relu")
x = (torch.randn(n, h, h, c, device=dev) * float(sys.argv[1]) if len(sys.argv) > 1 else torch.randn(n, h, h, c, device=dev)).to(torch.bfloat16)
def run(res, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv_forward(x, lay, res=x if res else None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for res in (False, True):
    run(res, 2)
    print("res", res, "burst(5) ms", round(run(res, 5), 3), "sustained(400) ms", round(run(res, 400), 3), "after ms", round(run(res, 5), 3))
