"""Runs the dominant conv (dv res-block 128x128, forward with residual) a few times: target for rocprofv3 --pmc."""
import math, os, sys
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
x = torch.randn(n, h, h, c, device=dev).to(torch.bfloat16)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for _ in range(3):
    if mode == "fwd":
        ops.conv_forward(x, lay, res=x)
    elif mode == "wgrad":
        ops.conv_wgrad(x, x, lay)
torch.cuda.synchronize()
