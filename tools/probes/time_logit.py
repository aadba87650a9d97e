"""The mask decoder's logit convolution as the headline runs it (fp16 input, fp32 logits, CoordConv): forward time.
Usage (GPU box): [UPS_ROWS_KERNEL=0] python tools/probes/time_logit.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops, lib
from bench_conv import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for n, h, P in ((128, 128, 10), (32, 256, 20)):
    P = min(P, 16) if os.environ.get("CLAMP16") else P
    V = (torch.randn(3, 3, 258, P, generator=g) / math.sqrt(9 * 258)).to(dev)
    b = torch.randn(P, generator=g).to(dev)
    lay = ops.ConvLayer("x%d/conv2d_0" % h, V, b, 3, 1, True, None)
    lay.f16 = True
    x = torch.randn(n, h, h, 256, device=dev).to(torch.float16).view(torch.bfloat16)
    t = timeit(lambda: ops.conv_forward(x, lay, out_f32=True, fmt=lib.F16), 20)
    print("logit conv %dx%dx%dx256 -> %d: %.3f ms  (%.2f TB/s of input)" % (n, h, h, P, t, n * h * h * 512 / t / 1e9))
