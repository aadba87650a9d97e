"""encoder_0's first `downsample` as the headline runs it (32 -> 64 channels, stride 2, CoordConv, 128 images of 128 x 128): forward time of
conv3x3_s2_kernel.   Usage (GPU box): python tools/probes/time_s2.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops, lib
from bench_conv import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for n, h, ci, co in ((128, 128, 32, 64), (128, 64, 64, 128)):
    V = (torch.randn(3, 3, ci + 2, co, generator=g) / math.sqrt(9 * (ci + 2))).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    lay = ops.ConvLayer("x%d/conv2d_0" % h, V, b, 3, 2, True, None)
    lay.out_act = lib.ACT_LRELU
    x = torch.randn(n, h, h, ci, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.conv_forward(x, lay), 20)
    print("downsample %dx%dx%dx%d -> %d (CoordConv): %.3f ms" % (n, h, h, ci, co, t))
