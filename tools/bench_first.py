"""First-layer launches as the model issues them: encoder_1's part-masked 3 -> 32 convolution on P x B part images (one view tensor +
hard-mask bit words in, P*B images out) and the plain 3 -> 32 / 3 -> 64 first layers.  Usage: python tools/bench_first.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops, lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def run(name, n, h, cout, coords, parts, post):
    cv = 3 + (2 if coords else 0)
    V = (torch.randn(3, 3, cv, cout, generator=g) / math.sqrt(9 * cv)).to(dev)
    lay = ops.ConvLayer("x/conv2d_0", V, torch.randn(cout, generator=g).to(dev), 3, 1, coords, None)
    if post:
        lay.out_act = lib.ACT_LRELU
    x = torch.randn(n, h, h, 8, generator=g).to(dev).to(torch.bfloat16)
    x[..., 3:] = 0
    mask = None
    if parts:
        bits = (1 << torch.randint(0, parts, (n, h, h), generator=g)).to(torch.int32).to(dev)
        mask = (bits, parts)
    f = lambda: ops.conv_forward(x, lay, mask=mask)
    y = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out_b = y.numel() * 2
    print("%-44s %7.3f ms   %6.0f GB/s written" % (name, ms, out_b / ms / 1e6))


run("ea_in masked: 64 views x 10 parts, 3 -> 32", 64, 128, 32, True, 10, True)
run("ea_in plain: 640 images, 3 -> 32", 640, 128, 32, True, 0, True)
run("e_pi: 128 images, 3 -> 32", 128, 128, 32, True, 0, True)
run("vgg block1_conv1: 128 images, 3 -> 64", 128, 128, 64, False, 0, True)
