"""Micro-benchmark of the convolution engine on representative layers of the CUB 128x128 / B=64 step.
Usage (GPU box): python tools/bench_conv.py [--check] [--only name,...]"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa: E402,F401
from upsparts_amd import ops, lib  # noqa: E402

CASES = [
    # name, n, h, cin, cout, k, stride, coords, act
    ("dv_rb128", 128, 128, 256, 256, 3, 1, True, "leaky_relu"),
    ("dv_rb128_noact", 128, 128, 256, 256, 3, 1, True, None),      # the same layer without activation-on-load (cost of the fused lrelu)
    ("dv_rb64", 128, 64, 256, 256, 3, 1, True, "leaky_relu"),
    ("dv_rb32", 128, 32, 256, 256, 3, 1, True, "leaky_relu"),
    ("dv_rb16", 128, 16, 256, 256, 3, 1, True, "leaky_relu"),
    ("dv_out", 128, 128, 256, 10, 3, 1, True, None),
    ("ea_in", 640, 128, 3, 32, 3, 1, False, None),
    ("ea_rb0", 640, 128, 32, 32, 3, 1, False, "leaky_relu"),
    ("ea_down0", 640, 128, 32, 64, 3, 2, False, None),
    ("ea_rb1", 640, 64, 64, 64, 3, 1, False, "leaky_relu"),
    ("ea_down1", 640, 64, 64, 128, 3, 2, False, None),
    ("ea_rb2", 640, 32, 128, 128, 3, 1, False, "leaky_relu"),
    ("ea_rb4", 640, 8, 256, 256, 3, 1, False, "leaky_relu"),
    ("vgg1_2", 64, 128, 64, 64, 3, 1, False, "relu"),
    ("vgg3_2", 64, 32, 256, 256, 3, 1, False, "relu"),
    ("vgg4_2", 64, 16, 512, 512, 3, 1, False, "relu"),
    ("crit_nin", 128, 1, 512, 512, 1, 1, False, "leaky_relu"),
    ("dv_rb4", 128, 4, 256, 256, 3, 1, True, "leaky_relu"),
    ("dv_rb8", 128, 8, 256, 256, 3, 1, True, "leaky_relu"),
    ("ea_rb4x4", 768, 4, 256, 256, 3, 1, False, "leaky_relu"),
    ("ea_rb8x8", 768, 8, 256, 256, 3, 1, False, "leaky_relu"),
    ("vgg5_1", 128, 8, 512, 512, 3, 1, False, "relu"),
    ("ea_down0", 640, 128, 32, 64, 3, 2, False, None),
    ("ea_down1", 640, 64, 64, 128, 3, 2, False, None),
]


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--f16", action="store_true", help="fp16 forward tensors (the mask decoder's format): forward in fp16, gradients bf16")
    ap.add_argument("--post", action="store_true", help="post-activation storage: the input holds act(x), the output is stored as act(y) "
                    "(layers with an activation only)")
    ap.add_argument("--bits", action="store_true", help="input gradient with act' from the producer's sign bytes, as the step runs it "
                    "(res_patch == 2 on the wide residual blocks)")
    ap.add_argument("--fp8", action="store_true", help="forward / input gradient of the eligible layers on the fp8 path (ops.Fp8)")
    ap.add_argument("--fp8-copy", action="store_true", help="with --fp8: the forward input arrives as the fp8 copy a producing layer "
                    "would have written (no conversion in the kernel, two blocks per CU)")
    ap.add_argument("--fp8-wgrad", action="store_true", help="with --fp8: the weight gradient of the eligible layers on the fp8 kernel "
                    "(the gradient arrives as its producer's e5m2 copy)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.Fp8.enabled = args.fp8
    ops.Fp8.COPY_ONLY = False        # isolated launches: convert in the kernel unless --fp8-copy supplies the quantised operand
    T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    only = set(args.only.split(",")) if args.only else None
    print("{:10s} {:>9s} {:>8s} {:>9s} {:>8s} {:>9s} {:>8s}".format("layer", "fwd ms", "TF/s", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
    tot = [0.0, 0.0, 0.0]
    for name, n, h, cin, cout, k, stride, coords, act in CASES:
        if only and name not in only:
            continue
        g = torch.Generator().manual_seed(1)
        cin_v = cin + (2 if coords else 0)
        V = (torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(cin_v * k * k)).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        lay = ops.ConvLayer(name + "/conv2d_0", V, b, k, stride, coords, act)
        fmt = lib.F16 if (args.f16 and k == 3) else None
        lay.f16 = fmt == lib.F16
        if args.post and act == "leaky_relu":
            lay.in_post, lay.out_act = True, lib.ACT_LRELU
        x = torch.randn(n, h, h, ops.round8(cin), device=dev).to(T)
        if ops.round8(cin) > cin:
            x[..., cin:] = 0
        if fmt == lib.F16:
            x = x.to(torch.float16).view(torch.bfloat16)
        res_self = act is not None and stride == 1 and cin == cout      # the residual blocks of the model
        _fwd = lambda: ops.conv_forward(x, lay, res=x if res_self else None, fmt=fmt, res_post=lay.in_post)
        y = _fwd()
        gy = torch.randn(y.shape, device=dev).to(T)
        ho = y.shape[1]
        flops = 2.0 * n * ho * ho * k * k * cin_v * cout
        if args.fp8 and args.fp8_copy and ops.Fp8.eligible(lay, x) and cout > 32:
            slot = ops.Fp8.slot(dev)
            xa = x.float()
            xa = torch.maximum(xa, 0.2 * xa) if act == "leaky_relu" else (torch.relu(xa) if act == "relu" else xa)
            ops.Fp8.scale[slot] = 224.0 / xa.abs().max()
            copy = {"t": (xa * ops.Fp8.scale[slot]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8), "act": lay.act_in, "slot": slot}

            def fwd():
                ops.Fp8.next_in = copy
                return ops.conv_forward(x, lay)
            tf = timeit(fwd, args.iters)
        else:
            tf = timeit(_fwd, args.iters)
        if fmt == lib.F16:
            gy = torch.randn(y.shape, device=dev).to(T)
        xb = None
        if args.bits and lay.in_post and x.shape[-1] % 8 == 0:
            pos = (x.view(torch.int16) > 0).view(*x.shape[:-1], -1, 8).to(torch.uint8)
            xb = (pos * (2 ** torch.arange(8, device=dev, dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()
            del pos
        td = timeit(lambda: ops.conv_dgrad(gy, x, lay, res=gy if res_self else None, x_bits=xb), args.iters)
        wsrc = None
        if args.fp8 and args.fp8_wgrad and ops.Fp8.eligible_wgrad(lay, gy, x, None):
            gslot = ops.Fp8.slot(dev)
            ops.Fp8.scale[gslot] = 57344.0 * 0.5 / gy.float().abs().max()
            wsrc = {"t": (gy.float() * ops.Fp8.scale[gslot]).clamp(-57344, 57344).to(torch.float8_e5m2).view(torch.uint8), "slot": gslot}
        tw = timeit(lambda: ops.conv_wgrad(gy, x, lay, fmt=fmt, f8_src=wsrc), args.iters)
        tot[0] += tf; tot[1] += td; tot[2] += tw
        print("{:10s} {:9.3f} {:8.1f} {:9.3f} {:8.1f} {:9.3f} {:8.1f}".format(
            name, tf, flops / tf / 1e9, td, flops / td / 1e9, tw, flops / tw / 1e9))
        if args.check:
            xa = x.float()[..., :cin]
            if act == "leaky_relu":
                xa = torch.nn.functional.leaky_relu(xa, 0.2)
            elif act == "relu":
                xa = torch.relu(xa)
            Vq = V.clone()
            Vq[:, :, :cin] = V[:, :, :cin].to(torch.float16 if fmt == lib.F16 else T).float()
            if coords:
                col = torch.arange(h, device=dev, dtype=torch.float32) / max(1, h - 1) * 2 - 1
                xx = col.view(1, 1, h, 1).expand(n, h, h, 1); yy = col.view(1, h, 1, 1).expand(n, h, h, 1)
                xa = torch.cat([xa, xx, yy], -1)
            oh = -(-h // stride); pt = max((oh - 1) * stride + k - h, 0)
            xp = torch.nn.functional.pad(xa.permute(0, 3, 1, 2), (pt // 2, pt - pt // 2, pt // 2, pt - pt // 2))
            ref = torch.nn.functional.conv2d(xp[:8], Vq.permute(3, 2, 0, 1), b, stride=stride).permute(0, 2, 3, 1)
            xin = (x.view(torch.float16) if fmt == lib.F16 else x).float()[:8, ..., :cin]
            if lay.in_post:        # the stored tensor is act(x): the reference operand is the tensor itself, the residual its inverse
                xa8 = xin
                xp = torch.nn.functional.pad(torch.cat([xa8, xa[:8, ..., cin:]], -1).permute(0, 3, 1, 2), (pt // 2, pt - pt // 2, pt // 2, pt - pt // 2))
                ref = torch.nn.functional.conv2d(xp, Vq.permute(3, 2, 0, 1), b, stride=stride).permute(0, 2, 3, 1)
                xin = torch.where(xin > 0, xin, xin / 0.2)
            if res_self:
                ref = ref + xin
            if lay.out_act:
                ref = torch.nn.functional.leaky_relu(ref, 0.2)
            yf = (y.view(torch.float16) if fmt == lib.F16 else y).float()
            err = float((yf[:8, ..., :cout] - ref).abs().max() / ref.abs().max())
            print("           fwd max-rel err vs torch fp32 (first 8 images): {:.2e}".format(err))
    print("{:10s} {:9.3f} {:8s} {:9.3f} {:8s} {:9.3f}".format("sum", tot[0], "", tot[1], "", tot[2]))


if __name__ == "__main__":
    main()
