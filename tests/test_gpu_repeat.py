"""Repeat-launch bit identity at the BENCH shapes (round 6; round-5 verdict, next 1d).

The one wrong result this tree has measured from a kernel whose code was right -- the row-stream forward's packed fp32 add beside a sibling
wave's MFMA section, docs/design/rows_hazard.md -- was invisible to every parity test: the values are right in most launches, at small
batches nearly always, and wrong by one residual in a few elements of one launch in a few.  What catches that class is what caught it by
hand: the SAME launch repeated on the same operands must return the same bits, at the sizes the headline step runs (where the kernels
are full of sibling waves), for every row-stream / patch-kernel instance that step launches.  Each case is also held to the alternative
kernel for the same layer (patch kernel / generic kernel) at a loose tolerance, so that "identically wrong" cannot pass either."""
import math

import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu
REPEATS = 6


def _mods():
    import upsparts_amd  # noqa: F401
    from upsparts_amd import lib, ops
    return lib, ops


def _pack_signs(t16):
    pos = (t16.view(torch.int16) > 0).view(*t16.shape[:-1], -1, 8).to(torch.uint8)
    return (pos * (2 ** torch.arange(8, device=t16.device, dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()


def _bits_equal(a, b):
    return torch.equal(a.view(torch.int16) if a.dtype in (torch.bfloat16, torch.float16) else a, b.view(torch.int16) if b.dtype in (torch.bfloat16, torch.float16) else b)


# name, n, h, w, cin, cout, stride, f16, coords, alternative-kernel environment
RESBLOCK_CASES = [
    ("ea_rb0 (rows, 32 ch @128)", 640, 128, 128, 32, 32, 1, False, False, {"UPS_ROWS_KERNEL": "0"}),
    ("ea_rb1 (rows, 64 ch @64)", 640, 64, 64, 64, 64, 1, False, False, {"UPS_ROWS_KERNEL": "0"}),
    ("dd / vgg1 (rows2, 64 ch @128)", 128, 128, 128, 64, 64, 1, False, False, {"UPS_ROWS_KERNEL": "0"}),
    ("ea_rb2 (patch, 128 ch @32)", 640, 32, 32, 128, 128, 1, False, False, None),
    ("dv_rb128 (patch, 258 -> 256 @128, fp16 forward)", 128, 128, 128, 256, 256, 1, True, True, None),
    ("dv_rb64 (patch, 258 -> 256 @64, fp16 forward)", 128, 64, 64, 256, 256, 1, True, True, None),
]


@pytest.mark.parametrize("case", RESBLOCK_CASES, ids=[c[0].split(" (")[0] for c in RESBLOCK_CASES])
def test_residual_block_launches_repeat_bit_identically(case, dev, monkeypatch):
    """Forward (post-activation storage, residual from the operand) and input gradient (from the producer's sign bytes, as the step runs
    it, and from the forward input) of the residual-block convolutions at the headline step's shapes: REPEATS launches, identical bits."""
    lib, ops = _mods()
    name, n, h, w, cin, cout, stride, f16, coords, alt = case
    g = torch.Generator().manual_seed(4242)
    cin_v = cin + (2 if coords else 0)
    V = (torch.randn(3, 3, cin_v, cout, generator=g) / math.sqrt(9 * cin_v)).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    lay = ops.ConvLayer("t/conv2d_0", V, b, 3, stride, coords, "leaky_relu")
    lay.f16 = f16
    lay.in_post, lay.out_act = True, lib.ACT_LRELU
    fmt = lib.F16 if f16 else None
    x = torch.randn(n, h, w, cin, device=dev)
    x = torch.where(x > 0, x, 0.2 * x)
    xs = x.to(torch.float16).view(torch.bfloat16) if f16 else x.to(torch.bfloat16)
    del x
    gy = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
    xb = _pack_signs(xs)

    def run():
        y = ops.conv_forward(xs, lay, res=xs, fmt=fmt, res_post=True)
        gx_bits = ops.conv_dgrad(gy, xs, lay, res=gy, x_bits=xb)
        gx = ops.conv_dgrad(gy, xs, lay, res=gy)
        torch.cuda.synchronize()
        return y, gx_bits, gx
    first = run()
    for rep in range(1, REPEATS):
        again = run()
        for what, a, b_ in zip(("forward", "input gradient (sign bytes)", "input gradient (forward input)"), first, again):
            nd = int((a.view(torch.int16) != b_.view(torch.int16)).sum())
            assert nd == 0, "{}: {} differs in {} elements between launch 0 and launch {}".format(name, what, nd, rep)
        del again
    # The two input gradients are the same numbers up to fp32 summation order: with sign bytes the wide instances take the residual from
    # the resident patch DURING the channel loop (res_patch == 2, round 6), without them it is added in the epilogue -- a bf16 rounding
    # flips in a few elements per thousand, by one unit in the last place; anything else is a bug.
    a, b_ = first[1].float(), first[2].float()
    diff = (a - b_).abs()
    ulp = b_.abs() * 2.0 ** -7 + 1e-5 * float(b_.pow(2).mean().sqrt())      # one bf16 unit, or (results near zero: cancellation) the fp32 noise of the terms
    assert bool((diff <= ulp).all()), "{}: input gradient from sign bytes vs from the forward input: max {} of the bound".format(name, float((diff / ulp).max()))
    assert float((diff > 0).float().mean()) < 0.02, "{}: {} of the elements differ".format(name, float((diff > 0).float().mean()))
    if alt:
        for k, v in alt.items():
            monkeypatch.setenv(k, v)
        other = run()
        for what, a, b_ in zip(("forward", "input gradient"), first[:2], other[:2]):
            a32 = (a.view(torch.float16) if (f16 and what == "forward") else a).float()
            b32 = (b_.view(torch.float16) if (f16 and what == "forward") else b_).float()
            assert_close(a32, b32, 1e-2, "{}: {} against the alternative kernel".format(name, what))


S2_CASES = [("ea_down0 (rows_s2, 32 -> 64 @128)", 640, 128, 32, 64), ("ea_down1 (rows_s2, 64 -> 128 @64)", 640, 64, 64, 128)]


@pytest.mark.parametrize("case", S2_CASES, ids=[c[0].split(" (")[0] for c in S2_CASES])
def test_stride2_row_stream_launches_repeat_bit_identically(case, dev, monkeypatch):
    lib, ops = _mods()
    name, n, h, cin, cout = case
    g = torch.Generator().manual_seed(77)
    V = (torch.randn(3, 3, cin, cout, generator=g) / math.sqrt(9 * cin)).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    lay = ops.ConvLayer("t/conv2d_0", V, b, 3, 2, False, None)
    lay.out_act = lib.ACT_LRELU
    xs = torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16)
    ops.SignBits.want, ops.SignBits.last = True, None
    y0 = ops.conv_forward(xs, lay)
    bits0 = ops.SignBits.take()
    for rep in range(1, REPEATS):
        ops.SignBits.want, ops.SignBits.last = True, None
        y = ops.conv_forward(xs, lay)
        bits = ops.SignBits.take()
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), y0.view(torch.int16)), "{}: launch {} differs".format(name, rep)
        assert (bits is None) == (bits0 is None) and (bits is None or torch.equal(bits, bits0))
    if bits0 is not None:
        assert torch.equal(bits0, _pack_signs(y0))
    monkeypatch.setenv("UPS_ROWS_KERNEL", "0")
    monkeypatch.setenv("UPS_S2_KERNEL", "0")
    assert_close(y0.float(), ops.conv_forward(xs, lay).float(), 1e-2, "{} against the generic kernel".format(name))


def test_logit_convolution_repeats_bit_identically(dev, monkeypatch):
    """decoder_visualize's 258 -> P logit convolution (conv3x3_thinout_kernel) at the headline shape: its output decides the masks."""
    lib, ops = _mods()
    g = torch.Generator().manual_seed(5)
    n, h, P = 128, 128, 10
    V = (torch.randn(3, 3, 258, P, generator=g) / math.sqrt(9 * 258)).to(dev)
    b = (torch.randn(P, generator=g) * 0.1).to(dev)
    lay = ops.ConvLayer("t/conv2d_0", V, b, 3, 1, True, None)
    lay.f16 = True
    xs = torch.randn(n, h, h, 256, device=dev).to(torch.float16).view(torch.bfloat16)
    y0 = ops.conv_forward(xs, lay, out_f32=True, fmt=lib.F16)
    for rep in range(1, REPEATS):
        y = ops.conv_forward(xs, lay, out_f32=True, fmt=lib.F16)
        torch.cuda.synchronize()
        assert torch.equal(y, y0), "logit convolution: launch {} differs".format(rep)
    monkeypatch.setenv("UPS_ROWS_KERNEL", "0")
    assert_close(y0[..., :P], ops.conv_forward(xs, lay, out_f32=True, fmt=lib.F16)[..., :P], 2e-3, "thin-out kernel against the patch kernel")


def test_part_mask_gradient_repeats_bit_identically(dev, monkeypatch):
    """conv3x3_rows_maskgrad_kernel at the headline shape (P = 10, B = 64, 128 x 128): the one kernel of the file that still brings an
    operand in by inline-asm register loads (gated at build time by tools/check_asm_loads.py)."""
    lib, ops = _mods()
    g = torch.Generator().manual_seed(11)
    P, B, H = 10, 64, 128
    view = (torch.rand(B, H, H, 3, generator=g) * 2 - 1).to(dev)
    mean = torch.randn(B, H, H, P, generator=g).to(dev)
    _, m, hard, _, bits = ops.part_softmax(mean, None, want_bits=True)
    V = torch.randn(3, 3, 3, 32, generator=g) / math.sqrt(27)
    b = torch.randn(32, generator=g) * 0.1
    view_act = torch.zeros(B, H, H, 8, dtype=torch.bfloat16, device=dev)
    view_act[..., :3] = view.to(torch.bfloat16)
    gy = torch.randn(P * B, H, H, 32, device=dev).to(torch.bfloat16)
    lay = ops.ConvLayer("t/conv2d_0", V.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), 3, 1, False, None)

    def run():
        hh = hard.clone().requires_grad_(True)
        y = ops.conv(view_act, lay, mask=(hh, bits, view))
        gh, = torch.autograd.grad([y], [hh], grad_outputs=[gy])
        torch.cuda.synchronize()
        return y.detach(), gh
    y0, g0 = run()
    for rep in range(1, REPEATS):
        y, gh = run()
        assert torch.equal(y.view(torch.int16), y0.view(torch.int16)), "part-masked first convolution: launch {} differs".format(rep)
        assert torch.equal(gh, g0), "mask gradient: launch {} differs".format(rep)
    monkeypatch.setenv("UPS_ROWS_KERNEL", "0")
    y1, g1 = run()
    # (both kernels round gx to bf16 before the dot product and sum their taps in different orders: a rounding flips now and then,
    # and over 10 M elements the largest flip is ~1.5x what the small parity shapes see at 5e-3)
    assert_close(g0, g1, 1e-2, "row-stream mask gradient against the patch kernel's epilogue")
