"""GPU parity tests: every HIP kernel family (through the C ABI) against the CPU oracle on identical
seeded inputs.  fp32 mode must meet north_star's 1e-3 relative tolerance (the kernel-level bars below
are tighter); bf16 mode is compared with the oracle evaluated on bf16-rounded inputs at 2e-2."""
import math

import numpy as np
import pytest
import torch

from util import assert_close, rel_err

pytestmark = pytest.mark.gpu

F32_TOL = 2e-4      # kernel-level bar for the fp32 (exact MFMA f32) path; north_star end-to-end bar is 1e-3
BF16_TOL = 2e-2


def _mods():
    import upsparts_amd  # noqa: F401
    from upsparts_amd import lib, ops
    from oracle import ref_model as R
    return lib, ops, R


def _layer(ops, lib, V, b, k, stride, coords, act, dev):
    return ops.ConvLayer("t/conv2d_0", V.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), k, stride, coords,
                         act)


def _oracle_conv(R, x, V, b, stride, coords, act, res_self, res):
    xa = x
    if act == "leaky_relu":
        xa = torch.nn.functional.leaky_relu(x, 0.2)
    elif act == "relu":
        xa = torch.relu(x)
    if coords:
        xa = R.Scope.add_coordinates(xa)
    y = R.conv2d_same(xa, V, b, stride)
    if res_self:
        y = y + x
    if res is not None:
        y = y + res
    return y


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, coords, act, res_self
    (2, 16, 16, 8, 16, 3, 1, False, None, False),
    (2, 16, 16, 16, 16, 3, 1, True, "leaky_relu", True),
    (3, 16, 16, 16, 24, 3, 2, True, None, False),
    (2, 12, 20, 32, 40, 3, 2, False, "leaky_relu", False),
    (2, 8, 8, 64, 130, 3, 1, True, "leaky_relu", False),
    (4, 1, 1, 16, 72, 1, 1, True, None, False),
    (5, 4, 4, 16, 16, 1, 1, False, "leaky_relu", True),
    (1, 32, 32, 8, 3, 3, 1, False, "relu", False),
    (2, 9, 7, 8, 8, 3, 2, True, None, False),       # odd sizes: pad_before = 1 for stride 2
    (1, 40, 40, 256, 256, 3, 1, True, "leaky_relu", True),
    # patch-tiled kernels (3x3 / stride 1 / 16-aligned): every tile-width variant, several tiles per image (image
    # borders inside and between tiles), ragged channel counts, CoordConv + residual epilogues
    (3, 32, 48, 64, 136, 3, 1, True, "leaky_relu", False),     # fwd BN=128 x2 N-tiles; wgrad3x3 <64,128>
    (2, 32, 32, 128, 128, 3, 1, True, "leaky_relu", True),     # residual block, staged epilogue
    (2, 48, 32, 72, 64, 3, 1, False, "relu", False),           # BN=64; wgrad3x3 <64,64>
    (5, 16, 32, 24, 40, 3, 1, True, None, False),              # wgrad3x3 <32,64>
    (2, 32, 32, 32, 96, 3, 1, False, "leaky_relu", False),     # wgrad3x3 <32,128>
    (2, 32, 16, 80, 16, 3, 1, False, None, False),             # wgrad3x3 <64,32>
    (2, 64, 64, 8, 32, 3, 1, False, None, False),              # first encoder conv (image input), <32,32>
    (8, 64, 64, 32, 32, 3, 1, False, "leaky_relu", True),      # thin res-block, many tiles (XCD remap path)
    # grids of >= 512 blocks: the two-blocks-per-CU configuration of the patch kernel (2-stage weight ring, single patch
    # buffer, sign-byte activation-derivative tile), forward with CoordConv + residual and dgrad with act' + residual
    (16, 64, 64, 136, 136, 3, 1, True, "leaky_relu", True),    # <bf16,128,2>, ragged second N-tile
    (34, 64, 64, 64, 64, 3, 1, False, "relu", True),           # <bf16,64,2>
    # whole 8x8 / 4x4 images packed 4 / 16 to a patch tile (each with its own zero halo): every N-tile width
    (8, 8, 8, 72, 72, 3, 1, True, "leaky_relu", True),         # SUB=8, 2 tiles, BN=32 (small grid)
    (256, 8, 8, 24, 72, 3, 1, False, "relu", False),           # SUB=8, BN=64
    (32, 4, 4, 136, 136, 3, 1, True, "leaky_relu", True),      # SUB=4, 2 tiles, BN=32
    (1536, 4, 4, 16, 136, 3, 1, True, None, False),            # SUB=4, BN=128 with a ragged second N-tile
    # the instantiations bench.py runs, at their real K depth (BASELINE config #2 widths)
    (8, 128, 128, 256, 256, 3, 1, True, "leaky_relu", True),   # decoder_visualize/conv2d_8 (37 % of the forward FLOPs): patch
                                                               # kernel <bf16,128,OCC=2>, kchunks 8, 1024 blocks through the XCD
                                                               # remap, CoordConv + residual epilogue; wgrad3x3 <64,128> at 128x128
    (64, 16, 16, 512, 512, 3, 1, False, "relu", False),        # VGG block 4 / 5 depth: kchunks 16, one block per CU (3-stage ring)
    (64, 8, 8, 512, 512, 3, 1, False, "relu", False),          # VGG block 5 at 8x8: whole images packed 4 to a tile, kchunks 16
    (4, 128, 128, 32, 32, 3, 1, True, "leaky_relu", True),     # encoder_0 first res-block: thin CoordConv layer at 128x128
    (2, 128, 128, 80, 32, 3, 1, False, None, False),           # decoder_delta input conv (74 + pad -> 32)
    (16, 64, 64, 256, 256, 3, 1, True, "leaky_relu", True),    # dv res-block at 64x64
    # wide layers on >= 256 tiles: ragged second N-tile, an odd number of 32-channel chunks, 512 outputs, dgrad with act' + residual
    (16, 64, 64, 96, 200, 3, 1, True, "leaky_relu", False),
    (4, 128, 128, 200, 200, 3, 1, False, "relu", True),
    (16, 64, 64, 64, 512, 3, 1, False, "relu", False),
    # thin single-chunk layers on large grids (>= 1536 tiles, three blocks per CU): the HBM-bound streams of encoder_1 on parts
    (24, 128, 128, 32, 32, 3, 1, True, "leaky_relu", True),    # <bf16,32,1,16>: res-block with CoordConv
    (25, 128, 128, 8, 32, 3, 1, False, None, False),           # image-input conv
    (24, 128, 128, 32, 64, 3, 1, False, "relu", False),        # <bf16,64,1,16>
    # 3x3 / stride-2 downsample layers whose bf16 input gradient runs as ONE depth-to-space convolution over the gradient lattice
    (4, 64, 64, 32, 64, 3, 2, False, None, False),             # encoder first downsample: 4 x 32 = 128 GEMM channels, one N-tile
    (2, 128, 128, 32, 64, 3, 2, False, "leaky_relu", False),   # with act' on the 2x lattice
    (3, 32, 32, 128, 256, 3, 2, True, None, False),            # CoordConv layer, 512 GEMM channels (4 N-tiles), kchunks 8
    (2, 64, 32, 64, 128, 3, 2, False, "relu", False),          # non-square, two N-tiles    # the critics' dense layers (128 rows): split-K forward / input gradient, single-split weight gradient written in place
    (128, 1, 1, 512, 512, 1, 1, False, "leaky_relu", True),
    (128, 1, 1, 64, 512, 1, 1, False, None, False),
    # ragged patch tiles (round 5): maps that are not a multiple of the 16-pixel tile -- the 56 / 28 / 14-pixel levels of the
    # perceptual trunk behind the 224 x 224 crop (`perceptual_input: resize256_crop224`) -- on the patch kernel with the pixels of
    # the overhanging tiles skipped in the epilogue (forward, input gradient with act'; (1, 40, 40, ...) above adds CoordConv + residual)
    (16, 28, 28, 256, 256, 3, 1, False, "relu", False),        # VGG block 4 at 28 x 28: 2 x 2 tiles, 12 of 16 pixels valid in the last
    (16, 14, 14, 512, 512, 3, 1, False, "relu", False),        # VGG block 5 at 14 x 14: one tile per image, kchunks 16
    (8, 56, 56, 128, 256, 3, 1, False, "relu", False),         # VGG block 3 at 56 x 56: 4 x 4 tiles, two N-tiles
    (3, 24, 40, 64, 136, 3, 1, True, "leaky_relu", False),     # non-square, both edges ragged, CoordConv, ragged second N-tile
    (40, 56, 56, 64, 64, 3, 1, False, "leaky_relu", True),     # >= 512 blocks: the two-blocks-per-CU instance, residual from the patch
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, dtype, dev):
    lib, ops, R = _mods()
    n, h, w, cin, cout, k, stride, coords, act, res_self = case
    g = torch.Generator().manual_seed(100 + CONV_CASES.index(case))
    cin_v = cin + (2 if coords else 0)
    x = torch.randn(n, h, w, cin, generator=g)
    V = torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(cin_v * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    T = torch.float32 if dtype == "fp32" else torch.bfloat16
    tol = F32_TOL if dtype == "fp32" else BF16_TOL
    if dtype == "bf16":     # the oracle sees the same rounded operands
        x = x.to(T).float()
    Vq = V.to(T).float() if dtype == "bf16" else V
    # oracle (fp64) with autograd
    xo = x.double().requires_grad_(True)
    Vo = V.double().clone()
    if dtype == "bf16":     # main channels are rounded by weight_prep, CoordConv rows stay fp32
        Vo[:, :, :cin] = Vq[:, :, :cin].double()
    Vo.requires_grad_(True)
    bo = b.double().requires_grad_(True)
    yo = _oracle_conv(R, xo, Vo, bo, stride, coords, act, res_self, None)
    go = torch.randn(yo.shape, generator=g)
    if dtype == "bf16":
        go = go.to(T).float()
    yo.backward(go.double())
    # HIP
    lay = _layer(ops, lib, V, b, k, stride, coords, act, dev)
    xd = x.to(dev, T).requires_grad_(True)
    y = ops.conv(xd, lay, res_self=res_self)
    assert y.shape[-1] == ops.round8(cout)
    assert_close(y[..., :cout].float(), yo.float(), tol, "conv fwd {}".format(case))
    if y.shape[-1] > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0
    gd = torch.zeros(y.shape, dtype=T, device=dev)
    gd[..., :cout] = go.to(dev, T)
    gx, gV, gb = torch.autograd.grad([y], [xd, lay.V, lay.b], grad_outputs=[gd])
    assert_close(gx.float(), xo.grad.float(), tol, "conv dgrad {}".format(case))
    assert_close(gV.float(), Vo.grad.float(), tol * (4 if dtype == "bf16" else 1), "conv wgrad {}".format(case))
    assert_close(gb.float(), bo.grad.float(), tol, "conv bias grad {}".format(case))


S2_CASES = [
    # n, h, w, cin, cout, coords, out_act
    (4, 64, 64, 32, 64, False, False),      # encoder first downsample
    (3, 32, 32, 64, 128, True, True),       # 64 input channels (two chunks), two 64-wide output blocks, CoordConv, stored activation
    (2, 32, 64, 32, 24, True, False),       # ragged output channels (24 of a 64-wide block), non-square
    (2, 31, 63, 32, 64, False, True),       # odd input sizes: pad_before = 1 (taps reach above / left of the image)
    (2, 128, 128, 32, 64, True, False),     # 128x128 -> 64x64 as in the model
]


@pytest.mark.parametrize("case", S2_CASES)
def test_conv_stride2_forward_kernel(case, dev, monkeypatch):
    """conv3x3_s2.hip: the forward of the large 3x3 / stride-2 `downsample` layers (N:816-817) with the taps read straight from
    global memory.  The model only routes launches of >= 128 Ki output pixels to it; UPS_S2_KERNEL=force takes the size gate away so
    that the parity shapes stay small.  Against the fp64 oracle on the bf16-rounded operands, and against the generic gather
    kernel (UPS_S2_KERNEL=0) on the same inputs."""
    lib, ops, R = _mods()
    n, h, w, cin, cout, coords, out_act = case
    g = torch.Generator().manual_seed(500 + S2_CASES.index(case))
    cin_v = cin + (2 if coords else 0)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    V = torch.randn(3, 3, cin_v, cout, generator=g) / math.sqrt(cin_v * 9)
    b = torch.randn(cout, generator=g) * 0.1
    Vo = V.double().clone()
    Vo[:, :, :cin] = V[:, :, :cin].to(torch.bfloat16).double()
    yo = _oracle_conv(R, x.double(), Vo, b.double(), 2, coords, None, False, None)
    if out_act:
        yo = torch.nn.functional.leaky_relu(yo, 0.2)
    lay = _layer(ops, lib, V, b, 3, 2, coords, None, dev)
    if out_act:
        lay.out_act = lib.ACT_LRELU
    xd = x.to(dev)
    monkeypatch.setenv("UPS_S2_KERNEL", "force")
    y = ops.conv_forward(xd, lay)
    monkeypatch.setenv("UPS_S2_KERNEL", "0")
    y_gen = ops.conv_forward(xd, lay)
    torch.cuda.synchronize()
    assert y.shape == y_gen.shape and y.shape[-1] == ops.round8(cout)
    assert_close(y[..., :cout].float(), yo.float(), BF16_TOL, "stride-2 kernel vs oracle {}".format(case))
    assert_close(y_gen[..., :cout].float(), yo.float(), BF16_TOL, "generic kernel vs oracle {}".format(case))
    assert_close(y[..., :cout].float(), y_gen[..., :cout].float(), 1e-2, "stride-2 kernel vs generic kernel {}".format(case))
    if y.shape[-1] > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0


ROWS_CASES = [
    # n, h (w = 128 for 32 channels, 64 for 64 channels), channels, fmt
    (3, 128, 32, "bf16"),       # encoder_1's first residual block on the part images (N:1042-1056 at 32 channels, 128x128)
    (3, 64, 64, "bf16"),        # its second (64 channels, 64x64: two 32-channel planes, two output groups)
    (2, 128, 32, "f16"),
    (2, 64, 64, "f16"),
    (5, 32, 32, "bf16"),        # a single band per image (32 x 128): top and bottom zero rows in the same block
    (1, 96, 64, "bf16"),        # three bands, non-square (96 x 64)
    # two column tiles per wave (conv3x3_rows2_kernel; the forward only -- the input gradient with act' stays on the patch kernel)
    (2, 64, 64, "bf16", 128),   # 64 channels at 128 columns: VGG block 1 / the hourglass decoder
    (2, 32, 32, "bf16", 256),   # 32 channels at 256 columns: encoder_1's first residual block of the 256 x 256 configs
    (1, 64, 64, "f16", 128),
]


@pytest.mark.parametrize("case", ROWS_CASES)
def test_conv_rows_kernel(case, dev, monkeypatch):
    """conv3x3_rows.hip: the thin residual blocks as row streams (whole input rows by LDS-DMA into a ring, weights in registers).
    The model routes only launches of >= 1024 row bands to it; UPS_ROWS_KERNEL=force takes the size gate away so that the parity
    shapes stay small.  Forward (post-activation storage, residual from the resident centre row, stored activation) and input
    gradient (flipped taps, act' from the stored forward input, residual gradient) against the fp64 oracle's autograd on the
    rounded operands, and against the patch kernel (UPS_ROWS_KERNEL=0) on the same inputs."""
    lib, ops, R = _mods()
    n, h, c, fmt_name = case[:4]
    w = case[4] if len(case) > 4 else (128 if c == 32 else 64)
    g = torch.Generator().manual_seed(700 + ROWS_CASES.index(case))
    f16 = fmt_name == "f16"
    TF = torch.float16 if f16 else torch.bfloat16
    V = torch.randn(3, 3, c, c, generator=g) / math.sqrt(c * 9)
    b = torch.randn(c, generator=g) * 0.1
    lay = _layer(ops, lib, V, b, 3, 1, False, "leaky_relu", dev)
    lay.f16 = f16
    lay.in_post, lay.out_act = True, lib.ACT_LRELU
    fmt = lib.F16 if f16 else None
    xs = torch.nn.functional.leaky_relu(torch.randn(n, h, w, c, generator=g), 0.2).to(TF)      # the stored tensor: act(x)
    xd = xs.to(dev).view(torch.bfloat16) if f16 else xs.to(dev)
    gy = torch.randn(n, h, w, c, generator=g).to(torch.bfloat16)
    gyd = gy.to(dev)

    def run():
        y = ops.conv_forward(xd, lay, res=xd, fmt=fmt, res_post=True)
        gx = ops.conv_dgrad(gyd, xd, lay, res=gyd)
        torch.cuda.synchronize()
        return (y.view(torch.float16) if f16 else y).float().cpu(), gx.float().cpu()
    monkeypatch.setenv("UPS_ROWS_KERNEL", "force")
    y1, g1 = run()
    for _ in range(4):      # (round 5: an edit that left the wait counts alone made the 64-channel instance differ from run to run)
        y2, g2 = run()
        assert torch.equal(y1, y2) and torch.equal(g1, g2), "not reproducible {}".format(case)
    monkeypatch.setenv("UPS_ROWS_KERNEL", "0")
    y0, g0 = run()
    # oracle: out = act(conv(act(x)) + b + x) with x = act^-1(stored); gx = d<out_pre, gy>/dx
    xo = xs.double()
    xpre = torch.where(xo > 0, xo, xo / 0.2).requires_grad_(True)
    Vo = V.to(TF).double()
    pre = _oracle_conv(R, torch.nn.functional.leaky_relu(xpre, 0.2), Vo, b.double(), 1, False, None, False, None) + xpre
    yo = torch.nn.functional.leaky_relu(pre, 0.2).detach()
    # the input gradient's weights are the bf16 rounding of V in either format
    xpre2 = xpre.detach().clone().requires_grad_(True)
    pre2 = _oracle_conv(R, torch.nn.functional.leaky_relu(xpre2, 0.2), V.to(torch.bfloat16).double(), b.double(), 1, False, None, False, None) + xpre2
    go, = torch.autograd.grad([pre2], [xpre2], grad_outputs=[gy.double()])
    tol = F16_TOL if f16 else BF16_TOL
    assert_close(y1, yo.float(), tol, "rows kernel forward vs oracle {}".format(case))
    assert_close(y0, yo.float(), tol, "patch kernel forward vs oracle {}".format(case))
    assert_close(g1, go.float(), BF16_TOL, "rows kernel input gradient vs oracle {}".format(case))
    assert_close(g0, go.float(), BF16_TOL, "patch kernel input gradient vs oracle {}".format(case))
    assert_close(y1, y0, 1e-2, "rows vs patch kernel forward {}".format(case))
    assert_close(g1, g0, 1e-2, "rows vs patch kernel input gradient {}".format(case))


ROWS_S2_CASES = [
    # n, h = w, cin, cout, stored activation
    (3, 128, 32, 64, False),     # encoder_1 / encoder_0 first downsample (N:816-817)
    (3, 64, 64, 128, True),      # second downsample: two 32-channel planes, four output groups
    (2, 128, 32, 64, True),
    (2, 256, 32, 64, False),     # two column tiles per wave: the same layers of the 256 x 256 configs
    (1, 128, 64, 128, True),
]


@pytest.mark.parametrize("case", ROWS_S2_CASES)
def test_conv_rows_stride2_kernel(case, dev, monkeypatch):
    """conv3x3_rows.hip, stride-2 form: the encoders' first two `downsample` forwards as row streams (even / odd pixel planes per
    input row).  Against the fp64 oracle on the bf16-rounded operands and against the generic gather kernel on the same inputs."""
    lib, ops, R = _mods()
    n, h, cin, cout, out_act = case
    g = torch.Generator().manual_seed(800 + ROWS_S2_CASES.index(case))
    x = torch.randn(n, h, h, cin, generator=g).to(torch.bfloat16)
    V = torch.randn(3, 3, cin, cout, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    yo = _oracle_conv(R, x.double(), V.to(torch.bfloat16).double(), b.double(), 2, False, None, False, None)
    if out_act:
        yo = torch.nn.functional.leaky_relu(yo, 0.2)
    lay = _layer(ops, lib, V, b, 3, 2, False, None, dev)
    if out_act:
        lay.out_act = lib.ACT_LRELU
    xd = x.to(dev)
    monkeypatch.setenv("UPS_ROWS_KERNEL", "force")
    y = ops.conv_forward(xd, lay)
    y_again = ops.conv_forward(xd, lay)
    monkeypatch.setenv("UPS_ROWS_KERNEL", "0")
    monkeypatch.setenv("UPS_S2_KERNEL", "0")
    y_gen = ops.conv_forward(xd, lay)
    torch.cuda.synchronize()
    assert y.shape == y_gen.shape == (n, h // 2, h // 2, cout)
    assert torch.equal(y.view(torch.int16), y_again.view(torch.int16))
    assert_close(y.float(), yo.float(), BF16_TOL, "row-streaming stride-2 kernel vs oracle {}".format(case))
    assert_close(y_gen.float(), yo.float(), BF16_TOL, "generic kernel vs oracle {}".format(case))
    assert_close(y.float(), y_gen.float(), 1e-2, "row-streaming stride-2 kernel vs generic kernel {}".format(case))


THINOUT_CASES = [
    # n, h, w, P, fmt, fp32 output, coords
    (3, 128, 128, 10, "f16", True, True),       # decoder_visualize's logit convolution as the headline runs it (M:154)
    (2, 64, 96, 16, "bf16", False, True),       # 16 outputs stored as bf16, non-square, three strips
    (2, 32, 32, 3, "bf16", True, False),        # one strip, one band, three parts, no CoordConv
    (1, 128, 256, 25, "f16", True, True),       # P = 25 does not fit 16 outputs: must fall through to the patch kernel
]


@pytest.mark.parametrize("case", THINOUT_CASES)
def test_conv_thinout_kernel(case, dev, monkeypatch):
    """conv3x3_rows.hip, K-deep thin-out form (256 -> P <= 16 channels, eight waves split K, partials summed through LDS): against
    the fp64 oracle on the rounded operands and against the patch kernel (UPS_ROWS_KERNEL=0) on the same inputs."""
    lib, ops, R = _mods()
    n, h, w, P, fmt_name, out_f32, coords = case
    g = torch.Generator().manual_seed(600 + THINOUT_CASES.index(case))
    f16 = fmt_name == "f16"
    TF = torch.float16 if f16 else torch.bfloat16
    cin = 256
    cin_v = cin + (2 if coords else 0)
    x = torch.randn(n, h, w, cin, generator=g).to(TF)
    V = torch.randn(3, 3, cin_v, P, generator=g) / math.sqrt(cin_v * 9)
    b = torch.randn(P, generator=g) * 0.1
    Vo = V.double().clone()
    Vo[:, :, :cin] = V[:, :, :cin].to(TF).double()
    yo = _oracle_conv(R, x.double(), Vo, b.double(), 1, coords, None, False, None)
    lay = _layer(ops, lib, V, b, 3, 1, coords, None, dev)
    lay.f16 = f16
    fmt = lib.F16 if f16 else None
    xd = x.to(dev).view(torch.bfloat16) if f16 else x.to(dev)

    def run():
        y = ops.conv_forward(xd, lay, out_f32=out_f32, fmt=fmt)
        torch.cuda.synchronize()
        if not out_f32:
            y = (y.view(torch.float16) if f16 else y).float()
        return y.cpu()
    monkeypatch.setenv("UPS_ROWS_KERNEL", "force")
    y1 = run()
    y2 = run()
    monkeypatch.setenv("UPS_ROWS_KERNEL", "0")
    y0 = run()
    assert torch.equal(y1, y2)
    tol = (F16_TOL if f16 else BF16_TOL) / (1 if out_f32 else 1)
    if not out_f32 and not f16:
        tol = BF16_TOL
    assert_close(y1[..., :P], yo.float(), tol, "thin-out kernel vs oracle {}".format(case))
    assert_close(y0[..., :P], yo.float(), tol, "patch kernel vs oracle {}".format(case))
    assert_close(y1[..., :P], y0[..., :P], 1e-2, "thin-out vs patch kernel {}".format(case))
    if y1.shape[-1] > P:
        assert float(y1[..., P:].abs().max()) == 0.0


F16_CASES = [c for c in CONV_CASES if c[6] == 1 and c[8] in (None, "leaky_relu")
             and c in ((2, 16, 16, 16, 16, 3, 1, True, "leaky_relu", True), (4, 1, 1, 16, 72, 1, 1, True, None, False),
                       (3, 32, 48, 64, 136, 3, 1, True, "leaky_relu", False), (2, 32, 32, 128, 128, 3, 1, True, "leaky_relu", True),
                       (5, 16, 32, 24, 40, 3, 1, True, None, False), (16, 64, 64, 136, 136, 3, 1, True, "leaky_relu", True),
                       (8, 8, 8, 72, 72, 3, 1, True, "leaky_relu", True), (32, 4, 4, 136, 136, 3, 1, True, "leaky_relu", True),
                       (1536, 4, 4, 16, 136, 3, 1, True, None, False), (8, 128, 128, 256, 256, 3, 1, True, "leaky_relu", True),
                       (16, 64, 64, 256, 256, 3, 1, True, "leaky_relu", True), (2, 40, 40, 256, 256, 3, 1, True, "leaky_relu", True),
                       (1, 40, 40, 256, 256, 3, 1, True, "leaky_relu", True))]
F16_TOL = 2e-3


@pytest.mark.parametrize("case", F16_CASES)
def test_conv_fp16_forward_bf16_gradients(case, dev):
    """The mask decoder's tensor format (`fmt = UPS_F16`): forward tensors and forward weights are fp16 (in bf16 containers),
    the gradients and the backward operands bf16.  Forward vs the fp64 oracle on the fp16-rounded operands at 2e-3 (bf16: 2e-2);
    input / weight gradients at the bf16 bar (their operands are bf16 roundings of the same numbers)."""
    lib, ops, R = _mods()
    n, h, w, cin, cout, k, stride, coords, act, res_self = case
    g = torch.Generator().manual_seed(300 + CONV_CASES.index(case))
    cin_v = cin + (2 if coords else 0)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.float16).float()
    V = torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(cin_v * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    xo = x.double().requires_grad_(True)
    Vo = V.double().clone()
    Vo[:, :, :cin] = V[:, :, :cin].to(torch.float16).double()
    Vo.requires_grad_(True)
    bo = b.double().requires_grad_(True)
    yo = _oracle_conv(R, xo, Vo, bo, stride, coords, act, res_self, None)
    go = torch.randn(yo.shape, generator=g).to(torch.bfloat16).float()
    yo.backward(go.double())
    lay = _layer(ops, lib, V, b, k, stride, coords, act, dev)
    lay.f16 = True
    xd = x.to(dev, torch.float16).view(torch.bfloat16).requires_grad_(True)
    y = ops.conv(xd, lay, res_self=res_self, fmt=lib.F16)
    assert y.dtype == torch.bfloat16 and y.shape[-1] == ops.round8(cout)
    yf = y.view(torch.float16).float()
    assert_close(yf[..., :cout], yo.float(), F16_TOL, "fp16 conv fwd {}".format(case))
    if y.shape[-1] > cout:
        assert float(yf[..., cout:].abs().max()) == 0.0
    gd = torch.zeros(y.shape, dtype=torch.bfloat16, device=dev)
    gd[..., :cout] = go.to(dev, torch.bfloat16)
    gx, gV, gb = torch.autograd.grad([y], [xd, lay.V, lay.b], grad_outputs=[gd])
    assert gx.dtype == torch.bfloat16
    assert_close(gx.float(), xo.grad.float(), BF16_TOL, "fp16-forward conv dgrad {}".format(case))
    assert_close(gV.float(), Vo.grad.float(), 4 * BF16_TOL, "fp16-forward conv wgrad {}".format(case))
    assert_close(gb.float(), bo.grad.float(), BF16_TOL, "fp16-forward conv bias grad {}".format(case))
    # fp32 logits out of an fp16 layer (the decoder's last convolution)
    if not res_self:
        y32 = ops.conv(xd, lay, out_f32=True, fmt=lib.F16)
        assert y32.dtype == torch.float32
        assert_close(y32[..., :cout], yo.float(), F16_TOL / 2, "fp16 conv fwd, fp32 out {}".format(case))


POST_CASES = [
    # n, h, w, cin, cout, k, stride, coords, res_self, dtype
    (2, 32, 32, 128, 128, 3, 1, True, True, "bf16"),      # patch kernel, staged epilogue
    (16, 64, 64, 128, 128, 3, 1, True, True, "bf16"),     # two blocks per CU: LDS-DMA patch on the forward (act_in none)
    (8, 128, 128, 256, 256, 3, 1, True, True, "bf16"),    # the dominant layer's shape
    (16, 64, 64, 256, 256, 3, 1, True, True, "f16"),      # mask decoder format
    (32, 4, 4, 136, 136, 3, 1, True, True, "bf16"),       # whole images packed into a tile
    (5, 4, 4, 16, 16, 1, 1, False, True, "bf16"),         # 1x1 residual block of the critics (generic kernel)
    (128, 1, 1, 512, 512, 1, 1, False, True, "bf16"),     # ... at the critics' width (split-K epilogue)
    (2, 16, 16, 16, 16, 3, 1, True, True, "fp32"),
    (3, 16, 16, 16, 24, 3, 2, True, False, "bf16"),       # a downsample whose output is stored post-activation
    (2, 12, 20, 32, 40, 3, 1, False, False, "fp32"),      # generic kernel, per-element epilogue
]


@pytest.mark.parametrize("case", POST_CASES)
def test_conv_post_activation_storage(case, dev):
    """ups_conv_desc.out_act / res_act: the input tensor holds a = lrelu(x) (in_post), the residual is recovered from it
    (x = a > 0 ? a : a / slope), the output is stored as lrelu(y); gradients stay with respect to the pre-activation values."""
    lib, ops, R = _mods()
    n, h, w, cin, cout, k, stride, coords, res_self, dtype = case
    g = torch.Generator().manual_seed(500 + POST_CASES.index(case))
    cin_v = cin + (2 if coords else 0)
    T = {"fp32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    tol = {"fp32": F32_TOL, "bf16": BF16_TOL, "f16": F16_TOL}[dtype]
    x = torch.randn(n, h, w, cin, generator=g)
    a = torch.nn.functional.leaky_relu(x, 0.2).to(T).float()          # what the producer stored
    V = torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(cin_v * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    Vo = V.double().clone()
    if dtype != "fp32":
        Vo[:, :, :cin] = V[:, :, :cin].to(T).double()
    xo = torch.where(a > 0, a, a / 0.2).double().requires_grad_(True)      # the pre-activation value the stored tensor stands for
    Vo.requires_grad_(True)
    bo = b.double().requires_grad_(True)
    ypre = _oracle_conv(R, xo, Vo, bo, stride, coords, "leaky_relu", res_self, None)
    go = torch.randn(ypre.shape, generator=g)
    if dtype != "fp32":
        go = go.to(torch.bfloat16).float()
    ypre.backward(go.double())
    want = torch.nn.functional.leaky_relu(ypre.detach(), 0.2)
    lay = _layer(ops, lib, V, b, k, stride, coords, "leaky_relu", dev)
    lay.in_post, lay.out_act, lay.f16 = True, lib.ACT_LRELU, dtype == "f16"
    fmt = lib.F16 if dtype == "f16" else None
    xd = (a.to(dev, torch.float16).view(torch.bfloat16) if dtype == "f16" else a.to(dev, T)).requires_grad_(True)
    y = ops.conv(xd, lay, res_self=res_self, fmt=fmt)
    yf = y.view(torch.float16).float() if dtype == "f16" else y.float()
    assert_close(yf[..., :cout], want.float(), tol, "post-activation conv fwd {}".format(case))
    TG = torch.float32 if dtype == "fp32" else torch.bfloat16
    gd = torch.zeros(y.shape, dtype=TG, device=dev)
    gd[..., :cout] = go.to(dev, TG)
    gx, gV, gb = torch.autograd.grad([y], [xd, lay.V, lay.b], grad_outputs=[gd])
    btol = F32_TOL if dtype == "fp32" else BF16_TOL
    assert_close(gx.float(), xo.grad.float(), btol, "post-activation conv dgrad {}".format(case))
    assert_close(gV.float(), Vo.grad.float(), btol * (1 if dtype == "fp32" else 4), "post-activation conv wgrad {}".format(case))
    assert_close(gb.float(), bo.grad.float(), btol, "post-activation conv bias grad {}".format(case))


def test_bilinear_post_activation_output(dev):
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(16)
    for T, fmt, tol in ((torch.float32, None, 1e-6), (torch.bfloat16, None, 1e-2), (torch.float16, lib.F16, 1e-3)):
        x = torch.randn(3, 6, 5, 16, generator=g).to(T).float()
        want = torch.nn.functional.leaky_relu(R.bilinear_up2(x.double()), 0.2)
        xd = x.to(dev, T).view(torch.bfloat16) if fmt == lib.F16 else x.to(dev, T)
        y = ops.BilinearFn.apply(xd, None, 0, 0.2, fmt, lib.ACT_LRELU)
        yf = y.view(torch.float16).float() if fmt == lib.F16 else y.float()
        assert_close(yf, want.float(), tol, "bilinear with post-activation output {}".format(T))


@pytest.mark.parametrize("shape", [(3, 16, 16, 3), (2, 32, 32, 10), (2, 128, 128, 25), (4, 64, 64, 10)])
def test_hard_mask_moments_from_the_softmax_pass(shape, dev):
    """ups_part_softmax_moments_fwd: the spatial soft-max moments of gamma * hard (the rectangle centres' input, M:437-441) in
    closed form from integer sums taken while the hard mask is produced == ups_spatial_moments on the hard map (to fp32
    rounding), identical integer centres, and the same l / m / hard / bit sets as the plain call."""
    lib, ops, R = _mods()
    n, h, w, P = shape
    g = torch.Generator().manual_seed(31 + P)
    mean = (torch.randn(n, h, w, P, generator=g) * 2).to(dev)
    mean[0, :, :, P - 1] = -50.0                      # a part that owns no pixel in image 0
    eps = torch.randn(n, h, w, P, generator=g).to(dev)
    l0, m0, h0, _, b0 = ops.part_softmax(mean, eps, want_bits=True)
    l1, m1, h1, _, b1, stats = ops.part_softmax(mean, eps, want_bits=True, moments_gamma=10.0)
    assert stats is not None
    assert torch.equal(l0, l1) and torch.equal(m0, m1) and torch.equal(h0, h1) and torch.equal(b0, b1)
    ref = ops.spatial_moments(h0, 10.0)
    for k in (1, 2, 3, 4, 5, 6):
        assert_close(stats[..., k], ref[..., k], 1e-3, "hard-mask moment {} (vs the fp32 summation kernel)".format(k), elementwise=False)
    # (a part that owns no pixel has its centre of mass exactly on the pixel border h/2: the closed form gives exactly 0, a
    # summation a rounding error of either sign -- the truncated centre of such a part is not defined to better than one pixel)
    dk = (ops.moments_to_px(stats, h) - ops.moments_to_px(ref, h)).abs()
    owned = (h0.sum(dim=(1, 2)) > 0)
    assert int(dk[owned].max()) == 0 and int(dk.max()) <= 1
    # against the oracle's centres (fp64; a centre of mass within fp32 rounding of a pixel border may truncate the other way)
    _, rc = R.patch_mask(h0.double().cpu(), 10.0, 6)
    d = (ops.moments_to_px(stats, h).cpu().long() - rc).abs()
    assert int(d.max()) <= 1 and float((d == 0).float().mean()) >= 0.9, (int(d.max()), float((d == 0).float().mean()))


def test_bilinear_fp16(dev):
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(15)
    x = torch.randn(3, 6, 5, 16, generator=g).to(torch.float16).float()
    yo = R.bilinear_up2(x.double())
    xd = x.to(dev, torch.float16).view(torch.bfloat16).requires_grad_(True)
    y = ops.BilinearFn.apply(xd, None, 0, 0.2, lib.F16)
    assert_close(y.view(torch.float16).float(), yo.float(), 1e-3, "bilinear fp16 fwd")
    go = torch.randn(yo.shape, generator=g).to(torch.bfloat16)
    (gx,) = torch.autograd.grad([y], [xd], grad_outputs=[go.to(dev)])
    xo = x.double().requires_grad_(True)
    R.bilinear_up2(xo).backward(go.double())
    assert_close(gx.float(), xo.grad.float(), 1e-2, "bilinear bwd (bf16 gradient of an fp16 tensor)")


def test_conv_f32_out_and_padded_input(dev):
    """fp32 output with ldo == co (logits / latent head) and an input whose physical width exceeds Cin."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(3)
    n, h, w, cin, cout = 2, 8, 8, 11, 10
    x = torch.randn(n, h, w, cin, generator=g)
    V = torch.randn(3, 3, cin, cout, generator=g) / 10
    b = torch.randn(cout, generator=g)
    yo = R.conv2d_same(x.double(), V.double(), b.double(), 1)
    lay = _layer(ops, lib, V, b, 3, 1, False, None, dev)
    xp = torch.zeros(n, h, w, 16)
    xp[..., :cin] = x
    xd = xp.to(dev).requires_grad_(True)
    y = ops.conv(xd, lay, out_f32=True)
    assert y.dtype == torch.float32 and y.shape[-1] == cout
    assert_close(y, yo.float(), F32_TOL, "conv f32-out")
    go = torch.randn(yo.shape, generator=g)
    xo = x.double().requires_grad_(True)
    Vo = V.double().requires_grad_(True)
    R.conv2d_same(xo, Vo, b.double(), 1).backward(go.double())
    gx, gV = torch.autograd.grad([y], [xd, lay.V], grad_outputs=[go.to(dev)])
    assert_close(gx[..., :cin], xo.grad.float(), F32_TOL, "dgrad padded")
    assert float(gx[..., cin:].abs().max()) == 0.0
    assert_close(gV, Vo.grad.float(), F32_TOL, "wgrad padded")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, k, coords, act
    (128, 1, 1, 256, 33152, 1, True, "leaky_relu"),     # encoder_0 head 258 -> 33152 (8.55 M weights): skinny GEMM, split-K dgrad
    (4, 128, 128, 256, 10, 3, True, None),              # decoder_visualize logit conv 258 -> P at 128x128, fp32 logits
    (128, 1, 1, 256, 4096, 1, True, None),              # decoder_visualize input nin 258 -> 4 x 4 x 256
])
def test_conv_heads_at_full_width(case, dtype, dev):
    """The fp32-output heads of the full-width CUB graph: forward, input gradient and weight gradient."""
    lib, ops, R = _mods()
    n, h, w, cin, cout, k, coords, act = case
    g = torch.Generator().manual_seed(77 + cout)
    cin_v = cin + (2 if coords else 0)
    T = torch.float32 if dtype == "fp32" else torch.bfloat16
    tol = F32_TOL if dtype == "fp32" else BF16_TOL
    x = torch.randn(n, h, w, cin, generator=g).to(T).float()
    V = torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(cin_v * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    Vo = V.double().clone()
    if dtype == "bf16":
        Vo[:, :, :cin] = V[:, :, :cin].to(T).double()
    xo = x.double().requires_grad_(True)
    Vo.requires_grad_(True)
    bo = b.double().requires_grad_(True)
    yo = _oracle_conv(R, xo, Vo, bo, 1, coords, act, False, None)
    go = torch.randn(yo.shape, generator=g)
    yo.backward(go.double())
    lay = _layer(ops, lib, V, b, k, 1, coords, act, dev)
    xd = x.to(dev, T).requires_grad_(True)
    y = ops.conv(xd, lay, out_f32=True)
    assert y.dtype == torch.float32 and y.shape[-1] == cout
    assert_close(y, yo.float(), tol, "head fwd {}".format(case))
    gx, gV, gb = torch.autograd.grad([y], [xd, lay.V, lay.b], grad_outputs=[go.to(dev)])
    assert_close(gx.float(), xo.grad.float(), tol, "head dgrad {}".format(case))
    assert_close(gV.float(), Vo.grad.float(), tol * (4 if dtype == "bf16" else 1), "head wgrad {}".format(case))
    assert_close(gb.float(), bo.grad.float(), tol, "head bias grad {}".format(case))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bilinear_actmean_maxpool(dtype, dev):
    lib, ops, R = _mods()
    from oracle import np_ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 5, 16, generator=g).to(dtype).float()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    xo = x.double().requires_grad_(True)
    yo = R.bilinear_up2(xo)
    assert np.allclose(yo.detach().numpy(), np_ops.bilinear_up2(x.double().numpy()), atol=1e-12)
    go = torch.randn(yo.shape, generator=g).to(dtype).float()
    yo.backward(go.double())
    xd = x.to(dev, dtype).requires_grad_(True)
    y = ops.BilinearFn.apply(xd)
    assert_close(y.float(), yo.float(), tol, "bilinear fwd")
    (gx,) = torch.autograd.grad([y], [xd], grad_outputs=[go.to(dev, dtype)])
    assert_close(gx.float(), xo.grad.float(), tol, "bilinear bwd")
    # activate + mean
    xo = x.double().requires_grad_(True)
    yo = torch.nn.functional.leaky_relu(xo, 0.2).mean(dim=(1, 2), keepdim=True)
    go = torch.randn(yo.shape, generator=g)
    yo.backward(go.double())
    xd = x.to(dev, dtype).requires_grad_(True)
    y = ops.ActMeanFn.apply(xd, lib.ACT_LRELU, 0.2)
    assert_close(y.float(), yo.float(), tol, "act_mean fwd")
    (gx,) = torch.autograd.grad([y], [xd], grad_outputs=[go.to(dev, dtype)])
    assert_close(gx.float(), xo.grad.float(), 2 * tol if tol > 1e-5 else 1e-5, "act_mean bwd")
    # max pool
    x2 = torch.randn(2, 6, 4, 16, generator=g).to(dtype).float()
    xo = x2.double().requires_grad_(True)
    yo = torch.nn.functional.max_pool2d(xo.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    go = torch.randn(yo.shape, generator=g).to(dtype).float()
    yo.backward(go.double())
    xd = x2.to(dev, dtype).requires_grad_(True)
    y = ops.MaxPoolFn.apply(xd)
    assert_close(y.float(), yo.float(), 1e-6, "maxpool fwd")
    (gx,) = torch.autograd.grad([y], [xd], grad_outputs=[go.to(dev, dtype)])
    assert_close(gx.float(), xo.grad.float(), 1e-6, "maxpool bwd")


def test_l1_and_vgg_preprocess(dev):
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(6)
    a = torch.randn(3, 5, 5, 8, generator=g)
    b = torch.randn(3, 5, 5, 8, generator=g)
    bo = b.double().requires_grad_(True)
    lo = (torch.relu(a.double()[..., :6]) - torch.relu(bo[..., :6])).abs().mean()
    (3.0 * lo).backward()
    bd = b.to(dev).requires_grad_(True)
    l = ops.L1MeanFn.apply(a.to(dev), bd, 6, lib.ACT_RELU)
    assert abs(float(l) - float(lo)) < 1e-6 * max(1.0, abs(float(lo)))
    (gb,) = torch.autograd.grad([3.0 * l], [bd])
    assert_close(gb[..., :6], bo.grad.float()[..., :6], 1e-5, "l1 bwd")
    assert float(gb[..., 6:].abs().max()) == 0.0
    img = torch.rand(2, 4, 4, 3, generator=g) * 2 - 1
    y = ops.VggPreFn.apply(img.to(dev), torch.float32)
    ref = torch.flip((img + 1) * 127.5, dims=[-1]) - torch.tensor(R.VGG_BGR_MEAN)
    assert_close(y[..., :3], ref, 1e-6, "vgg preprocess")
    assert float(y[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("P", [3, 10, 25])
def test_part_path(P, dev):
    lib, ops, R = _mods()
    from oracle import np_ops
    g = torch.Generator().manual_seed(7 + P)
    N, S, gamma, patch = 3, 16, 10.0, 6
    mean = torch.randn(N, S, S, P, generator=g)
    eps = torch.randn(N, S, S, P, generator=g)
    l, m, hard, _ = ops.part_softmax(mean.to(dev), eps.to(dev))
    lo = (mean + eps).double()
    mo = torch.softmax(lo, dim=-1)
    assert_close(l, lo.float(), 1e-6, "l")
    assert_close(m, mo.float(), 1e-5, "softmax")
    assert torch.equal(hard.cpu().argmax(-1), mo.argmax(-1))
    assert float(hard.sum(-1).min()) >= 1.0
    _, soft, _, am = ops.part_softmax(mean.to(dev), None, want_hard=False, want_argmax=True)
    assert torch.equal(am.cpu(), torch.softmax(mean.double(), -1).argmax(-1))
    # rectangle centres
    ho = R.hard_max(mo)
    rect_o, px_o = R.patch_mask(ho, gamma, patch)
    stats = ops.spatial_moments(hard, gamma)
    px = ops.moments_to_px(stats, S)
    mu_o, _ = R.probs_to_mu_sigma(R.spatial_softmax(ho * gamma))
    mu = torch.stack([stats[..., 3] / stats[..., 1], stats[..., 4] / stats[..., 1]], -1)
    assert_close(mu, mu_o.float(), 1e-4, "mu")
    frac = (mu_o * S / 2.0 + S / 2.0)
    stable = ((frac - frac.round()).abs() > 1e-3).all(-1)      # truncation is only compared away from integer boundaries
    assert torch.equal(px.cpu()[stable].long(), px_o[stable])
    rect = ops.draw_rect(px, S, S, patch // 2)
    # px is (row, column) of the box drawn; np_ops.draw_rect(order="yx") paints exactly that
    rect_np = np_ops.draw_rect(px.cpu().numpy().reshape(-1, 2), patch, patch, S, S, order="yx").reshape(N, P, S, S).transpose(0, 2, 3, 1)
    assert np.array_equal(rect.cpu().numpy(), rect_np)
    assert np.array_equal(rect_np[stable.all(-1)], rect_o.numpy()[stable.all(-1)])      # == the oracle's rectangles
    px_yx = ops.moments_to_px(stats, S, "yx")
    assert torch.equal(px_yx.flip(-1), px)
    # masked moments (variance-loss statistics)
    st2 = ops.spatial_moments(m, gamma, rect_px=px, half=patch // 2)
    c1 = R.spatial_softmax(mo * gamma) * (1 - torch.from_numpy(rect_np).double())
    mu2, sig2 = R.probs_to_mu_sigma(c1)
    v_o = sig2[:, :, 0, 0] + sig2[:, :, 1, 1]
    Z = st2[..., 1]
    v = st2[..., 5] / Z - (st2[..., 3] / Z) ** 2 - (st2[..., 4] / Z) ** 2
    assert_close(v, v_o.float(), 1e-4, "variance per part")


def _prior_oracle(R, variant, lm0, eps0, lm1, eps1, px_order, gamma, patch, w, g_hard0, g_hard1, entropy_func, ms_alpha, ms_lambda):
    """The mask priors of Trainer.make_loss_ops in torch fp64 from the oracle's own building blocks (M:652-797, N:1366-1451;
    variant 1: deepfashion/code/SB_model48c/model.py:719-776) + the straight-through term <g_hard, ste(hard, m)> that carries the
    reconstruction gradient.  Returns (dict of the logged forward quantities, total view 0, total view 1, rec-only v0, v1)."""
    B, S, _, P = lm0.shape
    l0, l1 = lm0 + eps0, lm1 + eps1
    m0, m1 = torch.softmax(l0, -1), torch.softmax(l1, -1)
    h0, h1 = R.ste(R.hard_max(m0), m0), R.ste(R.hard_max(m1), m1)
    q = {}
    kl0 = (m0 * torch.log(P * m0 + 1e-20)).sum(-1).mean()
    kl1 = (m1 * torch.log(P * m1 + 1e-20)).sum(-1).mean()
    q["mask0_kl"] = kl0 + kl1
    p_labels = torch.softmax(l0, -1)
    labels = R.ste(R.hard_max(p_labels), p_labels) if (variant == 1 or entropy_func == "cross_entropy") else p_labels
    q["weakly"] = (-(labels * torch.log_softmax(l0, -1)).sum(-1)).mean()
    dy, dx = lm0[:, 1:] - lm0[:, :-1], lm0[:, :, 1:] - lm0[:, :, :-1]
    q["gmrf"] = 0.5 * ((dy ** 2).sum(dim=(1, 2, 3)) + (dx ** 2).sum(dim=(1, 2, 3))).mean()
    sq = lambda t: (t.sum(dim=(1, 2)) ** 2).sum(dim=1).mean()
    if variant == 0:
        rect0, _ = R.patch_mask(h0, gamma, patch, px_order)
        rect1, _ = R.patch_mask(h1, gamma, patch, px_order)
        g = R.squared_grad(m0)
        r = torch.clamp(g, max=1.0e-2)
        q["ms"], q["area"] = sq(r), sq(m0)
        q["smooth"] = sq(torch.where(g < 1.0e-2, r, torch.zeros_like(r)))
        q["contour"] = sq(torch.where(g >= 1.0e-2, r, torch.zeros_like(r)))
        q["patch"] = (h0 * (1 - rect0).detach()).sum(dim=(1, 2, 3)).mean()
        c1 = R.spatial_softmax(torch.softmax(l1, -1) * gamma) * (1 - rect1).detach()
        _, sig = R.probs_to_mu_sigma(c1)
        q["var"] = (sig[:, :, 0, 0] + sig[:, :, 1, 1]).sum(dim=1).mean()
        t0 = (w["kl"] * kl0 + w["entropy"] * q["weakly"] + w["ms"] * q["ms"] + w["area"] * q["area"] + w["patch"] * q["patch"]
              + w["gmrf"] * q["gmrf"])
    else:
        q["msl"] = torch.clamp(ms_alpha * R.squared_grad(lm0), max=ms_lambda).sum(dim=(1, 2, 3)).mean()
        c1 = R.spatial_softmax(torch.softmax(l1, -1))
        c1 = c1 / c1.sum(dim=(1, 2), keepdim=True)
        _, sig = R.probs_to_mu_sigma(c1)
        q["var"] = (sig[:, :, 0, 0] ** 2 + sig[:, :, 1, 1] ** 2).sum(dim=1).mean()
        t0 = w["kl"] * kl0 + w["entropy"] * q["weakly"] + w["gmrf"] * q["gmrf"] + w["msl"] * q["msl"]
    t1 = w["kl"] * kl1 + w["var"] * q["var"]
    r0, r1 = (g_hard0 * h0).sum(), (g_hard1 * h1).sum()
    return q, t0 + r0, t1 + r1, r0, r1


PRIOR_CASES = [(0, 10, 32, "entropy"), (0, 25, 32, "cross_entropy"), (0, 10, 20, "entropy"),
               (1, 10, 32, "cross_entropy"), (1, 25, 32, "cross_entropy"), (0, 3, 48, "entropy"),
               # the pixel-per-lane forward of round 5 (P = 10 at 128- / 256-wide images)
               (0, 10, 128, "entropy"), (1, 10, 128, "cross_entropy"), (0, 10, 256, "cross_entropy"),
               # the direct-from-global forms of round 6 (P = 16 / 20 / 25 at 128- / 256-wide images)
               (0, 16, 128, "entropy"), (1, 20, 128, "cross_entropy"), (0, 25, 128, "cross_entropy"),
               (0, 20, 256, "entropy"), (1, 16, 256, "cross_entropy"),
               # (round 6, late: those shapes run chunk-per-lane -- five chunks of five parts per pixel at P = 25, twelve pixels per wave)
               (1, 25, 128, "entropy"), (0, 25, 256, "entropy")]
# px_bpi (UPS_PRIOR_PX_BPI: blocks per image of the pixel-per-lane kernels) only exists for the shapes those kernels take: the
# multi-tile variants are generated for exactly those cases (until round 6 they were generated for every case and skipped)
PRIOR_PARAMS = [c + (0,) for c in PRIOR_CASES] + [c + (b,) for c in PRIOR_CASES if c[1] == 10 and c[2] in (128, 256) for b in (4, 1)]


@pytest.mark.parametrize("variant,P,S,entropy_func,px_bpi", PRIOR_PARAMS)
def test_mask_priors_forward_and_backward(variant, P, S, entropy_func, px_bpi, dev, monkeypatch):
    """ups_prior_fwd / ups_prior_bwd ALONE (8a-12; until round 4 only covered through the whole-step tests): every logged prior
    and the fused analytic d/d logits -- total (`dl`, the decoder_visualize key) and reconstruction-only (`dl_rec`, what
    encoder_0 sees) -- for both views and both model variants against torch-fp64 autograd over the oracle's restatement, with
    non-trivial weights on EVERY term (the shipped schedules make area / Mumford-Shah ~1e-5 of the total, where an error in
    their backward would hide).  The forward sums are cross-checked against the independent NumPy restatement too."""
    import types
    lib, ops, R = _mods()
    from oracle import np_ops
    from upsparts_amd.model import Trainer
    # px_bpi (UPS_PRIOR_PX_BPI): blocks per image of the pixel-per-lane kernels -- 0: the launcher's choice (one tile per block for
    # three images), 4 / 1: 16 / 64 tiles per block at 128 x 128, the multi-tile loops of the 64-image benchmark shape
    if px_bpi:
        monkeypatch.setenv("UPS_PRIOR_PX_BPI", str(px_bpi))
    g = torch.Generator().manual_seed(100 * variant + P + S)
    B, gamma, patch = 3, 10.0, 8
    # logits with spatial structure (so that rectangles, Mumford-Shah contours and moments are non-degenerate) + unit noise
    low = torch.randn(2 * B, P, S // 4, S // 4, generator=g, dtype=torch.float64)
    lm = 2.0 * torch.nn.functional.interpolate(low, size=(S, S), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    eps = torch.randn(2 * B, S, S, P, generator=g, dtype=torch.float64)
    g_hard = torch.randn(2 * B, S, S, P, generator=g, dtype=torch.float64)
    w = {"kl": 0.7, "entropy": 1.3, "ms": 0.05, "area": 2.0e-3, "patch": 0.02, "gmrf": 0.3, "var": 1.7, "msl": 0.4}
    if variant == 1:       # SB_model48c has no Mumford-Shah-on-masks / area / patch terms (its trainer passes zero weights, DF:830-838)
        w.update({"ms": 0.0, "area": 0.0, "patch": 0.0})
    ms_alpha, ms_lambda = (1.0, 1.0e-2) if variant == 0 else (1.5, 0.05)
    lm0 = lm[:B].clone().requires_grad_(True)
    lm1 = lm[B:].clone().requires_grad_(True)
    q, t0, t1, r0, r1 = _prior_oracle(R, variant, lm0, eps[:B], lm1, eps[B:], "xy", gamma, patch, w, g_hard[:B], g_hard[B:],
                                      entropy_func, ms_alpha, ms_lambda)
    d_tot0, = torch.autograd.grad(t0, lm0, retain_graph=True)
    d_tot1, = torch.autograd.grad(t1, lm1, retain_graph=True)
    d_rec0, = torch.autograd.grad(r0, lm0, retain_graph=True)
    d_rec1, = torch.autograd.grad(r1, lm1)

    # ---- HIP: the product's own descriptor filling (Trainer._prior) on a stand-in trainer
    cfg = {"entropy_func": entropy_func, "gamma": gamma,
           "mumford_sha_alpha": {"var_type": "staircase", "options": {"start": 0, "start_value": ms_alpha, "step_size": 1, "stair_factor": 1.0,
                                                                      "clip_min": ms_alpha, "clip_max": ms_alpha}},
           "mumford_sha_lambda": {"var_type": "staircase", "options": {"start": 0, "start_value": ms_lambda, "step_size": 1, "stair_factor": 1.0,
                                                                       "clip_min": ms_lambda, "clip_max": ms_lambda}}}
    fake = types.SimpleNamespace(model=types.SimpleNamespace(patch_size=patch, df=variant == 1), config=cfg, global_step=0)
    lmd, epsd = lm.float().to(dev), eps.float().to(dev)
    if variant == 0:
        l, m, hard, _, _bits, hstats = ops.part_softmax(lmd, epsd, want_bits=True, moments_gamma=gamma)
        if hstats is None:
            hstats = ops.spatial_moments(hard, gamma)
        px = ops.moments_to_px(hstats, S, "xy")
        px0, px1 = px[:B].contiguous(), px[B:].contiguous()
    else:
        l, m, hard, _ = ops.part_softmax(lmd, epsd)
        px0 = px1 = None
    nfl = lib.load().ups_prior_sums_floats(B, P)
    sums0 = torch.empty(nfl, dtype=torch.float32, device=dev)
    sums1 = torch.empty(nfl, dtype=torch.float32, device=dev)
    per_np0 = torch.empty((B, P, 8), dtype=torch.float32, device=dev)
    l0, l1, m0, m1 = l[:B].contiguous(), l[B:].contiguous(), m[:B].contiguous(), m[B:].contiguous()
    h0 = hard[:B].contiguous()
    Trainer._prior(fake, 0, B, S, P, l0, lmd[:B].contiguous(), m0, h0, px0, per_np0, sums0, w)
    Trainer._prior(fake, 1, B, S, P, l1, None, m1, None, px1, None, sums1, w)
    stats_v = ops.spatial_moments(m1, 1.0 if variant == 1 else gamma, rect_px=px1, half=patch // 2)
    # the product path takes view 1's KL from the moments pass (ups_spatial_moments_kl): same statistics, same sum
    sums1b = torch.empty(nfl, dtype=torch.float32, device=dev)
    stats_vb = ops.spatial_moments(m1, 1.0 if variant == 1 else gamma, rect_px=px1, half=patch // 2, kl_sums=sums1b)
    assert torch.equal(stats_vb, stats_v)
    assert abs(float(sums1b[0]) - float(sums1[0])) <= 1e-5 * abs(float(sums1[0])) and float(sums1b[1:16].abs().max()) == 0.0
    dl_tot = torch.empty_like(lmd)
    dl_rec = torch.empty_like(lmd)
    gh = g_hard.float().to(dev)
    Trainer._prior(fake, 0, B, S, P, l0, lmd[:B].contiguous(), m0, h0, px0, per_np0, sums0, w, gh[:B].contiguous(), dl_tot[:B],
                   bwd=True, dl_rec=dl_rec[:B])
    Trainer._prior(fake, 1, B, S, P, l1, None, m1, None, px1, stats_v, sums1, w, gh[B:].contiguous(), dl_tot[B:], bwd=True,
                   dl_rec=dl_rec[B:])

    # ---- forward quantities, exactly as Trainer.build_logs derives them from the sums
    npx = float(B * S * S)
    s0, s1 = sums0.double().cpu(), sums1.double().cpu()
    got = {"mask0_kl": (s0[0] + s1[0]) / npx, "weakly": s0[1] / npx, "gmrf": s0[3] / B}
    Zs = stats_v.double().cpu()[..., 1]
    sv = stats_v.double().cpu()
    if variant == 0:
        got.update({"patch": s0[2] / B, "ms": s0[4] / B, "area": s0[5] / B, "smooth": s0[6] / B, "contour": s0[7] / B,
                    "var": (sv[..., 5] / Zs - (sv[..., 3] / Zs) ** 2 - (sv[..., 4] / Zs) ** 2).sum(dim=1).mean()})
    else:
        s00 = sv[..., 6] / Zs - (sv[..., 3] / Zs) ** 2
        s11 = (sv[..., 5] - sv[..., 6]) / Zs - (sv[..., 4] / Zs) ** 2
        got.update({"msl": s0[2] / B, "var": (s00 ** 2 + s11 ** 2).sum(dim=1).mean()})
    for k, v in got.items():
        ref = float(q[k])
        assert abs(float(v) - ref) <= 2e-4 * max(abs(ref), 1e-6), "prior {} (variant {}, P {}): oracle {} hip {}".format(k, variant, P, ref, float(v))
    # independent NumPy restatement of the forward terms (oracle/np_ops.py)
    m0n = torch.softmax(lm[:B] + eps[:B], -1).numpy()
    m1n = torch.softmax(lm[B:] + eps[B:], -1).numpy()
    assert abs(np_ops.categorical_kl(m0n) + np_ops.categorical_kl(m1n) - float(q["mask0_kl"])) <= 1e-9 * max(1.0, abs(float(q["mask0_kl"])))
    assert abs(np_ops.kl_improper_gmrf(lm[:B].numpy()) - float(q["gmrf"])) <= 1e-9 * abs(float(q["gmrf"]))
    if variant == 0:
        r_np = np_ops.mumford_shah(m0n, 1.0, 1.0e-2)
        r_np = r_np[0] if isinstance(r_np, tuple) else r_np
        assert abs(float(((r_np.sum(axis=(1, 2)) ** 2).sum(axis=1)).mean()) - float(q["ms"])) <= 1e-9 * abs(float(q["ms"]))

    # ---- backward: d total / d logits (decoder_visualize) and d rec / d logits (encoder_0)
    assert_close(dl_tot[:B], d_tot0.float(), 1e-3, "prior_bwd view 0 dl (variant {}, P {})".format(variant, P))
    assert_close(dl_tot[B:], d_tot1.float(), 1e-3, "prior_bwd view 1 dl")
    assert_close(dl_rec[:B], d_rec0.float(), 1e-3, "prior_bwd view 0 dl_rec")
    assert_close(dl_rec[B:], d_rec1.float(), 1e-3, "prior_bwd view 1 dl_rec")


def test_randn_philox_stream(dev):
    """ups_randn: Philox4x32-10 + Box-Muller.  Known answer of the generator (Random123's test vector: counter 0, key 0 ->
    6627e8d5 e169c58d bc57ac4c 9b00dbd8), the stream property (one call over n values == two calls over the halves), seed
    sensitivity, and the first four moments of 4 M draws."""
    lib, ops, R = _mods()
    ns = ops.NoiseStream(0)
    z = ns.randn(8, device=dev).cpu().double()
    words = [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    exp = []
    for h in range(2):
        u1 = ((words[2 * h] >> 8) + 0.5) / 16777216.0
        u2 = ((words[2 * h + 1] >> 8) + 0.5) / 16777216.0
        rad = math.sqrt(-2.0 * math.log(u1))
        exp += [rad * math.cos(2 * math.pi * u2), rad * math.sin(2 * math.pi * u2)]
    assert float((z[:4] - torch.tensor(exp, dtype=torch.float64)).abs().max()) <= 2e-5, (z[:4].tolist(), exp)
    a = ops.NoiseStream(1234)
    whole = a.randn(1 << 20, device=dev)
    b = ops.NoiseStream(1234)
    halves = torch.cat([b.randn(1 << 19, device=dev), b.randn(1 << 19, device=dev)])
    assert torch.equal(whole, halves) and a.offset == b.offset == (1 << 18)
    other = ops.NoiseStream(1235).randn(1 << 20, device=dev)
    assert float((other == whole).float().mean()) < 1e-3
    x = ops.NoiseStream(99).randn(1 << 22, device=dev).double()
    m, v = float(x.mean()), float(x.var())
    sk = float(((x - m) ** 3).mean() / v ** 1.5)
    ku = float(((x - m) ** 4).mean() / v ** 2)
    assert abs(m) <= 3e-3 and abs(v - 1.0) <= 5e-3 and abs(sk) <= 1e-2 and abs(ku - 3.0) <= 3e-2, (m, v, sk, ku)
    assert bool(torch.isfinite(x).all()) and float(x.abs().max()) < 6.5


@pytest.mark.parametrize("dtype,B", [(torch.float32, 8), (torch.bfloat16, 64), (torch.float32, 3)])
def test_critic_head(dtype, B, dev):
    """ups_critic_head_fwd / _bwd against the reference's formulation (M:159-173 last line: logit = sum(h_pi * h_alpha);
    M:524-529 logit_loss = mean softplus(-+logit); M:821-826 accuracy; M:532-536, 855 mean joint logit): values, and the gradients
    w.r.t. both embeddings for the two upstream gradients the trainer sends (d loss -- the critic's own key -- and d mean-joint --
    the adversarial term of encoder_0)."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(17 + B)
    K = 512
    hp = (torch.randn(2 * B, 1, 1, K, generator=g) * 0.2).to(dtype)
    ha = (torch.randn(2 * B, 1, 1, K, generator=g) * 0.2).to(dtype)
    hpo = hp.double().requires_grad_(True)
    hao = ha.double().requires_grad_(True)
    logits = (hpo * hao).sum(dim=(1, 2, 3))
    joint, marg = logits[:B], logits[B:]
    loss_o = 0.5 * (torch.nn.functional.softplus(-joint).mean() + torch.nn.functional.softplus(marg).mean())
    acc_o = ((joint > 0).sum() + (marg < 0).sum()).double() / (2 * B)
    mim_o = joint.mean()
    hpd = hp.to(dev).requires_grad_(True)
    had = ha.to(dev).requires_grad_(True)
    loss, acc, mim = ops.CriticHeadFn.apply(hpd, had, B, K)
    tol = 1e-5 if dtype == torch.float32 else 1e-5          # (inputs are the same rounded values; accumulation is fp32)
    assert abs(float(loss) - float(loss_o)) <= tol * max(1.0, abs(float(loss_o)))
    assert abs(float(mim) - float(mim_o)) <= tol * max(1.0, abs(float(mim_o)))
    assert abs(float(acc) - float(acc_o)) <= 1e-6
    gtol = 1e-5 if dtype == torch.float32 else 1e-2
    for wl, wm in ((1.0, 0.0), (0.0, 1.0), (0.7, -2.5)):
        go = torch.autograd.grad(wl * loss_o + wm * mim_o, [hpo, hao], retain_graph=True)
        gh = torch.autograd.grad(wl * loss + wm * mim, [hpd, had], retain_graph=True)
        assert_close(gh[0].float(), go[0].float(), gtol, "d / d h_pi ({}, {})".format(wl, wm))
        assert_close(gh[1].float(), go[1].float(), gtol, "d / d h_alpha ({}, {})".format(wl, wm))


def test_state_update_kernel(dev):
    """ups_state_update against the scalar expressions it replaces (M:28-35 EMA, M:890-909 loa, M:921-930 lor): bit-identical to
    the fp32 torch ops, with either multiplier frozen, and clipped where the reference clips."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(3)
    for trial in range(6):
        stats = (torch.randn(6, generator=g) * (3.0 if trial % 2 else 0.3)).to(dev)
        old = (torch.randn(9, generator=g).abs() * (5.0 if trial >= 3 else 1.0)).to(dev)
        up_loa, up_lor = trial % 3 != 1, trial % 3 != 2
        loa_lr, loa_t, lor_lr, lor_t, lo, hi = 4.0, (1.0 - 0.05) * 0.125, 0.05, 0.125, 1.0, 7.5
        out = torch.empty(9, device=dev)
        lib.call("ups_state_update", lib.ptr(stats), lib.ptr(old), lib.ptr(out), 0.99, 1.0 - 0.99, int(up_loa), loa_lr, loa_t,
                 int(up_lor), lor_lr, lor_t, lo, hi, lib.stream())
        mim, ind, a0, a1, l0, l1 = stats.unbind(0)
        ema = lambda o, v: 0.99 * o + (1.0 - 0.99) * v
        want = [ema(old[0], a0), ema(old[1], a1), ema(old[2], a1 - a0), ema(old[3], l0), ema(old[4], l1), ema(old[5], mim), ema(old[6], ind),
                torch.clamp(old[7] + loa_lr * (mim - loa_t), min=0.0) if up_loa else old[7],
                torch.clamp(old[8] + lor_lr * (ind - lor_t), lo, hi) if up_lor else old[8]]
        assert torch.equal(out, torch.stack(want)), (trial, out, torch.stack(want))


@pytest.mark.parametrize("M", [128, 6, 40])
def test_critic_towers_grouped(M, dev):
    """ops.TowersFn (ups_towers_fwd / _bwd: the same layer of every tower in one launch, every weight / bias gradient in one) against
    discriminator_model's tower (M:159-173: nin, 4 x residual_block(k = 1) = x + nin(lrelu(x)), nin(lrelu(x))) in fp64, and against
    the generic convolution path on the SAME layer objects (the storage form is the same, so the two differ by summation order
    only): outputs, the gradients of every kernel and bias, the input gradient of the 256-wide towers; a backward call that is given
    one tower's output gradient only (the adversarial term's) must leave the other towers out; ragged row counts."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(5 + M)
    widths, D, Ln = (256, 64, 256), 512, 6
    towers, ref = [], []
    for t, c in enumerate(widths):
        tw, rw = [], []
        for l in range(Ln):
            k = c if l == 0 else D
            bound = math.sqrt(1.0 / k)
            V = (torch.rand(1, 1, k, D, generator=g) * 2 - 1) * bound * (1.0 if l in (0, Ln - 1) else 0.5)
            b = (torch.rand(D, generator=g) * 2 - 1) * bound
            lay = ops.ConvLayer("t{}/conv2d_{}".format(t, l), V.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), 1, 1, False,
                                None if l == 0 else "leaky_relu")
            lay.in_post = l > 0
            lay.out_act = lib.ACT_LRELU if l < Ln - 1 else lib.ACT_NONE
            lay.grad_V, lay.grad_b = torch.zeros_like(lay.V), torch.zeros_like(lay.b)
            tw.append(lay)
            rw.append((V.double().requires_grad_(True), b.double().requires_grad_(True)))
        towers.append(tw); ref.append(rw)
    xs = [(torch.randn(M, 1, 1, c, generator=g)).to(torch.bfloat16) for c in widths]
    gos = [(torch.randn(M, 1, 1, D, generator=g) * 0.1).to(torch.bfloat16) for _ in widths]
    lrelu = lambda v: torch.nn.functional.leaky_relu(v, 0.2)

    # fp64 reference on the kernels' operands: bf16 kernels, and every stored tensor (lrelu(x), from which the residual x is
    # recovered) rounded to bf16 with a straight-through gradient
    xo = [x.double().requires_grad_(True) for x in xs]
    outs_o = []
    stored = lambda v: v + (v.detach().to(torch.bfloat16).double() - v.detach())
    wq = lambda v: v + (v.detach().to(torch.bfloat16).double() - v.detach())
    for t, rw in enumerate(ref):
        hs_ = stored(lrelu(xo[t].view(M, -1) @ wq(rw[0][0][0, 0]) + rw[0][1]))
        for l in range(1, Ln - 1):
            x_rec = torch.where(hs_ > 0, hs_, hs_ / 0.2)
            hs_ = stored(lrelu(x_rec + hs_ @ wq(rw[l][0][0, 0]) + rw[l][1]))
        outs_o.append(hs_ @ wq(rw[Ln - 1][0][0, 0]) + rw[Ln - 1][1])
    loss_o = sum((o * go.double().view(M, -1)).sum() for o, go in zip(outs_o, gos))
    par_o = [t_ for rw in ref for vb in rw for t_ in vb]
    gr_o = torch.autograd.grad(loss_o, par_o + [xo[0], xo[2]])

    # grouped launches
    assert ops.towers_eligible(towers, [x.to(dev) for x in xs])
    xd = [x.to(dev).requires_grad_(c == 256) for x, c in zip(xs, widths)]
    params = [t_ for tw in towers for lay in tw for t_ in (lay.V, lay.b)]
    hs = ops.TowersFn.apply(towers, *(xd + params))
    for t in range(3):
        assert_close(hs[t].float().view(M, -1), outs_o[t].float(), BF16_TOL, "tower {} output".format(t))
    loss = sum((h.float() * go.to(dev).float()).sum() for h, go in zip(hs, gos))
    gr = torch.autograd.grad(loss, params + [xd[0], xd[2]], retain_graph=True)
    gr = [t_.clone() for t_ in gr]
    for i, (a, b_) in enumerate(zip(gr, gr_o)):
        what = "d x" if i >= len(params) else "tower {} layer {} {}".format(i // (2 * Ln), (i // 2) % Ln, "b" if i % 2 else "V")
        # (the gradient is stored in bf16 after every layer, up to five roundings deep at the first one: 3e-2 here, and with
        # few rows (M = 6, 40), where nothing averages them, the max-norm and RMS bars only -- the element-wise check of that case is the
        # one against the generic path below, which rounds alike)
        assert_close(a.float().reshape(b_.shape), b_.float(), 1.5 * BF16_TOL, what, elementwise=M >= 128)
    # one tower's output gradient only, input gradients only (the adversarial term): the others' buffers stay untouched
    for tw in towers:
        for lay in tw:
            lay.grad_V.fill_(7.0); lay.grad_b.fill_(7.0)
    with ops.skip_wgrad():
        gx0, = torch.autograd.grad((hs[0].float() * gos[0].to(dev).float()).sum(), [xd[0]], retain_graph=True)
    assert_close(gx0.float().reshape(gr_o[-2].shape), gr_o[-2].float(), 1.5 * BF16_TOL, "d x of tower 0 alone", elementwise=M >= 128)
    assert all(bool((lay.grad_V == 7.0).all()) and bool((lay.grad_b == 7.0).all()) for tw in towers for lay in tw)
    # the critic whose key trains: towers 1 and 2 only
    par12 = [t_ for tw in towers[1:] for lay in tw for t_ in (lay.V, lay.b)]
    torch.autograd.grad(sum((h.float() * go.to(dev).float()).sum() for h, go in zip(hs[1:], gos[1:])), par12)
    assert all(bool((lay.grad_V == 7.0).all()) for lay in towers[0])
    for i, lay in enumerate(lay for tw in towers[1:] for lay in tw):
        assert_close(lay.grad_V.float(), gr[2 * Ln + 2 * i].float(), 1e-6, "repeat of layer {}".format(i))

    # the generic path on the same layers
    outs_g = []
    xg = [x.to(dev).requires_grad_(c == 256) for x, c in zip(xs, widths)]
    for t, tw in enumerate(towers):
        h = ops.conv(xg[t], tw[0])
        for l in range(1, Ln - 1):
            h = ops.conv(h, tw[l], res_self=True)
        outs_g.append(ops.conv(h, tw[Ln - 1]))
    for t in range(3):
        assert_close(hs[t].float(), outs_g[t].float(), BF16_TOL, "tower {} output vs the generic path".format(t))
    loss_g = sum((h.float() * go.to(dev).float()).sum() for h, go in zip(outs_g, gos))
    gg = torch.autograd.grad(loss_g, params + [xg[0], xg[2]])
    for i, (a, b_) in enumerate(zip(gr, gg)):
        assert_close(a.float(), b_.float().reshape(a.shape), BF16_TOL, "gradient {} vs the generic path".format(i))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_parts_unpool(dtype, dev):
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(11)
    B, S, P, F = 2, 8, 5, 16
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    view = torch.rand(B, S, S, 3, generator=g) * 2 - 1
    hard = torch.nn.functional.one_hot(torch.randint(0, P, (B, S, S), generator=g), P).float()
    hard[0, 0, 0] = torch.tensor([1.0, 1.0, 0, 0, 0])           # a tie
    feat = torch.randn(B, P, F, generator=g)
    ho = hard.double().requires_grad_(True)
    parts = (view.double().unsqueeze(3) * ho.unsqueeze(4)).permute(3, 0, 1, 2, 4).reshape(P * B, S, S, 3)
    gp = torch.randn(parts.shape, generator=g).to(dtype).float()
    parts.backward(gp.double())
    hd = hard.to(dev).requires_grad_(True)
    out = ops.MaskPartsFn.apply(view.to(dev), hd, dtype)
    assert_close(out[..., :3].float(), parts.float(), tol, "mask_parts fwd")
    assert float(out[..., 3:].float().abs().max()) == 0.0
    gd = torch.zeros(out.shape, dtype=dtype, device=dev)
    gd[..., :3] = gp.to(dev, dtype)
    (gh,) = torch.autograd.grad([out], [hd], grad_outputs=[gd])
    assert_close(gh, ho.grad.float(), tol, "mask_parts bwd")
    # unpool
    ho = hard.double().requires_grad_(True)
    fo = feat.double().requires_grad_(True)
    inj = torch.cat([torch.einsum("bhwp,bpf->bhwf", ho, fo), ho], dim=3)
    gi = torch.randn(inj.shape, generator=g).to(dtype).float()
    inj.backward(gi.double())
    hd = hard.to(dev).requires_grad_(True)
    fd = feat.to(dev).requires_grad_(True)
    out = ops.UnpoolFn.apply(hd, fd, dtype)
    ld = ops.round8(F + P)
    assert out.shape[-1] == ld
    assert_close(out[..., :F + P].float(), inj.float(), tol, "unpool fwd")
    gd = torch.zeros(out.shape, dtype=dtype, device=dev)
    gd[..., :F + P] = gi.to(dev, dtype)
    gh, gf = torch.autograd.grad([out], [hd, fd], grad_outputs=[gd])
    assert_close(gh, ho.grad.float(), tol, "unpool d hard")
    assert_close(gf, fo.grad.float(), tol, "unpool d feat")


@pytest.mark.parametrize("B,S,P", [(3, 16, 10), (2, 24, 25), (1, 10, 16), (2, 32, 20), (1, 7, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_parts_forward_over_tiles(B, S, P, dtype, dev):
    """mask_parts forward (model.py:176-187 + the part-major transpose, nn.py:97-103) over whole, partial and unaligned 256-pixel tiles
    at the part counts of the BASELINE configs and the shipped yamls: since round 6 the hard tile is staged through LDS (tile.h)."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(B * 100 + S + P)
    view = torch.rand(B, S, S, 3, generator=g) * 2 - 1
    hard = torch.nn.functional.one_hot(torch.randint(0, P, (B, S, S), generator=g), P).float()
    hard[0, 0, 0] = torch.randn(P, generator=g)               # an arbitrary fp32 row
    parts = (view.unsqueeze(3) * hard.unsqueeze(4)).permute(3, 0, 1, 2, 4).reshape(P * B, S, S, 3)
    out = ops.MaskPartsFn.apply(view.to(dev), hard.to(dev), dtype)
    assert out.shape == (P * B, S, S, 8)
    assert_close(out[..., :3].float(), parts.to(dtype).float(), 1e-6 if dtype == torch.float32 else 1e-2, "mask_parts fwd")
    assert float(out[..., 3:].float().abs().max()) == 0.0


def test_latent(dev):
    lib, ops, R = _mods()
    from oracle import np_ops
    g = torch.Generator().manual_seed(13)
    for Z, B, S in ((8, 3, 7), (256, 2, 7), (40, 2, 1)):
        NP = Z + Z * (Z + 1) // 2
        params = torch.randn(B, NP, generator=g) * 0.3
        eps = torch.randn(S, B, Z, generator=g)
        levels = [1.0, 0.7, 0.7, 1.0, 1.0, 1.0, 1.0][:S]
        po = params.double().requires_grad_(True)
        d = R.FullLatent(po, Z)
        so = torch.stack([d.sample(eps[s].double(), noise_level=levels[s]).reshape(B, Z) for s in range(S)])
        klo = d.kl()
        mean_np, L_np, ld_np = np_ops.full_latent(params.double().numpy(), Z)
        assert abs(np_ops.full_latent_kl(mean_np, L_np, ld_np) - float(klo)) < 1e-9 * max(1, abs(float(klo)))
        gs = torch.randn(S, B, Z, generator=g)
        w_kl = 1.7
        ((so * gs.double()).sum() + w_kl * klo).backward()
        samples, kl_rows = ops.latent_fwd(params.to(dev), eps.to(dev), levels, True)
        assert_close(samples, so.float(), 2e-5, "latent samples Z={}".format(Z))
        assert abs(float(kl_rows.sum(1).mean()) - float(klo)) < 1e-4 * max(1.0, abs(float(klo)))
        scale = torch.tensor([w_kl], device=dev)
        gp = ops.latent_bwd(params.to(dev), eps.to(dev), levels, gs.to(dev), scale, 1.0 / B)
        assert_close(gp, po.grad.float(), 5e-5, "latent bwd Z={}".format(Z))


def test_adam_and_gauss(dev):
    lib, ops, R = _mods()
    from oracle import np_ops
    g = torch.Generator().manual_seed(17)
    p = torch.randn(1000, generator=g); gr = torch.randn(1000, generator=g)
    m = torch.randn(1000, generator=g) * 0.1; v = torch.rand(1000, generator=g) * 0.1
    t, lr, b1, b2 = 3, 2e-4, 0.5, 0.9
    pn, mn, vn = np_ops.tf_adam_step(p.double().numpy(), gr.double().numpy(), m.double().numpy(), v.double().numpy(), t, lr, b1, b2)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    ops.adam_step(pd, gr.to(dev), md, vd, lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t), b1, b2, 1e-8)
    assert np.allclose(pd.cpu().numpy(), pn, rtol=1e-5, atol=1e-7)
    assert np.allclose(md.cpu().numpy(), mn, rtol=1e-5, atol=1e-7)
    assert np.allclose(vd.cpu().numpy(), vn, rtol=1e-5, atol=1e-7)
    pts = torch.rand(2, 3, 2, generator=g) * 10
    sd = torch.rand(2, 3, 2, generator=g) * 2 + 1
    hm = ops.gauss_hm(pts.to(dev), sd.to(dev), 12, 10)
    assert np.allclose(hm.cpu().numpy(), np_ops.tf_hm(pts.numpy(), 12, 10, sd.numpy()), rtol=1e-4, atol=1e-6)
    mu = torch.rand(2, 3, 2, generator=g) - 0.5
    Lt = torch.zeros(2, 3, 2, 2)
    Lt[..., 0, 0] = 0.3 + torch.rand(2, 3, generator=g); Lt[..., 1, 1] = 0.3 + torch.rand(2, 3, generator=g)
    Lt[..., 1, 0] = torch.randn(2, 3, generator=g) * 0.2
    hm3 = ops.gauss_hm3(mu.to(dev), Lt.to(dev), 9, 11)
    assert np.allclose(hm3.cpu().numpy(), np_ops.tf_hm3(9, 11, mu.numpy(), Lt.numpy()), rtol=1e-4, atol=1e-6)


def test_tps_warp_matches_oracle(dev):
    """ups_tps_warp + the host-side parameter / solve code (upsparts_amd.tps) against oracle/tps.py on identical uniforms."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd import tps as PT
    from oracle import tps as OT
    g = torch.Generator().manual_seed(11)
    P = dict(scal=0.8, tps_scal=0.15, rot_scal=0.2, off_scal=0.2, scal_var=0.1, augm_scal=1.0)
    B, S = 3, 40
    views = [torch.rand(B, S, S, 3, generator=g) * 2 - 1 for _ in range(3)]
    u = torch.rand(2 * B, OT.N_UNIFORMS, generator=g)
    want = OT.make_tps([v.double() for v in views], u.double(), P)
    got = PT.make_tps([v.to(dev) for v in views], P, uniforms=u.to(dev))
    for a, b, name in zip(got, want, ("view0", "view1", "target")):
        assert_close(a.float().cpu(), b.float(), 2e-4, "tps " + name)
    c, v = PT.make_input_tps_param(PT.tps_parameters(2 * B, uniforms=u.to(dev), **P))
    co, vo = OT.make_input_tps_param(OT.uniforms_to_params(u.double(), **P))
    assert_close(PT.solve_system(c, v).cpu(), OT.solve_system(co, vo).float(), 1e-5, "tps solve")


def test_integration_stub_runs(dev):
    """INTEGRATION.md section 2, executed as written: weight_prep + residual_block_conv through raw pointers."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host import _integration_stub
    lib, ops, R = _mods()
    ns = {}
    exec(_integration_stub(), ns)
    g = torch.Generator().manual_seed(11)
    n, h, w, c = 2, 32, 32, 64
    x = torch.randn(n, h, w, c, generator=g).bfloat16()
    V = (torch.randn(3, 3, c, c, generator=g) / math.sqrt(9 * c))
    b = torch.randn(c, generator=g) * 0.1
    xd, Vd, bd = x.to(dev), V.to(dev), b.to(dev)
    wq = torch.empty(9 * ((c + 31) // 32) * c * 32, dtype=torch.bfloat16, device=dev)
    out = torch.empty_like(xd)
    st = torch.cuda.current_stream().cuda_stream
    ns["weight_prep"](Vd.data_ptr(), wq.data_ptr(), c, c, st)
    ns["residual_block_conv"](xd.data_ptr(), wq.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, w, c, st)
    torch.cuda.synchronize()
    xo = x.double()
    yo = xo + R.conv2d_same(torch.nn.functional.leaky_relu(xo, 0.2), V.bfloat16().double(), b.double(), 1)
    assert_close(out.float(), yo.float(), BF16_TOL, "INTEGRATION.md stub")


@pytest.mark.parametrize("case", [
    # n, h, w, cout, coords, parts, post
    (4, 64, 48, 32, True, 10, True),         # encoder_1's first convolution: part-masked, CoordConv, stored as lrelu(y)
    (5, 32, 64, 64, False, 0, False),        # VGG block1_conv1 form: 3 -> 64
    (3, 16, 16, 24, True, 0, False),         # all-border tile, ragged channel tail
    (1, 48, 32, 64, True, 32, True),         # the most parts a mask word holds, 64 output channels
    # (round 6: strips of 16 columns x 64 / 32 rows, the fragments of the next four rows in flight; the CoordConv class of a first / last
    # column is the lane's own, only the first and the last ROW take the table path)
    (2, 128, 32, 32, True, 0, True),         # 16 rows per wave: encoder_0's first convolution form, both row borders in one strip's waves
    (2, 64, 48, 64, False, 0, False),        # 16 rows per wave, 64 output channels (VGG block1_conv1 form)
    (2, 64, 32, 32, True, 5, False),         # part-masked at 8 rows per wave, two strips per column
])
def test_first_layer_kernel(case, dev, monkeypatch):
    """conv3x3_first.hip (im2col in the MFMA fragment addressing, every part image of a tile from one read of the view) against
    the patch kernel on the same operands and against the fp64 oracle on the materialised part images."""
    lib, ops, R = _mods()
    n, h, w, cout, coords, parts, post = case
    g0 = torch.Generator().manual_seed(5)
    cv = 3 + (2 if coords else 0)
    V = torch.randn(3, 3, cv, cout, generator=g0) / math.sqrt(9 * cv)
    b = torch.randn(cout, generator=g0) * 0.1
    x = torch.zeros(n, h, w, 8, dtype=torch.bfloat16)
    x[..., :3] = torch.randn(n, h, w, 3, generator=g0).to(torch.bfloat16)
    mask, hard = None, None
    if parts:
        owner = torch.randint(0, parts, (n, h, w), generator=g0)
        bits = (1 << owner).to(torch.int32)
        bits[0, :3, :3] = 0b101                               # several parts own a pixel / (below) nobody does
        bits[0, 5, 5] = 0
        mask = (bits.to(dev), parts)
        hard = ((bits.unsqueeze(-1) >> torch.arange(parts)) & 1).double()
    out = {}
    for first in ("1", "0"):
        monkeypatch.setenv("UPS_FIRST_LAYER", first)
        lay = _layer(ops, lib, V, b, 3, 1, coords, None, dev)
        if post:
            lay.out_act = lib.ACT_LRELU
        out[first] = ops.conv_forward(x.to(dev), lay, mask=mask).float().cpu()
    assert_close(out["1"], out["0"], BF16_TOL, "first-layer kernel vs patch kernel")
    assert float((out["1"] - out["0"]).abs().max()) <= 2.0 ** -6 * float(out["0"].abs().max())      # at most a bf16 rounding flip
    xo = x[..., :3].double()
    if parts:
        xo = (xo.unsqueeze(3) * hard.unsqueeze(-1)).permute(3, 0, 1, 2, 4).reshape(parts * n, h, w, 3)
    Vo = V.double().clone()
    Vo[:, :, :3] = V[:, :, :3].to(torch.bfloat16).double()          # the image rows of the kernel's weights are bf16 (CoordConv rows fp32)
    ref = _oracle_conv(R, xo, Vo, b.double(), 1, coords, None, False, None)
    if post:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    assert_close(out["1"][..., :cout], ref.float(), BF16_TOL, "first-layer kernel vs oracle")
    if cout < out["1"].shape[-1]:
        assert float(out["1"][..., cout:].abs().max()) == 0.0


@pytest.mark.parametrize("P,B,S,coords", [(10, 3, 32, False), (25, 2, 48, True), (3, 5, 16, False)])
def test_part_masked_convolution_matches_the_materialised_path(P, B, S, coords, dev):
    """mask_parts fused into encoder_1's first convolution (model.py:176-187, nn.py:81-113): forward, weight gradient and the
    gradient w.r.t. the hard mask (reduced in the dgrad epilogue) must equal the path that builds the [P*B,S,S,8] part tensor
    (bit for bit: the same bf16 operands meet the same kernels), and the oracle's fp64 result within bf16 tolerance."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(40 + P)
    view = (torch.rand(B, S, S, 3, generator=g) * 2 - 1).to(dev)
    mean = torch.randn(B, S, S, P, generator=g).to(dev)
    mean[0, :4, :4] = 0.0                                            # ties: several parts own a pixel
    _, m, hard, _, bits = ops.part_softmax(mean, None, want_bits=True)
    assert torch.equal(((bits.unsqueeze(-1) >> torch.arange(P, device=dev)) & 1).float(), hard)
    cin_v = 3 + (2 if coords else 0)
    V = torch.randn(3, 3, cin_v, 32, generator=g) / math.sqrt(9 * cin_v)
    b = torch.randn(32, generator=g) * 0.1
    T = torch.bfloat16
    view_act = torch.zeros(B, S, S, 8, dtype=T, device=dev)
    view_act[..., :3] = view.to(T)
    res = {}
    for mode in ("fused", "materialised"):
        lay = _layer(ops, lib, V, b, 3, 1, coords, None, dev)
        h = hard.clone().requires_grad_(True)
        if mode == "fused":
            y = ops.conv(view_act, lay, mask=(h, bits, view))
        else:
            parts = ops.MaskPartsFn.apply(view, h, T)
            y = ops.conv(parts, lay)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).to(dev, T)
        gh, gV, gb = torch.autograd.grad([y], [h, lay.V, lay.b], grad_outputs=[gy])
        res[mode] = (y, gh, gV, gb)
    for a, c, name in zip(res["fused"], res["materialised"], ("forward", "d hard", "dV", "db")):
        if name == "d hard":      # three-term fp32 dot product: summation order differs by an ulp
            assert_close(a, c, 1e-5, "d hard: fused vs materialised")
        else:
            assert torch.equal(a, c), "{}: fused and materialised paths differ by {}".format(name, float((a.float() - c.float()).abs().max()))
    # oracle: part images in fp64 from the same bf16-rounded view / weights
    vq = view.to(T).double().cpu()
    xo = (vq.unsqueeze(3) * hard.double().cpu().unsqueeze(-1)).permute(3, 0, 1, 2, 4).reshape(P * B, S, S, 3)
    Vo = V.double().clone()
    Vo[:, :, :3] = V[:, :, :3].to(T).double()
    yo = _oracle_conv(R, xo, Vo, b.double(), 1, coords, None, False, None)
    assert_close(res["fused"][0][..., :32].float(), yo.float(), BF16_TOL, "part-masked conv vs oracle")


@pytest.mark.parametrize("P,B,H,W", [(10, 3, 128, 128), (3, 4, 32, 128), (25, 1, 64, 128), (5, 2, 64, 256), (20, 1, 32, 256)])
def test_part_mask_gradient_row_stream(P, B, H, W, dev, monkeypatch):
    """conv3x3_rows.hip, mask-gradient form: d loss / d hard of encoder_1's part-masked first convolution (M:176-187) at 128 columns as a
    row stream of the gradient tensor.  Against the patch kernel's epilogue (UPS_ROWS_KERNEL=0) on the same operands and against
    the fp64 oracle's autograd through the materialised part images."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(90 + P)
    view = (torch.rand(B, H, W, 3, generator=g) * 2 - 1).to(dev)
    mean = torch.randn(B, H, W, P, generator=g).to(dev)
    _, m, hard, _, bits = ops.part_softmax(mean, None, want_bits=True)
    V = torch.randn(3, 3, 3, 32, generator=g) / math.sqrt(27)
    b = torch.randn(32, generator=g) * 0.1
    T = torch.bfloat16
    view_act = torch.zeros(B, H, W, 8, dtype=T, device=dev)
    view_act[..., :3] = view.to(T)
    gy = torch.randn(P * B, H, W, 32, generator=g).to(dev, T)
    out = {}
    for mode in ("force", "0"):
        monkeypatch.setenv("UPS_ROWS_KERNEL", mode)
        lay = _layer(ops, lib, V, b, 3, 1, False, None, dev)
        h = hard.clone().requires_grad_(True)
        y = ops.conv(view_act, lay, mask=(h, bits, view))
        gh, = torch.autograd.grad([y], [h], grad_outputs=[gy])
        torch.cuda.synchronize()
        out[mode] = gh.cpu()
    # (gx is rounded to bf16 before the dot product in both kernels; their fp32 tap sums differ in order, so a rounding flips now and then)
    assert_close(out["force"], out["0"], 5e-3, "row-stream vs patch-kernel mask gradient")
    # oracle: g_hard[b, y, x, p] = sum_c gx[p * B + b, y, x, c] * view[b, y, x, c], gx rounded to bf16 as the tensor it replaces was
    vq = view.to(T).double().cpu()
    parts = (vq.unsqueeze(3) * hard.double().cpu().unsqueeze(-1)).permute(3, 0, 1, 2, 4).reshape(P * B, H, W, 3).requires_grad_(True)
    Vo = V.to(T).double()
    yo = _oracle_conv(R, parts, Vo, b.double(), 1, False, None, False, None)
    gx, = torch.autograd.grad([yo], [parts], grad_outputs=[gy.double().cpu()])
    gx = gx.to(T).double().reshape(P, B, H, W, 3)
    gho = (gx * view.double().cpu().unsqueeze(0)).sum(-1).permute(1, 2, 3, 0)
    assert_close(out["force"], gho.float(), BF16_TOL, "row-stream mask gradient vs oracle")


@pytest.mark.parametrize("case", [
    # n, h, cin, cout, coords, act, res_self
    (4, 32, 128, 128, False, "leaky_relu", True),
    (2, 128, 256, 256, True, "leaky_relu", True),      # the dominant decoder res-block layer (kchunks 4 at 64 channels each)
    (3, 16, 64, 96, True, None, False),                # one chunk, 64-wide N-tile with a ragged tail
    (2, 48, 192, 40, False, "relu", False),            # 32-wide N-tile is never taken (co >= 64 gate): stays bf16
])
def test_conv_fp8_forward(case, dev):
    """fp8 forward (BASELINE config #5): the kernel against the same arithmetic restated in torch -- activations and weights
    rounded to e4m3 (RNE) after the per-tensor / per-output-channel scaling, exact accumulation, dequantised, CoordConv rows
    and bias in fp32 -- at bf16 output resolution; and against the unquantised fp64 convolution at the fp8 error level."""
    lib, ops, R = _mods()
    n, h, cin, cout, coords, act, res_self = case
    g = torch.Generator().manual_seed(11)
    cv = cin + (2 if coords else 0)
    V = torch.randn(3, 3, cv, cout, generator=g) / math.sqrt(9 * cv)
    V[..., : cout // 3] *= 4.0                 # unequal channel ranges: the per-channel scales matter
    b = torch.randn(cout, generator=g) * 0.1
    x = (torch.randn(n, h, h, cin, generator=g) * 1.5).to(torch.bfloat16)
    lay = _layer(ops, lib, V, b, 3, 1, coords, act, dev)
    xd = x.to(dev)
    ops.Fp8.activate(ops.Fp8State(True, copy_only=False))      # (no producer here: the kernel converts its bf16 operand itself)
    try:
        eligible = ops.Fp8.eligible(lay, xd)
        y = ops.conv_forward(xd, lay, res=xd if res_self else None)
        torch.cuda.synchronize()
        if not eligible:
            assert cout < 64
            ref = _oracle_conv(R, x.double(), V.double(), b.double(), 1, coords, act, res_self, None)
            assert_close(y[..., :cout].float().cpu(), ref.float(), BF16_TOL, "bf16 path of an ineligible layer")
            return
        f8 = lay._cache["f8"]
        s_a = float(ops.Fp8.scale[f8["slot"]].cpu())
        amax_seen = float(ops.Fp8.amax[f8["slot"]].max().cpu())
    finally:
        ops.Fp8.activate(ops.Fp8State(False))
    xf = x.float()
    xa = torch.maximum(xf, 0.2 * xf) if act == "leaky_relu" else (torch.relu(xf) if act == "relu" else xf)
    assert abs(amax_seen - float(xa.abs().max())) <= 1e-6 * amax_seen, "recorded activation maximum"
    assert abs(s_a - 448.0 * ops.Fp8.MARGIN / float(xf.abs().max())) <= 1e-5 * s_a
    xq = (xa * s_a).clamp(-448, 448).to(torch.float8_e4m3fn).double()
    wmax = V[:, :, :cin].abs().amax(dim=(0, 1, 2))
    wq = (V[:, :, :cin] * (448.0 / wmax)).clamp(-448, 448).to(torch.float8_e4m3fn).double()
    assert_close(f8["deq"].cpu(), (wmax / 448.0), 1e-6, "per-channel dequantisation factors")
    acc = R.conv2d_same(xq, wq, torch.zeros(cout, dtype=torch.float64), 1) * (wmax.double() / 448.0) / s_a
    extra = b.double().view(1, 1, 1, -1).expand_as(acc).clone()
    if coords:      # CoordConv rows stay fp32 (ups_coord_table): their contribution = conv of the coordinate planes alone
        cz = torch.zeros(n, h, h, cin, dtype=torch.float64)
        Vc = V.double().clone(); Vc[:, :, :cin] = 0
        extra = extra + R.conv2d_same(R.Scope.add_coordinates(cz), Vc, torch.zeros(cout, dtype=torch.float64), 1)
    ref_q = acc + extra + (x.double() if res_self else 0)
    got = y[..., :cout].float().cpu()
    assert_close(got, ref_q.float(), 1e-2, "fp8 kernel vs e4m3 emulation")          # bf16 output rounding
    ref = _oracle_conv(R, x.double(), V.double(), b.double(), 1, coords, act, res_self, None)
    err = float((got.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert err < 0.05, "fp8 forward vs unquantised convolution: rel RMS {:.3g}".format(err)


@pytest.mark.parametrize("case", [
    # n, h, cin, cout, coords, act
    (4, 32, 128, 128, False, "leaky_relu"),
    (2, 128, 256, 256, True, "leaky_relu"),
    (3, 16, 64, 192, False, None),
])
def test_conv_fp8_input_gradient(case, dev):
    """Input gradient of an fp8 layer: the gradient tensor staged as e5m2 (per-tensor scale), the transposed weights as e4m3
    scaled per input channel, fp32 accumulation, act' from the bf16 forward input -- against the same arithmetic in torch."""
    lib, ops, R = _mods()
    n, h, cin, cout, coords, act = case
    g0 = torch.Generator().manual_seed(12)
    cv = cin + (2 if coords else 0)
    V = torch.randn(3, 3, cv, cout, generator=g0) / math.sqrt(9 * cv)
    V[:, :, : cin // 2] *= 3.0
    b = torch.zeros(cout)
    x = (torch.randn(n, h, h, cin, generator=g0)).to(torch.bfloat16)
    gy = (torch.randn(n, h, h, cout, generator=g0) * 0.02).to(torch.bfloat16)
    lay = _layer(ops, lib, V, b, 3, 1, coords, act, dev)
    ops.Fp8.activate(ops.Fp8State(True, copy_only=False))
    try:
        assert ops.Fp8.eligible_grad(lay, gy.to(dev), x.to(dev))
        gx = ops.conv_dgrad(gy.to(dev), x.to(dev), lay)
        torch.cuda.synchronize()
        f8 = lay._cache["f8g"]
        s_g = float(ops.Fp8.scale[f8["slot"]].cpu())
        amax_seen = float(ops.Fp8.amax[f8["slot"]].max().cpu())
    finally:
        ops.Fp8.activate(ops.Fp8State(False))
    gf = gy.float()
    assert abs(amax_seen - float(gf.abs().max())) <= 1e-6 * amax_seen
    assert abs(s_g - 57344.0 * ops.Fp8.MARGIN / float(gf.abs().max())) <= 1e-5 * s_g
    gq = (gf * s_g).clamp(-57344, 57344).to(torch.float8_e5m2).double()
    wmax = V[:, :, :cin].abs().amax(dim=(0, 1, 3))                              # per input channel
    wq = (V[:, :, :cin] * (448.0 / wmax).view(1, 1, -1, 1)).clamp(-448, 448).to(torch.float8_e4m3fn).double()
    assert_close(f8["deq"].cpu(), wmax / 448.0, 1e-6, "per-row dequantisation factors")
    # conv^T with the quantised operands: gradient of sum(conv(z, wq) * gq) w.r.t. z
    z = torch.zeros(n, h, h, cin, dtype=torch.float64, requires_grad=True)
    y = R.conv2d_same(z, wq, torch.zeros(cout, dtype=torch.float64), 1)
    ref_q, = torch.autograd.grad((y * gq).sum(), z)
    ref_q = ref_q * (wmax.double() / 448.0) / s_g
    xf = x.double()
    if act == "leaky_relu":
        ref_q = ref_q * torch.where(xf > 0, torch.ones_like(xf), torch.full_like(xf, 0.2))
    assert_close(gx[..., :cin].float().cpu(), ref_q.float(), 1e-2, "fp8 input gradient vs e5m2 / e4m3 emulation")
    # and against the unquantised gradient at the e5m2 error level
    z2 = torch.zeros(n, h, h, cin, dtype=torch.float64, requires_grad=True)
    y2 = R.conv2d_same(z2, V[:, :, :cin].double(), torch.zeros(cout, dtype=torch.float64), 1)
    ref, = torch.autograd.grad((y2 * gf.double()).sum(), z2)
    if act == "leaky_relu":
        ref = ref * torch.where(xf > 0, torch.ones_like(xf), torch.full_like(xf, 0.2))
    err = float((gx[..., :cin].double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert err < 0.08, "fp8 input gradient vs unquantised: rel RMS {:.3g}".format(err)


@pytest.mark.parametrize("case", [
    # (n, h, cin, cout, coords, act, fp16 forward input, image width != height)
    (2, 32, 64, 128, False, None, False, 32),
    (3, 16, 128, 256, True, "leaky_relu", False, 48),
    (2, 32, 256, 256, True, None, True, 32),            # the mask decoder's form: fp16 post-activation input, CoordConv rows beside it
    (8, 128, 256, 256, True, None, True, 128),          # the bench's own instance: 258 -> 256 at 128 x 128, 256 blocks, split-K 128
])
def test_conv_fp8_weight_gradient(case, dev):
    """Weight gradient on the block-scaled fp8 MFMA (conv_wgrad3x3_f8.hip; round 5): the gradient arrives as its producer's e5m2
    copy (per-tensor scale), the forward input (bf16, or fp16 in a bf16 container) is quantised to e4m3 while it is staged, fp32
    accumulation over all pixels -- against the same arithmetic restated in torch from the very bytes the kernel reads (1e-3: the
    products are exact in fp32, only the summation order differs), the bias gradient against the sum of the quantised gradient,
    the recorded maximum against the tensor's, and the unquantised fp64 weight gradient at the e5m2 / e4m3 error level."""
    lib, ops, R = _mods()
    n, h, cin, cout, coords, act, f16in, w = case
    g0 = torch.Generator().manual_seed(31)
    cv = cin + (2 if coords else 0)
    V = torch.randn(3, 3, cv, cout, generator=g0) / math.sqrt(9 * cv)
    lay = _layer(ops, lib, V, torch.zeros(cout), 3, 1, coords, act, dev)
    xf = torch.randn(n, h, w, cin, generator=g0)
    xf = xf * (1.0 + 2.0 * torch.rand(1, 1, 1, cin, generator=g0))                   # channels on different scales
    if f16in:
        x16 = xf.to(torch.float16)
        x = x16.view(torch.bfloat16).to(dev)
        xs = x16.float()
    else:
        x = xf.to(torch.bfloat16).to(dev)
        xs = x.float().cpu()
    gy = (torch.randn(n, h, w, cout, generator=g0) * 0.02 * (1.0 + 3.0 * torch.rand(1, 1, 1, cout, generator=g0))).to(torch.bfloat16)
    gyd = gy.to(dev)
    with ops.fp8_scope(enabled=True) as F:
        # the producer's side of the hand-off, restated: e5m2(g * scale) with the delayed scale of that tensor's slot
        slot = F.slot(dev)
        F.fmax[slot] = F.E5M2_MAX
        s_g = F.E5M2_MAX * F.MARGIN / float(gy.float().abs().max())
        F.scale[slot] = s_g
        s_g = float(F.scale[slot].cpu())
        g8 = (gy.float() * s_g).clamp(-57344, 57344).to(torch.float8_e5m2)
        copy = {"t": g8.view(torch.uint8).to(dev), "slot": slot}
        assert F.eligible_wgrad(lay, gyd, x, None)
        gV, gb = ops.conv_wgrad(gyd, x, lay, fmt=lib.F16 if f16in else None, f8_src=copy)
        torch.cuda.synchronize()
        assert F.stats["wgrad_f8"] == 1
        ew = lay._cache["f8w"]
        s_x = float(F.scale[ew["slot"]].cpu())
        amax_seen = float(F.amax[ew["slot"]].max().cpu())
    xa = xs if act is None else torch.where(xs > 0, xs, 0.2 * xs)
    assert abs(amax_seen - float(xa.abs().max())) <= 1e-6 * amax_seen, (amax_seen, float(xa.abs().max()))
    assert abs(s_x - 448.0 * 0.5 / float(xs.abs().max())) <= 1e-5 * s_x          # first launch: primed from the tensor at hand
    xq = (xa * s_x).clamp(-448, 448).to(torch.float8_e4m3fn).double()
    gq = g8.double()
    # dV[r][s][ci][co] = sum_pix xq[pix + (r-1, s-1)][ci] * gq[pix][co] / (s_x * s_g)
    xp = torch.nn.functional.pad(xq, (0, 0, 1, 1, 1, 1))
    ref_q = torch.zeros(3, 3, cin, cout, dtype=torch.float64)
    for r in range(3):
        for c in range(3):
            ref_q[r, c] = torch.einsum("nhwi,nhwo->io", xp[:, r:r + h, c:c + w], gq)
    ref_q /= (s_x * s_g)
    assert_close(gV[:, :, :cin].cpu(), ref_q.float(), 1e-3, "fp8 weight gradient vs e4m3 / e5m2 emulation")
    assert_close(gb.cpu(), (gq.sum(dim=(0, 1, 2)) / s_g).float(), 1e-3, "fp8 bias gradient vs the sum of the e5m2 copy")
    # unquantised reference: the CoordConv rows (written by the fp32 coordinate kernels) to 2e-3 as in the bf16 tests, the main
    # rows at the quantisation error level
    xpf = torch.nn.functional.pad(xa.double(), (0, 0, 1, 1, 1, 1))
    ref = torch.zeros(3, 3, cin, cout, dtype=torch.float64)
    for r in range(3):
        for c in range(3):
            ref[r, c] = torch.einsum("nhwi,nhwo->io", xpf[:, r:r + h, c:c + w], gy.double())
    err = float((gV[:, :, :cin].double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    cos = float((gV[:, :, :cin].double().cpu() * ref).sum() / (gV[:, :, :cin].double().cpu().norm() * ref.norm()))
    print("fp8 weight gradient {}: rel RMS {:.4f}, cosine {:.5f} vs the unquantised gradient".format(case, err, cos))
    assert err < 0.08 and cos > 0.997, (err, cos)
    if coords:
        with ops.fp8_scope(enabled=False):
            gV16, _ = ops.conv_wgrad(gyd, x, lay, fmt=lib.F16 if f16in else None)
        assert torch.equal(gV[:, :, cin:], gV16[:, :, cin:]) or float((gV[:, :, cin:] - gV16[:, :, cin:]).abs().max()) <= \
            1e-5 * float(gV16[:, :, cin:].abs().max()), "CoordConv rows must be the fp32 coordinate kernels' in both modes"


@pytest.mark.parametrize("case", [(4, 64, 128, 128, True), (2, 128, 256, 256, True), (8, 32, 64, 192, False)])
def test_conv_fp8_copy_handed_from_producer_to_consumer(case, dev):
    """fp8 copies between layers: the producing convolution's epilogue writes e4m3(act(out) * scale) next to its bf16 output (one
    step after that tensor's first maximum was recorded), the consuming convolution stages those bytes as they are.  The copy must
    equal the quantisation of the stored bf16 tensor bit for bit, and the consumer must produce what it produces when it converts
    the bf16 tensor itself with the same scale."""
    lib, ops, R = _mods()
    n, h, cin, cout, coords = case
    g = torch.Generator().manual_seed(21)
    cv = cin + (2 if coords else 0)
    V1 = torch.randn(3, 3, cv, cin, generator=g) / math.sqrt(9 * cv)
    V2 = torch.randn(3, 3, cv, cout, generator=g) / math.sqrt(9 * cv)
    l1 = _layer(ops, lib, V1, torch.randn(cin, generator=g) * 0.1, 3, 1, coords, "leaky_relu", dev)
    l2 = _layer(ops, lib, V2, torch.randn(cout, generator=g) * 0.1, 3, 1, coords, "leaky_relu", dev)
    x = torch.randn(n, h, h, cin, generator=g).to(torch.bfloat16).to(dev)
    F = ops.Fp8.activate(ops.Fp8State(True))          # a fresh state for this test (scale slots, hand-off, counters)
    producer_was, copy_only_was = F.PRODUCER, F.COPY_ONLY
    F.PRODUCER, F.COPY_ONLY = True, False          # (the producing layer itself converts its bf16 input in the kernel here)
    try:
        F.next_out_act = lib.ACT_LRELU
        y0 = ops.conv_forward(x, l1, res=x)                   # first call: records max |act(y)| only
        assert F.last_out is None
        F.update()
        F.next_out_act = lib.ACT_LRELU
        y = ops.conv_forward(x, l1, res=x)
        copy = F.last_out
        assert copy is not None and copy["act"] == lib.ACT_LRELU and copy["t"].shape == y.shape
        # (y differs from y0 in the last bits: the input's own scale moved from the primed to the recorded maximum)
        assert rel_err(y.float(), y0.float()) <= 5e-2, "second call"     # (two fp8 quantisations of the same tensor: a sanity bound)
        scale = float(F.scale[copy["slot"]].cpu())
        yf = y.float()
        ya = torch.maximum(yf, 0.2 * yf)
        y0f = y0.float()
        assert abs(scale - 448.0 * F.MARGIN / float(torch.maximum(y0f, 0.2 * y0f).abs().max())) <= 1e-5 * scale   # delayed: first call's maximum
        want = (ya * scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(copy["t"], want), "fp8 copy differs from the quantised bf16 tensor in {} bytes".format(int((copy["t"] != want).sum()))
        # consumer: bytes as they are (two blocks per CU) vs its own conversion of the bf16 tensor with the same scale
        F.next_in = copy
        za = ops.conv_forward(y, l2)
        zb0 = ops.conv_forward(y, l2)                         # primes the layer's own slot from the tensor ...
        F.scale[l2._cache["f8"]["slot"]] = F.scale[copy["slot"]]      # ... which is then set to the producer's scale
        zb = ops.conv_forward(y, l2)
        torch.cuda.synchronize()
    finally:
        ops.Fp8.activate(ops.Fp8State(False))
        F.PRODUCER, F.COPY_ONLY = producer_was, copy_only_was
        F.next_in = F.next_out_act = F.last_out = None
    assert_close(za[..., :cout].float(), zb[..., :cout].float(), 1e-6, "consumer of the fp8 copy vs in-kernel conversion")
    assert float((za[..., :cout].float() - zb0[..., :cout].float()).abs().max()) < 0.1 * float(zb0.float().abs().max())


@pytest.mark.parametrize("case", [
    # n, h, cin, cout, coords, res_self      (grids of >= 512 blocks with whole 128-channel double chunks)
    (8, 128, 128, 128, True, True),          # one double chunk, 128-wide N-tile
    (4, 128, 256, 256, True, True),          # the dominant decoder layer's shape: two double chunks, two N-tiles
    (32, 64, 128, 64, False, False),         # 64-wide N-tile
    (2, 128, 256, 256, True, True),          # small grid: stays on the K = 32 path (same expectation)
    (6, 112, 384, 192, False, False),        # three double chunks, a ragged second N-tile (192 = 128 + 64), 7x7 tiles per image
])
def test_conv_fp8_block_scaled_mfma(case, dev):
    """The block-scaled K = 128 MFMA path (v_mfma_scale_f32_16x16x128_f8f6f4, conv3x3_patch.hip f8s_tap): a consumer of a
    pre-quantised e4m3 copy (forward) / e5m2 copy (input gradient) on a grid of two blocks per CU, against the same arithmetic
    restated in torch from the very bytes the kernel reads (exact products, fp64 accumulation; the kernel accumulates in fp32),
    and against the unquantised convolution at the fp8 error level."""
    lib, ops, R = _mods()
    n, h, cin, cout, coords, res_self = case
    g0 = torch.Generator().manual_seed(31)
    cv = cin + (2 if coords else 0)
    V = torch.randn(3, 3, cv, cout, generator=g0) / math.sqrt(9 * cv)
    V[..., : cout // 3] *= 4.0
    V[:, :, : cin // 2] *= 2.0
    b = torch.randn(cout, generator=g0) * 0.1
    x = (torch.randn(n, h, h, cin, generator=g0) * 1.5).to(torch.bfloat16)
    gy = (torch.randn(n, h, h, cout, generator=g0) * 0.02).to(torch.bfloat16)
    lay = _layer(ops, lib, V, b, 3, 1, coords, "leaky_relu", dev)
    xd, gd = x.to(dev), gy.to(dev)
    F = ops.Fp8.activate(ops.Fp8State(True))          # a fresh state for this test (scale slots, hand-off, counters)
    try:
        # ---- forward: e4m3(lrelu(x) * s_a) handed in as a producer's copy
        xa = torch.maximum(x.float(), 0.2 * x.float())
        s_a = 448.0 * F.MARGIN / float(xa.abs().max())
        sl = F.slot(dev)
        F.scale[sl] = s_a
        xq8 = (xa * s_a).clamp(-448, 448).to(torch.float8_e4m3fn)
        F.next_in = {"t": xq8.view(torch.uint8).to(dev), "slot": sl, "act": lib.ACT_LRELU, "site": None}
        y = ops.conv_forward(xd, lay, res=xd if res_self else None)
        assert F.stats["fwd_copy_in"] == 1
        # ---- input gradient: e5m2(gy * s_g) registered as the copy of exactly this gradient tensor
        s_g = 57344.0 * F.MARGIN / float(gy.float().abs().max())
        sg = F.slot(dev)
        F.scale[sg] = s_g
        gq8 = (gy.float() * s_g).clamp(-57344, 57344).to(torch.float8_e5m2)
        F.register_grad_copy(gd, {"t": gq8.view(torch.uint8).to(dev), "slot": sg, "site": None})
        gx = ops.conv_dgrad(gd, xd, lay)
        assert F.stats["dgrad_copy_in"] == 1
        torch.cuda.synchronize()
    finally:
        ops.Fp8.activate(ops.Fp8State(False))
    # forward expectation
    wmax = V[:, :, :cin].abs().amax(dim=(0, 1, 2))
    wq = (V[:, :, :cin] * (448.0 / wmax)).clamp(-448, 448).to(torch.float8_e4m3fn).double()
    acc = R.conv2d_same(xq8.double(), wq, torch.zeros(cout, dtype=torch.float64), 1) * (wmax.double() / 448.0) / s_a
    extra = b.double().view(1, 1, 1, -1).expand_as(acc).clone()
    if coords:
        Vc = V.double().clone(); Vc[:, :, :cin] = 0
        extra = extra + R.conv2d_same(R.Scope.add_coordinates(torch.zeros(n, h, h, cin, dtype=torch.float64)), Vc,
                                      torch.zeros(cout, dtype=torch.float64), 1)
    ref_q = acc + extra + (x.double() if res_self else 0)
    got = y[..., :cout].float().cpu()
    assert_close(got, ref_q.float(), 1e-2, "block-scaled fp8 forward vs e4m3 emulation")      # (bf16 output rounding)
    ref = _oracle_conv(R, x.double(), V.double(), b.double(), 1, coords, "leaky_relu", res_self, None)
    err = float((got.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert err < 0.05, "fp8 forward vs unquantised convolution: rel RMS {:.3g}".format(err)
    # input-gradient expectation
    wmax_i = V[:, :, :cin].abs().amax(dim=(0, 1, 3))
    wqi = (V[:, :, :cin] * (448.0 / wmax_i).view(1, 1, -1, 1)).clamp(-448, 448).to(torch.float8_e4m3fn).double()
    z = torch.zeros(n, h, h, cin, dtype=torch.float64, requires_grad=True)
    yz = R.conv2d_same(z, wqi, torch.zeros(cout, dtype=torch.float64), 1)
    ref_g, = torch.autograd.grad((yz * gq8.double()).sum(), z)
    ref_g = ref_g * (wmax_i.double() / 448.0) / s_g
    xf = x.double()
    ref_g = ref_g * torch.where(xf > 0, torch.ones_like(xf), torch.full_like(xf, 0.2))
    assert_close(gx[..., :cin].float().cpu(), ref_g.float(), 1e-2, "block-scaled fp8 input gradient vs e5m2 / e4m3 emulation")


def test_bilinear_fp8_copies(dev):
    """The bilinear x2 kernels as fp8 producers: same bf16 result as the plain kernels, the forward copy = e4m3(lrelu(y) * scale), the
    backward copy = e5m2(gx * scale), the recorded maxima exact."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 16, 24, 128, generator=g) * 2).to(torch.bfloat16).to(dev)
    gy = (torch.randn(3, 32, 48, 128, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    y_ref = ops.BilinearFn.apply(x)
    F = ops.Fp8.activate(ops.Fp8State(True))          # a fresh state for this test (scale slots, hand-off, counters)
    producer_was = F.PRODUCER
    F.PRODUCER = True
    try:
        site = {}
        xr = x.clone().requires_grad_(True)
        y0 = ops.BilinearFn.apply(xr, site, lib.ACT_LRELU, 0.2)          # first step: maxima only
        assert F.last_out is None and torch.equal(y0, y_ref)
        y0.backward(gy)
        gx_ref = xr.grad.clone()
        F.update()
        xr.grad = None
        y1 = ops.BilinearFn.apply(xr, site, lib.ACT_LRELU, 0.2)
        copy = F.last_out
        assert copy is not None and torch.equal(y1, y_ref)
        yf = y_ref.float(); ya = torch.maximum(yf, 0.2 * yf)
        sc = float(F.scale[copy["slot"]].cpu())
        assert abs(sc - 448.0 * F.MARGIN / float(ya.abs().max())) <= 1e-5 * sc
        assert torch.equal(copy["t"], (ya * sc).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
        y1.backward(gy)
        gx = xr.grad
        assert torch.equal(gx, gx_ref)
        ent = F.grad_side.get(gx.data_ptr())
        assert ent is not None, "no e5m2 copy registered for the returned gradient"
        gcopy = ent[2]
        sg = float(F.scale[gcopy["slot"]].cpu())
        assert abs(sg - 57344.0 * F.MARGIN / float(gx_ref.float().abs().max())) <= 1e-5 * sg
        want = (gx_ref.float() * sg).clamp(-57344, 57344).to(torch.float8_e5m2).view(torch.uint8)
        assert torch.equal(gcopy["t"], want)
    finally:
        ops.Fp8.activate(ops.Fp8State(False))
        F.PRODUCER = producer_was
        F.last_out = None
        F.grad_side.clear()


@pytest.mark.parametrize("k,stride,ci,co,hw", [(3, 1, 16, 16, 16), (3, 2, 8, 16, 32), (3, 1, 256, 10, 128), (1, 1, 32, 64, 8)])
def test_weight_copies_per_layer_and_batched_are_bit_identical(k, stride, ci, co, hw, dev):
    """A layer's converted weights and CoordConv table come from ups_weight_prep + ups_coord_table the first time (and after a
    restore), from ups_weight_prep_batch after every optimizer step: both must give the SAME bits, or a run restored from a
    checkpoint drifts an ulp from the run that wrote it (round 5: the table's arithmetic was left to per-call-site contraction)."""
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(k * 100 + ci + co)
    V = (torch.randn(k, k, ci + 2, co, generator=g) / math.sqrt(k * k * (ci + 2))).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    lay = ops.ConvLayer("t/conv2d_0", V, b, k, stride, True, "leaky_relu")
    reg = ops.PrepRegistry()
    lay.registry = reg
    ops.WeightVersion.value += 1
    ent = lay.prepared(lib.BF16, hw, hw, True)              # per-layer launches (registers the entry)
    first = {n: ent[n].clone() for n in ("w_fwd", "w_dgrad", "ctab") if ent[n] is not None}
    assert "ctab" in first and float(first["ctab"].abs().max()) > 0
    for t in (ent["w_fwd"], ent["w_dgrad"], ent["ctab"]):
        if t is not None:
            t.view(torch.uint8).fill_(0x55)
    ops.WeightVersion.value += 1
    reg.refresh()                                           # the batched launch
    torch.cuda.synchronize()
    for n, t in first.items():
        assert torch.equal(t.view(torch.uint8), ent[n].view(torch.uint8)), n


@pytest.mark.parametrize("B,S,P", [(3, 128, 10), (1, 32, 10), (2, 64, 10), (2, 64, 16), (1, 32, 16), (2, 64, 20), (3, 32, 20), (2, 64, 25),
                                   (1, 32, 25), (1, 128, 25)])
def test_unpool_backward_on_the_matrix_cores(B, S, P, dev, monkeypatch):
    """ups_unpool_bwd at the benchmark shape classes (bf16 gradient, 64 features + P parts in round8(64 + P)-channel rows; P = 10 since
    round 5, 16 / 20 / 25 -- BASELINE configs #3 / #5, the shipped yamls -- since round 6: two 16-part blocks, a partial last DMA piece at
    P = 25): the MFMA form against torch-fp64 and against the VALU form it replaces (UPS_UNPOOL_MFMA=0).  hard: one-hot rows (what the
    model passes), a tie, and a block of arbitrary fp32 values (the hi + lo bf16 pair of the mask operand must keep them to ~1e-5)."""
    lib, ops, R = _mods()
    L = lib
    F, ld = 64, (64 + P + 7) // 8 * 8
    g = torch.Generator().manual_seed(5 + B + S + P)
    hard = torch.nn.functional.one_hot(torch.randint(0, P, (B, S, S), generator=g), P).float()
    hard[0, 0, 0] = torch.tensor([1.0, 1.0] + [0.0] * (P - 2))
    hard[0, 1, :8] = torch.randn(8, P, generator=g)
    feat = torch.randn(B, P, F, generator=g).bfloat16().float()          # bf16 mode: the float view of a bf16 tensor
    gi = torch.zeros(B, S, S, ld)
    gi[..., :F + P] = torch.randn(B, S, S, F + P, generator=g)
    gi = gi.bfloat16()
    gd = gi.double()
    want_h = torch.einsum("bhwf,bpf->bhwp", gd[..., :F], feat.double()) + gd[..., F:F + P]
    want_f = torch.einsum("bhwp,bhwf->bpf", hard.double(), gd[..., :F])
    hd, fd, gdv = hard.to(dev), feat.to(dev), gi.to(dev)
    nfl = L.load().ups_unpool_bwd_floats(B, P, F)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UPS_UNPOOL_MFMA", mode)
        gh = torch.full((B, S, S, P), float("nan"), device=dev)
        gf = torch.full((nfl,), float("nan"), dtype=torch.float32, device=dev)
        L.call("ups_unpool_bwd", L.ptr(hd), L.ptr(fd), L.ptr(gdv), L.ptr(gh), L.ptr(gf), L.dt(gdv), B, S * S, P, F, ld, L.stream())
        torch.cuda.synchronize()
        res[mode] = (gh.cpu(), gf[:B * P * F].view(B, P, F).cpu())
        assert_close(res[mode][0], want_h.float(), 1e-5, "unpool d hard (UPS_UNPOOL_MFMA={})".format(mode))
        assert_close(res[mode][1], want_f.float(), 2e-5, "unpool d feat (UPS_UNPOOL_MFMA={})".format(mode), elementwise=False)
    assert_close(res["1"][0], res["0"][0], 1e-5, "MFMA vs VALU form, d hard")


@pytest.mark.parametrize("N,H,W,rect", [(3, 128, 128, True), (2, 64, 256, True), (5, 128, 128, False), (64, 32, 128, True)])
def test_spatial_moments_pixel_per_lane_matches_the_slab_form(N, H, W, rect, dev, monkeypatch):
    """ups_spatial_moments(_kl) at P = 10 / 128- and 256-wide maps runs pixel-per-lane since round 5 (moments_px_kernel): the same
    statistics {max, Z, S0, Sy, Sx, Q, Qy} and the same categorical-KL sum as the (part, sub-lane) form it replaces
    (UPS_MOMENTS_PX=0) and as a torch-fp64 evaluation, one to many 512-pixel tiles per block."""
    lib, ops, R = _mods()
    L = lib
    P, gamma = 10, 10.0
    g = torch.Generator().manual_seed(N + H + W)
    x = torch.softmax(2.0 * torch.randn(N, H, W, P, generator=g), -1)
    px = torch.stack([torch.randint(0, H, (N, P), generator=g), torch.randint(0, W, (N, P), generator=g)], -1).int() if rect else None
    hh, hw_ = 9, 5
    xd = x.to(dev)
    pxd = px.to(dev).contiguous() if rect else None
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UPS_MOMENTS_PX", mode)
        nfl = L.load().ups_spatial_moments_floats(N, P)
        st = torch.full((nfl,), float("nan"), dtype=torch.float32, device=dev)
        kl = torch.full((16,), float("nan"), dtype=torch.float32, device=dev)
        L.call("ups_spatial_moments_kl", L.ptr(xd), N, H, W, P, gamma, L.ptr(pxd) if rect else None, hh, hw_, L.ptr(st), L.ptr(kl), L.stream())
        torch.cuda.synchronize()
        out[mode] = (st[:N * P * 8].view(N, P, 8).cpu().double(), float(kl[0]), kl[1:].cpu())
    xv = x.double()
    e = torch.exp(gamma * xv - (gamma * xv).amax(dim=(1, 2), keepdim=True))
    gy = torch.linspace(-1, 1, H, dtype=torch.float64).view(1, H, 1, 1)
    gx = torch.linspace(-1, 1, W, dtype=torch.float64).view(1, 1, W, 1)
    k = torch.ones_like(xv)
    if rect:
        yy = torch.arange(H).view(1, H, 1, 1); xx = torch.arange(W).view(1, 1, W, 1)
        inside = ((yy - px[:, :, 0].view(N, 1, 1, P)).abs() <= hh) & ((xx - px[:, :, 1].view(N, 1, 1, P)).abs() <= hw_)
        k = (~inside).double()
    ek = e * k
    want = torch.stack([(gamma * xv).amax(dim=(1, 2)), e.sum((1, 2)), ek.sum((1, 2)), (ek * gy).sum((1, 2)), (ek * gx).sum((1, 2)),
                        (ek * (gy * gy + gx * gx)).sum((1, 2)), (ek * gy * gy).sum((1, 2))], -1)
    want_kl = float((xv * torch.log(P * xv + 1e-20)).sum())
    for mode, (st, klv, rest) in out.items():
        assert float(rest.abs().max()) == 0.0
        assert abs(klv - want_kl) <= 2e-5 * abs(want_kl), (mode, klv, want_kl)
        scale = want[..., 1:2].abs()           # sums relative to Z (Sy / Sx cross zero)
        err = ((st[..., :7] - want).abs() / torch.cat([torch.ones_like(scale), scale.expand(-1, -1, 6)], -1)).max()
        assert float(err) <= 2e-5, (mode, float(err))
    assert torch.equal(out["1"][0][..., 0], out["0"][0][..., 0])


@pytest.mark.parametrize("N,H,W,eps_on", [(4, 128, 128, True), (2, 64, 256, True), (3, 128, 128, False), (130, 16, 128, True)])
def test_part_softmax_pixel_per_lane_matches_the_lds_walking_form(N, H, W, eps_on, dev, monkeypatch):
    """ups_part_softmax(_moments)_fwd at P = 10 runs pixel-per-lane since round 5 (part_softmax_px_kernel): l bit-identical, m to
    ~1 ulp (exp / rcp + Newton instead of expf and a division), hard mask / bit set / arg-max and the hard-mask moments identical to
    the form it replaces (UPS_SOFTMAX_PX=0), m against torch-fp64; one to sixteen tiles per block, a tie included."""
    lib, ops, R = _mods()
    P, gamma = 10, 10.0
    g = torch.Generator().manual_seed(N + H + W)
    mean = torch.randn(N, H, W, P, generator=g) * 2.0
    eps = torch.randn(N, H, W, P, generator=g) if eps_on else None
    mean[0, 0, 0, :] = 0.0
    if eps_on:
        eps[0, 0, 0, :] = 0.0                      # a ten-way tie: every part is a maximum
    md, ed = mean.to(dev), (eps.to(dev) if eps_on else None)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UPS_SOFTMAX_PX", mode)
        l, m, hard, am, bits, stats = ops.part_softmax(md, ed, want_argmax=True, want_bits=True, moments_gamma=gamma)
        l2, m2, hard2, _ = ops.part_softmax(md, ed)                     # the launch without moments
        torch.cuda.synchronize()
        assert torch.equal(m2, m) and torch.equal(hard2, hard) and torch.equal(l2, l)
        out[mode] = (l.cpu(), m.cpu(), hard.cpu(), am.cpu(), bits.cpu(), stats.cpu())
    a, b = out["1"], out["0"]
    lo = (mean + eps).double() if eps_on else mean.double()
    mo = torch.softmax(lo, -1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], lo.float())
    assert_close(a[1], mo.float(), 2e-6, "softmax (pixel-per-lane)")
    assert float((a[1] - b[1]).abs().max()) <= 3e-7
    assert torch.equal(a[2].argmax(-1), mo.argmax(-1)) or float((a[2].argmax(-1) != mo.argmax(-1)).float().mean()) < 1e-5
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert float(a[2][0, 0, 0].sum()) == float(P) and int(a[4][0, 0, 0]) == (1 << P) - 1 and int(a[3][0, 0, 0]) == 0
    assert torch.equal(a[5], b[5])


def _pack_signs(t16):
    """[.., c] 16-bit tensor (bf16, or fp16 in a bf16 container) -> [.., c / 8] uint8, bit e of byte j = element 8 j + e > 0."""
    pos = (t16.view(torch.int16) > 0).view(*t16.shape[:-1], -1, 8).to(torch.uint8)
    return (pos * (2 ** torch.arange(8, device=t16.device, dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()


@pytest.mark.parametrize("n,h,cin,cout,k,stride,f16", [(4, 32, 64, 64, 3, 1, False), (2, 64, 256, 256, 3, 1, True), (3, 28, 64, 64, 3, 1, False),
                                                       (8, 8, 128, 128, 3, 1, False), (4, 32, 64, 128, 1, 1, False), (4, 32, 64, 64, 3, 2, False),
                                                       (2, 32, 16, 24, 3, 1, False),
                                                       # (round 6: the tile's sign bytes are loaded by the prologue -- PatchK.sgn_res -- here with a
                                                       # partial second N-tile, 192 = 128 + 64 channels, on the descriptor-tap instance)
                                                       (2, 32, 192, 24, 3, 1, False), (2, 32, 192, 192, 3, 1, False),
                                                       # the row-stream kernels (UPS_ROWS_KERNEL=force takes them at small batches): the one-tile
                                                       # forms write / read the bytes (LDS-DMA ring), the two-tile form does neither
                                                       (4, 128, 32, 32, 3, 1, False), (4, 64, 64, 64, 3, 1, False), (2, 128, 64, 64, 3, 1, False),
                                                       (4, 128, 32, 64, 3, 2, False), (4, 64, 64, 128, 3, 2, False), (3, 96, 64, 64, 3, 1, False)])
def test_sign_bits_written_by_the_producer_and_read_by_the_input_gradient(n, h, cin, cout, k, stride, f16, dev, monkeypatch):
    """ups_conv_desc.sign_out / dact_bits (ABI 4): a forward launch also writes one bit per stored element (> 0), whatever kernel
    runs it (the patch kernel's epilogue, or the pass over the output behind the other kernels); an input-gradient launch that is
    handed the bits of its forward input returns exactly what it returns from the input itself."""
    lib, ops, R = _mods()
    rows = cin in (32, 64) and h in (64, 128) and k == 3
    if rows:
        monkeypatch.setenv("UPS_ROWS_KERNEL", "force")
    g = torch.Generator().manual_seed(n + h + cin + cout)
    V = (torch.randn(k, k, cin, cout, generator=g) / math.sqrt(k * k * cin)).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    lay = ops.ConvLayer("t/conv2d_0", V, b, k, stride, False, "leaky_relu" if stride == 1 else None)
    lay.f16 = f16
    lay.in_post, lay.out_act = stride == 1, lib.ACT_LRELU          # (a `downsample` takes its input as it is and stores act(out))
    fmt = lib.F16 if f16 else None
    x = torch.randn(n, h, h, ops.round8(cin), device=dev)
    x = torch.where(x > 0, x, 0.2 * x)                      # a post-activation tensor
    x[0, 0, 0, :8] = 0.0                                    # zeros are not positive
    xs = x.to(torch.float16).view(torch.bfloat16) if f16 else x.to(torch.bfloat16)
    res = xs if (cin == cout and stride == 1 and k == 3) else None
    ops.SignBits.want, ops.SignBits.last = True, None
    y = ops.conv_forward(xs, lay, res=res, fmt=fmt, res_post=res is not None)
    bits = ops.SignBits.take()
    two_tile = rows and cin == 64 and h == 128    # (64 channels at 128 columns: conv3x3_rows2_kernel, no sign bytes)
    native = (k == 3 and stride == 1 and not rows) or (rows and stride == 2)     # best effort: patch epilogue, stride-2 row kernel
    assert (bits is not None) == native
    if bits is None:                              # ... and ups_sign_pack packs what another kernel stored
        bits = torch.empty(tuple(y.shape[:-1]) + (y.shape[-1] // 8,), dtype=torch.uint8, device=dev)
        lib.call("ups_sign_pack", lib.ptr(y), lib.F16 if f16 else lib.BF16, y.numel() // 8, lib.ptr(bits), lib.stream())
    assert tuple(bits.shape) == tuple(y.shape[:-1]) + (y.shape[-1] // 8,)
    assert torch.equal(bits, _pack_signs(y))
    assert float((y.view(torch.float16).float() if f16 else y.float())[..., :cout].abs().max()) > 0
    if stride != 1:
        return                                    # (no activation on a downsample's input: its input gradient needs no signs)
    # the input gradient: bits of x instead of x
    gy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
    xb = _pack_signs(xs)
    ref = ops.conv_dgrad(gy, xs, lay, res=gy if res is not None else None)
    got = ops.conv_dgrad(gy, xs, lay, res=gy if res is not None else None, x_bits=xb)
    assert torch.equal(ref, got)
    if k == 3 and stride == 1 and cin % 16 == 0 and cin >= 32 and not two_tile and (rows or cin >= 64):   # where the bits are what is read: flip them all
        flipped = ops.conv_dgrad(gy, xs, lay, res=gy if res is not None else None, x_bits=~xb)
        assert not torch.equal(ref, flipped)


def test_bilinear_sign_bits(dev):
    lib, ops, R = _mods()
    g = torch.Generator().manual_seed(3)
    for f16 in (False, True):
        x = torch.randn(2, 16, 16, 64, generator=g).to(dev)
        xs = x.to(torch.float16).view(torch.bfloat16) if f16 else x.to(torch.bfloat16)
        fmt = lib.F16 if f16 else None
        ops.SignBits.want, ops.SignBits.last = False, None
        y0 = ops.BilinearFn.apply(xs, None, 0, 0.2, fmt, lib.ACT_LRELU)
        assert ops.SignBits.take() is None
        ops.SignBits.want = True
        y1 = ops.BilinearFn.apply(xs, None, 0, 0.2, fmt, lib.ACT_LRELU)
        bits = ops.SignBits.take()
        assert torch.equal(y0, y1) and bits is not None and torch.equal(bits, _pack_signs(y1))
