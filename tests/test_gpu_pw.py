"""The 1x1-convolution kernel of the eval path (csrc/conv_pw.hip) through the C-ABI hook `spk_op_conv1x1`.

Stands in for `Conv2d(k=1, bias=False) -> BatchNorm2d.eval() (-> + shortcut) (-> ReLU)` inside `net(x)`
(sykepic/compute/probability.py:189).  Oracle: the same chain in torch fp32 on the CPU, evaluated on the operands the
kernel gets (activations rounded to fp16 once; weights exact: the hi + lo images carry ~22 bits).  Tolerance: the fp16
rounding of the OUTPUT (2^-11 relative) plus fp32 accumulation noise - far inside the 1e-3 probability tolerance the
network-level tests hold; every tile configuration must agree, including on ragged pixel counts."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(n, h, w, cin, cout, stride, res, relu, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g)).half()
    wgt = (torch.rand(cout, cin, generator=g) * 2 - 1) * (6.0 / cin) ** 0.5
    scale = 0.5 + torch.rand(cout, generator=g)
    bias = torch.rand(cout, generator=g) - 0.5
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r = torch.relu(torch.randn(n, cout, ho, wo, generator=g)).half() if res else None
    ref = torch.nn.functional.conv2d(x.float(), wgt[:, :, None, None], stride=stride)
    ref = ref * scale[None, :, None, None] + bias[None, :, None, None]
    if r is not None:
        ref = ref + r.float()
    if relu:
        ref = torch.relu(ref)
    return x, wgt, scale, bias, r, ref


SHAPES = [
    # n, h, w, cin, cout, stride, res, relu        (what ResNet-50 has, scaled down, plus ragged pixel counts)
    (2, 56, 56, 64, 256, 1, True, True),
    (2, 56, 56, 64, 64, 1, False, True),
    (3, 28, 28, 256, 64, 1, False, True),
    (2, 28, 28, 128, 512, 1, True, True),
    (2, 56, 56, 256, 512, 2, False, False),      # downsample branch: stride 2, no ReLU
    (5, 14, 14, 256, 1024, 1, True, True),
    (3, 14, 14, 1024, 256, 1, False, True),
    (2, 13, 11, 512, 128, 1, False, True),       # 286 pixels: not a multiple of any tile
    (1, 7, 7, 512, 2048, 1, True, True),         # 49 pixels: one partial tile
    (3, 15, 9, 128, 128, 2, False, True),        # odd sizes under stride 2
    (1, 1, 1, 64, 64, 1, True, False),           # a single pixel
    (2, 7, 7, 2048, 512, 1, False, True),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_%dx%d_c%d-%d_s%d%s%s" % (s[0], s[1], s[2], s[3], s[4], s[5], "_res" if s[6] else "", "_relu" if s[7] else ""))
@pytest.mark.parametrize("split", [True, False], ids=["hi+lo", "plain"])
def test_every_configuration_matches_the_fp32_chain(shape, split):
    from sykepic_hip import ops
    n, h, w, cin, cout, stride, res, relu = shape
    x, wgt, scale, bias, r, ref = _case(n, h, w, cin, cout, stride, res, relu, seed=cin * 7 + cout)
    dev = "cuda:0"
    ran = 0
    # plain fp16 weights: the rounding of (scale * w) to fp16 is part of what that mode is (2^-11 per weight)
    tol_rel, tol_abs = (1.5e-3, 2e-3) if split else (6e-3, 8e-3)
    outs = []
    for cfg in range(-1, ops.conv1x1_num_configs()):
        try:
            y = ops.conv1x1(x.to(dev), wgt.to(dev), scale.to(dev), bias.to(dev), stride=stride, relu=relu,
                            res=r.to(dev) if r is not None else None, split=split, cfg=cfg)
        except RuntimeError as e:
            assert "does not fit" in str(e), str(e)
            continue
        y = y.float().cpu()
        assert torch.isfinite(y).all(), f"cfg {cfg}: an output element was never written"
        err = (y - ref).abs()
        bound = tol_abs + tol_rel * ref.abs()
        assert (err <= bound).all(), (cfg, float(err.max()), int((err > bound).sum()))
        outs.append((cfg, y))
        ran += cfg >= 0
    assert ran >= 2, "at least two configurations of the direct-operand kernel must cover every ResNet shape"
    # every configuration of the new kernel AND the implicit-GEMM kernel accumulate each output in the same order (K
    # steps of 32 ascending, hi product then lo product) and apply the same fp32 epilogue: bit-identical results, so
    # the per-problem choice of the tuner never shows in a probability
    for cfg, y in outs[1:]:
        assert torch.equal(y, outs[0][1]), f"cfg {cfg} differs from cfg {outs[0][0]}"


def test_saturates_instead_of_overflowing():
    """fp16 has no room above 65504: the epilogue clamps (ReLU floor and ceiling in one v_med3_f32) instead of writing
    inf, as the implicit-GEMM kernel does."""
    from sykepic_hip import ops
    dev = "cuda:0"
    x = torch.full((1, 64, 4, 4), 200.0).half()
    wgt = torch.full((64, 64), 10.0)
    y = ops.conv1x1(x.to(dev), wgt.to(dev), torch.ones(64).to(dev), torch.zeros(64).to(dev), relu=True, cfg=9)
    assert torch.isfinite(y).all() and float(y.max()) == 65504.0
    y = ops.conv1x1(x.to(dev), (-wgt).to(dev), torch.ones(64).to(dev), torch.zeros(64).to(dev), relu=False, cfg=9)
    assert torch.isfinite(y).all() and float(y.min()) == -65504.0


@pytest.mark.parametrize("shape", [(2, 56, 56, 64, 64), (3, 20, 17, 64, 128), (1, 9, 7, 128, 64), (2, 28, 28, 128, 128),
                                   (1, 1, 1, 64, 64)],
                         ids=lambda s: "n%d_%dx%d_c%d-256-%d" % s)
def test_chained_pair_equals_two_launches(shape):
    """A block-closing conv (cout 256, + shortcut, ReLU) and the conv that reads its output as ONE launch: the trunk
    tile goes from the epilogue's registers into the second product.  Both tensors are bit-identical to two stand-alone
    launches of the kernel (same fp16 values in between, same K order), on ragged pixel counts too."""
    from sykepic_hip import ops
    n, h, w, cin, coutz = shape
    dev = "cuda:0"
    x, wgt, scale, bias, r, ref = _case(n, h, w, cin, 256, 1, True, True, seed=cin + coutz)
    g = torch.Generator().manual_seed(coutz)
    wz = (torch.rand(coutz, 256, generator=g) * 2 - 1) * (6.0 / 256) ** 0.5
    sz, bz = 0.5 + torch.rand(coutz, generator=g), torch.rand(coutz, generator=g) - 0.5
    y, z = ops.conv1x1_chain(x.to(dev), wgt.to(dev), scale.to(dev), bias.to(dev), r.to(dev), wz.to(dev), sz.to(dev), bz.to(dev))
    y1 = ops.conv1x1(x.to(dev), wgt.to(dev), scale.to(dev), bias.to(dev), relu=True, res=r.to(dev), split=False, cfg=-1)
    z1 = ops.conv1x1(y1, wz.to(dev), sz.to(dev), bz.to(dev), relu=True, split=False, cfg=-1)
    assert torch.isfinite(y.float()).all() and torch.isfinite(z.float()).all()
    assert torch.equal(y, y1)
    assert torch.equal(z, z1)
    err = (y.float().cpu() - ref).abs()
    assert (err <= 8e-3 + 6e-3 * ref.abs()).all()


DUAL_SHAPES = [
    # n, ho, wo, cin (block-closing conv), cin2 (shortcut conv), cout, stride2     (ResNet-50's four stage openers, scaled down)
    (2, 56, 56, 64, 64, 256, 1),
    (3, 28, 28, 128, 256, 512, 2),
    (5, 14, 14, 256, 512, 1024, 2),
    (7, 7, 7, 512, 1024, 2048, 2),
    (2, 13, 9, 256, 256, 512, 2),           # ragged pixel count, odd second source (25 x 17)
]


@pytest.mark.parametrize("shape", DUAL_SHAPES, ids=lambda s: "n%d_%dx%d_c%d+%d-%d_s%d" % s)
def test_every_configuration_of_the_fused_shortcut_conv_agrees(shape):
    """The block-closing conv + shortcut conv as one K-concatenated GEMM (PwConvArgs::x2) on every configuration that fits
    - conv_pw.hip's and, for the deep stages, conv_pwr.hip's: within fp16 output rounding of the fp32 chain, and the same
    bits from all of them (the tuner's choice must not show)."""
    from sykepic_hip import ops
    n, ho, wo, cin, cin2, cout, s2 = shape
    g = torch.Generator().manual_seed(cin * 3 + cin2)
    h2, w2 = (ho - 1) * s2 + 1 + (s2 - 1), (wo - 1) * s2 + 1          # (rows: one spare line under stride 2)
    x = torch.relu(torch.randn(n, cin, ho, wo, generator=g)).half()
    x2 = torch.relu(torch.randn(n, cin2, h2, w2, generator=g)).half()
    w1 = (torch.rand(cout, cin, generator=g) * 2 - 1) * (6.0 / cin) ** 0.5
    w2_ = (torch.rand(cout, cin2, generator=g) * 2 - 1) * (6.0 / cin2) ** 0.5
    sc1, sc2 = 0.5 + torch.rand(cout, generator=g), 0.5 + torch.rand(cout, generator=g)
    b1, b2 = torch.rand(cout, generator=g) - 0.5, torch.rand(cout, generator=g) - 0.5
    F = torch.nn.functional
    ref = F.conv2d(x.float(), w1[:, :, None, None]) * sc1[None, :, None, None] + b1[None, :, None, None]
    ref = ref + F.conv2d(x2.float(), w2_[:, :, None, None], stride=s2)[:, :, :ho, :wo] * sc2[None, :, None, None] + b2[None, :, None, None]
    ref = torch.relu(ref)
    dev = "cuda:0"
    outs = []
    for cfg in range(ops.conv1x1_num_configs()):
        try:
            y = ops.conv1x1_dual(x.to(dev), w1.to(dev), sc1.to(dev), b1.to(dev), x2.to(dev), w2_.to(dev), sc2.to(dev), b2.to(dev),
                                 stride2=s2, relu=True, split=False, cfg=cfg)
        except RuntimeError as e:
            assert "does not fit" in str(e), str(e)
            continue
        y = y.float().cpu()
        assert torch.isfinite(y).all(), f"cfg {cfg}: an output element was never written"
        err = (y - ref).abs()
        bound = 8e-3 + 6e-3 * ref.abs()      # plain fp16 weights with the BatchNorm scales folded in (as the `plain` rows above)
        assert (err <= bound).all(), (cfg, float(err.max()), int((err > bound).sum()))
        outs.append((cfg, y))
    assert len(outs) >= 2, [c for c, _ in outs]
    for cfg, y in outs[1:]:
        assert torch.equal(y, outs[0][1]), f"cfg {cfg} differs from cfg {outs[0][0]}"
