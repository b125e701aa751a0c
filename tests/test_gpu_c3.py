"""The 3x3 stride-1 pad-1 convolution kernel of the eval path (csrc/conv_c3.hip) through `spk_op_conv3x3`.

Stands in for `Conv2d(k=3, padding=1, bias=False) -> BatchNorm2d.eval() (-> ReLU)`, the middle conv of a bottleneck
block inside `net(x)` (sykepic/compute/probability.py:189).  Oracle: torch fp32 on the CPU on the same fp16-rounded
activations.  The kernel keeps a halo window of the input in LDS with explicit zero columns / rows for the padding, so
the cases lean on the borders: tiles that start and end mid-row, span several images (7x7: two and a half per tile),
a single image smaller than a tile, non-square maps."""

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # n, h, w, cin, cout
    (5, 14, 14, 256, 256),
    (7, 7, 7, 512, 512),
    (1, 7, 7, 64, 256),        # one image, fewer pixels than a tile
    (3, 28, 28, 128, 256),
    (2, 13, 9, 64, 256),       # non-square, odd sizes
    (4, 3, 5, 128, 512),       # maps smaller than a pixel tile: every tile spans images
    (2, 1, 1, 64, 256),        # 1x1 maps: only the centre tap is inside the image
    (1, 56, 56, 64, 256),
    (3, 56, 56, 64, 64),       # ResNet-50 stage 1 (one 64-channel chunk: one window stage, 64-cout blocks)
    (5, 28, 28, 128, 128),     # stage 2
    (2, 10, 6, 64, 128),
]


def _ref(x, wgt, scale, bias, relu, res=None):
    y = torch.nn.functional.conv2d(x.float(), wgt, padding=1) * scale[None, :, None, None] + bias[None, :, None, None]
    if res is not None:
        y = y + res.float()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_%dx%d_c%d-%d" % s)
@pytest.mark.parametrize("split", [False, True], ids=["plain", "hi+lo"])
def test_every_configuration_matches_the_fp32_conv(shape, split):
    from sykepic_hip import ops
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + 3 * cout + h)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g)).half()
    wgt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * (6.0 / (9 * cin)) ** 0.5
    scale = 0.5 + torch.rand(cout, generator=g)
    bias = torch.rand(cout, generator=g) - 0.5
    relu = (h + cout) % 3 != 0
    # every other shape carries a shortcut operand (the block-closing conv of a ResNet-18/34 basic block)
    res = torch.randn(n, cout, h, w, generator=g).half() if (n + h) % 2 else None
    ref = _ref(x, wgt, scale, bias, relu, res)
    tol_rel, tol_abs = (1.5e-3, 2e-3) if split else (6e-3, 8e-3)
    dev = "cuda:0"
    outs = []
    for cfg in range(-1, ops.conv3x3_num_configs()):
        try:
            y = ops.conv3x3(x.to(dev), wgt.to(dev), scale.to(dev), bias.to(dev), relu=relu, split=split, cfg=cfg,
                            res=res.to(dev) if res is not None else None)
        except RuntimeError as e:
            assert "does not fit" in str(e), str(e)
            continue
        y = y.float().cpu()
        assert torch.isfinite(y).all(), f"cfg {cfg}: an output element was never written"
        err = (y - ref).abs()
        bound = tol_abs + tol_rel * ref.abs()
        assert (err <= bound).all(), (cfg, float(err.max()), int((err > bound).sum()))
        if cfg >= 0:
            outs.append((cfg, y))
    # (1x1 maps: one zero row per image makes the window outgrow LDS - no configuration fits, the eval path then runs
    # the implicit GEMM, whose result is checked above as cfg -1)
    assert len(outs) >= 2 or h * w == 1
    # every configuration of the window kernel sums in the same order (chunk -> tap -> half): bit-identical
    for cfg, y in outs[1:]:
        assert torch.equal(y, outs[0][1]), f"cfg {cfg} differs from cfg {outs[0][0]}"
