"""A whole identity bottleneck block as ONE kernel (csrc/conv_bneck.hip) through `spk_op_bottleneck`.

Stands in for torchvision's `Bottleneck.forward` without a downsample branch - conv1x1 + BN + ReLU, conv3x3 + BN + ReLU,
conv1x1 + BN, + x, ReLU - inside `net(x)` (sykepic/compute/probability.py:189).  Two checks per shape:
  * against the same block as the eval path's three launches (conv_pw / conv_c3 / conv_pw): the fused kernel keeps their K
    orders and epilogue arithmetic and rounds the two mid tensors to fp16 exactly where they are stored, so the outputs
    are BIT-IDENTICAL;
  * against torch fp32 on the CPU with the mid tensors rounded to fp16 at the same two points."""

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # n, hw, cm
    (3, 14, 256),      # ResNet-50 stage 3 (a block owns a whole image)
    (1, 14, 256),
    (9, 14, 256),      # more blocks than one XCD's share of a small grid
    (3, 28, 128),      # stage 2: two bands per image, one halo row each side (top band: none above, bottom band: none below)
    (1, 28, 128),
]


def _block(n, hw, cm, seed):
    g = torch.Generator().manual_seed(seed)
    c4 = 4 * cm
    x = torch.relu(torch.randn(n, c4, hw, hw, generator=g)).half()
    w1 = (torch.rand(cm, c4, 1, 1, generator=g) * 2 - 1) * (6.0 / c4) ** 0.5
    w2 = (torch.rand(cm, cm, 3, 3, generator=g) * 2 - 1) * (6.0 / (9 * cm)) ** 0.5
    w3 = (torch.rand(c4, cm, 1, 1, generator=g) * 2 - 1) * (6.0 / cm) ** 0.5
    bns = []
    for c in (cm, cm, c4):
        bns.append((0.5 + torch.rand(c, generator=g), torch.rand(c, generator=g) - 0.5))
    return x, w1, w2, w3, bns


def _ref(x, w1, w2, w3, bns):
    F = torch.nn.functional
    # weights as the kernels see them (fp16), activations fp32 between the rounding points
    bn = lambda t, p: t * p[0][None, :, None, None] + p[1][None, :, None, None]  # noqa: E731
    y1 = torch.relu(bn(F.conv2d(x.float(), w1.half().float()), bns[0])).half().float()
    y2 = torch.relu(bn(F.conv2d(y1, w2.half().float(), padding=1), bns[1])).half().float()
    return torch.relu(bn(F.conv2d(y2, w3.half().float()), bns[2]) + x.float())


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_%dx%d_cm%d" % (s[0], s[1], s[1], s[2]))
def test_fused_bottleneck_equals_the_three_launches(shape):
    from sykepic_hip import ops
    n, hw, cm = shape
    x, w1, w2, w3, bns = _block(n, hw, cm, seed=hw + cm + n)
    dev = "cuda:0"
    d = lambda t: t.to(dev)  # noqa: E731
    dbns = [(d(a), d(b)) for a, b in bns]
    three = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=False).float().cpu()
    fused = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=True).float().cpu()
    assert torch.isfinite(fused).all(), "an output element was never written"
    ref = _ref(x, w1, w2, w3, bns)
    err = (three - ref).abs()
    bound = 2e-2 + 1e-2 * ref.abs()       # (a flipped fp16 rounding of a mid value moves an output by ~ its weight)
    assert (err <= bound).all(), float(err.max())
    assert torch.equal(fused, three), (float((fused - three).abs().max()), int((fused != three).sum()))


TAIL_SHAPES = [
    # n, hw, cm, coutz (0: no chained conv)
    (2, 56, 64, 0),       # ResNet-50 stage 1: bands of 4 rows, 14 per image
    (3, 56, 64, 64),      # + the next block's conv1 (256 -> 64) from the output tile in registers
    (1, 56, 64, 128),     # + the next STAGE's conv1 (256 -> 128)
]


@pytest.mark.parametrize("shape", TAIL_SHAPES, ids=lambda s: "n%d_%dx%d_cm%d_z%d" % (s[0], s[1], s[1], s[2], s[3]))
def test_block_tail_kernel_equals_the_launches_it_replaces(shape):
    """conv2 + conv3 + shortcut (+ the conv that reads the block's output) as ONE kernel (conv_btail_kernel) on the stage whose
    trunk is too wide for the whole-block kernel: bit-identical to conv_c3 -> conv_pw (-> conv_pw) on the same conv1 output."""
    from sykepic_hip import ops
    n, hw, cm, coutz = shape
    x, w1, w2, w3, bns = _block(n, hw, cm, seed=hw + cm + n + coutz)
    g = torch.Generator().manual_seed(7 + coutz)
    c4 = 4 * cm
    wz = ((torch.rand(coutz, c4, 1, 1, generator=g) * 2 - 1) * (6.0 / c4) ** 0.5) if coutz else None
    bnz = (0.5 + torch.rand(coutz, generator=g), torch.rand(coutz, generator=g) - 0.5) if coutz else None
    dev = "cuda:0"
    d = lambda t: t.to(dev)  # noqa: E731
    dbns = [(d(a), d(b)) for a, b in bns]
    kw = dict(wz=d(wz), bnz=(d(bnz[0]), d(bnz[1]))) if coutz else {}
    ref = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=0, **kw)
    got = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=2, **kw)
    if not coutz:
        ref, got = (ref,), (got,)
    for name, r, t in zip(("out", "z"), ref, got):
        r, t = r.float().cpu(), t.float().cpu()
        assert torch.isfinite(t).all(), f"{name}: an element was never written"
        assert torch.equal(t, r), (name, float((t - r).abs().max()), int((t != r).sum()))
    want = _ref(x, w1, w2, w3, bns)
    err = (got[0].float().cpu() - want).abs()
    assert (err <= 2e-2 + 1e-2 * want.abs()).all(), float(err.max())
