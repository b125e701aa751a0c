"""Single-layer parity of the training kernels (through the C-ABI test hooks ``spk_op_*``) with torch autograd
evaluated on the SAME bf16-rounded operands: what is left is the rounding of the kernel's own output (bf16:
2^-9 per element, ~1.1e-3 relative L2; float32 outputs: accumulation order only).

Reference lines these stand in for: ``out = net(x)`` in train mode and ``loss.backward()``,
/root/reference/sykepic/train/train.py:240,242 (torch Conv2d / BatchNorm2d / ReLU autograd).

Tolerances (relative L2 over the whole tensor): bf16 outputs 5e-3, float32 outputs 1e-3; maximum element error
is bounded too, so that a wrong halo tap / parity class / split-K slab on a few elements cannot hide in the norm.
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _rand_bf16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).bfloat16()


def _check_bf16(got, want, what, tol=5e-3):
    """got: bf16 tensor from the GPU, want: fp32 reference.  Relative L2 and a per-element bound of 1.5 bf16
    ulps of the reference magnitude plus a small absolute floor tied to the tensor scale."""
    got, want = got.float().cpu(), want.float()
    assert torch.isfinite(got).all(), f"{what}: non-finite output (unwritten elements?)"
    r = _rel(got, want)
    scale = float(want.abs().max())
    err = (got - want).abs()
    bound = want.abs() * 2.0 ** -7 + scale * 2e-3
    worst = float((err - bound).max())
    assert r < tol, f"{what}: relative L2 {r:.3e}"
    assert worst <= 0, f"{what}: element error exceeds bound by {worst:.3e} (scale {scale:.3e})"
    return r


# (n, h, w, cin, cout, k, stride, pad): stride 1 and stride 2 (all four output parity classes, odd and even
# sizes), 1x1 and 3x3, the ResNet-50 channel widths at small spatial sizes
CONV_CASES = [
    (4, 14, 14, 64, 64, 3, 1, 1),
    (3, 9, 11, 128, 64, 3, 1, 1),
    (4, 14, 14, 256, 64, 1, 1, 0),
    (2, 7, 7, 64, 256, 1, 1, 0),
    (4, 16, 16, 128, 128, 3, 2, 1),
    (3, 15, 13, 64, 128, 3, 2, 1),
    (4, 16, 16, 256, 512, 1, 2, 0),
    (3, 15, 13, 64, 128, 1, 2, 0),
    (2, 28, 28, 128, 128, 3, 1, 1),
    (1, 30, 30, 64, 64, 3, 2, 1),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%ds%dp%d" % c)
def test_conv_dgrad_matches_autograd(case):
    from sykepic_hip import ops
    n, h, w, cin, cout, k, stride, pad = case
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    wt = _rand_bf16((cout, cin, k, k), 1, (2.0 / (cin * k * k)) ** 0.5).float()
    dy = _rand_bf16((n, cout, oh, ow), 2)
    want = torch.nn.grad.conv2d_input((n, cin, h, w), wt, dy.float(), stride, pad)
    got = ops.conv_dgrad(dy.cuda(), wt.cuda(), (h, w), stride, pad)
    r = _check_bf16(got, want, "dgrad")
    # accumulating form (a tensor with two consumers: shortcut + main branch)
    base = _rand_bf16((n, cin, h, w), 3)
    got2 = ops.conv_dgrad(dy.cuda(), wt.cuda(), (h, w), stride, pad, accumulate_into=base.cuda())
    r2 = _check_bf16(got2, want + base.float(), "dgrad (accumulate)")
    print(f"dgrad {case}: rel-L2 {r:.2e} / accumulate {r2:.2e}")


@pytest.mark.parametrize("case", CONV_CASES + [(4, 32, 32, 3, 64, 7, 2, 3), (3, 37, 29, 3, 64, 7, 2, 3),
                                               (16, 28, 28, 128, 128, 3, 1, 1)],
                         ids=lambda c: "n%d_%dx%d_c%d-%d_k%ds%dp%d" % c)
def test_conv_wgrad_matches_autograd(case):
    from sykepic_hip import ops
    n, h, w, cin, cout, k, stride, pad = case
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = _rand_bf16((n, cin, h, w), 4)
    dy = _rand_bf16((n, cout, oh, ow), 5)
    want = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, k, k), dy.float(), stride, pad)
    got = ops.conv_wgrad(x.cuda(), dy.cuda(), k, stride, pad).cpu()
    assert torch.isfinite(got).all()
    r = _rel(got, want)
    worst = float((got - want).abs().max() / want.abs().max())
    print(f"wgrad {case}: rel-L2 {r:.2e}, max element error / max {worst:.2e}")
    assert r < 1e-3 and worst < 1e-3


BN_CASES = [(6, 64, 8, 8, True, False), (5, 256, 7, 5, True, True), (4, 128, 2, 2, True, True),
            (8, 512, 1, 1, False, True), (3, 2048, 2, 2, True, True), (300, 64, 3, 3, True, False)]


def _mask_bits(pos):
    """[N,C,H,W] bool -> [M][C/8] uint8, bit j = channel 8*cc + j (the layout bn_apply writes)."""
    n, c, h, w = pos.shape
    p = pos.permute(0, 2, 3, 1).reshape(n * h * w, c // 8, 8).to(torch.uint8)
    weights = (2 ** torch.arange(8, dtype=torch.int32)).to(torch.uint8)
    return (p * weights).sum(-1).to(torch.uint8)


@pytest.mark.parametrize("case", BN_CASES, ids=lambda c: "n%d_c%d_%dx%d_relu%d_res%d" % tuple(int(v) for v in c))
def test_bn_backward_matches_autograd(case):
    """bn_bwd_reduce / finalize / apply on given operands.  The statistics handed to the kernel are the exact
    batch statistics of the bf16 `raw` tensor, so autograd through F.batch_norm(raw) is the same function."""
    from sykepic_hip import ops
    n, c, h, w, relu, with_res = case
    raw = (_rand_bf16((n, c, h, w), 6).float() * 1.7 + 0.4).bfloat16()
    g = _rand_bf16((n, c, h, w), 7, 0.01)
    gamma = torch.randn(c, generator=torch.Generator().manual_seed(8)) * 0.5 + 1.0
    beta = torch.randn(c, generator=torch.Generator().manual_seed(9)) * 0.3
    res = _rand_bf16((n, c, h, w), 10) if with_res else None
    y = raw.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    v = F.batch_norm(y, None, None, gm, bt, True, 0.1, 1e-5)
    if with_res:
        v = v + res.float()
    pos = (v > 0) if relu else torch.ones_like(v, dtype=torch.bool)
    dz = g.float() * pos
    v.backward(dz)
    mean = raw.double().mean((0, 2, 3))
    var = raw.double().var((0, 2, 3), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    got = ops.bn_backward(g.cuda(), _mask_bits(pos).cuda(), raw.cuda(), mean.float().cuda(), invstd.cuda(),
                          gamma.cuda(), relu=relu, want_res=with_res)
    r = _check_bf16(got["dy"], y.grad, "bn backward dy")
    rg, rb = _rel(got["dgamma"].cpu(), gm.grad), _rel(got["dbeta"].cpu(), bt.grad)
    print(f"bn_bwd {case}: dy rel-L2 {r:.2e}, dgamma {rg:.2e}, dbeta {rb:.2e}")
    assert rg < 1e-4 and rb < 1e-4
    if with_res:
        assert torch.equal(got["g_res"].float().cpu(), dz.bfloat16().float())
        prev = _rand_bf16((n, c, h, w), 11, 0.01)
        acc = ops.bn_backward(g.cuda(), _mask_bits(pos).cuda(), raw.cuda(), mean.float().cuda(), invstd.cuda(),
                              gamma.cuda(), relu=relu, g_res=prev.cuda(), res_accumulate=True)
        assert torch.equal(acc["g_res"].float().cpu(), (prev.float() + dz).bfloat16().float())


@pytest.mark.parametrize("case", [(6, 14, 14, 64, 64, 3, 1, 1, True, False), (5, 14, 14, 64, 256, 1, 1, 0, True, True),
                                  (4, 16, 16, 128, 128, 3, 2, 1, True, False), (4, 33, 33, 3, 64, 7, 2, 3, True, False),
                                  (4, 16, 16, 256, 512, 1, 2, 0, False, False)],
                         ids=lambda c: "n%d_%dx%d_c%d-%d_k%ds%dp%d_relu%d_res%d" % tuple(int(v) for v in c))
def test_conv_bn_train_forward_matches_torch(case):
    """conv (raw output + statistics partials in the epilogue) -> bn_finalize -> bn_apply, against
    F.conv2d + F.batch_norm(training=True) on the same bf16 operands: batch statistics, running statistics
    (momentum 0.1, unbiased variance), the bf16 raw tensor, the normalised output and the ReLU mask."""
    from sykepic_hip import ops
    n, h, w, cin, cout, k, stride, pad, relu, with_res = case
    x = _rand_bf16((n, cin, h, w), 12).float().abs().bfloat16()
    wt = _rand_bf16((cout, cin, k, k), 13, (2.0 / (cin * k * k)) ** 0.5).float()
    gamma = torch.rand(cout, generator=torch.Generator().manual_seed(14)) + 0.5
    beta = torch.randn(cout, generator=torch.Generator().manual_seed(15)) * 0.2
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = _rand_bf16((n, cout, oh, ow), 16) if with_res else None
    rm, rv = torch.zeros(cout), torch.ones(cout)
    y32 = F.conv2d(x.float(), wt, None, stride, pad)
    mean = y32.double().mean((0, 2, 3))
    var = y32.double().var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    v = (y32.bfloat16().double() - mean.view(1, -1, 1, 1)) * (invstd * gamma.double()).view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    if with_res:
        v = v + res.double()
    if relu:
        v = v.clamp_min(0)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    got = ops.conv_bn_train_forward(x.cuda(), wt.cuda(), gamma.cuda(), beta.cuda(), rm_d, rv_d,
                                    res.cuda() if with_res else None, relu, stride, pad)
    _check_bf16(got["raw"], y32, "raw conv output")
    assert torch.allclose(got["mean"].cpu().double(), mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(got["invstd"].cpu().double(), invstd, rtol=1e-4)
    m = n * oh * ow
    assert torch.allclose(rm_d.cpu().double(), 0.1 * mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(rv_d.cpu().double(), 0.9 + 0.1 * var * m / (m - 1), rtol=1e-4)
    # the GPU normalises ITS bf16 raw tensor: compare on that (a 1-ulp difference of raw is not an error of bn_apply)
    vg = (got["raw"].double().cpu() - mean.view(1, -1, 1, 1)) * (invstd * gamma.double()).view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    if with_res:
        vg = vg + res.double()
    pos = vg > 0
    if relu:
        vg = vg.clamp_min(0)
    _check_bf16(got["out"], vg.float(), "normalised output", tol=4e-3)
    if relu:
        near = vg.abs() < 1e-3 * float(vg.abs().max())   # mask may differ only where the value is ~0
        gm = got["mask"].cpu()
        assert torch.equal(gm[~_any8(near)], _mask_bits(pos)[~_any8(near)])


def _any8(flag):
    n, c, h, w = flag.shape
    return flag.permute(0, 2, 3, 1).reshape(n * h * w, c // 8, 8).any(-1)


@pytest.mark.parametrize("case", [(4, 14, 14, 64, 64, 3, 1, True, False), (3, 9, 11, 128, 64, 3, 1, True, True),
                                  (4, 14, 14, 256, 64, 1, 0, True, True), (2, 28, 28, 128, 128, 3, 1, False, False),
                                  (5, 7, 7, 64, 256, 1, 0, True, False), (2, 33, 21, 64, 128, 3, 1, True, True)],
                         ids=lambda c: "n%d_%dx%d_c%d-%d_k%dp%d_relu%d_acc%d" % tuple(int(v) for v in c))
def test_dgrad_with_the_producer_bn_reduction_matches_autograd(case):
    """The data gradient of a stride-1 conv whose epilogue also makes the BatchNorm-backward sums of the layer that
    produced its input (round 4: that layer's reduce pass is gone), then that layer's finalize + apply.  Oracle: autograd
    through ReLU(BatchNorm(raw)) -> conv on the same bf16 operands.  The sums come from the fp32 gradient BEFORE it is
    rounded to bf16, so dgamma / dbeta are closer to autograd than the three-pass form (which sums the rounded tensor);
    with accumulate the stored gradient already holds another consumer's contribution (a shortcut add)."""
    from sykepic_hip import ops
    n, h, w, cin, cout, k, pad, relu, acc = case
    raw = (_rand_bf16((n, cin, h, w), 21).float() * 1.3 + 0.2).bfloat16()
    gamma = torch.randn(cin, generator=torch.Generator().manual_seed(22)) * 0.5 + 1.0
    beta = torch.randn(cin, generator=torch.Generator().manual_seed(23)) * 0.3
    wgt = _rand_bf16((cout, cin, k, k), 24, (2.0 / (cin * k * k)) ** 0.5)
    dy = _rand_bf16((n, cout, h, w), 25, 0.01)
    other = _rand_bf16((n, cin, h, w), 26, 0.01) if acc else None
    y = raw.float().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    v = F.batch_norm(y, None, None, gm, bt, True, 0.1, 1e-5)
    a = torch.relu(v) if relu else v
    a.retain_grad()
    out = F.conv2d(a, wgt.float(), padding=pad)
    loss = (out * dy.float()).sum() + ((a * other.float()).sum() if acc else 0.0)
    loss.backward()
    pos = (v > 0) if relu else torch.ones_like(v, dtype=torch.bool)
    mean = raw.double().mean((0, 2, 3))
    invstd = (1.0 / torch.sqrt(raw.double().var((0, 2, 3), unbiased=False) + 1e-5)).float()
    got = ops.conv_dgrad_bn_backward(dy.cuda(), wgt.cuda(), (h, w), raw.cuda(), _mask_bits(pos).cuda(), mean.float().cuda(),
                                     invstd.cuda(), gamma.cuda(), pad=pad, relu=relu,
                                     accumulate_into=other.cuda() if acc else None)
    rg = _check_bf16(got["g"], a.grad, "input gradient")
    rd = _check_bf16(got["dy"], y.grad, "producer raw-output gradient")
    eg, eb = _rel(got["dgamma"].cpu(), gm.grad), _rel(got["dbeta"].cpu(), bt.grad)
    print(f"dgrad+bn_bwd {case}: g {rg:.2e}, dy {rd:.2e}, dgamma {eg:.2e}, dbeta {eb:.2e}")
    assert eg < 1e-4 and eb < 1e-4
    # the stored gradient is the same tensor the plain data-gradient launch writes
    plain = ops.conv_dgrad(dy.cuda(), wgt.cuda(), (h, w), 1, pad, accumulate_into=other.cuda() if acc else None)
    assert torch.equal(plain, got["g"])
    if acc:
        # the shortcut gradient picked up at its source: `other` = g3 * ReLU bits of a block-closing conv.  Handing the
        # kernel g3 and the bits gives, bit for bit, what accumulating into the pre-masked tensor gives
        g3 = _rand_bf16((n, cin, h, w), 27, 0.01)
        pos3 = torch.rand((n, cin, h, w), generator=torch.Generator().manual_seed(28)) > 0.4
        pre = (g3.float() * pos3).bfloat16()
        a = ops.conv_dgrad_bn_backward(dy.cuda(), wgt.cuda(), (h, w), raw.cuda(), _mask_bits(pos).cuda(), mean.float().cuda(),
                                       invstd.cuda(), gamma.cuda(), pad=pad, relu=relu, accumulate_into=pre.cuda())
        b = ops.conv_dgrad_bn_backward(dy.cuda(), wgt.cuda(), (h, w), raw.cuda(), _mask_bits(pos).cuda(), mean.float().cuda(),
                                       invstd.cuda(), gamma.cuda(), pad=pad, relu=relu, res_src=g3.cuda(),
                                       res_bits=_mask_bits(pos3).cuda())
        for key in ("g", "dy", "dgamma", "dbeta"):
            assert torch.equal(a[key], b[key]), key
