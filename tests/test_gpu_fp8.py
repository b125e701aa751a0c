"""fp8 (OCP e4m3) eval mode of the EfficientNet MBConv blocks — BASELINE config 5 ("EfficientNet-B4 fp8
inference").  Two levels:

* the fp8 pointwise-conv kernel (csrc/pw_fp8.hip, through the C-ABI hook ``spk_op_pw_fp8``) against a CPU
  restatement of the SAME arithmetic with torch's float8_e4m3fn (operands rounded to e4m3 exactly where the kernel
  rounds them, fp32 accumulation): what is left is accumulation order, so the bound is tight (2e-3 of the tensor
  scale for fp16 outputs, at most one e4m3 ulp on <1 % of the bytes for e4m3 outputs);
* the whole network in fp8 mode against the fp32 oracle / the reference's golden probabilities.  Three mantissa
  bits do NOT reach the reference's 1e-3 probability tolerance: the measured distance is asserted as a bound and
  reported, the fp16 path stays the parity mode (tests/test_gpu_effnet.py) and must be unchanged by a detour
  through the fp8 mode.
"""

import numpy as np
import pytest
import torch

from sykepic_hip import synth

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def _q(t):
    """fp32 -> e4m3 -> fp32 with the kernel's saturation."""
    return t.clamp(-448, 448).to(F8).float()


def _bytes(t):
    return t.clamp(-448, 448).to(F8).view(torch.uint8)


CASES = [
    # M, cin, cout, a_fp8, out_fp8, gated, res, act
    (300, 24, 144, False, True, False, False, 2),      # expand: fp16 trunk -> e4m3 expanded tensor, SiLU
    (1000, 56, 336, False, True, False, False, 2),
    (257, 448, 2688, False, True, False, False, 2),
    (300, 144, 24, True, False, True, False, 0),       # project: e4m3 x gate -> fp16 trunk
    (640, 336, 56, True, False, True, True, 0),        # ... + shortcut
    (129, 2688, 448, True, False, True, True, 0),
    (200, 192, 32, True, False, False, False, 0),
    (64, 64, 64, False, False, False, False, 1),
    # more tiles than resident blocks: the persistent blocks walk several tiles, loads run ahead across tile borders
    (40000, 24, 144, False, True, False, False, 2),
    (100000, 192, 32, True, False, True, True, 0),
    (33000, 160, 960, False, True, False, False, 2),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "M%d_%d-%d_a8%d_o8%d_g%d_r%d_act%d" % tuple(int(v) for v in c))
def test_pw_fp8_kernel_matches_e4m3_reference(case):
    from sykepic_hip import ops
    m, cin, cout, a_fp8, out_fp8, gated, with_res, act = case
    g = torch.Generator().manual_seed(m + cin)
    hw = 40
    a_scale, y_scale = 0.037, 0.021
    w = torch.randn((cout, cin), generator=g) * (2.0 / cin) ** 0.5
    w[min(3, cout - 1)] *= 40.0                          # per-output-channel scales must absorb an outlier row
    bn_s = torch.rand(cout, generator=g) + 0.5
    bn_b = torch.randn(cout, generator=g) * 0.2
    xf = torch.randn((m, cin), generator=g) * 2.0
    gate = torch.rand(((m + hw - 1) // hw, cin), generator=g) if gated else None
    res = (torch.randn((m, cout), generator=g)).half() if with_res else None
    # ---- reference: the same roundings, fp32 accumulate (double here) ----
    ws = (w.abs().amax(1) / 448.0).clamp_min(1e-30)
    wq = _q(w / ws[:, None])
    if a_fp8:
        xb = _bytes(xf / a_scale)
        xq = xb.view(F8).float()
        if gated:
            img = torch.arange(m) // hw
            xq = _q(xq * gate[img])
        x_dev = xb.cuda()
    else:
        xh = xf.half()
        xq = _q(xh.float() * (1.0 / a_scale))
        x_dev = xh.cuda()
    y = (xq.double() @ wq.double().t()) * (bn_s * ws * a_scale).double() + bn_b.double()
    if with_res:
        y = y + res.double()
    if act == 1:
        y = y.clamp_min(0)
    elif act == 2:
        y = y * torch.sigmoid(y)
    got = ops.pw_fp8(x_dev, w.cuda(), bn_s.cuda(), bn_b.cuda(), act=act, a_scale=a_scale, y_scale=y_scale,
                     out_fp8=out_fp8, res=res.cuda() if with_res else None, gate=gate.cuda() if gated else None, hw=hw).cpu()
    if out_fp8:
        want_b = _bytes((y / y_scale).float())
        same = float((got == want_b).float().mean())
        gv, wv = got.view(F8).float(), want_b.view(F8).float()
        # a differing byte is a value that sat on a rounding boundary: either one e4m3 ulp apart, or (small values
        # of large cancelling sums) within the fp32 accumulation noise of the tensor scale
        ulp = torch.maximum(wv.abs(), torch.tensor(2.0 ** -6)) * 2.0 ** -3
        diff = (gv - wv).abs()
        ok = (diff <= ulp * (1 + 1e-6)) | (diff <= 2e-3 * float(wv.abs().max()))
        print(f"{case}: {same * 100:.2f} % identical e4m3 bytes, worst difference {float((diff / ulp).max()):.2f} ulp, "
              f"{float(diff.max() / wv.abs().max()):.2e} of the tensor maximum")
        assert same > 0.995 and bool(ok.all())
    else:
        err = float((got.double() - y).abs().max() / y.abs().max())
        print(f"{case}: max error / max |y| = {err:.2e}")
        assert err < 2e-3


# fp8 default mode vs the fp16 parity mode, 256 fresh images (measured, tests/archive/diagnostics/fp8_decided.py: B0 top-1 0.953 /
# 0.987 at margin >= 0.05, |dp| median 2.3e-3; B4 0.844 / 1.000 at margin >= 0.2 (0.87 at >= 0.05), median 4.2e-3)
DECIDED = {"efficientnet_b0": 0.05, "efficientnet_b4": 0.2}
# (B4: ~24 of the 256 random-weight images have a margin >= 0.2; one image is 4 points.  Measured over the rounds 0.917-1.00
# on them; the statement that matters - top-1 on a TRAINED net's decided images - is tests/test_gpu_trained.py: 256 / 256)
BOUNDS = {"efficientnet_b0": (0.90, 0.95, 5e-3), "efficientnet_b4": (0.78, 0.90, 8e-3)}


def _state(network, golden_dir):
    from effnet_util import calibrated_state
    gold = np.load(golden_dir / "net_pass_effnet.npz")
    return gold, calibrated_state(network, 224, gold)


@pytest.mark.parametrize("network", ["efficientnet_b0", "efficientnet_b4"])
def test_efficientnet_fp8_mode(golden_dir, network):
    """The fp8 mode end to end: calibrate on 16 images, classify the golden images and 32 fresh ones.
    Asserted: the expanded tensors really are e4m3 (every value of a read-back tensor is a representable e4m3
    multiple of its scale), they track the fp16 path's tensors (relative L2 within fp8's 2^-4 per element), the
    arg-max of the reference survives on the images whose margin is above the measured fp8 error, max |dp| stays
    within the stated fp8 bound (NOT the reference's 1e-3: three mantissa bits cannot reach it — the bound below is
    the measured one with head-room), and switching the mode off restores the fp16 results bit for bit."""
    from oracle import refnet
    from sykepic_hip.net import HipNet
    gold, (g, sd, ref) = _state(network, golden_dir)
    tag = f"{network}_224"
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    n = len(gold[f"{tag}_rois_in"])
    x = torch.from_numpy(synth.synth_images(n, 3, 224, 224, seed=0)).cuda()
    fresh = torch.cat([torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=21 + i)) for i in range(2)])
    p16 = net.probabilities(x).cpu()
    p16_fresh = net.probabilities(fresh.cuda()).cpu()
    calib = torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=99)).cuda()
    net.set_fp8(True, calibration_batch=calib)
    p8 = net.probabilities(x).cpu()
    # the project convs leave an e4m3 copy of the trunk for the next expand conv (no conversion in its loader): it must
    # hold exactly the bytes the loader would have made, i.e. the network output does not change by a bit
    import os
    os.environ["SPK_FP8_SHADOW"] = "0"
    try:
        p8_converting = net.probabilities(x).cpu()
    finally:
        del os.environ["SPK_FP8_SHADOW"]
    assert torch.equal(p8, p8_converting)
    # the expanded tensor (expand conv output) of the second MBConv block with an expand conv, read back dequantised
    from oracle import graph_eval
    exp_ops = [o for o in g.ops if o.kind == 1 and o.k == 1 and int(o.relu) == 2 and o.cout % 16 == 0 and o.src != 0]
    op = exp_ops[1]
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    shape = (n,) + tuple(graph_eval.run(g, tsd, x[:1].cpu())[op.dst].shape[1:])
    t8 = net.read_activation(op.dst, n, shape)
    net.set_fp8(False)
    p16_again = net.probabilities(x).cpu()
    t16 = net.read_activation(op.dst, n, tuple(t8.shape))
    assert torch.equal(p16_again, p16)                                    # fp16 path untouched
    rel = float((t8 - t16).norm() / t16.norm())
    assert len(torch.unique(t8)) <= 255                                   # at most 255 distinct e4m3 codes x one scale
    print(f"{tag}: expanded tensor {op.name}: fp8 vs fp16 relative L2 {rel:.3e}")
    assert rel < 0.15                                                     # e4m3 input, weights and output: ~0.10
    # probabilities
    want = torch.from_numpy(gold[f"{tag}_probs"].astype(np.float32))
    order = np.argsort(gold[f"{tag}_rois_in"])
    dp_gold = float((p8[order] - want).abs().max())
    net.set_fp8(True)
    p8_fresh = net.probabilities(fresh.cuda()).cpu()
    with torch.no_grad():
        pr = torch.cat([refnet.probabilities(ref, fresh[i:i + 16]) for i in (0, 16)])
    dp = (p8_fresh - pr).abs().max(1).values
    dp16 = (p16_fresh - pr).abs().max(1).values
    top2 = pr.topk(2, 1).values
    margin = top2[:, 0] - top2[:, 1]
    agree = float((p8_fresh.argmax(1) == pr.argmax(1)).float().mean())
    z = torch.cat([refnet.probabilities(ref, fresh[i:i + 16], base=0) for i in (0, 16)])
    z8 = net.forward(fresh.cuda()).cpu()
    zerr = float(((z8 - z) ** 2).mean().sqrt() / z.std())
    print(f"{tag}: fp8 max |dp| vs reference golden {dp_gold:.2e}; fresh images: fp8 |dp| median {float(dp.median()):.2e} "
          f"max {float(dp.max()):.2e} (fp16 path: median {float(dp16.median()):.2e} max {float(dp16.max()):.2e}); "
          f"logit rms error / logit std {zerr:.3f}; top-1 agreement {agree:.2f}")
    # Not a parity mode: e4m3 (3 mantissa bits, ~10 % relative L2 per block interior) on this random-weight 16/32-block
    # SiLU network, whose fp16 top-1 MARGIN is 0.011 in the median (p1 - p2 at the reference's base 1.3) and which
    # amplifies even fp16 rounding 40-fold (test_gpu_effnet.py).  Round 3: the mode's default covers the blocks WITH a
    # shortcut only - a block without one (the first of each stage) replaces the trunk by its e4m3-computed output and
    # alone flips more arg-maxes than all the others together (tests/archive/diagnostics/fp8_block_sweep.py, fp8_decided.py).
    assert torch.isfinite(p8_fresh).all() and torch.allclose(p8_fresh.sum(1), torch.ones(32), atol=1e-4)
    # agreement with the fp16 parity mode on 256 fresh images, overall and on the DECIDED ones
    more = [torch.from_numpy(synth.synth_images(32, 3, 224, 224, seed=300 + i)).cuda() for i in range(8)]
    net.set_fp8(False)
    q16 = torch.cat([net.probabilities(xb).cpu() for xb in more])
    net.set_fp8(True)
    q8 = torch.cat([net.probabilities(xb).cpu() for xb in more])
    net.set_fp8(True, calibration_batch=calib, blocks="all")
    q8_all = torch.cat([net.probabilities(xb).cpu() for xb in more])
    t2 = q16.topk(2, 1).values
    marg = t2[:, 0] - t2[:, 1]
    same, same_all = q8.argmax(1) == q16.argmax(1), q8_all.argmax(1) == q16.argmax(1)
    ddp = (q8 - q16).abs().max(1).values
    thr = DECIDED[network]
    dec = marg >= thr
    a_all, a_dec = float(same.float().mean()), float(same[dec].float().mean())
    print(f"{tag}: fp8 default (blocks with a shortcut) vs fp16 on 256 images: top-1 {a_all:.3f}, on the {int(dec.sum())} images "
          f"with margin >= {thr}: {a_dec:.3f}; |dp| median {float(ddp.median()):.2e} p90 {float(ddp.quantile(0.9)):.2e}; "
          f"every block on the e4m3 path: top-1 {float(same_all.float().mean()):.3f}")
    lo_all, lo_dec, hi_med = BOUNDS[network]
    assert a_all >= lo_all and a_dec >= lo_dec and float(ddp.median()) < hi_med
    assert float(dp.median()) < 3e-2 and zerr < 0.6 and agree >= 0.5      # against the reference itself, 32 images
