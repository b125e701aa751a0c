"""Parity on TRAINED networks.

Every other inference test uses random-weight nets whose BatchNorm running statistics do not describe their
activations and whose top-1 margins are ~0.01.  Here ResNet-18 and EfficientNet-B0 are trained with the HIP path itself
for a few hundred Adam steps on a separable synthetic labelled set (seeded), the resulting ``state_dict`` goes into the
fp32 oracle (the torch module of reference ``sykepic/train/network.py:11-72``), and the HIP eval modes are compared with
it on 256 fresh images: probabilities (north_star: 1e-3), top-1 on every decided image.  A trained net has the
statistics and the margins a deployed classifier has, so this is where the claims about each mode are settled:
``mixed`` / ``calibrated`` / ``precise`` for the ResNet, fp16 / calibrated / fp8 for the EfficientNet.
"""

import numpy as np
import pytest
import torch

from sykepic_hip import arch, synth

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-3
CLASSES, HW = 10, 96


def labelled_images(n, seed, hw=None):
    """class k = a light IFCB-like frame with a dark blob whose size, position band and stripe period depend on k, plus
    per-image jitter and pixel noise: separable, but not by one pixel.  Values are k/255 as ToTensor gives them.
    `hw`: the 96 x 96 frame blown up to hw x hw (nearest pixel: the same task at the headline size)."""
    if hw is not None and hw != HW:
        x, y = labelled_images(n, seed)
        idx = (torch.arange(hw) * HW) // hw
        return x[:, :, idx][:, :, :, idx].contiguous(), y
    rng = np.random.RandomState(seed)
    y = rng.randint(0, CLASSES, n)
    yy, xx = np.mgrid[0:HW, 0:HW]
    x = np.empty((n, 3, HW, HW), np.float32)
    for i in range(n):
        k = int(y[i])
        bg = 170 + rng.randint(-25, 25)
        img = np.full((HW, HW), bg, np.float64)
        cy = HW * (0.25 + 0.05 * (k % 5)) + rng.randint(-6, 7) + (HW // 3 if k >= 5 else 0)
        cx = HW * 0.5 + rng.randint(-12, 13)
        r = 8 + 2 * (k % 4) + rng.randint(-1, 2)
        blob = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        img[blob] = 60 + 8 * k
        period = 4 + k
        img += 14.0 * np.sin(2 * np.pi * (xx if k % 2 else yy) / period)
        img += rng.randint(-8, 9, (HW, HW))
        x[i] = np.clip(np.rint(img), 0, 255)[None] / 255.0
    return torch.from_numpy(x), torch.from_numpy(y.astype(np.int64))


def train_hip(network, steps, lr, seed, hw=None, batch=64):
    from sykepic_hip.net import HipNet
    from sykepic_hip.optim import HipOptimizer
    net = HipNet(network, CLASSES, weights=None, head=(64, 32))
    net.reset_parameters(seed=seed)
    net.set_seed(seed)
    for p in net.parameters():
        p.requires_grad = True
    opt = HipOptimizer(net, "Adam", [{"params": list(net.parameters()), "lr": lr}])
    net.train()
    acc = []
    # the last quarter at a tenth of the learning rate, then 30 steps without an update: the BatchNorm running statistics
    # are a momentum-0.1 average over the last ~20 steps and must describe the FINAL weights (one run in which Adam was
    # still moving fast at the end gave 0.99 train-mode and chance eval-mode accuracy - in the fp32 oracle too: a property
    # of the schedule, not of a kernel)
    for s in range(steps + 30):
        x, y = labelled_images(batch, 10_000 + s, hw)
        if s == steps * 3 // 4:
            opt.param_groups[0]["lr"] = lr / 10
        net.reset_stats()
        net.forward_backward(x.cuda(), y.cuda())
        if s < steps:
            opt.step()
        loss, correct = net.read_stats()
        acc.append(correct / batch)
    return net.eval(), float(np.mean(acc[-20:]))


def compare(net, ref, x, label):
    from oracle import refnet
    pr = refnet.probabilities(ref, x).numpy().astype(np.float64)
    p = net.probabilities(x.cuda()).cpu().numpy().astype(np.float64)
    d = np.abs(p - pr).max(1)
    top2 = np.sort(pr, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 2 * PROB_TOL
    same = p.argmax(1) == pr.argmax(1)
    out = {"max": d.max(), "p90": np.percentile(d, 90), "median": np.median(d), "decided": int(decided.sum()),
           "top1_decided": float(same[decided].mean()) if decided.any() else 1.0, "top1_all": float(same.mean()),
           "margin_median": float(np.median(top2[:, 1] - top2[:, 0]))}
    print(f"  {label:34s} max |dp| {out['max']:.2e}  p90 {out['p90']:.2e}  median {out['median']:.2e}  "
          f"top-1 {out['top1_all']:.3f} (decided {out['decided']}: {out['top1_decided']:.3f})")
    return out


def test_trained_resnet18_every_mode_within_tolerance():
    from oracle import refnet
    net, acc = train_hip("resnet18", 300, 1e-3, seed=11)
    ref = refnet.RefNet("resnet18", CLASSES, head=(64, 32))
    ref.load_state_dict(net.state_dict())
    ref.eval()
    x, y = labelled_images(256, 77)
    ref_acc = float((refnet.probabilities(ref, x).argmax(1) == y).float().mean())
    print(f"resnet18 trained 300 steps: train accuracy (last 20 steps) {acc:.3f}, oracle accuracy on fresh images {ref_acc:.3f}")
    assert acc > 0.8 and ref_acc > 0.7            # it really learned the task: margins and statistics are a trained net's
    res = {}
    res["mixed"] = compare(net.set_precision(split_weights=3), ref, x, "mixed (default, 16 of 20 convs hi+lo)")
    res["precise"] = compare(net.set_precision(split_weights=1), ref, x, "precise (every conv hi+lo)")
    res["fast"] = compare(net.set_precision(split_weights=0), ref, x, "fast (plain fp16, nearest)")
    net.calibrate(labelled_images(64, 5555)[0].cuda())
    res["calibrated"] = compare(net.set_precision("calibrated"), ref, x, "calibrated (single pass, zero-sum)")
    res["precise+res"] = compare(net.set_precision(split_weights=1, precise_residual=True), ref, x,
                                 "precise + trunk rounding remainders")
    net.set_precision(split_weights=3)
    # Since the input is stored as exact pixel values (x 255, csrc/spk_common.h SPK_INPUT_SCALE) a trained net sits far
    # inside the tolerance: measured mixed / precise 5.7e-5, calibrated 9.2e-5, plain fp16 1.4e-4 (worst of 256 images).
    # Before that change every mode read 1.2e-3 here - the fp16 rounding of k / 255, one systematic error per grey level
    # (tests/archive/diagnostics/input_rounding.py), which hid the differences between the weight modes altogether.
    for mode in ("mixed", "precise", "calibrated"):
        assert res[mode]["max"] <= 3e-4 and res[mode]["p90"] <= 1e-4, (mode, res[mode])
        assert res[mode]["top1_decided"] == 1.0, (mode, res[mode])
    assert res["fast"]["max"] <= PROB_TOL
    # the single-pass mode is as close to fp32 as the two-pass one: the systematic part of the weight rounding is gone
    # (trained on the spot: bounds with room for another box's trajectory - measured 3.1e-5 vs 2.1e-5 and 3.9e-5 vs 2.1e-5)
    assert res["calibrated"]["p90"] <= 2.0 * res["precise"]["p90"] + 2e-5
    assert res["calibrated"]["max"] <= res["fast"]["max"]


def test_trained_efficientnet_b0_fp16_calibrated_and_fp8():
    """The fp16 EfficientNet path sits at p90 1.4e-3 / max 4.6e-3 on random-weight nets (tests/test_gpu_effnet.py).  On a net
    whose BatchNorm statistics describe its data - and with the input stored as exact pixel values - it holds the
    reference's tolerance with a wide margin: measured max 2.0e-4, p90 5.8e-5, median 1.5e-5 over 256 fresh images, top-1
    identical on all of them (before the exact input: 1.3e-3 / 3.7e-4 / 8e-5).  The worst image of the 256 depends on the
    training trajectory: the net is trained HERE, by the HIP path, and the kernel tuner's tile choices - made per box - group
    the BatchNorm partial sums differently, so every box trains another, equally valid net.  Seen over the runs of rounds
    4-5: fp16 2.0e-4 ... 3.5e-4, calibrated 2.0e-4 ... 6.6e-4, p90 6e-5 ... 9e-5, median 1.5e-5 ... 2.7e-5.  Asserted: max <= 8e-4
    (the reference's tolerance is 1e-3), p90 <= 2e-4, median <= 5e-5.  fp8 (e4m3 inside the MBConv blocks) is judged on
    top-1 of decided images: a throughput mode, not a drop-in one."""
    from oracle import refnet
    net, acc = train_hip("efficientnet_b0", 400, 2e-3, seed=12)
    ref = refnet.RefNet("efficientnet_b0", CLASSES, head=(64, 32))
    ref.load_state_dict(net.state_dict())
    ref.eval()
    x, y = labelled_images(256, 78)
    ref_acc = float((refnet.probabilities(ref, x).argmax(1) == y).float().mean())
    print(f"efficientnet_b0 trained 400 steps: train accuracy {acc:.3f}, oracle accuracy on fresh images {ref_acc:.3f}")
    assert acc > 0.7 and ref_acc > 0.6
    res = {}
    res["fp16"] = compare(net.set_precision(split_weights=3), ref, x, "fp16 (default split rule)")
    res["precise"] = compare(net.set_precision(split_weights=1), ref, x, "fp16, every conv hi+lo")
    try:
        compare(net.set_precision(split_weights=3, precise_residual=True), ref, x, "fp16 + trunk rounding remainders")
    except RuntimeError as e:   # (diagnostic only)
        print("  precise_residual:", e)
    net.set_precision(split_weights=3)
    net.calibrate(labelled_images(64, 5556)[0].cuda())
    res["calibrated"] = compare(net.set_precision("calibrated"), ref, x, "calibrated (single pass, zero-sum)")
    net.set_precision(split_weights=3)
    net.set_fp8(True, calibration_batch=labelled_images(64, 5557)[0])
    res["fp8"] = compare(net, ref, x, "fp8 (e4m3 MBConv interior)")
    net.set_fp8(False)
    for mode in ("fp16", "precise", "calibrated"):
        assert res[mode]["max"] <= 8e-4 and res[mode]["p90"] <= 2e-4 and res[mode]["median"] <= 5e-5, (mode, res[mode])
        assert res[mode]["top1_decided"] == 1.0, (mode, res[mode])
    assert res["fp8"]["top1_decided"] >= 0.99 and res["fp8"]["p90"] <= 1.5e-2, res["fp8"]


def test_trained_resnet18_gradients_follow_fp32_autograd():
    """Training parity on a TRAINED net.  On random-weight nets with batch statistics over a handful
    of images the bf16 forward flips enough ReLU masks that whole gradient tensors decorrelate from fp32 autograd (min
    cosine 0.79 for ResNet-50, tests/test_gpu_train.py) - for ANY bf16 implementation.  Here the HIP step is compared with
    the reference's pure-fp32 forward / backward (sykepic/train/train.py:240-242) on a net that has been trained for 300
    steps and a batch of 64 images: what a user's training run looks like."""
    import torch.nn.functional as F
    from oracle import graph_eval, refnet
    net, acc = train_hip("resnet18", 300, 1e-3, seed=11)
    g = arch.build_graph("resnet18", CLASSES, head=(64, 32))
    specs = arch.param_specs(g)
    kinds = {k: kind for k, _, kind in specs}
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x, y = labelled_images(64, 4321)
    tsd = {k: v.clone().requires_grad_(v.dtype == torch.float32 and kinds[k] not in ("bn_mean", "bn_var"))
           for k, v in state.items()}
    out = graph_eval.run(g, tsd, x, train=True)[g.ops[-1].dst]
    loss = F.cross_entropy(out, y)
    loss.backward()
    want = {k: t.grad for k, t in tsd.items() if t.grad is not None}
    net.train()
    for p in net.parameters():
        p.requires_grad = True
    net.reset_stats()
    logits = net.forward_backward(x.cuda(), y.cuda(), want_logits=True).cpu()
    loss_n, _ = net.read_stats()
    cos, ratio = {}, {}
    for name, wg in want.items():
        got = net._read_grad(name, tuple(wg.shape)).double().flatten()
        w = wg.double().flatten()
        cos[name] = float(got @ w / (got.norm() * w.norm() + 1e-30))
        ratio[name] = float(got.norm() / (w.norm() + 1e-30))
    worst = min(cos, key=cos.get)
    vals = np.array(list(cos.values()))
    print(f"trained resnet18, batch 64: loss {loss_n / 64:.5f} vs fp32 {float(loss.detach()):.5f}; logits rel-L2 "
          f"{float((logits - out.detach()).norm() / out.detach().norm()):.2e}; gradient cosine vs fp32 autograd min {vals.min():.4f} "
          f"({worst}) median {np.median(vals):.4f}; norm ratio {min(ratio.values()):.3f}..{max(ratio.values()):.3f}")
    assert abs(loss_n / 64 - float(loss.detach())) < 2e-2 * max(1.0, float(loss.detach()))
    # measured: min cosine 0.9979 (a BatchNorm bias of layer2), median 0.9996, norms within 1 %, logits 6.6e-4
    assert vals.min() >= 0.99 and np.median(vals) >= 0.999
    assert 0.97 <= min(ratio.values()) and max(ratio.values()) <= 1.03


def test_trained_resnet50_at_the_headline_size_mixed_and_calibrated():
    """the headline model, trained (160 Adam steps at 224 x 224, batches of 32), then classified in
    batches of 256 - two half batches on two streams, chained stage-1 convs, the identity bottlenecks of stages 2-3 as
    single kernels (csrc/conv_bneck.hip) - in the mode a reference-trained directory starts in (`mixed`) and in the one
    `prob` switches to after its first batch (`calibrated`).  Oracle: the reference's torch module with the same state_dict
    on the CPU (sykepic/train/network.py:11-72, sykepic/compute/probability.py:180-197)."""
    from oracle import refnet
    net, acc = train_hip("resnet50", 160, 1e-3, seed=21, hw=224, batch=32)
    ref = refnet.RefNet("resnet50", CLASSES, head=(64, 32))
    ref.load_state_dict(net.state_dict())
    ref.eval()
    x, y = labelled_images(256, 79, 224)
    ref_acc = float((refnet.probabilities(ref, x).argmax(1) == y).float().mean())
    print(f"resnet50 trained 160 steps at 224: train accuracy (last 20 steps) {acc:.3f}, oracle accuracy on fresh images {ref_acc:.3f}")
    assert acc > 0.8 and ref_acc > 0.7
    res = {}
    res["mixed"] = compare(net.set_precision(split_weights=3), ref, x, "mixed (37 of 53 convs hi+lo)")
    net.calibrate(labelled_images(64, 5558, 224)[0].cuda())
    net.set_precision("calibrated")
    net.probabilities(x.cuda())                   # (the first forward of a mode tunes on one stream; the next ones use two)
    res["calibrated"] = compare(net, ref, x, "calibrated (single pass, zero-sum)")
    res["fast"] = compare(net.set_precision(split_weights=0), ref, x, "fast (plain fp16, nearest)")
    # measured (worst of 256 images) on two boxes - the net is trained here and every box's tuner choices give another
    # trajectory: mixed 1.1e-4 / 1.3e-4 (p90 4.7e-5 / 3.4e-5), calibrated 3.0e-4 / 1.6e-4 (p90 1.4e-4 / 9.0e-5), plain fp16
    # 4.9e-4 / 2.1e-4 (p90 - / 9.1e-5); top-1 256 / 256.  (No ordering between calibrated and plain fp16 is asserted: on a
    # trained ResNet-50 the two are within each other's box-to-box spread.)
    assert res["mixed"]["max"] <= 3e-4 and res["calibrated"]["max"] <= 6e-4
    for mode in ("mixed", "calibrated", "fast"):
        assert res[mode]["max"] <= PROB_TOL and res[mode]["top1_decided"] == 1.0, (mode, res[mode])


def test_trained_efficientnet_b4_fp16():
    """Config 5's model, trained (300 Adam steps at 96 x 96), fp16 parity mode against the fp32 oracle on 256 fresh images:
    is the random-weight spread of tests/test_gpu_effnet.py (p90 ~1e-3 at base 1.3) a property of untrained 32-block SiLU
    nets, as the trained B0 says, or of B4?"""
    from oracle import refnet
    net, acc = train_hip("efficientnet_b4", 300, 2e-3, seed=22)
    ref = refnet.RefNet("efficientnet_b4", CLASSES, head=(64, 32))
    ref.load_state_dict(net.state_dict())
    ref.eval()
    x, y = labelled_images(256, 80)
    ref_acc = float((refnet.probabilities(ref, x).argmax(1) == y).float().mean())
    print(f"efficientnet_b4 trained 300 steps: train accuracy {acc:.3f}, oracle accuracy on fresh images {ref_acc:.3f}")
    assert acc > 0.6 and ref_acc > 0.5
    res = compare(net.set_precision(split_weights=3), ref, x, "fp16 (default: no conv split)")
    # measured: max 6.0e-4, p90 9.0e-5, median 2.4e-5, top-1 256 / 256 - the fp16 path holds the reference's tolerance on B4
    assert res["max"] <= PROB_TOL and res["p90"] <= 3e-4 and res["top1_decided"] == 1.0, res
