"""Parity of the HIP training step (through the C-ABI) with the oracle's
torch fp32 autograd, and with the golden vectors the reference's own
train_net produced.  bf16 activations/gradients vs fp32: tolerances are
relative L2 per tensor (stated at each check)."""

import numpy as np
import pytest
import torch

from sykepic_hip import arch, schedule, synth
from sykepic_hip.optim import HipOptimizer

pytestmark = pytest.mark.gpu


def _pair(network, classes, seed, gain=2.0):
    from oracle import refnet
    from sykepic_hip.net import HipNet
    g = arch.build_graph(network, classes)
    specs = arch.param_specs(g)
    sd = synth.synth_state_dict(specs, seed=seed, logit_gain=gain)
    ref = refnet.load_numpy_state(refnet.RefNet(network, classes), sd)
    net = HipNet(network, classes, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return g, specs, ref, net


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("network,hw,n", [("resnet18", 64, 8), ("resnet50", 64, 6), ("resnet18", 75, 5)])
def test_backward_kernels_at_the_gpu_operating_point(network, hw, n):
    """Autograd evaluated AT the activations the GPU produced (teacher-forced
    oracle graph: same ReLU masks, same batch statistics) isolates the
    backward kernels: what remains is the bf16 rounding of the gradient
    tensors (2^-9 each, a few per layer).  Tolerance: relative L2 <= 0.15 per
    parameter tensor in the backbone, 1e-4 in the fp32 head."""
    import torch.nn.functional as F
    from oracle import graph_eval
    classes = 10
    g, specs, ref, net = _pair(network, classes, seed=5)
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10))
    y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
    net.train()
    net.forward_backward(x.cuda(), y.cuda())
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    probe = graph_eval.run(g, sd, x, train=True)
    forced = {op.dst: net.read_activation(op.dst, n, tuple(probe[op.dst].shape)) for op in g.ops}
    tsd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in ref.state_dict().items()}
    acts = graph_eval.run_train_forced(g, tsd, x, forced)
    F.cross_entropy(acts[g.ops[-1].dst], y).backward()
    worst = ("", 0.0)
    for k, _, kind in specs:
        if tsd[k].grad is None:
            continue
        r = _rel(net._read_grad(k, tuple(tsd[k].shape)), tsd[k].grad)
        if r > worst[1]:
            worst = (k, r)
        assert r < (1e-4 if k.startswith("head.") else 0.15), f"{k}: relative L2 gradient error {r:.3e}"
    print(f"{network}@{hw}x{n}: worst gradient rel-L2 {worst[1]:.3e} at {worst[0]}")


@pytest.mark.parametrize("network,hw,n", [("resnet18", 64, 8), ("resnet50", 96, 16)])
def test_gradients_vs_fp32_autograd(network, hw, n):
    """Against the reference's pure-fp32 forward/backward.  A bf16 forward
    differs from the fp32 one by ~1e-2 deep in the net, which flips the ReLU
    mask of the ~1 % of activations that sit next to zero; every flipped
    element moves the gradient by its full magnitude, so per-tensor relative
    L2 lands at 0.1-0.3 for ANY bf16 training path.  Checked here: direction
    (cosine >= 0.75; measured 0.95 on ResNet-18, 0.79-0.85 on ResNet-50) and size (norm within 20 %) of every gradient, the loss
    to 2e-2, the logits to 8e-2 relative L2 and the accuracy counter exactly."""
    classes = 10
    g, specs, ref, net = _pair(network, classes, seed=5)
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10))
    y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
    ref.train()
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    net.train()
    net.reset_stats()
    logits = net.forward_backward(x.cuda(), y.cuda(), want_logits=True).cpu()
    loss_n, correct = net.read_stats()
    assert abs(loss_n / n - float(loss.detach())) < 2e-2 * max(1.0, abs(float(loss.detach())))
    assert correct == float((out.argmax(1) == y).sum())
    assert _rel(logits, out.detach()) < 8e-2
    worst = 1.0
    for name, p in ref.named_parameters():
        got = net._read_grad(name, tuple(p.shape)).double().flatten()
        want = p.grad.double().flatten()
        cos = float(got @ want / (got.norm() * want.norm() + 1e-30))
        ratio = float(got.norm() / (want.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > 0.75 and 0.8 < ratio < 1.2, f"{name}: cos {cos:.4f} ratio {ratio:.3f}"
    print(f"{network}: min gradient cosine vs fp32 autograd {worst:.4f}")
    # BatchNorm running statistics (momentum 0.1, unbiased variance) and counter
    sd_ref, sd_hip = ref.state_dict(), net.state_dict()
    for k, _, kind in specs:
        if kind in ("bn_mean", "bn_var"):
            assert torch.allclose(sd_hip[k], sd_ref[k], rtol=2e-2, atol=2e-3), k
        if kind == "bn_nbt":
            assert int(sd_hip[k]) == int(sd_ref[k]) == 1


@pytest.mark.parametrize("optim_name", ["SGD", "Adam"])
def test_unfreeze_schedule_matches_reference_golden(golden_dir, optim_name):
    """Same 3-epoch run as tests/golden/make_golden.py drove through the
    reference's train_net: freeze -> LRWarmup steps at epochs 1,2,3."""
    gold = np.load(golden_dir / f"train_{optim_name.lower()}.npz")
    n, hw, classes = 8, 64, 10
    g, specs, ref, net = _pair("resnet18", classes, seed=5)
    schedule.freeze(net.base)
    first = [p for p in net.parameters() if p.requires_grad]
    assert sum(p.numel() for p in first) == int(gold["group_sizes"][0])
    opt = HipOptimizer(net, optim_name, [{"params": first, "lr": 0.01}, {"params": [], "lr": 0.0},
                                         {"params": [], "lr": 0.0}])
    warm = schedule.LRWarmup(net, opt, 0.1, 0.5, 1, 2, 3, verbose=False)
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10)).cuda()
    y = torch.from_numpy(synth.synth_labels(n, classes, seed=11)).cuda()
    xv = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=12)).cuda()
    yv = torch.from_numpy(synth.synth_labels(n, classes, seed=13)).cuda()
    keys = [k for k, _, _ in specs]
    for epoch in (1, 2, 3):
        warm(epoch)
        net.train()
        net.reset_stats()
        net.forward_backward(x, y)
        opt.step()
        loss_n, _ = net.read_stats()
        # loss within 3 % of the reference's fp32 run (bf16 forward)
        assert abs(loss_n / n - gold["train_loss"][epoch - 1]) < 0.03 * gold["train_loss"][epoch - 1] + 0.02
        net.eval()
        net.reset_stats()
        net.eval_step(xv, yv)
        vloss_n, _ = net.read_stats()
        assert abs(vloss_n / n - gold["val_loss"][epoch - 1]) < 0.05 * gold["val_loss"][epoch - 1] + 0.02
        sd = net.state_dict()
        l2 = np.array([float(sd[k].double().norm()) for k in keys])
        assert np.allclose(l2, gold[f"e{epoch}_l2"], rtol=2e-2, atol=1e-3), epoch
        assert int(sd["base.1.num_batches_tracked"]) == int(gold[f"e{epoch}_nbt"])
    assert np.allclose([gp["lr"] for gp in opt.param_groups], gold["group_lr"][0])
    assert [sum(p.numel() for p in gp["params"]) for gp in opt.param_groups] == gold["group_sizes"].tolist()


def test_optimizers_match_torch_update_rule():
    """One tensor-level check of the fused Adam / SGD kernels: after a step
    every updated parameter equals torch.optim's result on the HIP gradients."""
    classes, n, hw = 10, 8, 64
    for name in ("SGD", "Adam"):
        g, specs, ref, net = _pair("resnet18", classes, seed=7)
        x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10)).cuda()
        y = torch.from_numpy(synth.synth_labels(n, classes, seed=11)).cuda()
        params = list(net.parameters())
        opt = HipOptimizer(net, name, [{"params": params, "lr": 0.01}])
        net.train()
        before = net.state_dict()
        tparams = {p.key: torch.nn.Parameter(before[p.key].clone()) for p in params}
        topt = getattr(torch.optim, name)(list(tparams.values()), lr=0.01)
        for step in range(2):
            net.forward_backward(x, y)
            for p in params:
                tparams[p.key].grad = net._read_grad(p.key, p.shape)
            opt.step()
            topt.step()
            after = net.state_dict()
            for k, tp in tparams.items():
                assert torch.allclose(after[k], tp.detach(), rtol=1e-5, atol=1e-6), (name, step, k)
            # keep the two trajectories on identical parameters
            for k, tp in tparams.items():
                tp.data.copy_(after[k])


def test_training_is_bitwise_reproducible():
    classes, n, hw = 10, 8, 64
    grads = []
    for _ in range(2):
        g, specs, ref, net = _pair("resnet18", classes, seed=5)
        x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10)).cuda()
        y = torch.from_numpy(synth.synth_labels(n, classes, seed=11)).cuda()
        net.train()
        net.forward_backward(x, y)
        grads.append({p.key: net._read_grad(p.key, p.shape) for p in net.parameters()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def test_fresh_network_is_randomly_initialised_and_learns():
    """A HipNet built with weights=None starts from torch/torchvision's initial distributions (the
    reference's TorchVisionNet(weights=None)), reproducibly under torch.manual_seed, and a few Adam steps on one
    batch reduce the loss (an all-zero start would leave every conv dead)."""
    from sykepic_hip.net import HipNet
    from sykepic_hip.optim import HipOptimizer
    torch.manual_seed(7)
    net = HipNet("resnet18", 5, weights=None, head=(32,))
    sd = net.state_dict()
    w = sd["base.4.0.conv1.weight"]                     # [64, 64, 3, 3]: std = sqrt(2 / (64*9))
    assert abs(float(w.std()) - (2.0 / (64 * 9)) ** 0.5) < 0.1 * (2.0 / (64 * 9)) ** 0.5 and abs(float(w.mean())) < 2e-3
    assert torch.all(sd["base.1.weight"] == 1) and torch.all(sd["base.1.bias"] == 0)
    assert torch.all(sd["base.1.running_var"] == 1) and int(sd["base.1.num_batches_tracked"]) == 0
    hw_ = sd["head.0.weight"]                           # U(+-1/sqrt(512))
    assert float(hw_.abs().max()) <= 512 ** -0.5 + 1e-7 and float(hw_.std()) > 0.5 * 512 ** -0.5 / 3 ** 0.5
    torch.manual_seed(7)
    again = HipNet("resnet18", 5, weights=None, head=(32,)).state_dict()
    assert all(torch.equal(again[k], v) for k, v in sd.items())
    x = torch.from_numpy(synth.synth_images(16, 3, 64, 64, seed=3)).cuda()
    y = torch.from_numpy(synth.synth_labels(16, 5, seed=4)).cuda()
    opt = HipOptimizer(net, "Adam", [{"params": list(net.parameters()), "lr": 1e-3}])
    net.train()
    losses = []
    for _ in range(12):
        net.reset_stats()
        net.forward_backward(x, y)
        opt.step()
        losses.append(net.read_stats()[0] / 16)
    print("losses", [round(l, 3) for l in losses])
    assert losses[-1] < 0.7 * losses[0]


def test_weights_argument_loads_a_local_backbone_checkpoint(tmp_path):
    """`weights=<path to a torchvision checkpoint>` fills `base` (what `weights="DEFAULT"` downloads in the
    reference), the head keeps its fresh initialisation."""
    from oracle import backbones
    from sykepic_hip.net import HipNet
    tv = backbones.make("resnet18")
    torch.manual_seed(3)
    for p in tv.parameters():
        p.data.normal_(0, 0.05)
    path = tmp_path / "resnet18-local.pth"
    torch.save(tv.state_dict(), path)
    net = HipNet("resnet18", 4, weights=str(path), head=(16,))
    sd = net.state_dict()
    assert torch.equal(sd["base.0.weight"], tv.state_dict()["conv1.weight"])
    assert torch.equal(sd["base.7.1.bn2.running_var"], tv.state_dict()["layer4.1.bn2.running_var"])
    assert float(sd["head.0.weight"].abs().max()) <= 512 ** -0.5 + 1e-7
    with pytest.raises(RuntimeError, match="not a resnet50 backbone"):
        HipNet("resnet50", 4, weights=str(path))
