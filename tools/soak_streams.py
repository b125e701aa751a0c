"""Soak of the multi-stream paths: the training step (weight gradients on a second stream) and the eval forward (two half
batches on two streams) must give bit-identical results call after call.  usage: soak_streams.py [iterations]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# (efficientnet_b0: the two-stream eval with the squeeze-excitation scaling inside the project conv, per-half gate state)
for network, n, hw in (("resnet50", 64, 96), ("resnet18", 128, 64), ("efficientnet_b0", 64, 96)):
    g = arch.build_graph(network, 10)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
    net = HipNet(network, 10, weights=None)
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    net.load_state_dict(state)
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=1)).cuda()
    y = torch.from_numpy(synth.synth_labels(n, 10, seed=2)).cuda()
    ptr, numel = net.grad_buffer() if hasattr(net, "grad_buffer") else (None, 0)
    ref_g = ref_p = None
    bad = 0
    for it in range(iters):
        net.load_state_dict(state)          # same weights and running statistics every iteration
        if network.startswith("efficientnet"):
            # (train-mode EfficientNets draw a new stochastic-depth pattern every step: their gradients - and the running
            # statistics the eval forward below would then use - differ by design; the eval forward alone is soaked)
            gsum = torch.zeros(1)
        else:
            net.train()
            net.forward_backward(x, y)
            gsum = torch.cat([net._read_grad(k, tuple(v.shape)).flatten() for k, v in state.items()
                              if k.endswith(".weight") and v.dim() == 4][:12])
        net.eval()
        p = net.probabilities(x).cpu()
        if it == 0:
            # warm-up: the per-problem tuners run inside this iteration (EfficientNet: the depthwise kernel variants group
            # the fp32 pool partial sums differently, so this iteration may differ from the steady state in the last bits)
            first_g, first_p = gsum, p
        elif ref_g is None:
            ref_g, ref_p = gsum, p
            print(f"{network}: warm-up vs steady state: max |dp| {float((p - first_p).abs().max()):.2e}, "
                  f"max |dg| / max |g| {float((gsum - first_g).abs().max() / gsum.abs().max().clamp_min(1e-30)):.2e}")
        else:
            bad += int(not torch.equal(gsum, ref_g)) + int(not torch.equal(p, ref_p))
    print(f"{network} n={n} {hw}x{hw}: {iters} iterations, {bad} mismatches")
    assert bad == 0
print("soak ok")
