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
for network, n, hw in (("resnet50", 64, 96), ("resnet18", 128, 64)):
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
        net.train()
        net.forward_backward(x, y)
        gsum = torch.cat([net._read_grad(k, tuple(v.shape)).flatten() for k, v in state.items()
                          if k.endswith(".weight") and v.dim() == 4][:12])
        net.eval()
        p = net.probabilities(x).cpu()
        if ref_g is None:
            ref_g, ref_p = gsum, p
        else:
            bad += int(not torch.equal(gsum, ref_g)) + int(not torch.equal(p, ref_p))
    print(f"{network} n={n} {hw}x{hw}: {iters} iterations, {bad} mismatches")
    assert bad == 0
print("soak ok")
