"""Diagnostic: per-tensor gradient error of the HIP training step vs torch autograd."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
from oracle import refnet
network, hw, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
classes = 10
g = arch.build_graph(network, classes)
sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
ref = refnet.load_numpy_state(refnet.RefNet(network, classes), sd)
net = HipNet(network, classes, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10)); y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
ref.train(); out = ref(x); loss = torch.nn.functional.cross_entropy(out, y); loss.backward()
net.train(); net.reset_stats(); logits = net.forward_backward(x.cuda(), y.cuda(), want_logits=True).cpu()
print("loss", float(loss), net.read_stats()[0] / n)
for name, p in ref.named_parameters():
    got = net._read_grad(name, tuple(p.shape)).double(); want = p.grad.double()
    cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-30))
    print(f"{name:34s} |g| {float(want.norm()):.3e} rel {float((got - want).norm() / (want.norm() + 1e-30)):.3e} cos {cos:.5f} ratio {float(got.norm() / (want.norm() + 1e-30)):.4f}")
