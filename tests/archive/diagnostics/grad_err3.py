"""Diagnostic: HIP gradients vs autograd evaluated at the GPU's own activations."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch, torch.nn.functional as F
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
from oracle import graph_eval
network, hw, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
classes = 10
g = arch.build_graph(network, classes)
sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
net = HipNet(network, classes, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10)); y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
net.train(); net.reset_stats(); net.forward_backward(x.cuda(), y.cuda())
shapes = {}
probe = graph_eval.run(g, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, x, train=True)
forced = {op.dst: net.read_activation(op.dst, n, tuple(probe[op.dst].shape)) for op in g.ops}
tsd = {k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(v.dtype == np.float32) for k, v in sd.items()}
acts = graph_eval.run_train_forced(g, tsd, x, forced)
for v in acts.values():
    if v.requires_grad: v.retain_grad()
loss = F.cross_entropy(acts[g.ops[-1].dst], y); loss.backward()
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for op in reversed(g.ops):
    got_g = net.read_activation_grad(op.dst, n, tuple(probe[op.dst].shape))
    print(f"{op.dst:3d} {op.name or op.kind!s:26s} grad rel {rel(got_g, acts[op.dst].grad):.3e}")
worst = 0
for k, _, kind in arch.param_specs(g):
    if tsd[k].grad is not None:
        r = rel(net._read_grad(k, tuple(tsd[k].shape)), tsd[k].grad); worst = max(worst, r)
        print(f"{k:34s} rel {r:.3e}")
print("worst param grad rel", worst)
