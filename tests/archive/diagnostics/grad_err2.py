"""Diagnostic: HIP gradients vs autograd through the bf16-emulating oracle graph."""
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
tsd = {k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(v.dtype == np.float32) for k, v in sd.items()}
acts = graph_eval.run_train_bf16(g, tsd, x)
for v in acts.values():
    if v.requires_grad: v.retain_grad()
loss = F.cross_entropy(acts[g.ops[-1].dst], y); loss.backward()
net.train(); net.reset_stats(); net.forward_backward(x.cuda(), y.cuda())
print("loss", float(loss), net.read_stats()[0] / n)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for op in reversed(g.ops):
    want_a, want_g = acts[op.dst].detach(), acts[op.dst].grad
    got_a = net.read_activation(op.dst, n, tuple(want_a.shape)); got_g = net.read_activation_grad(op.dst, n, tuple(want_a.shape))
    print(f"{op.dst:3d} {op.name or op.kind!s:26s} act rel {rel(got_a, want_a):.3e}   grad rel {rel(got_g, want_g):.3e}")
for k, _, kind in arch.param_specs(g):
    if tsd[k].grad is not None:
        print(f"{k:34s} rel {rel(net._read_grad(k, tuple(tsd[k].shape)), tsd[k].grad):.3e}")
