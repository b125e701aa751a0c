"""Diagnostic: activation and activation-gradient error per layer (train mode)."""
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
# interpreter with retained grads
acts = {0: x}
for op in g.ops:
    a = acts[op.src]
    if op.kind == arch.OP_CONV:
        v = F.conv2d(a, tsd[op.name + ".weight"], None, op.stride, op.pad)
        v = F.batch_norm(v, None, None, tsd[op.bn + ".weight"], tsd[op.bn + ".bias"], True, 0.1, 1e-5)
        if op.res >= 0: v = v + acts[op.res]
        if op.relu: v = F.relu(v)
    elif op.kind == arch.OP_MAXPOOL: v = F.max_pool2d(a, op.k, op.stride, op.pad)
    elif op.kind == arch.OP_GAVGPOOL: v = a.mean((2, 3))
    elif op.kind == arch.OP_LINEAR: v = F.linear(a, tsd[op.name + ".weight"], tsd[op.name + ".bias"])
    else: v = a
    v.retain_grad(); acts[op.dst] = v
loss = F.cross_entropy(acts[g.ops[-1].dst], y); loss.backward()
net.train(); net.reset_stats(); net.forward_backward(x.cuda(), y.cuda())
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for op in reversed(g.ops):
    want_a, want_g = acts[op.dst].detach(), acts[op.dst].grad
    got_a = net.read_activation(op.dst, n, tuple(want_a.shape))
    got_g = net.read_activation_grad(op.dst, n, tuple(want_a.shape))
    print(f"{op.dst:3d} {op.name or op.kind!s:26s} act rel {rel(got_a, want_a):.3e}   grad |g| {float(want_g.norm()):.3e} rel {rel(got_g, want_g):.3e}")
