import sys
sys.path[:0] = ["/root/repo", "/root/repo/syke-pic_amd"]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
from oracle import graph_eval
g = arch.build_graph("efficientnet_b0", 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
net = HipNet("efficientnet_b0", 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
x = synth.synth_images(2, 3, 64, 64, seed=0)
acts = graph_eval.run(g, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, torch.from_numpy(x))
net.forward(torch.from_numpy(x).cuda())
want = acts[1]; got = net.read_activation(1, 2, tuple(want.shape))
d = (got - want).abs()
print("shape", want.shape, "max err", d.max().item())
idx = (d > 0.01).nonzero()
print("n bad", len(idx), "of", d.numel())
print(idx[:20].tolist())
import collections
print("bad by channel", collections.Counter(idx[:,1].tolist()).most_common(8))
print("bad by row", collections.Counter(idx[:,2].tolist()).most_common(8))
print("bad by col", collections.Counter(idx[:,3].tolist()).most_common(8))
