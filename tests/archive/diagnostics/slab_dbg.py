import sys, os, numpy as np, torch
import pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
g = arch.build_graph('resnet18', 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
net = HipNet('resnet18', 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
x = torch.from_numpy(synth.synth_images(9, 3, 100, 84, seed=5)).cuda()
net.forward(x)
acts = {}
h, w = 100, 84
dims = {0: (h, w)}
for op in g.ops:
    ih, iw = dims.get(op.src, (1, 1))
    if op.kind in (arch.OP_CONV, arch.OP_MAXPOOL):
        oh = (ih + 2 * op.pad - op.k) // op.stride + 1; ow = (iw + 2 * op.pad - op.k) // op.stride + 1
        dims[op.dst] = (oh, ow)
        acts[op.dst] = net.read_activation(op.dst, 9, (9, op.cout, oh, ow)).numpy()
np.savez(sys.argv[1], **{str(k): v for k, v in acts.items()})
