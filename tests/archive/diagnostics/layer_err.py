"""Diagnostic: per-layer error of the HIP path vs the torch fp32 interpreter."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
from oracle import graph_eval

network, hw, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = arch.build_graph(network, 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
if network.startswith("efficientnet"):   # calibrated BatchNorm statistics (tests/effnet_util.py)
    from oracle import refnet
    ref = refnet.load_numpy_state(refnet.RefNet(network, 50), sd)
    refnet.calibrate_bn(ref, torch.from_numpy(synth.synth_images(16, 3, hw, hw, seed=99)))
    sd = {k: v.numpy().copy() for k, v in ref.state_dict().items()}
if len(sys.argv) > 4:
    pass
net = HipNet(network, 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
net.eval()
if len(sys.argv) > 4:
    net.set_precision(split_weights=int(sys.argv[4]))
x = synth.synth_images(n, 3, hw, hw, seed=0)
tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
acts = graph_eval.run(g, tsd, torch.from_numpy(x))
z = net.forward(torch.from_numpy(x).cuda()).cpu()
for op in g.ops:
    want = acts[op.dst]
    got = net.read_activation(op.dst, n, tuple(want.shape))
    d = (got - want)
    print(f"{op.dst:3d} {op.name or op.kind!s:28s} max|x| {float(want.abs().max()):9.3f} rms|x| {float(want.pow(2).mean().sqrt()):8.4f} "
          f"max err {float(d.abs().max()):.3e} rms err {float(d.pow(2).mean().sqrt()):.3e} rel-rms {float(d.pow(2).mean().sqrt()/want.pow(2).mean().sqrt()):.2e}")
