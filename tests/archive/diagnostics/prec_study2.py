"""Diagnostic: max |dp| of the eval precision modes over many synthetic images (ResNet-50)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
from oracle import refnet
torch.set_num_threads(16)
network, hw = "resnet50", 224
gold = np.load(ROOT / "tests/golden/net_pass.npz")
g = arch.build_graph(network, 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
last = [k for k in sd if k.startswith("head.") and k.endswith(".bias")][-1]
sd[last] = sd[last] + gold["resnet50_224_bias_adj"]
ref = refnet.load_numpy_state(refnet.RefNet(network, 50), sd)
net = HipNet(network, 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
worst = {}
for seed in range(200, 204):
    x = torch.from_numpy(synth.synth_images(32, 3, hw, hw, seed=seed))
    pr = refnet.probabilities(ref, x).numpy()
    for name, kw in (("f16", dict(split_weights=0)), ("trunk", dict(split_weights=2)), ("full", dict(split_weights=1))):
        net.set_precision(**kw)
        p = net.probabilities(x.cuda()).cpu().numpy()
        e = np.abs(p - pr).max(1)
        worst.setdefault(name, []).extend(e.tolist())
    print(seed, {k: f"{max(v):.2e}" for k, v in worst.items()}, "pmax range", pr.max(1).min().round(3), pr.max(1).max().round(3), flush=True)
for k, v in worst.items():
    v = np.array(v); print(k, "max", f"{v.max():.2e}", "p99", f"{np.percentile(v,99):.2e}", "median", f"{np.median(v):.2e}", "n>1e-3:", int((v>1e-3).sum()), "of", len(v))
