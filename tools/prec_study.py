"""Diagnostic: probability error of the eval-path precision modes vs golden."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
gold = np.load(ROOT / "tests/golden/net_pass.npz")
for network, hw in (("resnet18", 224), ("resnet50", 224)):
    tag = f"{network}_{hw}"
    g = arch.build_graph(network, 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    last = [k for k in sd if k.startswith("head.") and k.endswith(".bias")][-1]
    sd[last] = sd[last] + gold[f"{tag}_bias_adj"]
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    x = torch.from_numpy(synth.synth_images(8, 3, hw, hw, seed=0)).cuda()
    order = np.argsort(gold[f"{tag}_rois_in"])
    for name, kw in (("bf16", dict(bf16=True, split_weights=False)), ("f16", dict(split_weights=False)),
                     ("f16+res_lo", dict(split_weights=False, precise_residual=True)),
                     ("f16+w_lo(trunk)", dict(split_weights=2)),
                     ("f16+w_lo", dict(split_weights=1)), ("f16+w_lo+res_lo", dict(split_weights=1, precise_residual=True))):
        net.set_precision(**kw)
        p = net.probabilities(x).cpu().numpy()[order]
        z = net.forward(x).cpu().numpy()
        print(f"{tag:14s} {name:16s} max|dp| {np.abs(p - gold[f'{tag}_probs']).max():.2e}  logits rms err {np.sqrt(((z - gold[f'{tag}_logits'])**2).mean()):.2e}")
