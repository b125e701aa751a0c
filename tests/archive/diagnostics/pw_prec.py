"""Diagnostic: does running the 1x1 convs on conv_pw.hip (SPK_PW=2) change the probability error of the default
`mixed` mode?  ResNet-50 @224, 3 synthetic nets x 128 images vs the fp32 oracle.  Run once per SPK_PW value:
  SPK_PW=0 python tests/archive/diagnostics/pw_prec.py ; SPK_PW=2 python tests/archive/diagnostics/pw_prec.py"""
import os
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
g = arch.build_graph(network, 50)
allv = []
for wseed in (2, 3, 4):
    sd = synth.synth_state_dict(arch.param_specs(g), seed=wseed)
    ref = refnet.load_numpy_state(refnet.RefNet(network, 50), sd)
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
    worst = []
    for seed in range(300, 304):
        x = torch.from_numpy(synth.synth_images(32, 3, hw, hw, seed=seed))
        pr = refnet.probabilities(ref, x).numpy()
        p = net.probabilities(x.cuda()).cpu().numpy()
        worst.extend(np.abs(p - pr).max(1).tolist())
    v = np.array(worst); allv.extend(worst)
    print(f"SPK_PW={os.environ.get('SPK_PW', 'default')} net seed {wseed}: max {v.max():.2e} p99 {np.percentile(v, 99):.2e} median {np.median(v):.2e} n>1e-3: {int((v > 1e-3).sum())} of {len(v)}", flush=True)
v = np.array(allv)
print(f"SPK_PW={os.environ.get('SPK_PW', 'default')} ALL: max {v.max():.2e} p99 {np.percentile(v, 99):.2e} median {np.median(v):.2e} mean {v.mean():.2e}")
