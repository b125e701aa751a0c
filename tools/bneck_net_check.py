"""Diagnostics: ResNet-50 logits with the whole-bottleneck kernel forced on / off, one and two streams (they must be equal)."""
import hashlib, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = f"""
import sys, hashlib, numpy as np, torch
sys.path[:0] = [{str(ROOT)!r}, {str(ROOT / 'syke-pic_amd')!r}]
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
g = arch.build_graph('resnet50', 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
net = HipNet('resnet50', 50, weights=None)
net.load_state_dict({{k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}); net.eval()
net.calibrate(torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=9000)).cuda())
net.set_precision('calibrated')
for n in (64, 33):
    x = torch.from_numpy(synth.synth_images(n, 3, 224, 224, seed=5)).cuda()
    for it in range(3):
        z = net.forward(x).cpu().numpy()
        print('SHA', n, it, hashlib.sha256(z.tobytes()).hexdigest()[:12], float(np.abs(z).max()))
"""
for env in ({"SPK_BNECK": "0"}, {"SPK_BNECK": "2"}, {"SPK_BNECK": "3"}, {"SPK_BNECK": "3", "SPK_EVAL_STREAMS": "1"}, {"SPK_BNECK": "1"}):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPK_TUNE_CACHE="off", **env), capture_output=True, text=True)
    print(env, out.returncode, out.stderr[-300:] if out.returncode else "")
    print("\n".join(l for l in out.stdout.splitlines() if l.startswith("SHA")))
