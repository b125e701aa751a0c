"""Edge shapes: batch 1, tiny / odd image sizes, all networks (GPU diagnostic; asserts against the oracle)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from oracle import refnet
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
for name in ("resnet18", "resnet50", "efficientnet_b0", "efficientnet_b3"):
    g = arch.build_graph(name, 10, head=(16,))
    sd = synth.synth_state_dict(arch.param_specs(g), seed=3, logit_gain=4.0)
    ref = refnet.load_numpy_state(refnet.RefNet(name, 10, head=(16,)), sd)
    net = HipNet(name, 10, weights=None, head=(16,))
    for n, h, w in ((1, 32, 32), (1, 33, 47), (3, 64, 37), (2, 224, 224), (7, 97, 131), (1, 300, 500)):
        x = torch.from_numpy(synth.synth_images(n, 3, h, w, seed=h + w))
        if name.startswith("eff"):
            # a random-weight EfficientNet is only well conditioned with BatchNorm statistics that match its
            # inputs (activations reach 7e3 otherwise): calibrate on images of the tested size
            ref = refnet.load_numpy_state(refnet.RefNet(name, 10, head=(16,)), sd)
            refnet.calibrate_bn(ref, torch.from_numpy(synth.synth_images(8, 3, h, w, seed=9)))
            cur = {k: v.numpy() for k, v in ref.state_dict().items()}
        else:
            cur = sd
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in cur.items()}); net.eval()
        z = refnet.probabilities(ref, x, base=0).numpy()
        zg = net.forward(x.cuda()).cpu().numpy()
        rel = np.sqrt(np.mean((zg - z) ** 2)) / max(z.std(), 1e-6)
        pr = refnet.probabilities(ref, x).numpy()
        pg = net.probabilities(x.cuda()).cpu().numpy()
        err = np.abs(pg - pr).max()
        print(f"{name:16s} n{n} {h}x{w}: logit rel err {rel:.2e}  max|dp| {err:.2e}", "" if rel < 1e-2 else "  <<<<<", flush=True)
