"""Diagnostic (not a test): the calibrated single-pass mode (zero-sum rounded fp16 weights, csrc/zero_sum.hip) on the
EfficientNets - golden vectors of the reference's net_pass and fresh images against the oracle, beside `mixed` (hi + lo
weights on every 1x1 conv).  Run on the GPU box from the repo root:  python tests/archive/diagnostics/effnet_calibrated.py"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd"), str(ROOT / "tests")]
from sykepic_hip import synth  # noqa: E402
from sykepic_hip.net import HipNet  # noqa: E402
from sykepic_hip.prob import net_pass  # noqa: E402
from effnet_util import calibrated_state  # noqa: E402
from oracle import refnet  # noqa: E402

gold = np.load(ROOT / "tests" / "golden" / "net_pass_effnet.npz")
for network in ("efficientnet_b0", "efficientnet_b4"):
    tag = f"{network}_224"
    g, sd, ref = calibrated_state(network, 224, gold)
    n = len(gold[f"{tag}_rois_in"])
    x = torch.from_numpy(synth.synth_images(n, 3, 224, 224, seed=0))
    paths = [f"/x/D20180712T065600_IFCB114_{int(r):05d}.png" for r in gold[f"{tag}_rois_in"]]
    want = gold[f"{tag}_probs"].astype(np.float64)
    x2 = torch.cat([torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=21 + i)) for i in range(2)])
    z = np.concatenate([refnet.probabilities(ref, x2[i:i + 16], base=0).numpy() for i in (0, 16)])
    for mode in ("mixed", "calibrated", "fast"):
        net = HipNet(network, 50, weights=None)
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        net.eval()
        if mode == "calibrated":
            net.calibrate(torch.from_numpy(synth.synth_images(32, 3, 224, 224, seed=9000)).cuda())
            net.set_precision("calibrated")
        elif mode == "fast":
            net.set_precision(split_weights=0)
        res = net_pass(net, [(x.cuda(), paths)], "cuda:0")
        p = np.array([q for _, q in res], dtype=np.float64)
        zg = net.forward(x2.cuda()).cpu().numpy()
        per_img = np.sqrt(np.mean((zg - z) ** 2, 1)) / z.std()
        s = float(np.log(1.3)) / 4
        pr = torch.softmax(torch.from_numpy(z) * s, 1).numpy()
        pg = torch.softmax(torch.from_numpy(zg) * s, 1).numpy()
        d = np.abs(pg - pr).max(1)
        print(f"{network} {mode:10s} golden max|dp| {np.abs(p - want).max():.2e}   fresh: logit rms rel median {np.median(per_img):.2e} "
              f"max {per_img.max():.2e}   |dp| median {np.median(d):.2e} p90 {np.quantile(d, 0.9):.2e} max {d.max():.2e}")
