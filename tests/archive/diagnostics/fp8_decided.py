"""fp8 mode of EfficientNet-B4 / B0 on DECIDED images: top-1 agreement with the fp16 parity mode as a function of the fp16
top-1 margin (p1 - p2 at the reference's softmax base 1.3), 256 fresh images.  Sets: every block on the e4m3 path; every
block but the first of each stage (the stride-2 / widening blocks, the sensitive ones in fp8_block_sweep.py).
Usage (GPU box, repo root): python tests/archive/diagnostics/fp8_decided.py [network]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "syke-pic_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))

from effnet_util import calibrated_state  # noqa: E402
from sykepic_hip import arch, synth  # noqa: E402
from sykepic_hip.net import HipNet  # noqa: E402


def main(network="efficientnet_b4"):
    gold = np.load(ROOT / "tests" / "golden" / "net_pass_effnet.npz")
    g, sd, _ = calibrated_state(network, 224, gold)
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    fresh = [torch.from_numpy(synth.synth_images(32, 3, 224, 224, seed=300 + i)).cuda() for i in range(8)]
    calib = torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=99)).cuda()
    p16 = torch.cat([net.probabilities(x).cpu() for x in fresh])
    top2 = p16.topk(2, 1).values
    margin = top2[:, 0] - top2[:, 1]
    nb = net.num_fp8_blocks()
    # first qualifying block of every stage: its expand conv reads a trunk tensor no other block of the stage wrote
    exp = [o for o in g.ops if o.kind == arch.OP_CONV and o.k == 1 and int(o.relu) == arch.ACT_SILU and o.src != 0
           and o.name != "base.0.8.0"]
    assert len(exp) == nb, (len(exp), nb)
    first = [o.name.split(".")[3] == "0" for o in exp]
    sets = {"all blocks": [1] * nb, "all but the first block of each stage": [0 if f else 1 for f in first]}
    for name, flags in sets.items():
        net.set_fp8(True, calibration_batch=calib, blocks=flags)
        p8 = torch.cat([net.probabilities(x).cpu() for x in fresh])
        net.set_fp8(False)
        same = p8.argmax(1) == p16.argmax(1)
        dp = (p8 - p16).abs().max(1).values
        print(f"{network}, {name} ({sum(flags)} of {nb}): |dp| median {float(dp.median()):.2e} p90 {float(dp.quantile(0.9)):.2e}")
        for thr in (0.0, 0.01, 0.02, 0.05, 0.1, 0.2):
            sel = margin >= thr
            if int(sel.sum()):
                print(f"   fp16 margin >= {thr:4.2f}: {int(sel.sum()):3d} images, top-1 agreement {float(same[sel].float().mean()):.3f}")


if __name__ == "__main__":
    main(*sys.argv[1:])
