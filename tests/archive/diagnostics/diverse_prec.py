"""Diagnostic (not a test): max |dp| on the diverse-top-1 golden fixture per eval precision mode."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd"), str(ROOT / "tests")]
import numpy as np
import torch
from test_oracle_golden import diverse_case
from sykepic_hip.net import HipNet

for network, hw in (("resnet18", 180), ("resnet50", 224)):
    g, sd, x, paths, rois_out, ref = diverse_case(ROOT / "tests" / "golden", network, hw)
    order = np.argsort([int(p.split("_")[-1].split(".")[0]) for p in paths])
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    for name, sw, pr in (("fast", 0, False), ("balanced", 2, False), ("mixed", 3, False), ("precise", 1, False),
                         ("precise+res", 1, True), ("mixed+res", 3, True)):
        net.set_precision(split_weights=sw, precise_residual=pr)
        p = net.probabilities(x.cuda()).cpu().numpy().astype(np.float64)[order]
        print(f"{network}_{hw} {name:12s} max |dp| = {np.abs(p - ref).max():.2e}  top-1 ok {(p.argmax(1) == ref.argmax(1)).all()}")
