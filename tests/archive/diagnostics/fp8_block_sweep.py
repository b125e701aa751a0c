"""Which MBConv blocks can run on the e4m3 path?  (VERDICT r2 item 4 / DESIGN.md section 5.)
EfficientNet-B4 at 224x224 with the calibrated synthetic weights of the goldens: for several sets of fp8 blocks,
agreement of the fp8 mode with the fp16 parity mode on 64 fresh images (top-1, |dp| median / p90 / max at the reference's
softmax base 1.3) and the forward time at batch 128.  Usage (GPU box, repo root): python tests/archive/diagnostics/fp8_block_sweep.py
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "syke-pic_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))

from effnet_util import calibrated_state  # noqa: E402
from sykepic_hip import synth  # noqa: E402
from sykepic_hip.net import HipNet  # noqa: E402


def main(network="efficientnet_b4"):
    gold = np.load(ROOT / "tests" / "golden" / "net_pass_effnet.npz")
    g, sd, _ = calibrated_state(network, 224, gold)
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    fresh = torch.cat([torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=21 + i)) for i in range(4)]).cuda()
    calib = torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=99)).cuda()
    xb = torch.from_numpy(synth.synth_images(128, 3, 224, 224, seed=5)).cuda()
    p16 = net.probabilities(fresh).cpu()
    nb = net.num_fp8_blocks()
    top2 = p16.topk(2, 1).values
    print(f"{network}: {nb} qualifying blocks; fp16 top-1 margins on the 64 images: median {float((top2[:, 0] - top2[:, 1]).median()):.3f}")

    def timed():
        for _ in range(3):
            net.probabilities(xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net.probabilities(xb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3

    net.set_fp8(False)
    t16 = timed()
    print(f"{'set':28s} {'top-1':>6s} {'med|dp|':>9s} {'p90|dp|':>9s} {'max|dp|':>9s} {'ms@128':>7s}")
    print(f"{'none (fp16)':28s} {1.0:6.3f} {0:9.2e} {0:9.2e} {0:9.2e} {t16:7.2f}")

    def run(name, flags):
        net.set_fp8(True, calibration_batch=calib, blocks=flags)
        p8 = net.probabilities(fresh).cpu()
        dp = (p8 - p16).abs().max(1).values
        agree = float((p8.argmax(1) == p16.argmax(1)).float().mean())
        ms = timed()
        net.set_fp8(False)
        print(f"{name:28s} {agree:6.3f} {float(dp.median()):9.2e} {float(dp.quantile(0.9)):9.2e} {float(dp.max()):9.2e} {ms:7.2f}",
              flush=True)
        return agree, float(dp.quantile(0.9))

    run("all", [1] * nb)
    for k in (1, 2, 4, 8, 12, 16, 24):
        if k < nb:
            run(f"last {k}", [0] * (nb - k) + [1] * k)
    for k in (1, 2, 4, 8, 12, 16, 24):
        if k < nb:
            run(f"first {k}", [1] * k + [0] * (nb - k))
    for i in range(nb):
        run(f"only block {i}", [1 if j == i else 0 for j in range(nb)])


if __name__ == "__main__":
    main(*sys.argv[1:])
