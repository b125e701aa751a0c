"""Soak run for intermittent faults (races, dead loads, stale hand-offs) in the eval and training paths: the same inputs
again and again, every result compared bit for bit with the first one.  usage: python3 tools/soak.py [iterations]
Prints one line per configuration: mismatching iterations / iterations."""
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
from sykepic_hip import arch, synth  # noqa: E402
from sykepic_hip.net import HipNet  # noqa: E402
from sykepic_hip.optim import HipOptimizer  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def eval_soak(network, batches, hw, precision):
    g = arch.build_graph(network, 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=3)
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    if precision == "calibrated":
        net.calibrate(torch.from_numpy(synth.synth_images(32, 3, hw, hw, seed=99)).cuda())
    net.set_precision("calibrated" if precision == "calibrated" else 3)   # 3 = the default split rule ("mixed")
    for n in batches:
        x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=n)).cuda()
        for _ in range(3):
            ref = sha(net.forward(x))     # (the first forwards of a shape tune; every candidate gives the same bits)
        bad = sum(sha(net.forward(x)) != ref for _ in range(ITERS))
        print(f"eval  {network} {precision} batch {n} @{hw}: {bad} / {ITERS} mismatches", flush=True)


def train_soak(network, n, hw, steps):
    finals = []
    for _ in range(2):
        net = HipNet(network, 10, weights=None, head=(64, 32))
        net.reset_parameters(seed=7)
        net.set_seed(7)
        for p in net.parameters():
            p.requires_grad = True
        opt = HipOptimizer(net, "Adam", [{"params": list(net.parameters()), "lr": 1e-3}])
        net.train()
        for s in range(steps):
            x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=1000 + s % 4)).cuda()
            y = torch.from_numpy(synth.synth_labels(n, 10, seed=2000 + s % 4)).cuda()
            net.forward_backward(x, y)
            opt.step()
        h = hashlib.sha256()
        for k, v in sorted(net.state_dict().items()):
            h.update(v.cpu().numpy().tobytes())
        finals.append(h.hexdigest())
    print(f"train {network} batch {n} @{hw}, {steps} steps twice: {'identical' if finals[0] == finals[1] else 'DIFFERENT'} weights", flush=True)


if __name__ == "__main__":
    eval_soak("resnet50", (256, 70, 33), 224, "calibrated")
    eval_soak("resnet50", (128,), 224, "mixed")
    eval_soak("resnet18", (512, 65), 224, "calibrated")
    eval_soak("efficientnet_b4", (128,), 224, "mixed")
    train_soak("resnet50", 64, 224, 40)
    train_soak("resnet18", 128, 128, 40)
