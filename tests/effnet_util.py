"""Synthetic EfficientNet test networks: generator-seeded weights whose BatchNorm running
statistics are calibrated on a generator-seeded batch (exactly what tests/golden/make_golden.py
did with the reference's own TorchVisionNet), plus the stored logit-centring bias shift."""

import numpy as np
import torch

from oracle import refnet
from sykepic_hip import arch, synth


def calibrated_state(network, hw, gold, classes=50):
    g = arch.build_graph(network, classes)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = refnet.load_numpy_state(refnet.RefNet(network, classes), sd)
    refnet.calibrate_bn(net, torch.from_numpy(synth.synth_images(16, 3, hw, hw, seed=99)))
    sd = {k: v.numpy().copy() for k, v in net.state_dict().items()}
    last = [k for k in sd if k.startswith("head.") and k.endswith(".bias")][-1]
    sd[last] = sd[last] + gold[f"{network}_{hw}_bias_adj"]
    return g, sd, refnet.load_numpy_state(net, sd)
