"""Is the training step bound by the host's launch rate?  Prints, per step, the time the (asynchronous) C call takes
to return and the step time with a synchronisation behind every step.
Usage: python3 tools/launch_bound_probe.py [network=efficientnet_b0] [batch=128] [size=224]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "syke-pic_amd"))
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
from sykepic_hip.optim import HipOptimizer

network = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
dev = torch.device("cuda", 0)
g = arch.build_graph(network, 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
net = HipNet(network, 50, weights=None, device=dev)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
net.set_precision(split_weights=False, bf16=True)
net.train()
for p in net.parameters():
    p.requires_grad = True
opt = HipOptimizer(net, "Adam", [{"params": list(net.parameters()), "lr": 1e-4}, {"params": [], "lr": 0.0},
                                 {"params": [], "lr": 0.0}])
x = torch.from_numpy(synth.synth_images(batch, 3, size, size, seed=0)).to(dev)
y = torch.from_numpy(synth.synth_labels(batch, 50, seed=1)).to(dev)
for _ in range(8):
    net.forward_backward(x, y)
    opt.step()
torch.cuda.synchronize()
host, full = [], []
for _ in range(10):
    t0 = time.perf_counter()
    net.forward_backward(x, y)
    opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    full.append((t2 - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(10):
    net.forward_backward(x, y)
    opt.step()
torch.cuda.synchronize()
back = (time.perf_counter() - t0) * 100
print(f"{network} batch {batch} @{size}: host enqueue {np.median(host):.2f} ms, step with a sync behind it "
      f"{np.median(full):.2f} ms, back-to-back {back:.2f} ms per step")
