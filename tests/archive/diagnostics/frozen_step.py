"""Training-step time with the base frozen (head only; the BatchNorm layers of the base stay trainable and in the
optimizer, as sykepic/train/network.py:149-172 + train.py:131 leave them), the last two base modules unfrozen, and
everything unfrozen - the three phases of the reference's LRWarmup schedule - and with the BatchNorm layers frozen
too, where the backward pass stops at the head (GPU diagnostic)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import torch
from sykepic_hip import schedule
from sykepic_hip.net import HipNet
from sykepic_hip.optim import HipOptimizer

net = HipNet("resnet50", 50, weights=None, head=(256, 128)).to("cuda:0")
net.train()
x = torch.rand((256, 3, 224, 224), device="cuda:0")
y = torch.randint(0, 50, (256,), device="cuda:0")


def timed(tag):
    params = [p for p in net.parameters() if p.requires_grad]
    opt = HipOptimizer(net, "Adam", [{"params": params, "lr": 1e-3}])
    for it in range(13):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.time()
        opt.zero_grad()
        net.forward_backward(x, y)
        opt.step()
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 10 * 1e3
    print(f"{tag}: {len(params)} trainable tensors, {ms:.2f} ms / step = {256 / ms * 1e3:.0f} images/s", flush=True)


schedule.freeze(net.base)
timed("base frozen (head only)")
schedule.make_trainable(net.base[-2:])
timed("+ last two base modules")
schedule.make_trainable(net.base[:-2])
timed("everything trainable")
for p in net.base.parameters():
    p.requires_grad = False
timed("base frozen including its BatchNorm layers")
