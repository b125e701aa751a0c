"""ORACLE (test infrastructure, never shipped on the product path).

CPU restatement, in torch fp32 NCHW, of the reference's CNN classification
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.

Pinned against the real reference: ``tests/golden/make_golden.py`` runs the
reference's own ``TorchVisionNet`` / ``net_pass`` / ``train_net`` /
``LRWarmup`` (imported from ``/root/reference`` with a ``sys.modules`` shim
for the two absent third-party packages) on the same generator-seeded
tensors and stores their outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against those vectors.

What follows what (reference paths relative to ``/root/reference``):
  RefNet               sykepic/train/network.py:11-72
  freeze_base          sykepic/train/network.py:133-172
  non_bn_trainable     sykepic/train/network.py:175-187
  warmup_step          sykepic/train/network.py:98-130
  net_pass             sykepic/compute/probability.py:180-197
  train_step/eval_step sykepic/train/train.py:237-248, 262-270
"""

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import backbones

SOFTMAX_BASE = 1.3  # sykepic/compute/probability.py:18
_BN = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


class RefNet(nn.Module):
    """backbone minus its last child + affine head (Linear stack, no
    activations); Dropout inserted by list index."""

    def __init__(self, name, num_classes, head=(256, 128), dropout=()):
        super().__init__()
        children = list(backbones.make(name).children())
        last = children[-1]
        if isinstance(last, nn.Sequential):   # EfficientNet classifier = [Dropout, Linear] (network.py:50-55)
            last = next(m for m in last if isinstance(m, nn.Linear))
        feat = last.in_features
        widths = [feat] + [int(h) for h in head] + [int(num_classes)]
        mods = [nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])]
        for idx, p in dropout:
            mods.insert(int(idx), nn.Dropout(float(p)))
        self.base = nn.Sequential(*children[:-1])
        self.head = nn.Sequential(*mods)

    def forward(self, x):
        f = self.base(x)
        return self.head(f.reshape(f.shape[0], -1))


def _leaves(module):
    kids = list(module.children())
    if not kids:
        yield module
    for k in kids:
        yield from _leaves(k)


def freeze_base(base):
    """Every leaf of ``base``: BatchNorm stays trainable (train mode), the
    rest is frozen (eval mode)."""
    for leaf in _leaves(base):
        keep = isinstance(leaf, _BN)
        for p in leaf.parameters():
            p.requires_grad = keep
        leaf.train(keep)


def unfreeze(part):
    for p in part.parameters():
        p.requires_grad = True
    part.train()


def non_bn_trainable(part):
    return [p for leaf in _leaves(part) if not isinstance(leaf, _BN)
            for p in leaf.parameters() if p.requires_grad]


def make_optimizer(net, name, lr):
    """Three param groups: trainable-now / empty / empty
    (sykepic/train/train.py:131-138)."""
    first = [p for p in net.parameters() if p.requires_grad]
    return getattr(torch.optim, name)([
        {"params": first, "lr": lr},
        {"params": [], "lr": 0.0},
        {"params": [], "lr": 0.0},
    ])


def warmup_step(net, opt, epoch, f1, f2, s1, s2, s3):
    g = opt.param_groups
    if epoch == s1:
        g[0]["lr"] *= f1
    elif epoch == s2:
        part = net.base[-2:]
        unfreeze(part)
        g[1]["params"] = non_bn_trainable(part)
        g[1]["lr"] = g[0]["lr"] * f1
        g[0]["lr"] *= f2
    elif epoch == s3:
        part = net.base[:-2]
        unfreeze(part)
        g[2]["params"] = non_bn_trainable(part)
        g[2]["lr"] = g[1]["lr"] * f1
        g[0]["lr"] *= f2


def roi_of(path):
    stem = str(path).rsplit("/", 1)[-1].rsplit(".", 1)[0]
    return int(stem.split("_")[-1])


def net_pass(net, batches, base=SOFTMAX_BASE):
    """batches: iterable of (x [B,C,H,W] float32, paths).  Returns
    [(roi, [p...])] sorted by roi id."""
    rows = []
    net.eval()
    with torch.no_grad():
        for x, paths in batches:
            z = net(x)
            if base:
                z = z * np.log(base)
            p = F.softmax(z, dim=1)
            rows.extend(zip((roi_of(q) for q in paths), p.tolist()))
    return sorted(rows)


def probabilities(net, x, base=SOFTMAX_BASE):
    net.eval()
    with torch.no_grad():
        z = net(x)
        return F.softmax(z * math.log(base), dim=1) if base else z


def train_step(net, opt, x, y):
    """One iteration of the hot loop; returns (loss, n_correct, logits)."""
    opt.zero_grad()
    out = net(x)
    loss = F.cross_entropy(out, y)
    loss.backward()
    opt.step()
    correct = int((out.argmax(1) == y).sum())
    return float(loss.detach()), correct, out.detach()


def eval_step(net, x, y):
    with torch.no_grad():
        out = net(x)
        loss = F.cross_entropy(out, y)
    return float(loss), int((out.argmax(1) == y).sum()), out


def calibrate_bn(net, x):
    """Replace every BatchNorm's running statistics by the statistics of batch ``x`` (one
    train-mode forward with momentum 1), as a trained network's would match its data.  Used
    to give deep random-weight test networks (EfficientNet) O(1) activations at every depth;
    returns the net in eval mode."""
    bns = [m for m in net.modules() if isinstance(m, _BN)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    net.train()
    with torch.no_grad():
        net(x)
    for m, mom in zip(bns, old):
        m.momentum = mom
        m.num_batches_tracked.zero_()
    return net.eval()


def load_numpy_state(net, state):
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}
    net.load_state_dict(sd)
    return net
