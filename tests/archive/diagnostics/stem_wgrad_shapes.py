"""Diagnostic (not a test): stem weight gradient over image shapes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import torch
from sykepic_hip import ops

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))

for n, h, w in [(3, 37, 37), (3, 36, 28), (3, 37, 30), (3, 36, 29), (3, 37, 29), (1, 37, 29), (2, 64, 32), (2, 32, 64), (5, 75, 75)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((n, 3, h, w), generator=g).bfloat16()
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    dy = torch.randn((n, 64, oh, ow), generator=g).bfloat16()
    want = torch.nn.grad.conv2d_weight(x.float(), (64, 3, 7, 7), dy.float(), 2, 3)
    got = ops.conv_wgrad(x.cuda(), dy.cuda(), 7, 2, 3).cpu()
    per_tap = [(r, s, rel(got[:, :, r, s], want[:, :, r, s])) for r in range(7) for s in range(7)]
    bad = [(r, s) for r, s, e in per_tap if e > 1e-3]
    print(f"n{n} {h}x{w}: rel {rel(got, want):.2e}; bad taps {bad[:12]}{'...' if len(bad) > 12 else ''}")
