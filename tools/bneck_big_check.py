"""Diagnostics: the one-kernel bottleneck against the three launches at a batch that makes blocks share CUs."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd"), str(ROOT / "tests")]
import torch
from sykepic_hip import ops
from test_gpu_bneck import _block
for (n, hw, cm) in ((256, 14, 256), (160, 28, 128), (300, 14, 256)):
    x, w1, w2, w3, bns = _block(n, hw, cm, seed=3)
    d = lambda t: t.cuda()
    dbns = [(d(a), d(b)) for a, b in bns]
    three = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=0).float().cpu()
    for it in range(3):
        fused = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=1).float().cpu()
        bad = (fused != three)
        print(n, hw, cm, "iteration", it, "equal" if not bad.any() else f"DIFFERENT in {int(bad.sum())} elements, images {sorted(set(bad.nonzero()[:, 0].tolist()))[:12]}", flush=True)
