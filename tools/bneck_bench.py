"""Times one identity bottleneck block of the eval path: the one-kernel form (csrc/conv_bneck.hip) against the three
launches of the eval path's own kernels, through spk_op_bottleneck.  usage: bneck_bench.py [n ...]"""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import torch
from sykepic_hip import ops

SHAPES = {"stage3": (14, 256), "stage2": (28, 128), "stage1": (56, 64)}
ns = [int(v) for v in sys.argv[1:]] or [256, 128]
for name, (hw, cm) in SHAPES.items():
    for n in ns:
        g = torch.Generator().manual_seed(1)
        c4 = 4 * cm
        x = torch.relu(torch.randn(n, c4, hw, hw, generator=g)).half().cuda()
        w1 = ((torch.rand(cm, c4, 1, 1, generator=g) * 2 - 1) * (6.0 / c4) ** 0.5).cuda()
        w2 = ((torch.rand(cm, cm, 3, 3, generator=g) * 2 - 1) * (6.0 / (9 * cm)) ** 0.5).cuda()
        w3 = ((torch.rand(c4, cm, 1, 1, generator=g) * 2 - 1) * (6.0 / cm) ** 0.5).cuda()
        bns = [((0.5 + torch.rand(c, generator=g)).cuda(), (torch.rand(c, generator=g) - 0.5).cuda()) for c in (cm, cm, c4)]
        gflop = 2.0 * n * hw * hw * (2 * cm * c4 + 9 * cm * cm) / 1e9
        mbytes = 2.0 * n * hw * hw * c4 * 2 / 1e6
        line = f"{name} n={n} {hw}x{hw} cm={cm}: {gflop:.1f} GFLOP, {mbytes:.0f} MB in+out;"
        for fused, what in ((0, "three launches"), (1, "one kernel"), (2, "conv1 + tail kernel")):
            try:
                _, ms = ops.bottleneck(x, w1, w2, w3, *bns, fused=fused, iters=20)
                line += f"  {what} {ms * 1e3:.1f} us = {gflop / ms:.0f} TFLOP/s, {mbytes / ms:.0f} GB/s;"
            except RuntimeError as e:
                line += f"  {what}: {str(e)[-40:]};"
        if hw == 56:   # with the next block's conv1 chained (256 -> 64): four launches against conv1 + one kernel
            wz = ((torch.rand(64, c4, 1, 1, generator=g) * 2 - 1) * (6.0 / c4) ** 0.5).cuda()
            bnz = ((0.5 + torch.rand(64, generator=g)).cuda(), (torch.rand(64, generator=g) - 0.5).cuda())
            for fused, what in ((0, "four launches (+ next conv1)"), (2, "conv1 + tail kernel with the chained conv")):
                _, ms = ops.bottleneck(x, w1, w2, w3, *bns, fused=fused, iters=20, wz=wz, bnz=bnz)
                line += f"  {what} {ms * 1e3:.1f} us;"
        print(line, flush=True)
        if "STAMPS" in __import__("os").environ:
            # phase breakdown of the fused kernel from shader-clock stamps (one launch, median over the blocks)
            blocks = n * (hw // {14: 14, 28: 14, 56: 8}[hw])
            blocks *= 2 if os.environ.get("SPK_BNECK_FLAGS", "0") in ("4", "5") else 1
            st = torch.zeros(blocks, 16, dtype=torch.int64, device="cuda")
            try:
                ops.bottleneck(x, w1, w2, w3, *bns, fused=True, stamps=st)
            except RuntimeError:
                continue
            for half, o in (("first wave", 0), ("first wave of the second half", 8)):
                t = st.cpu().double()[:, o:o + 8]
                d = lambda a, b: float((t[:, b] - t[:, a]).median())  # noqa: E731
                print(f"   {half}, clocks (median over {blocks} blocks): setup->phase1 K loop {d(0, 1):.0f}, y1 epilogue {d(1, 2):.0f}, phase 2 K loop "
                      f"{d(2, 3):.0f}, y2 epilogue {d(3, 4):.0f}, phase 3 {d(4, 5):.0f} (pass 0: K loop {d(4, 7):.0f}, epilogue {d(7, 6):.0f}); "
                      f"whole {d(0, 5):.0f}", flush=True)
