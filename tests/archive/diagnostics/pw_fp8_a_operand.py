"""How much of an fp8 expand conv is the fp16 -> e4m3 conversion of its A operand?  Times spk_op_pw_fp8 on the
EfficientNet-B4 expand shapes (batch 256) with the A operand given as fp16 (converted in the loader, once per N tile)
and as e4m3 bytes (no conversion) - GPU diagnostic."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import torch
from sykepic_hip import ops

shapes = [(56, 32, 192), (14, 112, 672), (14, 160, 960), (7, 272, 1632), (7, 448, 2688)]   # cin % 16 == 0
for hw, cin, cout in shapes:
    m = 256 * hw * hw
    g = torch.Generator().manual_seed(cin)
    w = torch.randn((cout, cin), generator=g).cuda() * (2.0 / cin) ** 0.5
    s = torch.rand(cout, generator=g).cuda() + 0.5
    b = torch.randn(cout, generator=g).cuda() * 0.1
    x16 = (torch.randn((m, cin), generator=g) * 1.5).half().cuda()
    x8 = (x16.float() / 0.02).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    out = {}
    for tag, x in (("fp16 A", x16), ("e4m3 A", x8)):
        for it in range(6):
            if it == 1:
                torch.cuda.synchronize(); t0 = time.time()
            y = ops.pw_fp8(x, w, s, b, act=2, a_scale=0.02, y_scale=0.03, out_fp8=True)
        torch.cuda.synchronize()
        out[tag] = (time.time() - t0) / 5 * 1e6
    print(f"{hw}x{hw} {cin}->{cout}: fp16 A {out['fp16 A']:.0f} us, e4m3 A {out['e4m3 A']:.0f} us (includes the hook's weight packing)", flush=True)
