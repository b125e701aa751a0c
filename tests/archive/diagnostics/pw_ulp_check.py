import sys, torch
sys.path.insert(0, "syke-pic_amd"); sys.path.insert(0, "tests")
from sykepic_hip import ops
from test_gpu_pw import _case
dev = "cuda:0"
for shape in [(2, 28, 28, 128, 512, 1, True, True), (3, 14, 14, 1024, 256, 1, False, True), (2, 56, 56, 64, 256, 1, False, False)]:
    n, h, w, cin, cout, stride, res, relu = shape
    x, wgt, scale, bias, r, ref = _case(n, h, w, cin, cout, stride, res, relu, seed=3)
    ref64 = torch.nn.functional.conv2d(x.double(), wgt.double()[:, :, None, None], stride=stride) * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]
    if r is not None: ref64 = ref64 + r.double()
    if relu: ref64 = torch.relu(ref64)
    for split in (True, False):
        for cfg in (-1, 1, 7, 9):
            try:
                y = ops.conv1x1(x.to(dev), wgt.to(dev), scale.to(dev), bias.to(dev), stride=stride, relu=relu, res=r.to(dev) if r is not None else None, split=split, cfg=cfg).double().cpu()
            except RuntimeError as e:
                continue
            ulp = torch.maximum(ref64.abs(), torch.tensor(2.0 ** -14, dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -10
            e = ((y - ref64).abs() / ulp)
            print(shape, "split" if split else "plain", "cfg", cfg, "max err %.3f ulp, mean %.4f ulp, rms rel %.3e" % (float(e.max()), float(e.mean()), float(((y - ref64) ** 2).mean().sqrt() / (ref64 ** 2).mean().sqrt())))
