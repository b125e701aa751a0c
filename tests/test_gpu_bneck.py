"""A whole identity bottleneck block as ONE kernel (csrc/conv_bneck.hip) through `spk_op_bottleneck`.

Stands in for torchvision's `Bottleneck.forward` without a downsample branch - conv1x1 + BN + ReLU, conv3x3 + BN + ReLU,
conv1x1 + BN, + x, ReLU - inside `net(x)` (sykepic/compute/probability.py:189).  Two checks per shape:
  * against the same block as the eval path's three launches (conv_pw / conv_c3 / conv_pw): the fused kernel keeps their K
    orders and epilogue arithmetic and rounds the two mid tensors to fp16 exactly where they are stored, so the outputs
    are BIT-IDENTICAL;
  * against torch fp32 on the CPU with the mid tensors rounded to fp16 at the same two points."""

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # n, hw, cm
    (3, 14, 256),      # ResNet-50 stage 3 (a block owns a whole image)
    (1, 14, 256),
    (9, 14, 256),      # more blocks than one XCD's share of a small grid
    (3, 28, 128),      # stage 2: two bands per image, one halo row each side (top band: none above, bottom band: none below)
    (1, 28, 128),
]


def _block(n, hw, cm, seed):
    g = torch.Generator().manual_seed(seed)
    c4 = 4 * cm
    x = torch.relu(torch.randn(n, c4, hw, hw, generator=g)).half()
    w1 = (torch.rand(cm, c4, 1, 1, generator=g) * 2 - 1) * (6.0 / c4) ** 0.5
    w2 = (torch.rand(cm, cm, 3, 3, generator=g) * 2 - 1) * (6.0 / (9 * cm)) ** 0.5
    w3 = (torch.rand(c4, cm, 1, 1, generator=g) * 2 - 1) * (6.0 / cm) ** 0.5
    bns = []
    for c in (cm, cm, c4):
        bns.append((0.5 + torch.rand(c, generator=g), torch.rand(c, generator=g) - 0.5))
    return x, w1, w2, w3, bns


def _ref(x, w1, w2, w3, bns):
    F = torch.nn.functional
    # weights as the kernels see them (fp16), activations fp32 between the rounding points
    bn = lambda t, p: t * p[0][None, :, None, None] + p[1][None, :, None, None]  # noqa: E731
    y1 = torch.relu(bn(F.conv2d(x.float(), w1.half().float()), bns[0])).half().float()
    y2 = torch.relu(bn(F.conv2d(y1, w2.half().float(), padding=1), bns[1])).half().float()
    return torch.relu(bn(F.conv2d(y2, w3.half().float()), bns[2]) + x.float())


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_%dx%d_cm%d" % (s[0], s[1], s[1], s[2]))
def test_fused_bottleneck_equals_the_three_launches(shape):
    from sykepic_hip import ops
    n, hw, cm = shape
    x, w1, w2, w3, bns = _block(n, hw, cm, seed=hw + cm + n)
    dev = "cuda:0"
    d = lambda t: t.to(dev)  # noqa: E731
    dbns = [(d(a), d(b)) for a, b in bns]
    three = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=False).float().cpu()
    fused = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=True).float().cpu()
    assert torch.isfinite(fused).all(), "an output element was never written"
    ref = _ref(x, w1, w2, w3, bns)
    err = (three - ref).abs()
    bound = 2e-2 + 1e-2 * ref.abs()       # (a flipped fp16 rounding of a mid value moves an output by ~ its weight)
    assert (err <= bound).all(), float(err.max())
    assert torch.equal(fused, three), (float((fused - three).abs().max()), int((fused != three).sum()))


TAIL_SHAPES = [
    # n, hw, cm, coutz (0: no chained conv)
    (2, 56, 64, 0),       # ResNet-50 stage 1: bands of 4 rows, 14 per image
    (3, 56, 64, 64),      # + the next block's conv1 (256 -> 64) from the output tile in registers
    (1, 56, 64, 128),     # + the next STAGE's conv1 (256 -> 128)
]


@pytest.mark.parametrize("shape", TAIL_SHAPES, ids=lambda s: "n%d_%dx%d_cm%d_z%d" % (s[0], s[1], s[1], s[2], s[3]))
def test_block_tail_kernel_equals_the_launches_it_replaces(shape):
    """conv2 + conv3 + shortcut (+ the conv that reads the block's output) as ONE kernel (conv_btail_kernel) on the stage whose
    trunk is too wide for the whole-block kernel: bit-identical to conv_c3 -> conv_pw (-> conv_pw) on the same conv1 output."""
    from sykepic_hip import ops
    n, hw, cm, coutz = shape
    x, w1, w2, w3, bns = _block(n, hw, cm, seed=hw + cm + n + coutz)
    g = torch.Generator().manual_seed(7 + coutz)
    c4 = 4 * cm
    wz = ((torch.rand(coutz, c4, 1, 1, generator=g) * 2 - 1) * (6.0 / c4) ** 0.5) if coutz else None
    bnz = (0.5 + torch.rand(coutz, generator=g), torch.rand(coutz, generator=g) - 0.5) if coutz else None
    dev = "cuda:0"
    d = lambda t: t.to(dev)  # noqa: E731
    dbns = [(d(a), d(b)) for a, b in bns]
    kw = dict(wz=d(wz), bnz=(d(bnz[0]), d(bnz[1]))) if coutz else {}
    ref = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=0, **kw)
    got = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=2, **kw)
    if not coutz:
        ref, got = (ref,), (got,)
    for name, r, t in zip(("out", "z"), ref, got):
        r, t = r.float().cpu(), t.float().cpu()
        assert torch.isfinite(t).all(), f"{name}: an element was never written"
        assert torch.equal(t, r), (name, float((t - r).abs().max()), int((t != r).sum()))
    want = _ref(x, w1, w2, w3, bns)
    err = (got[0].float().cpu() - want).abs()
    assert (err <= 2e-2 + 1e-2 * want.abs()).all(), float(err.max())


def test_network_output_does_not_depend_on_the_form_the_blocks_run_in():
    """ResNet-50 at 224 x 224 (stage 2: 28 x 28 x 512, stage 3: 14 x 14 x 1024): the identity bottlenecks of stages 2-3 as
    three launches (SPK_BNECK=0), as 14-row blocks (2) and as 7-row blocks (3), on one stream and on two, for a batch that
    takes the two-stream path (64) and a ragged single-stream one (33): the same logits bit for bit, call after call - a
    row of probabilities does not depend on the batch it was computed in or on what the tuner picked."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, hashlib, numpy as np, torch\n"
        f"sys.path[:0] = [{str(root)!r}, {str(root / 'syke-pic_amd')!r}]\n"
        "from sykepic_hip import arch, synth\n"
        "from sykepic_hip.net import HipNet\n"
        "g = arch.build_graph('resnet50', 50)\n"
        "sd = synth.synth_state_dict(arch.param_specs(g), seed=2)\n"
        "net = HipNet('resnet50', 50, weights=None)\n"
        "net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()\n"
        "net.calibrate(torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=9000)).cuda())\n"
        "net.set_precision('calibrated')\n"
        "for n in (64, 33):\n"
        "    x = torch.from_numpy(synth.synth_images(n, 3, 224, 224, seed=5)).cuda()\n"
        "    for it in range(3):\n"
        "        z = net.forward(x).cpu().numpy()\n"
        "        print('SHA', n, hashlib.sha256(z.tobytes()).hexdigest())\n")
    seen = {}
    for env in ({"SPK_BNECK": "0"}, {"SPK_BNECK": "2"}, {"SPK_BNECK": "3"}, {"SPK_BNECK": "3", "SPK_EVAL_STREAMS": "1"},
                {"SPK_BNECK": "1"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPK_TUNE_CACHE="off", **env), capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        for ln in out.stdout.splitlines():
            if ln.startswith("SHA"):
                _, n, h = ln.split()
                seen.setdefault(n, set()).add(h)
    assert set(seen) == {"64", "33"} and all(len(v) == 1 for v in seen.values()), seen


@pytest.mark.parametrize("flags", ["0", "4"], ids=["14-row blocks", "7-row blocks"])
def test_blocks_that_share_and_inherit_cus(flags):
    """More blocks than the chip holds at once (7-row form: two per CU, later blocks start beside running ones): still the
    three launches' bits, run after run.  (Round 5 found one wrong image in a few hundred blocks here: phase 1 issued a last,
    dead weight load through the instruction form the compiler does not track - buffer_load_b128_untracked - and it landed
    in registers that were live again.)"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, torch\n"
        f"sys.path[:0] = [{str(root)!r}, {str(root / 'syke-pic_amd')!r}, {str(root / 'tests')!r}]\n"
        "from sykepic_hip import ops\n"
        "from test_gpu_bneck import _block\n"
        "for (n, hw, cm) in ((300, 14, 256), (160, 28, 128)):\n"
        "    x, w1, w2, w3, bns = _block(n, hw, cm, seed=3)\n"
        "    d = lambda t: t.cuda()\n"
        "    dbns = [(d(a), d(b)) for a, b in bns]\n"
        "    three = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=0).cpu()\n"
        "    for it in range(3):\n"
        "        one = ops.bottleneck(d(x), d(w1), d(w2), d(w3), *dbns, fused=1).cpu()\n"
        "        print('CHECK', n, hw, it, int((one != three).sum()))\n")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPK_BNECK_FLAGS=flags), capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("CHECK")]
    assert len(lines) == 6 and all(ln[-1] == "0" for ln in lines), lines
