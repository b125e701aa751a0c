"""The calibrated single-pass eval mode (split_weights = 5, csrc/zero_sum.hip): every conv as ONE fp16 product whose
weights were zero-sum rounded against per-channel activation means.

Parity bar as for every eval mode (north_star): probabilities within 1e-3 of the fp32 CPU reference, top-1 identical
wherever the reference's margin exceeds the tolerance - on the goldens the reference's own net_pass produced, on the
class-diverse fixture, on fresh images, and on images from ANOTHER distribution than the calibration batch."""

import numpy as np
import pytest
import torch

from sykepic_hip import arch, synth

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-3


def _state(network, golden, tag, classes=50):
    g = arch.build_graph(network, classes)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    last = [k for k in sd if k.startswith("head.") and k.endswith(".bias")][-1]
    sd[last] = sd[last] + golden[f"{tag}_bias_adj"]
    return g, sd


def _hipnet(network, sd, classes=50):
    from sykepic_hip.net import HipNet
    net = HipNet(network, classes, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return net.eval()


def _calibrated(net, hw, n=32, seed=9000):
    """calibration images never overlap the test images (other generator seeds)"""
    net.calibrate(torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=seed)).cuda())
    return net.set_precision("calibrated")


# ---- the rounding kernel alone -------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,row_len,period", [(7, 64, 64), (5, 576, 64), (3, 2304, 256), (4, 147, 147), (2, 3072, 3072),
                                                 (3, 40, 8)])
def test_zero_sum_kernel_equals_its_restatement(rows, row_len, period):
    from oracle import zero_sum as oz
    from sykepic_hip import ops
    rng = np.random.default_rng(rows * 1000 + row_len)
    w = (rng.uniform(-1, 1, (rows, row_len)) * np.sqrt(6.0 / row_len)).astype(np.float32)
    w[0, :3] = [0.0, 0.25, -0.5]                      # exactly representable values have nothing to re-round
    mu = rng.uniform(0.05, 1.2, period).astype(np.float32)
    for m in (mu, None):
        got = ops.zero_sum_round(torch.from_numpy(w).cuda(), None if m is None else torch.from_numpy(m).cuda(),
                                 period).cpu().numpy()
        ref = oz.zero_sum_round(w, m, period)
        assert np.array_equal(got, ref)
        # properties: fp16 numbers, each the nearest or the other neighbour of w, weighted sum driven towards zero
        assert np.array_equal(got.astype(np.float16).astype(np.float32), got)
        q = w.astype(np.float16).astype(np.float32)
        ulp = np.abs(np.nextafter(q.astype(np.float16), np.float16(np.inf)).astype(np.float32) - q)
        assert (np.abs(got - w) <= 1.0001 * np.maximum(ulp, np.abs(q - np.nextafter(q.astype(np.float16), np.float16(-np.inf)).astype(np.float32)))).all()
        s0, s1 = np.abs(oz.weighted_sum(w, q, m, period)), np.abs(oz.weighted_sum(w, got, m, period))
        assert (s1 <= s0 + 1e-12).all()
        if row_len >= 64:
            assert np.sqrt((s1 ** 2).mean()) < 0.05 * np.sqrt((s0 ** 2).mean())
            e0, e1 = ((q - w) ** 2).sum(), ((got - w) ** 2).sum()
            assert e1 <= 1.05 * e0                     # the squared rounding error hardly grows


# ---- network level ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("network,hw", [("resnet18", 180), ("resnet18", 224), ("resnet50", 224)])
def test_calibrated_mode_matches_reference_golden(golden_dir, network, hw):
    gold = np.load(golden_dir / "net_pass.npz")
    tag = f"{network}_{hw}"
    g, sd = _state(network, gold, tag)
    net = _calibrated(_hipnet(network, sd), hw)
    n = len(gold[f"{tag}_rois_in"])
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=0)).cuda()
    from sykepic_hip.prob import net_pass
    paths = [f"/x/D20180712T065600_IFCB114_{int(r):05d}.png" for r in gold[f"{tag}_rois_in"]]
    half = n // 2
    res = net_pass(net, [(x[:half], paths[:half]), (x[half:], paths[half:])], "cuda:0")
    assert [r for r, _ in res] == gold[f"{tag}_rois_out"].tolist()
    p = np.array([q for _, q in res], dtype=np.float64)
    ref = gold[f"{tag}_probs"].astype(np.float64)
    err = np.abs(p - ref).max()
    print(f"{tag} calibrated: max |dp| = {err:.2e}")
    assert err <= PROB_TOL
    top2 = np.sort(ref, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 2 * PROB_TOL
    assert (p.argmax(1)[decided] == ref.argmax(1)[decided]).all()


@pytest.mark.parametrize("network,hw", [("resnet18", 180), ("resnet50", 224)])
def test_calibrated_mode_on_the_class_diverse_fixture(golden_dir, network, hw):
    """8 images with 8 DIFFERENT arg-max classes (tests/golden/net_pass_diverse.npz, made by the reference's net_pass),
    through `prob.net_pass` in ragged batches of 3 + 5."""
    from test_oracle_golden import diverse_case
    from sykepic_hip.prob import net_pass
    g, sd, x, paths, rois_out, ref = diverse_case(golden_dir, network, hw)
    net = _calibrated(_hipnet(network, sd), hw)
    xc = x.cuda()
    res = net_pass(net, [(xc[:3], paths[:3]), (xc[3:], paths[3:])], "cuda:0")
    assert [r for r, _ in res] == rois_out
    p = np.array([q for _, q in res], dtype=np.float64)
    err = np.abs(p - ref).max()
    print(f"{network}_{hw} (diverse), calibrated: max |dp| = {err:.2e}")
    assert err <= PROB_TOL
    assert (p.argmax(1) == ref.argmax(1)).all()


def _shifted(x, kind):
    if kind == "same":
        return x
    if kind == "inverted":
        return (1.0 - x).contiguous()
    n, _, h, w = x.shape           # "ifcb": light grey background, one dark blob, three identical channels
    g = 0.8 + 0.1 * (x[:, :1] - 0.5)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    for i in range(n):
        cy, cx, r = h // 4 + (37 * i) % (h // 2), w // 4 + (53 * i) % (w // 2), 6 + (i % 5) * 4
        g[i, 0][(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.35
    return (torch.round(g.expand(n, 3, h, w) * 255) / 255).contiguous()


@pytest.mark.parametrize("network,hw,n", [("resnet50", 224, 64), ("resnet18", 160, 64)])
def test_calibrated_mode_on_fresh_and_shifted_images(golden_dir, network, hw, n):
    """The means come from 32 synthetic images; the test images are others of that kind, their negatives, and
    IFCB-like frames (near-constant background): the mode must hold the tolerance on all of them, and be as close to
    the fp32 oracle as the most accurate mode (every conv hi + lo) is - the rest is fp16 activation rounding."""
    from oracle import refnet
    gold = np.load(golden_dir / "net_pass.npz")
    tag = f"{network}_{hw}" if f"{network}_{hw}_bias_adj" in gold.files else f"{network}_224"
    g, sd = _state(network, gold, tag)
    ref_net = refnet.load_numpy_state(refnet.RefNet(network, 50), sd)
    net = _calibrated(_hipnet(network, sd), hw)
    precise = _hipnet(network, sd).set_precision(split_weights=1)
    for kind in ("same", "inverted", "ifcb"):
        x = _shifted(torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=4242)), kind)
        lg = refnet.probabilities(ref_net, x, base=0).numpy()
        pr = torch.softmax(torch.from_numpy(lg) * float(np.log(1.3)), 1).numpy()
        p = net.probabilities(x.cuda()).cpu().numpy()
        z = net.forward(x.cuda()).cpu().numpy()
        zp = precise.forward(x.cuda()).cpu().numpy()
        err = np.abs(p - pr).max()
        rms, rms_p = np.sqrt(np.mean((z - lg) ** 2)), np.sqrt(np.mean((zp - lg) ** 2))
        print(f"{network} {kind}: calibrated max |dp| {err:.2e}, logit rms {rms:.2e} (every conv split: {rms_p:.2e})")
        assert err <= PROB_TOL
        assert rms <= 1.25 * rms_p + 2e-4
        top2 = np.sort(pr, axis=1)[:, -2:]
        decided = (top2[:, 1] - top2[:, 0]) > 2 * PROB_TOL
        assert (p.argmax(1)[decided] == pr.argmax(1)[decided]).all()


def test_means_round_trip_and_accumulate(golden_dir):
    gold = np.load(golden_dir / "net_pass.npz")
    g, sd = _state("resnet18", gold, "resnet18_180")
    a = _hipnet("resnet18", sd)
    with pytest.raises(RuntimeError, match="activation means"):
        a.set_precision("calibrated").probabilities(torch.zeros(2, 3, 96, 96).cuda())
    xc = torch.from_numpy(synth.synth_images(24, 3, 96, 96, seed=9000)).cuda()
    a.calibrate(xc)
    means = a.act_means()
    assert means.numel() == sum(op.cin for op in g.ops if op.kind == arch.OP_CONV)
    assert torch.isfinite(means).all() and (means >= 0).all() and means.max() > 0     # post-ReLU inputs, pixel values
    # the stem's three values are the image's channel means
    assert torch.allclose(means[:3], xc.float().mean((0, 2, 3)).cpu(), rtol=2e-3)
    # two calls that accumulate == one call over both halves (up to fp32 summation order)
    b = _hipnet("resnet18", sd)
    b.calibrate(xc[:8]).calibrate(xc[8:], reset=False)
    assert torch.allclose(b.act_means(), means, rtol=2e-4, atol=1e-6)
    # stored means restore the same model: bit-identical probabilities, whatever batch an image arrives in
    c = _hipnet("resnet18", sd).set_act_means(means).set_precision("calibrated")
    x = torch.from_numpy(synth.synth_images(9, 3, 96, 96, seed=77)).cuda()
    pa, pc = a.probabilities(x), c.probabilities(x)
    assert torch.equal(pa, pc)
    assert torch.equal(a.probabilities(x[3:5]), pa[3:5])
    with pytest.raises(RuntimeError):
        c.set_act_means(means[:-1])
    # forgetting the means makes the mode refuse again; another mode still works
    c.set_act_means(None)
    with pytest.raises(RuntimeError, match="activation means"):
        c.probabilities(x)
    c.set_precision(split_weights=3).probabilities(x)


def test_calibrated_mode_with_the_fused_shortcut_conv(golden_dir):
    """The K-concatenated (block-closing + shortcut) conv rounds scale-folded rows: balanced as one row against both
    sources' means.  Same tolerance with the fusion on (default) and off; the two differ only by rounding realisation."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path[:0] = [{str(root)!r}, {str(root / 'syke-pic_amd')!r}]\n"
        "from oracle import refnet\n"
        "from sykepic_hip import arch, synth\n"
        "from sykepic_hip.net import HipNet\n"
        "g = arch.build_graph('resnet50', 50)\n"
        "sd = synth.synth_state_dict(arch.param_specs(g), seed=2)\n"
        "net = HipNet('resnet50', 50, weights=None)\n"
        "net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()\n"
        "net.calibrate(torch.from_numpy(synth.synth_images(32, 3, 128, 128, seed=9000)).cuda()).set_precision('calibrated')\n"
        "x = torch.from_numpy(synth.synth_images(32, 3, 128, 128, seed=5))\n"
        "ref = refnet.probabilities(refnet.load_numpy_state(refnet.RefNet('resnet50', 50), sd), x).numpy()\n"
        "p = net.probabilities(x.cuda()).cpu().numpy()\n"
        "print('ERR', float(np.abs(p - ref).max()))\n")
    errs = []
    for fuse in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPK_FUSE_DS=fuse), capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        errs.append(float([ln for ln in out.stdout.splitlines() if ln.startswith("ERR")][0].split()[1]))
    print(f"fused shortcut conv on / off: {errs[0]:.2e} / {errs[1]:.2e}")
    assert max(errs) <= PROB_TOL


def test_chained_convs_do_not_change_a_bit():
    """Single-weight modes run a bottleneck's block-closing conv and the next block's first conv as one launch where that
    is faster (csrc/conv_pw.hip, PwConvArgs::wpz).  Forced on (SPK_CHAIN=2) against off (SPK_CHAIN=0): identical logits,
    in the calibrated and in the plain fp16 mode, at a batch that takes the two-stream path and at a ragged one."""
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
        "net.calibrate(torch.from_numpy(synth.synth_images(16, 3, 96, 96, seed=9000)).cuda())\n"
        "for mode in ('calibrated', 0):\n"
        "    net.set_precision(mode)\n"
        "    for n in (70, 5):\n"
        "        x = torch.from_numpy(synth.synth_images(n, 3, 96, 96, seed=5)).cuda()\n"
        "        z = net.forward(x).cpu().numpy()\n"
        "        z2 = net.forward(x).cpu().numpy()\n"
        "        assert np.array_equal(z, z2)\n"
        "        print('SHA', mode, n, hashlib.sha256(z.tobytes()).hexdigest())\n")
    outs = []
    for chain in ("2", "0"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPK_CHAIN=chain, SPK_TUNE_CACHE="off"),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append([ln for ln in out.stdout.splitlines() if ln.startswith("SHA")])
    assert len(outs[0]) == 4 and outs[0] == outs[1], (outs[0], outs[1])


def test_chained_convs_decide_per_half_batch(tmp_path):
    """the two-stream forward runs every layer for half 0 and then for half 1, and the chained-conv decision
    (tuner cache keyed with the half's image count) may differ between the halves of an odd batch: n = 129 splits into 64
    and 65 images.  With a tuner cache that says "two kernels" for 64 images and "chained" for 65, half 1's decision must
    not make half 0 skip the conv its own launch did not compute (the flag is per half since round 5): the logits equal
    the never-chained ones bit for bit."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    cache = tmp_path / "tune.txt"
    lines = []
    for cin2, coutz in ((64, 64), (0, 64), (0, 128)):       # the three stage-1 pairs of ResNet-50 at 96 x 96 input (24 x 24)
        lines += [f"chain 24 24 64 {cin2} {coutz} 64 0", f"chain 24 24 64 {cin2} {coutz} 65 1"]
    code = (
        "import sys, hashlib, numpy as np, torch\n"
        f"sys.path[:0] = [{str(root)!r}, {str(root / 'syke-pic_amd')!r}]\n"
        "from sykepic_hip import arch, synth\n"
        "from sykepic_hip.net import HipNet\n"
        "g = arch.build_graph('resnet50', 50)\n"
        "sd = synth.synth_state_dict(arch.param_specs(g), seed=2)\n"
        "net = HipNet('resnet50', 50, weights=None)\n"
        "net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()\n"
        "net.calibrate(torch.from_numpy(synth.synth_images(16, 3, 96, 96, seed=9000)).cuda())\n"
        "net.set_precision('calibrated')\n"
        "x = torch.from_numpy(synth.synth_images(129, 3, 96, 96, seed=6)).cuda()\n"
        "for it in range(3):\n"
        "    z = net.forward(x).cpu().numpy()\n"
        "    print('SHA', it, hashlib.sha256(z.tobytes()).hexdigest())\n")
    outs = []
    for chain in ("1", "0"):
        cache.write_text("\n".join(lines) + "\n")
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPK_CHAIN=chain, SPK_TUNE_CACHE=str(cache)),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append([ln.split()[2] for ln in out.stdout.splitlines() if ln.startswith("SHA")])
    assert len(outs[0]) == 3 and len(set(outs[0] + outs[1])) == 1, (outs[0], outs[1])
