"""Parity of the HIP inference path (through the C-ABI) with the oracle and
with the golden vectors produced by the reference's own net_pass.

Tolerances (north_star): per-class probabilities within 1e-3 of the fp32 CPU
reference; top-1 identical wherever the reference's top-2 margin exceeds the
probability tolerance (SURVEY.md §7 "hard parts": near-ties below the error
bound are excluded explicitly)."""

import numpy as np
import pytest
import torch

from sykepic_hip import arch, synth

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-3
GOLD = __import__("pathlib").Path(__file__).resolve().parent / "golden" / "net_pass.npz"
ROOT_DIR = __import__("pathlib").Path(__file__).resolve().parent.parent


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


@pytest.mark.parametrize("network,hw", [("resnet18", 180), ("resnet18", 224), ("resnet50", 224)])
def test_probabilities_match_reference_golden(golden_dir, network, hw):
    gold = np.load(golden_dir / "net_pass.npz")
    tag = f"{network}_{hw}"
    g, sd = _state(network, gold, tag)
    net = _hipnet(network, sd)
    n = len(gold[f"{tag}_rois_in"])
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=0)).cuda()
    from sykepic_hip.prob import net_pass
    rois = [int(r) for r in gold[f"{tag}_rois_in"]]
    paths = [f"/x/D20180712T065600_IFCB114_{r:05d}.png" for r in rois]
    half = n // 2
    res = net_pass(net, [(x[:half], paths[:half]), (x[half:], paths[half:])], "cuda:0")
    assert [r for r, _ in res] == gold[f"{tag}_rois_out"].tolist()
    p = np.array([q for _, q in res], dtype=np.float64)
    ref = gold[f"{tag}_probs"].astype(np.float64)
    err = np.abs(p - ref).max()
    print(f"{tag}: max |dp| = {err:.2e}")
    assert err <= PROB_TOL
    top2 = np.sort(ref, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 2 * PROB_TOL
    assert decided.sum() >= n // 2
    assert (p.argmax(1)[decided] == ref.argmax(1)[decided]).all()
    assert np.allclose(p.sum(1), 1.0, atol=1e-5)


@pytest.mark.parametrize("network,hw", [("resnet18", 180), ("resnet50", 224)])
def test_probabilities_match_reference_golden_diverse_top1(golden_dir, network, hw):
    """The reference's net_pass on 8 images with 8 DIFFERENT arg-max classes (net_pass_diverse.npz): the top-1
    assertion cannot pass by always answering the same class; ragged batches of 3 + 5, unsorted sparse ROI ids."""
    from test_oracle_golden import diverse_case
    from sykepic_hip.prob import net_pass
    g, sd, x, paths, rois_out, ref = diverse_case(golden_dir, network, hw)
    assert len(set(ref.argmax(1).tolist())) >= 6
    net = _hipnet(network, sd)
    xc = x.cuda()
    res = net_pass(net, [(xc[:3], paths[:3]), (xc[3:], paths[3:])], "cuda:0")
    assert [r for r, _ in res] == rois_out
    p = np.array([q for _, q in res], dtype=np.float64)
    err = np.abs(p - ref).max()
    print(f"{network}_{hw} (diverse): max |dp| = {err:.2e}, top-1 {p.argmax(1).tolist()}")
    assert err <= PROB_TOL
    top2 = np.sort(ref, axis=1)[:, -2:]
    assert ((top2[:, 1] - top2[:, 0]) > 2 * PROB_TOL).all()      # every image is decided in this fixture
    assert (p.argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("network,hw,n", [("resnet18", 96, 5), ("resnet50", 64, 3), ("resnet34", 64, 2)])
def test_layerwise_vs_oracle(network, hw, n):
    """Every activation of the graph against the torch fp32 interpreter: maximum element error <= 4e-3 of the
    layer's largest value (measured worst case 1.1e-3: fp16 storage 2^-11 per tensor plus accumulation over
    the depth; a wrong halo tap or a dropped K chunk on ONE edge pixel is an O(1) error of that element)."""
    from oracle import graph_eval, refnet
    g = arch.build_graph(network, 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=3)
    net = _hipnet(network, sd)
    x = synth.synth_images(n, 3, hw, hw, seed=21)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    acts = graph_eval.run(g, tsd, torch.from_numpy(x))
    # the interpreter itself agrees with the reference-pinned oracle module
    ref = refnet.load_numpy_state(refnet.RefNet(network, 50), sd)
    z_ref = refnet.probabilities(ref, torch.from_numpy(x), base=None)
    assert torch.allclose(acts[g.ops[-1].dst], z_ref, atol=1e-4, rtol=1e-4)
    z = net.forward(torch.from_numpy(x).cuda()).cpu()
    worst = 0.0
    for op in g.ops:
        want = acts[op.dst]
        got = net.read_activation(op.dst, n, tuple(want.shape))
        scale = float(want.abs().max()) + 1e-6
        err = float((got - want).abs().max()) / scale
        worst = max(worst, err)
        assert err < 4e-3, f"{network} {op.name or op.kind} id {op.dst}: rel err {err:.3e}"
    print(f"{network}@{hw}: worst layer rel err {worst:.2e}")
    assert float((z - z_ref).abs().max()) / (float(z_ref.abs().max()) + 1e-6) < 4e-3


def test_ragged_batches_and_odd_sizes():
    """batch of 1, batch not a multiple of any tile, odd image width."""
    from oracle import refnet
    g = arch.build_graph("resnet18", 7)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=4)
    net = _hipnet("resnet18", sd, classes=7)
    ref = refnet.load_numpy_state(refnet.RefNet("resnet18", 7), sd)
    for n, h, w in [(1, 64, 64), (3, 75, 101), (13, 40, 56)]:
        x = torch.from_numpy(synth.synth_images(n, 3, h, w, seed=n))
        p = net.probabilities(x.cuda()).cpu()
        pr = refnet.probabilities(ref, x)
        assert float((p - pr).abs().max()) <= PROB_TOL, (n, h, w)


def test_state_dict_round_trip():
    g = arch.build_graph("resnet18", 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    sd["base.1.num_batches_tracked"] = np.array(1234567890123, dtype=np.int64)
    net = _hipnet("resnet18", sd)
    out = net.state_dict()
    assert list(out.keys()) == [k for k, _, _ in arch.param_specs(g)]
    for k, v in sd.items():
        assert np.array_equal(out[k].numpy(), np.asarray(v)), k
    with pytest.raises(RuntimeError):
        net.load_state_dict({"nope": torch.zeros(1)})


def test_resnet50_probability_tolerance_over_64_images():
    """The 1e-3 bound is checked on more inputs than the 8 golden images: 64
    fresh synthetic ROIs against the oracle run on this host (default eval
    precision: fp16 storage, hi/lo split weights).  Measured worst case over
    128 images: 3.6e-4; plain fp16 reaches 1.4e-3 and is NOT the default."""
    from oracle import refnet
    torch.set_num_threads(min(16, torch.get_num_threads()))
    gold = np.load(GOLD)
    g, sd = _state("resnet50", gold, "resnet50_224")
    net = _hipnet("resnet50", sd)
    ref = refnet.load_numpy_state(refnet.RefNet("resnet50", 50), sd)
    worst = 0.0
    for seed in (300, 301):
        x = torch.from_numpy(synth.synth_images(32, 3, 224, 224, seed=seed))
        p = net.probabilities(x.cuda()).cpu()
        pr = refnet.probabilities(ref, x)
        worst = max(worst, float((p - pr).abs().max()))
        decided = (pr.topk(2, 1).values[:, 0] - pr.topk(2, 1).values[:, 1]) > 2 * PROB_TOL
        assert (p.argmax(1)[decided] == pr.argmax(1)[decided]).all()
    print(f"resnet50, 64 images: max |dp| = {worst:.2e}")
    assert worst <= PROB_TOL


def test_split_weight_modes_and_per_op_mask():
    """The precision modes of the eval path: the per-op mask API reproduces the
    built-in modes bit for bit, and the default ("all but the 3x3 convs inside a
    residual block") sits between plain fp16 and the every-conv split."""
    from oracle import refnet
    gold = np.load(GOLD)
    g, sd = _state("resnet50", gold, "resnet50_224")
    net = _hipnet("resnet50", sd)
    ops = {op.name: op for op in g.ops if op.kind == arch.OP_CONV}
    convs = [n for _, n in net.conv_ops()]
    assert convs == list(ops) and len(convs) == 53
    trunk = {n for n in convs if n == "base.0" or "downsample" in n or ops[n].res >= 0}
    inner3 = {n for n in convs if ops[n].k == 3 and n not in trunk}
    assert len(inner3) == 16
    x = torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=41)).cuda()
    by_mode = {}
    for mode, keep in ((3, set(convs) - inner3), (1, set(convs)), (2, trunk), (0, set())):
        net.set_precision(split_weights=mode)
        a = net.forward(x)
        net.set_split_ops(keep)
        b = net.forward(x)
        assert torch.equal(a, b), f"mode {mode} != its mask"
        by_mode[mode] = a.cpu().numpy()
    with pytest.raises(KeyError):
        net.set_split_ops(["base.9.conv1"])
    ref = refnet.load_numpy_state(refnet.RefNet("resnet50", 50), sd)
    z = refnet.probabilities(ref, x.cpu(), base=0).numpy()
    rms = {m: float(np.sqrt(np.mean((v - z) ** 2))) for m, v in by_mode.items()}
    print("logit rms by split mode", rms)
    assert rms[1] < rms[3] < rms[2] < rms[0]
    net.set_precision()   # back to the default
    assert torch.equal(net.forward(x), torch.from_numpy(by_mode[3]).cuda())


def test_every_main_loop_flavour_gives_the_same_network_output():
    """Each main-loop flavour of the implicit-GEMM kernel (register-staged, LDS-DMA 3-4 stages, 2 stages,
    hybrid, 32-deep K steps, single stage) forced for every conv of ResNet-50: all accumulate each output
    element in the same K order, so the logits are bit-identical whichever the autotuner picks."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = [%r, %r]\n"
        "from sykepic_hip import arch, synth\n"
        "from sykepic_hip.net import HipNet\n"
        "g = arch.build_graph('resnet50', 50)\n"
        "sd = synth.synth_state_dict(arch.param_specs(g), seed=2)\n"
        "net = HipNet('resnet50', 50, weights=None)\n"
        "net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()\n"
        "x = torch.from_numpy(synth.synth_images(6, 3, 96, 128, seed=5)).cuda()\n"
        "np.save(sys.argv[1], net.forward(x).cpu().numpy())\n"
    ) % (str(ROOT_DIR), str(ROOT_DIR / "syke-pic_amd"))
    outs = {}
    for cfg, dma in ((3, 3), (3, 0), (0, 1), (0, 4), (3, 5), (0, 5), (3, 6), (4, 6), (2, 3)):
        env = dict(os.environ)
        env["SPK_CONV_CFG"], env["SPK_CONV_DMA"] = str(cfg), str(dma)
        path = f"/tmp/flavour_{cfg}_{dma}.npy"
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
        outs[(cfg, dma)] = np.load(path)
    ref = outs[(3, 3)]
    assert np.abs(ref).max() > 0.1
    for key, z in outs.items():
        assert np.array_equal(z, ref), f"cfg {key[0]} flavour {key[1]}: max diff {np.abs(z - ref).max()}"


@pytest.mark.parametrize("shape", [(5, 224, 224), (3, 180, 180), (2, 75, 101), (4, 64, 64), (1, 37, 53)])
def test_stem_kernel_with_fused_maxpool_equals_stem_then_pool(shape):
    """The eval path computes the ResNet stem (7x7/2 conv + BN + ReLU) and the 3x3/2 max-pool behind it in one kernel
    (conv_stem.hip, POOL variant: only the pooled tensor is written).  After a forward the pooled tensor comes from the
    fused kernel; reading the stem output makes the library run the stem layer alone (un-fused kernel).  Max-pooling
    that with torch must give the fused kernel's tensor exactly - same roundings, same values - for tile-aligned,
    odd and non-square sizes (tiles that end mid-image, pooled rows whose window hangs over the border)."""
    import torch.nn.functional as F
    n, h, w = shape
    g = arch.build_graph("resnet18", 5)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=6)
    net = _hipnet("resnet18", sd, classes=5)
    x = torch.from_numpy(synth.synth_images(n, 3, h, w, seed=h))
    net.forward(x.cuda())
    stem = [op for op in g.ops if op.kind == arch.OP_CONV][0]
    pool = [op for op in g.ops if op.kind == arch.OP_MAXPOOL][0]
    assert pool.src == stem.dst
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    pooled = net.read_activation(pool.dst, n, (n, 64, hp, wp))       # written by the fused kernel
    stem_out = net.read_activation(stem.dst, n, (n, 64, ho, wo))     # recomputed by the un-fused kernel
    assert float(stem_out.abs().max()) > 0.1 and float(stem_out.min()) >= 0.0
    want = F.max_pool2d(stem_out, 3, 2, 1)
    assert torch.equal(pooled, want), float((pooled - want).abs().max())


@pytest.mark.parametrize("shape", [(6, 224, 224), (3, 75, 101), (1, 64, 64)])
def test_block_closing_conv_with_fused_shortcut_conv(golden_dir, shape, monkeypatch):
    """Round 3: in the first block of every ResNet-50 stage the block-closing 1x1 conv and the 1x1 shortcut (downsample)
    conv run as ONE K-concatenated GEMM (conv_pw.hip, PwConvArgs::x2): the shortcut tensor is never written.  Against a
    handle built with SPK_FUSE_DS=0 (two kernels, shortcut rounded to fp16 in between): probabilities within 6e-4 (two
    valid roundings of the same function on a net that amplifies fp16 rounding ~40x: 3.7e-4 measured; each of them is
    within 1e-3 of the reference, test_probabilities_match_reference_golden*), every block output within 2e-3 relative L2, and a
    fused-away shortcut tensor read back through read_activation is recomputed by the stand-alone kernel - bit-equal to
    the unfused handle's where both see the same input (the first stage)."""
    from oracle import graph_eval
    n, h, w = shape
    gold = np.load(golden_dir / "net_pass.npz")
    g, sd = _state("resnet50", gold, "resnet50_224")
    x = torch.from_numpy(synth.synth_images(n, 3, h, w, seed=31))
    fused = _hipnet("resnet50", sd)
    monkeypatch.setenv("SPK_FUSE_DS", "0")
    plain = _hipnet("resnet50", sd)
    monkeypatch.delenv("SPK_FUSE_DS")
    pf, pp = fused.probabilities(x.cuda()).cpu(), plain.probabilities(x.cuda()).cpu()
    assert float((pf - pp).abs().max()) < 6e-4
    assert not torch.equal(pf, pp)            # (the fusion is active: identical bits would mean it never ran)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    shapes = {t: tuple(v.shape) for t, v in graph_eval.run(g, tsd, x).items()}
    closers = [op for op in g.ops if op.kind == arch.OP_CONV and op.res >= 0 and
               any(d.dst == op.res and "downsample" in d.name for d in g.ops)]
    assert len(closers) == 4
    for op in closers:
        a, b = fused.read_activation(op.dst, n, shapes[op.dst]), plain.read_activation(op.dst, n, shapes[op.dst])
        rel = float((a - b).norm() / b.norm())
        assert rel < 2e-3, (op.name, rel)
        ra, rb = fused.read_activation(op.res, n, shapes[op.res]), plain.read_activation(op.res, n, shapes[op.res])
        if op is closers[0]:
            assert torch.equal(ra, rb)      # same input (the max-pool output), same stand-alone kernel
        else:
            assert float((ra - rb).norm() / rb.norm()) < 2e-3    # (its input already differs by the stage before)


@pytest.mark.parametrize("network,n,hw", [("resnet50", 96, 64), ("resnet18", 65, 75)])
def test_two_stream_forward_equals_the_halves_run_alone(golden_dir, network, n, hw):
    """Round 3: a batch of >= 64 images runs as two halves on two streams that share the activation tensors (images
    [0, n/2) and [n/2, n)).  Per-image results must not depend on that: the output equals, bit for bit, what the same
    images give in batches small enough to take the single-stream path - also for an odd batch, repeated calls (the first
    call of a shape runs both halves on one stream while the kernel tuners time their candidates), and for the tensors
    read back afterwards."""
    gold = np.load(golden_dir / "net_pass.npz")
    g, sd = _state(network, gold, f"{network}_224")
    net = _hipnet(network, sd)
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=77)).cuda()
    first = net.probabilities(x).cpu()          # tuning pass: both halves on the caller's stream
    second = net.probabilities(x).cpu()         # two streams
    third = net.probabilities(x).cpu()
    assert torch.equal(first, second) and torch.equal(second, third)
    small = torch.cat([net.probabilities(x[i:i + 32]).cpu() for i in range(0, n, 32)])   # < 64 images: one stream
    assert torch.equal(second, small)
    assert torch.isfinite(second).all() and torch.allclose(second.sum(1), torch.ones(n), atol=1e-4)
