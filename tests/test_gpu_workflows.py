"""End-to-end workflows through the drop-in entry points on the GPU:
`sykepic prob` on the reference's own raw fixture (counterpart of the
reference's tests/test_probability.py:10-37, plus numerical parity with the
oracle) and `sykepic train` on a tiny synthetic dataset."""

import shutil
from collections import namedtuple
from configparser import ConfigParser

import numpy as np
import pytest
import torch
from PIL import Image

from sykepic_hip import arch, synth

pytestmark = pytest.mark.gpu

Args = namedtuple("Args", "raw samples image_dir images model out batch_size num_workers force")


def _model_dir(tmp_path, golden_dir):
    d = tmp_path / "model"
    d.mkdir()
    shutil.copy(golden_dir / "ref_data" / "class_names.txt", d / "class_names.txt")
    shutil.copy(golden_dir / "ref_data" / "config.ini", d / "config.ini")   # legacy file: no `weights` key (Q7)
    g = arch.build_graph("resnet18", 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, d / "best_state.pth")
    return d, sd


def test_prob_call_on_reference_fixture(tmp_path, golden_dir):
    from sykepic_hip import prob
    raw = tmp_path / "raw" / "valid"
    raw.mkdir(parents=True)
    for ext in ("adc", "hdr", "roi"):
        shutil.copy(golden_dir / "ref_data" / f"D20180712T065600_IFCB114.{ext}", raw)
    model, sd = _model_dir(tmp_path, golden_dir)
    out_dir = tmp_path / "out"
    prob.call(Args(raw=str(raw), samples=None, image_dir=None, images=None, model=str(model), out=out_dir,
                   batch_size=64, num_workers=2, force=False))
    # --- the reference test's assertions, verbatim in meaning ---
    out_csvs = list(out_dir.glob("**/*.csv"))
    assert len(out_csvs) == 1
    assert out_csvs[0] == out_dir / "2018" / "07" / "12" / "D20180712T065600_IFCB114.prob.csv"
    lines = out_csvs[0].read_text().splitlines(keepends=True)
    assert len(lines) == 3
    header = lines[0].split(",")
    assert len(header) == 51 and header[0] == "roi"
    roi_2 = list(filter(None, lines[1].split(",")))
    roi_3 = list(filter(None, lines[2].split(",")))
    assert len(roi_2) == len(header) and len(roi_3) == len(header)
    assert int(roi_2[0]) == 2 and int(roi_3[0]) == 3
    # --- no PNGs left beside the raw data (quirk Q8) ---
    assert sorted(p.name for p in raw.iterdir()) == sorted(f"D20180712T065600_IFCB114.{e}" for e in ("adc", "hdr", "roi"))
    # --- numbers: same preprocessing, oracle forward ---
    from oracle import refnet
    from sykepic_hip import ifcb
    from sykepic_hip.config import get_img_shape, get_transforms
    cfg = ConfigParser()
    cfg.read(model / "config.ini")
    _, ev = get_transforms(cfg, get_img_shape(cfg))
    rois = ifcb.read_rois(raw / "D20180712T065600_IFCB114.adc", raw / "D20180712T065600_IFCB114.roi")
    x = torch.stack([ev(np.repeat(img[:, :, None], 3, axis=2)) for _, img in rois])
    ref = refnet.probabilities(refnet.load_numpy_state(refnet.RefNet("resnet18", 50), sd), x).numpy()
    got = np.array([[float(v) for v in ln.split(",")[1:]] for ln in lines[1:]])
    assert np.abs(got - ref).max() <= 1e-3 + 5e-6          # 5 printed decimals
    # second run without --force keeps the file; with --force rewrites it
    stamp = out_csvs[0].stat().st_mtime_ns
    prob.call(Args(str(raw), None, None, None, str(model), out_dir, 64, 2, False))
    assert out_csvs[0].stat().st_mtime_ns == stamp


def _synthetic_sample(raw_dir, name, n_roi, seed):
    """An IFCB sample (.adc + .roi) of n_roi random-sized ROIs with a blob on a light background, as utils/ifcb.py reads
    them (columns 16-18 of the .adc: width, height, byte offset)."""
    rng = np.random.RandomState(seed)
    adc, blobs, off = [], [], 0
    for _ in range(n_roi):
        h, w = int(rng.randint(24, 90)), int(rng.randint(30, 160))
        img = np.clip(rng.normal(200, 6, (h, w)), 0, 255)
        cy, cx, r = rng.randint(4, h - 4), rng.randint(4, w - 4), rng.randint(3, 12)
        yy, xx = np.ogrid[:h, :w]
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = rng.randint(30, 120)
        cols = ["0"] * 24
        cols[15], cols[16], cols[17] = str(w), str(h), str(off)
        adc.append(",".join(cols))
        blobs.append(img.astype(np.uint8).reshape(-1))
        off += h * w
    raw_dir.mkdir(parents=True, exist_ok=True)
    (raw_dir / f"{name}.adc").write_text("\n".join(adc) + "\n")
    np.concatenate(blobs).tofile(raw_dir / f"{name}.roi")


def test_prob_calibrates_a_reference_trained_directory_on_its_first_sample(tmp_path, golden_dir, caplog):
    """a model directory as the REFERENCE leaves it (best_state.pth, config.ini, class_names.txt:
    probability.py:118-130) has no act_means.pth.  `prob` measures the means on the first <= 256 ROIs it classifies,
    switches to the calibrated single-pass mode (the benchmarked mode) and writes the sidecar; both the first sample - the
    one calibrated on - and a later one are within 1e-3 of the oracle; a second process starts calibrated from the
    sidecar and reproduces the CSVs byte for byte; a sidecar of OTHER weights is ignored."""
    import logging
    from oracle import refnet
    from sykepic_hip import ifcb, prob
    from sykepic_hip.config import get_img_shape, get_transforms
    raw = tmp_path / "raw"
    names = ["D20200101T000000_IFCB114", "D20200101T000100_IFCB114"]
    _synthetic_sample(raw, names[0], 300, seed=1)
    _synthetic_sample(raw, names[1], 90, seed=2)
    model, sd = _model_dir(tmp_path, golden_dir)
    assert not (model / prob.ACT_MEANS_FILE).exists()
    out = tmp_path / "out"
    with caplog.at_level(logging.INFO, logger="prob"):
        prob.call(Args(str(raw), None, None, None, str(model), out, 64, 2, False))
    assert any("calibrated single-pass mode from here on" in r.message for r in caplog.records)
    side = torch.load(model / prob.ACT_MEANS_FILE)
    # (whichever sample `list_sample_paths` hands over first: its first <= 256 ROIs)
    assert side["images"] in (256, 90) and side["state_sha256"] == prob.state_digest(model)
    cfg = ConfigParser()
    cfg.read(model / "config.ini")
    _, ev = get_transforms(cfg, get_img_shape(cfg))
    ref_net = refnet.load_numpy_state(refnet.RefNet("resnet18", 50), sd)
    texts = {}
    for name in names:
        csv = out / "2020" / "01" / "01" / f"{name}.prob.csv"
        texts[name] = csv.read_bytes()
        rois = ifcb.read_rois(raw / f"{name}.adc", raw / f"{name}.roi")
        x = torch.stack([ev(np.repeat(img[:, :, None], 3, axis=2)) for _, img in rois])
        ref = refnet.probabilities(ref_net, x).numpy()
        got = np.array([[float(v) for v in ln.split(",")[1:]] for ln in csv.read_text().splitlines()[1:]])
        err = float(np.abs(got - ref).max())
        print(f"{name}: {len(rois)} ROIs, calibrated on the first sample, max |dp| = {err:.2e}")
        assert err <= 1e-3 + 5e-6
        assert (got.argmax(1) == ref.argmax(1))[np.sort(ref, 1)[:, -1] - np.sort(ref, 1)[:, -2] > 2e-3].all()
    # second run: starts calibrated from the sidecar, same bytes
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="prob"):
        prob.call(Args(str(raw), None, None, None, str(model), out, 64, 2, True))
    assert any("activation means from" in r.message for r in caplog.records)
    for name in names:
        assert (out / "2020" / "01" / "01" / f"{name}.prob.csv").read_bytes() == texts[name]
    # other weights in the directory: the sidecar no longer belongs to them
    g = arch.build_graph("resnet18", 50)
    sd2 = synth.synth_state_dict(arch.param_specs(g), seed=3)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd2.items()}, model / "best_state.pth")
    net, *_ = prob.prepare_model(model)
    assert getattr(net, "_auto_calibration_dir", None) is not None     # ignored: armed to measure again


def test_prob_from_png_images(tmp_path, golden_dir):
    from sykepic_hip import ifcb, prob
    model, _ = _model_dir(tmp_path, golden_dir)
    img_dir = tmp_path / "imgs"
    ifcb.raw_to_png(golden_dir / "ref_data" / "D20180712T065600_IFCB114.adc",
                    golden_dir / "ref_data" / "D20180712T065600_IFCB114.roi", out_dir=img_dir, force=True)
    assert sorted(p.name for p in img_dir.iterdir()) == ["D20180712T065600_IFCB114_00002.png",
                                                        "D20180712T065600_IFCB114_00003.png"]
    out = tmp_path / "o"
    prob.call(Args(None, None, str(img_dir), None, str(model), out, 64, 0, False))
    text = (out / "D20180712T065600_IFCB114.prob.csv").read_text().splitlines()
    assert len(text) == 3 and text[1].startswith("2,") and text[2].startswith("3,")


INI = """[dataset]
path = {ds}
split = 0.6, 0.2, 0.2
external_test =
min_N =
max_N =
exclude =
random_seed = 42
oversample_until = 12
oversample_with_decay =
[model]
path = {models}
network = {network}
weights =
id = auto
exist_ok = no
head = 32, 16
dropout =
[image]
shape = 3, 64, 64
augmentations = flip, translate, zoom, brightness
imagenet_normalization = no
border = mode
zoom_range = 0.8, 1.2
brightness_range = 0.95, 1.1
max_rotation = 10
batch_size = 16
num_workers = 0
[train]
gpu = yes
max_epochs = 8
early_stop_patience = 12
learning_rate = 0.01
optimizer = Adam
[lr_warmup]
use = yes
factor_1 = 0.1
factor_2 = 0.5
step_1 = 3
step_2 = 5
step_3 = 7
verbose = no
[lr_reduction]
use = yes
factor = 0.1
patience = 4
verbose = yes
"""


@pytest.mark.parametrize("network", ["resnet18", "efficientnet_b0"])
def test_train_main_end_to_end(tmp_path, capsys, network):
    """8 epochs on 3 synthetic classes (every phase of the unfreeze schedule): artefacts, checkpoint
    interchangeable with the torch module of the reference, loss goes down.  `network = efficientnet_b0` takes the
    MBConv training path (depthwise / squeeze-excitation / SiLU backward, stochastic depth)."""
    import random
    from oracle import refnet
    from sykepic_hip import train
    random.seed(1234)          # the split / oversampling / augmentations draw from the global generators
    np.random.seed(1234)
    torch.manual_seed(1234)
    rng = np.random.RandomState(0)
    ds = tmp_path / "ds"
    for ci, name in enumerate(("blob", "bars", "flat")):
        (ds / name).mkdir(parents=True)
        for i in range(20):
            h, w = rng.randint(30, 70), rng.randint(30, 90)
            img = np.full((h, w), 180, np.uint8)
            if ci == 0:
                img[h // 4: h // 2, w // 4: w // 2] = 40
            elif ci == 1:
                img[:, ::6] = 60
            img = np.clip(img.astype(np.int32) + rng.randint(-10, 10, (h, w)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(ds / name / f"{name}_{i:02d}.png")
    ini = tmp_path / "train.ini"
    ini.write_text(INI.format(ds=ds, models=tmp_path / "models", network=network))
    train.main(namedtuple("A", "config collage dist save_images")(str(ini), None, None, None))
    out = capsys.readouterr().out
    assert "[ERROR]" not in out, out
    mdir = tmp_path / "models" / f"{network}_1"
    for f in ("config.ini", "class_names.txt", "class_distribution.csv", "best_state.pth", "test_report.txt"):
        assert (mdir / f).is_file(), f
    assert (mdir / "class_names.txt").read_text().split("\n") == ["bars", "blob", "flat"]
    stats = [ln for ln in out.splitlines() if ln.startswith("[STAT] Train")]
    assert len(stats) == 8
    losses = [float(s.split("Train Loss: ")[1]) for s in stats]
    # resnet18 improves on the head-only first epoch; the randomly initialised EfficientNet (stochastic depth, 36 training
    # images) can spike when the base unfreezes (0.87 -> 2.65 in one run) and must then come back down
    assert all(np.isfinite(losses)), losses
    assert min(losses[1:]) < losses[0] - 0.005 or (network != "resnet18" and losses[-1] < max(losses) - 0.5), losses
    # the checkpoint is a plain state_dict the reference's torch module accepts
    sd = torch.load(mdir / "best_state.pth")
    ref = refnet.RefNet(network, 3, head=(32, 16))
    ref.load_state_dict(sd)
    assert int(sd["base.1.num_batches_tracked" if network == "resnet18" else "base.0.0.1.num_batches_tracked"]) > 0
    assert "accuracy" in (mdir / "test_report.txt").read_text()
    # the run leaves activation means of the validation images beside the checkpoint; `prob.prepare_model` picks them up
    # (calibrated single-pass mode) and stays within the tolerance of the fp32 torch module on the trained weights
    from sykepic_hip import prob
    assert (mdir / prob.ACT_MEANS_FILE).is_file() and "Activation means" in out
    net, classes, img_shape, ev, dev = prob.prepare_model(mdir)
    assert classes == ["bars", "blob", "flat"]
    imgs = sorted(ds.rglob("*.png"))[::3]
    x = torch.stack([ev(np.repeat(np.array(Image.open(p))[:, :, None], 3, axis=2)) for p in imgs])
    ref.eval()
    pr = refnet.probabilities(ref, x).numpy()
    p_cal = net.probabilities(x.cuda()).cpu().numpy()
    net.set_precision(split_weights=1)
    p_two = net.probabilities(x.cuda()).cpu().numpy()
    e_cal, e_two = np.abs(p_cal - pr).max(), np.abs(p_two - pr).max()
    print(f"{network} trained by train.main: calibrated max |dp| {e_cal:.2e}, every conv hi+lo {e_two:.2e}")
    assert e_cal <= (1e-3 if network == "resnet18" else 4e-3)
    assert (p_cal.argmax(1) == pr.argmax(1))[np.sort(pr, 1)[:, -1] - np.sort(pr, 1)[:, -2] > 2e-3].all()
    # `sykepic calibrate` re-measures the means of an existing directory on images selected as for `prob`
    from sykepic_hip.__main__ import build_parser
    args = build_parser().parse_args(["calibrate", "-m", str(mdir), "--image-dir", str(ds), "-n", "24", "-b", "8"])
    assert args.func(args) == 24
    stored = torch.load(mdir / prob.ACT_MEANS_FILE)
    assert stored["images"] == 24 and stored["network"] == network
    net2, *_ = prob.prepare_model(mdir)
    assert np.abs(net2.probabilities(x.cuda()).cpu().numpy() - pr).max() <= (1e-3 if network == "resnet18" else 4e-3)
