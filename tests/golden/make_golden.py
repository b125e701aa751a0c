#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's
own code (imported from /root/reference, read-only) in this container.

The reference needs two third-party packages that are not installed here
(``torchvision``, ``cv2``); they are satisfied by a ``sys.modules`` shim:
``torchvision.models.<name>`` returns the torch.nn restatement in
``oracle/backbones.py`` and ``cv2`` only has to exist at import time.  All
arithmetic below the shim is the reference's: ``TorchVisionNet.__init__/
forward``, ``freeze``, ``LRWarmup``, ``probability.net_pass``,
``train.train_net`` and ``prediction.row_prediction`` run unmodified.

Run:  python tests/golden/make_golden.py         (needs /root/reference)
Only data (inputs are generator-seeded, outputs are arrays) is written.
"""

import contextlib
import io
import json
import shutil
import sys
import types
from configparser import ConfigParser
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "syke-pic_amd"))
sys.path.insert(0, str(REF))

from oracle import backbones  # noqa: E402
from sykepic_hip import arch, synth  # noqa: E402


def install_shims():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.BORDER_CONSTANT = 0
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    for n in backbones.names():
        setattr(models, n, (lambda n: lambda weights=None, **kw: backbones.make(n))(n))
    transforms = types.ModuleType("torchvision.transforms")

    class ToTensor:
        def __call__(self, a):
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float() / 255.0

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(-1, 1, 1)
            self.std = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    transforms.ToTensor = ToTensor
    transforms.Normalize = Normalize
    tv.models = models
    tv.transforms = transforms
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.transforms"] = transforms


def ref_config(network, shape):
    cfg = ConfigParser()
    cfg.read_dict({
        "model": {"network": network, "weights": "", "head": "256, 128", "dropout": ""},
        "image": {"shape": ",".join(str(s) for s in shape)},
    })
    return cfg


def build_ref_net(network, num_classes, seed, logit_gain=60.0):
    from sykepic.train.config import get_network
    net = get_network(ref_config(network, (3, 224, 224)), num_classes)
    g = arch.build_graph(network, num_classes)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=seed, logit_gain=logit_gain)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net


def golden_net_pass(effnet=False):
    """effnet=True: the EfficientNet cases (SURVEY.md section 4 golden (1)), written to their own file so
    that the ResNet fixture stays byte-identical."""
    from sykepic.compute.probability import net_pass
    out = {}
    cases = [("resnet18", 180, 8), ("resnet18", 224, 8), ("resnet50", 224, 8)]
    if effnet:
        cases = [("efficientnet_b0", 224, 8), ("efficientnet_b4", 224, 8)]
    for network, hw, n in cases:
        net = build_ref_net(network, 50, seed=2)
        if effnet:
            # BatchNorm running statistics := statistics of the calibration batch (as a trained net's match
            # its data): keeps the activations of the 32-block random-weight net O(1) at every depth.  The
            # tests repeat this with oracle.refnet.calibrate_bn on the same generator-seeded batch.
            from oracle import refnet
            refnet.calibrate_bn(net, torch.from_numpy(synth.synth_images(16, 3, hw, hw, seed=99)))
        # centre the logits on a calibration batch so that the arg-max varies
        # from image to image (a random-weight net otherwise always favours
        # one class); the shift is stored and re-applied by the tests.
        with torch.no_grad():
            net.eval()
            xc = torch.from_numpy(synth.synth_images(16, 3, hw, hw, seed=99))
            adj = -net(xc).mean(0)
            net.head[-1].bias += adj
        x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=0))
        # ROI ids deliberately unsorted and sparse (quirk Q11): 7,3,12,...
        rois = [int(r) for r in (synth.hash_u32(n, 77) % 1000 + 2)]
        paths = [f"/x/D20180712T065600_IFCB114_{r:05d}.png" for r in rois]
        half = n // 2
        loader = [(x[:half], paths[:half]), (x[half:], paths[half:])]
        res = net_pass(net, loader, "cpu")
        tag = f"{network}_{hw}"
        out[f"{tag}_rois_in"] = np.array(rois, dtype=np.int64)
        out[f"{tag}_bias_adj"] = adj.numpy()
        out[f"{tag}_rois_out"] = np.array([r for r, _ in res], dtype=np.int64)
        out[f"{tag}_probs"] = np.array([p for _, p in res], dtype=np.float32)
        with torch.no_grad():
            net.eval()
            out[f"{tag}_logits"] = net(x).numpy()
        print(tag, "top1", out[f"{tag}_probs"].argmax(1), "pmax", out[f"{tag}_probs"].max(1))
    np.savez_compressed(HERE / ("net_pass_effnet.npz" if effnet else "net_pass.npz"), **out)


def golden_net_pass_diverse():
    """A second net_pass fixture whose 8 images have (almost) all-different arg-max classes, so that the tests'
    top-1 assertion bites: 48 candidate images go through the reference's net_pass; 8 are picked greedily for
    distinct top-1 classes with a top-2 margin above 4e-3; those 8 go through the reference's net_pass again
    (two ragged batches, unsorted sparse ROI ids) and are stored with their indices into the candidate batch."""
    from sykepic.compute.probability import net_pass
    out = {}
    for network, hw in (("resnet18", 180), ("resnet50", 224)):
        net = build_ref_net(network, 50, seed=2)
        with torch.no_grad():
            # standardise every class's logit over a calibration batch (row scale of the last Linear + bias
            # shift, both stored and re-applied by the tests): a random-weight net otherwise lets the two or
            # three classes with the largest logit variance win every image
            net.eval()
            xc = torch.from_numpy(synth.synth_images(32, 3, hw, hw, seed=99))
            z = net(xc)
            row_scale = 4.0 / z.std(0)
            net.head[-1].weight *= row_scale[:, None]
            net.head[-1].bias *= row_scale
            adj = -net(xc).mean(0)
            net.head[-1].bias += adj
        cand = torch.from_numpy(synth.synth_images(48, 3, hw, hw, seed=7))
        res = net_pass(net, [(cand[i:i + 16], [f"/x/S_{j:05d}.png" for j in range(i, i + 16)]) for i in (0, 16, 32)], "cpu")
        probs = np.array([p for _, p in res], dtype=np.float64)      # ROI ids == candidate indices
        top2 = np.sort(probs, axis=1)[:, -2:]
        margin = top2[:, 1] - top2[:, 0]
        pick, seen = [], set()
        for i in np.argsort(-margin):
            c = int(probs[i].argmax())
            if c not in seen and margin[i] > 4e-3:
                pick.append(int(i))
                seen.add(c)
            if len(pick) == 8:
                break
        for i in np.argsort(-margin):   # fill up if fewer than 8 distinct classes exist
            if len(pick) == 8:
                break
            if int(i) not in pick:
                pick.append(int(i))
        x = cand[pick]
        rois = [int(r) for r in (synth.hash_u32(8, 78) % 1000 + 2)]
        paths = [f"/x/D20180712T065600_IFCB114_{r:05d}.png" for r in rois]
        res = net_pass(net, [(x[:3], paths[:3]), (x[3:], paths[3:])], "cpu")
        tag = f"{network}_{hw}"
        out[f"{tag}_index"] = np.array(pick, dtype=np.int64)
        out[f"{tag}_rois_in"] = np.array(rois, dtype=np.int64)
        out[f"{tag}_bias_adj"] = adj.numpy()
        out[f"{tag}_row_scale"] = row_scale.numpy()
        out[f"{tag}_rois_out"] = np.array([r for r, _ in res], dtype=np.int64)
        out[f"{tag}_probs"] = np.array([p for _, p in res], dtype=np.float32)
        distinct = len(set(out[f"{tag}_probs"].argmax(1).tolist()))
        print(tag, "top1", out[f"{tag}_probs"].argmax(1), "distinct", distinct)
        assert distinct >= 6, "fixture must have at least 6 distinct arg-max classes"
    np.savez_compressed(HERE / "net_pass_diverse.npz", **out)


class SnapshotLoader:
    """Validation 'dataloader' that snapshots the net each time the
    reference's train_net starts its validation phase (i.e. right after the
    epoch's training steps)."""

    def __init__(self, net, batches, sink):
        self.net, self.batches, self.sink = net, batches, sink

    def __iter__(self):
        self.sink.append({k: v.detach().clone() for k, v in self.net.state_dict().items()})
        return iter(self.batches)


class RecordingLoss(torch.nn.Module):
    def __init__(self, sink):
        super().__init__()
        self.inner = torch.nn.CrossEntropyLoss()
        self.sink = sink

    def forward(self, out, y):
        v = self.inner(out, y)
        self.sink.append((float(v), out.detach().clone(), bool(out.requires_grad)))
        return v


def golden_train(optim_name, tmp):
    """3 epochs x 1 batch through the reference's train_net with LRWarmup
    steps at epochs 1,2,3 => one step per phase of the unfreeze schedule."""
    from sykepic.train import network as rnet
    from sykepic.train.train import train_net
    import sykepic.analyze.plot as rplot
    rplot.plot_stats = lambda *a, **k: None  # cosmetic PNG writer (out of scope)

    torch.manual_seed(0)
    n, hw, classes = 8, 64, 10
    from sykepic.train.config import get_network
    net = get_network(ref_config("resnet18", (3, hw, hw)), classes)
    g = arch.build_graph("resnet18", classes)
    specs = arch.param_specs(g)
    sd0 = synth.synth_state_dict(specs, seed=5, logit_gain=2.0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
    rnet.freeze(net.base)
    lr = 0.01
    first = [p for p in net.parameters() if p.requires_grad]
    opt = getattr(torch.optim, optim_name)([
        {"params": first, "lr": lr}, {"params": [], "lr": 0.0}, {"params": [], "lr": 0.0}])
    warm = rnet.LRWarmup(net, opt, 0.1, 0.5, 1, 2, 3, verbose=False)
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10))
    y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
    xv = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=12))
    yv = torch.from_numpy(synth.synth_labels(n, classes, seed=13))
    snaps, losses = [], []
    val = SnapshotLoader(net, [(xv, yv)], snaps)
    with contextlib.redirect_stdout(io.StringIO()) as log, contextlib.redirect_stderr(io.StringIO()):
        best = train_net(net, [(x, y)], val, opt, RecordingLoss(losses), 3, 12, Path(tmp), "cpu",
                         None, warm)
    text = log.getvalue()
    assert "[ERROR]" not in text, text
    out = {}
    train_rec = [l for l in losses if l[2]]
    val_rec = [l for l in losses if not l[2]]
    out["train_loss"] = np.array([l[0] for l in train_rec], dtype=np.float64)
    out["val_loss"] = np.array([l[0] for l in val_rec], dtype=np.float64)
    out["train_logits"] = np.stack([l[1].numpy() for l in train_rec])
    out["val_logits"] = np.stack([l[1].numpy() for l in val_rec])
    out["group_lr"] = np.array([[gp["lr"] for gp in opt.param_groups]], dtype=np.float64)
    out["group_sizes"] = np.array([sum(p.numel() for p in gp["params"]) for gp in opt.param_groups])
    keys = [k for k, _, _ in specs]
    for e, snap in enumerate(snaps):
        out[f"e{e+1}_l2"] = np.array([float(snap[k].double().norm()) for k in keys])
        out[f"e{e+1}_sum"] = np.array([float(snap[k].double().sum()) for k in keys])
        for k in ("base.0.weight", "base.1.weight", "base.1.running_mean", "base.1.running_var",
                  "base.4.0.conv1.weight", "base.7.1.conv2.weight", "base.7.1.bn2.bias",
                  "head.0.weight", "head.2.weight", "head.2.bias"):
            out[f"e{e+1}_{k}"] = snap[k].flatten()[:64].numpy().copy()
        out[f"e{e+1}_nbt"] = np.array(int(snap["base.1.num_batches_tracked"]))
    print(optim_name, "train loss", out["train_loss"], "val loss", out["val_loss"],
          "lrs", out["group_lr"], "sizes", out["group_sizes"], "best exists", Path(best).exists())
    np.savez_compressed(HERE / f"train_{optim_name.lower()}.npz", **out)


def golden_schedules():
    from sykepic.train import network as rnet
    net = build_ref_net("resnet18", 50, seed=2)
    rnet.freeze(net.base)
    first = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.Adam([{"params": first, "lr": 0.01}, {"params": [], "lr": 0.0},
                            {"params": [], "lr": 0.0}])
    warm = rnet.LRWarmup(net, opt, 0.1, 0.5, 4, 14, 24, verbose=False)
    traj = []
    for epoch in range(1, 27):
        warm(epoch)
        traj.append({"epoch": epoch,
                     "lr": [g["lr"] for g in opt.param_groups],
                     "n_tensors": [len(g["params"]) for g in opt.param_groups],
                     "n_elems": [sum(p.numel() for p in g["params"]) for g in opt.param_groups]})
    # quirk Q3: ReduceLROnPlateau(optimizer, "min", factor, patience, verbose)
    # passes `verbose` positionally into `threshold` (train/train.py:159-161)
    opt2 = torch.optim.SGD([{"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 1.0}], lr=1.0)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt2, "min", 0.1, 4, True)
    vals = [1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0.09, 0.08, 0.07]
    plateau = []
    for v in vals:
        sched.step(v)
        plateau.append(opt2.param_groups[0]["lr"])
    (HERE / "schedules.json").write_text(json.dumps(
        {"lr_warmup": traj, "plateau_q3": {"factor": 0.1, "patience": 4, "threshold": True,
                                            "val_loss": vals, "lr_after": plateau}}, indent=1))
    print("warmup final", traj[-1]["lr"], "plateau", plateau)


def golden_prediction():
    from sykepic.compute.prediction import prediction_dataframe
    data = HERE / "ref_data"
    data.mkdir(exist_ok=True)
    # data files the reference's own tests hold (fixtures, not source)
    for rel in ("tests/data/prob/D20180712T065600_IFCB114.prob.csv",
                "tests/model/thresholds-2021.txt", "tests/model/thresholds-zero.txt",
                "tests/model/resnet18_20201022/class_names.txt",
                "tests/model/resnet18_20201022/config.ini",
                "tests/data/raw/valid/D20180712T065600_IFCB114.adc",
                "tests/data/raw/valid/D20180712T065600_IFCB114.hdr",
                "tests/data/raw/valid/D20180712T065600_IFCB114.roi"):
        shutil.copy(REF / rel, data / Path(rel).name)
    res = {}
    for thr in ("thresholds-2021.txt", "thresholds-zero.txt", 0.0, 0.3):
        arg = str(data / thr) if isinstance(thr, str) else thr
        df = prediction_dataframe(data / "D20180712T065600_IFCB114.prob.csv", arg)
        res[str(thr)] = {"roi": [int(i) for i in df.index],
                         "prediction": [str(s) for s in df["prediction"]],
                         "classified": [bool(b) for b in df["classified"]]}
    (HERE / "prediction.json").write_text(json.dumps(res, indent=1))
    print(res)


def golden_input_helpers():
    """The cv2-free input helpers of the reference, run unmodified: `image.get_new_dims` over a size grid,
    `ifcb.raw_to_numpy` / `next_roi` on the valid raw fixture, and `image.Compose` (+ every transform class) with the
    stand-in primitives of tests/golden/standins.py installed as `cv2` - the call log and the output of each run pin the
    control flow around the pixel interpolation (sykepic/train/image.py:9-180, sykepic/utils/ifcb.py:121-145)."""
    import random
    import cv2
    import standins as S
    from sykepic.train import image as rimg
    from sykepic.utils import ifcb as rifcb
    out = {}
    grid = []
    for th, tw in ((180, 180), (224, 224), (299, 299), (64, 96), (96, 64)):
        for h in (1, 2, 3, 7, 16, 41, 42, 56, 99, 100, 179, 180, 181, 223, 224, 225, 500, 1023):
            for w in (1, 2, 5, 17, 42, 55, 56, 57, 100, 180, 224, 333, 1024):
                nh, nw = rimg.get_new_dims(h, w, th, tw)
                grid.append([h, w, th, tw, int(nh), int(nw)])
    out["get_new_dims"] = grid
    data = HERE / "ref_data"
    rois = []
    for i, a in rifcb.raw_to_numpy(data / "D20180712T065600_IFCB114.adc", data / "D20180712T065600_IFCB114.roi"):
        rois.append({"id": int(i), "shape": [int(v) for v in a.shape], "sum": int(a.astype(np.int64).sum()),
                     "crc": S.crc(a), "dtype": str(a.dtype)})
    out["raw_to_numpy"] = rois
    # ---- Compose under recording stand-ins ----
    log = []

    def resize(img, dsize, fx=0, fy=0, interpolation=None):
        h, w = img.shape[:2]
        if dsize is None:
            nw, nh = int(np.rint(w * fx)), int(np.rint(h * fy))     # cv2: saturate_cast<int>(src * f)
            log.append(["resize", nw, nh, round(float(fx), 9)])
        else:
            nw, nh = int(dsize[0]), int(dsize[1])
            log.append(["resize", nw, nh, None])
        return S.nn_resize(img, nw, nh)

    def copy_make_border(img, top, bot, left, right, borderType=None, value=None):
        log.append(["pad", int(top), int(bot), int(left), int(right), S.border_list(value)])
        return S.pad(img, top, bot, left, right, S.border_list(value))

    def calc_hist(imgs, channels, mask, sizes, ranges):
        return np.bincount(imgs[0][..., channels[0]].reshape(-1), minlength=256).astype(np.float32)

    def flip(img, code):
        log.append(["flip", int(code)])
        return np.ascontiguousarray(img[:, ::-1] if code == 1 else img[::-1])

    def warp_affine(img, m, dsize, borderValue=None):
        log.append(["warp", S.mat_list(m), [int(dsize[0]), int(dsize[1])], S.border_list(borderValue)])
        return S.warp(img, m, S.border_list(borderValue))

    def get_rot(center, angle, scale):
        log.append(["rot", [int(center[0]), int(center[1])], int(angle), float(scale)])
        return S.rotation_matrix(center, angle, scale)

    cv2.resize, cv2.copyMakeBorder, cv2.calcHist, cv2.flip = resize, copy_make_border, calc_hist, flip
    cv2.warpAffine, cv2.getRotationMatrix2D = warp_affine, get_rot
    runs = []
    for pname, dims, border, spec in S.PIPELINES:
        ts = [getattr(rimg, t[0])(*t[1:]) for t in spec]
        comp = rimg.Compose(ts, dims, border)
        for iname, img in S.test_images():
            for seed in (0, 1, 2):
                random.seed(seed * 1000 + len(iname))
                del log[:]
                mode = rimg.mode_pixel_value(img)
                res = comp(img.copy())
                runs.append({"pipeline": pname, "image": iname, "seed": seed, "mode": int(mode),
                             "trace": json.loads(json.dumps(log)), "out_shape": [int(v) for v in res.shape],
                             "out_crc": S.crc(res), "rand_after": random.random()})
    out["compose"] = runs
    (HERE / "input_helpers.json").write_text(json.dumps(out))
    print("input helpers:", len(grid), "size pairs,", len(rois), "ROIs,", len(runs), "Compose runs")


if __name__ == "__main__":
    sys.path.insert(0, str(HERE))
    install_shims()
    torch.set_num_threads(8)
    import tempfile
    which = sys.argv[1:] or ["net_pass", "train", "sched", "pred"]
    if "net_pass" in which:
        golden_net_pass()
    if "effnet" in which:
        golden_net_pass(effnet=True)
    if "diverse" in which:
        golden_net_pass_diverse()
    if "train" in which:
        for name in ("SGD", "Adam"):
            with tempfile.TemporaryDirectory() as tmp:
                golden_train(name, tmp)
    if "sched" in which:
        golden_schedules()
    if "pred" in which:
        golden_prediction()
    if "helpers" in which:
        golden_input_helpers()
