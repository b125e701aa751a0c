"""Host logic of the training workflow (no GPU): schedule mirrors against the
goldens the reference produced, dataset split/oversampling semantics, the
training transform pipeline."""

import json
import random
from collections import OrderedDict

import numpy as np
import pytest
from PIL import Image

from sykepic_hip import arch, data, preprocess, schedule
from sykepic_hip.net import HipNet


class StubNet:
    """Module views of HipNet without the GPU library."""

    def __init__(self, network, classes):
        self.graph = arch.build_graph(network, classes)
        self._specs = arch.param_specs(self.graph)
        self._params = OrderedDict()
        self.groups = {}
        HipNet._build_views(self)

    def _set_requires_grad(self, key, flag):
        pass

    def set_param_group(self, key, group):
        self.groups[key] = group

    def parameters(self):
        return iter(self._params.values())


class StubOpt:
    def __init__(self, groups):
        self.param_groups = groups


def test_lr_warmup_matches_reference_trajectory(golden_dir):
    traj = json.loads((golden_dir / "schedules.json").read_text())["lr_warmup"]
    net = StubNet("resnet18", 50)
    schedule.freeze(net.base)
    first = [p for p in net.parameters() if p.requires_grad]
    opt = StubOpt([{"params": first, "lr": 0.01}, {"params": [], "lr": 0.0}, {"params": [], "lr": 0.0}])
    warm = schedule.LRWarmup(net, opt, 0.1, 0.5, 4, 14, 24, verbose=False)
    for rec in traj:
        warm(rec["epoch"])
        assert np.allclose([g["lr"] for g in opt.param_groups], rec["lr"])
        assert [len(g["params"]) for g in opt.param_groups] == rec["n_tensors"]
        assert [sum(p.numel() for p in g["params"]) for g in opt.param_groups] == rec["n_elems"]
    # after step_3 everything is trainable, BN stays in group 0
    assert all(p.requires_grad for p in net.parameters())
    assert all(p.is_bn or p.key.startswith("head.") for p in opt.param_groups[0]["params"])


def test_plateau_scheduler_reproduces_quirk_q3(golden_dir):
    gold = json.loads((golden_dir / "schedules.json").read_text())["plateau_q3"]
    opt = StubOpt([{"params": [], "lr": 1.0}])
    sched = schedule.ReduceLROnPlateau(opt, "min", gold["factor"], gold["patience"], gold["threshold"])
    got = []
    for v in gold["val_loss"]:
        sched.step(v)
        got.append(opt.param_groups[0]["lr"])
    assert np.allclose(got, gold["lr_after"])
    # with a sane threshold an improving loss never triggers a cut
    opt2 = StubOpt([{"params": [], "lr": 1.0}])
    s2 = schedule.ReduceLROnPlateau(opt2, "min", 0.1, 4, 1e-4)
    for v in gold["val_loss"]:
        s2.step(v)
    assert opt2.param_groups[0]["lr"] == 1.0


def _make_dataset(root, per_class):
    rng = np.random.RandomState(0)
    for name, n in per_class.items():
        (root / name).mkdir(parents=True)
        for i in range(n):
            h, w = rng.randint(20, 60), rng.randint(20, 90)
            Image.fromarray(rng.randint(0, 255, (h, w), dtype=np.uint8)).save(root / name / f"{name}_{i:03d}.png")


def test_model_data_split_and_oversampling(tmp_path):
    _make_dataset(tmp_path / "ds", {"Beta": 10, "alpha": 20, "Gamma": 5})
    md = data.ModelData(tmp_path / "ds", (0.6, 0.2, 0.2), None, None, ["Unclassified"], 42)
    assert list(md.le.classes_) == ["Beta", "Gamma", "alpha"]          # sorted like LabelEncoder
    assert md.distribution["alpha"] == [20, 12, 4, 4] and md.distribution["Gamma"] == [5, 3, 1, 1]
    assert len(md.train_x) == 21 and len(md.val_x) == 7 and len(md.test_x) == 7
    assert not (set(md.train_x) & set(md.val_x)) and not (set(md.train_x) & set(md.test_x))
    md2 = data.ModelData(tmp_path / "ds", (0.6, 0.2, 0.2), None, None, [], 42)
    assert md2.train_x == md.train_x                                    # seeded: reproducible
    md.oversample(15, None)
    assert md.distribution["Gamma"] == [5, 15, 1, 1, 12] and md.distribution["alpha"][4] == 3
    md.save(tmp_path / "model")
    lines = (tmp_path / "model" / "class_distribution.csv").read_text().splitlines()
    assert lines[0] == "class,total,train,validation,test,oversampled" and lines[1].startswith("alpha,20,15,")
    assert (tmp_path / "model" / "class_names.txt").read_text() == "Beta\nGamma\nalpha"
    assert data.auto_id("resnet18", tmp_path) == 1
    (tmp_path / "resnet18_3").mkdir()
    assert data.auto_id("resnet18", tmp_path) == 4
    with pytest.raises(ValueError):
        data.oversample([1], [1])


def test_train_transform_pipeline():
    random.seed(1)
    t = preprocess.Compose([preprocess.Resize(), preprocess.FlipHorizontal(), preprocess.FlipVertical(),
                            preprocess.Translate(), preprocess.Zoom((0.6, 1.4)), preprocess.Rotate(10),
                            preprocess.ChangeBrightness((0.95, 1.1)), preprocess.ToTensor()], (64, 64), "mode")
    rng = np.random.RandomState(0)
    for shape in ((30, 80), (80, 30), (64, 64), (5, 200)):
        img = np.repeat(rng.randint(0, 255, shape + (1,), dtype=np.uint8), 3, axis=2)
        out = t(img)
        assert tuple(out.shape) == (3, 64, 64) and 0.0 <= float(out.min()) and float(out.max()) <= 1.0
    # same seed, same decisions
    img = np.repeat(rng.randint(0, 255, (40, 50, 1), dtype=np.uint8), 3, axis=2)
    random.seed(7)
    a = t(img)
    random.seed(7)
    assert (a == t(img)).all()
