"""GPU ROI preprocessing (`spk_preprocess_rois`) against
the host pipeline (sykepic_hip/preprocess.py, the restated
Compose/Resize/mode-border of the reference): byte-identical uint8 output on
the reference's own raw fixture and on synthetic ROIs of every shape class
(wide, tall, square, exact 2x, identity, 1-pixel, larger than the target)."""

import numpy as np
import pytest
import torch

from sykepic_hip import gpu_preprocess, preprocess

pytestmark = pytest.mark.gpu


def _host(img, th, tw, border):
    t = preprocess.Compose([preprocess.Resize()], (th, tw), border)
    return t(np.repeat(img[:, :, None], 3, axis=2))


def _write_sample(tmp_path, name, imgs):
    adc, blob, off = [], [], 0
    for im in imgs:
        h, w = (0, 0) if im is None else im.shape
        cols = ["0"] * 24
        cols[15], cols[16], cols[17] = str(w), str(h), str(off)
        adc.append(",".join(cols))
        if im is not None:
            blob.append(im.reshape(-1))
            off += im.size
    (tmp_path / f"{name}.adc").write_text("\n".join(adc) + "\n")
    np.concatenate(blob).astype(np.uint8).tofile(tmp_path / f"{name}.roi")
    return tmp_path / f"{name}.adc", tmp_path / f"{name}.roi"


@pytest.mark.parametrize("border", ["mode", "black", "white"])
def test_matches_host_pipeline_bytewise(tmp_path, border):
    rng = np.random.RandomState(3)
    shapes = [(42, 56), (53, 128), (180, 180), (360, 360), (90, 45), (1, 1), (1, 300), (300, 1), (7, 500),
              (400, 77), (64, 64), (200, 100), (179, 181), (360, 90)]
    imgs = [None]
    for h, w in shapes:
        im = rng.randint(0, 256, (h, w)).astype(np.uint8)
        im[rng.rand(h, w) < 0.4] = rng.randint(100, 220)   # a clear modal grey level
        imgs.append(im)
    adc, roi = _write_sample(tmp_path, "D20200101T000000_IFCB114", imgs)
    th, tw = 180, 180
    gs = gpu_preprocess.SampleOnGpu(adc, roi, "cuda:0")
    assert gs.numbers.tolist() == list(range(2, len(imgs) + 1))          # the empty trigger is skipped
    tr = preprocess.Compose([preprocess.Resize(), preprocess.ToTensor()], (th, tw), border)
    assert gpu_preprocess.supported(tr, 3)
    got = gs.batch(0, len(gs), th, tw, gpu_preprocess.border_code(tr)).cpu().numpy()
    for i, im in enumerate(imgs[1:]):
        want = _host(im, th, tw, border)
        assert got[i].shape == want.shape == (th, tw, 3)
        assert np.array_equal(got[i], want), f"ROI shape {im.shape}: {np.abs(got[i].astype(int) - want).max()}"


def test_reference_fixture_and_forward(golden_dir):
    from sykepic_hip import arch, synth
    from sykepic_hip.net import HipNet
    d = golden_dir / "ref_data"
    gs = gpu_preprocess.SampleOnGpu(d / "D20180712T065600_IFCB114.adc", d / "D20180712T065600_IFCB114.roi", "cuda:0")
    assert gs.numbers.tolist() == [2, 3]
    x8 = gs.batch(0, 2, 180, 180, -1)
    assert int(x8[0, 0, 0, 0]) == 164 and int(x8[1, 0, 0, 0]) == 206     # modal greys
    # the uint8 NHWC batch drives the forward exactly like the float NCHW tensor the host pipeline builds
    g = arch.build_graph("resnet18", 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = HipNet("resnet18", 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    xf = (x8.permute(0, 3, 1, 2).float() / 255.0).contiguous()
    assert float((net.probabilities(x8) - net.probabilities(xf)).abs().max()) < 2e-5


def test_gpu_prediction_matches_reference_golden(golden_dir):
    """spk_predict_rows vs the golden the reference's prediction module produced
    and vs the host implementation on random probabilities."""
    import json

    import pandas as pd

    from sykepic_hip import prediction
    gold = json.loads((golden_dir / "prediction.json").read_text())
    d = golden_dir / "ref_data"
    df = pd.read_csv(d / "D20180712T065600_IFCB114.prob.csv", index_col=0)
    classes = list(df.columns)
    probs = torch.tensor(df.to_numpy(), dtype=torch.float32).cuda()
    for key, want in gold.items():
        thr = prediction.threshold_dictionary(d / key) if key.endswith(".txt") else float(key)
        pred, ok = prediction.predict_gpu(probs, classes, thr)
        assert [classes[i] for i in pred.tolist()] == want["prediction"]
        assert ok.tolist() == want["classified"]
    rng = np.random.RandomState(0)
    p = rng.dirichlet(np.ones(50) * 0.3, size=4000).astype(np.float32)
    thr = {c: float(t) for c, t in zip(classes, rng.uniform(0.05, 0.9, 50)) if rng.rand() < 0.8}
    for t in (thr, 0.3, 0.0):
        want_i, want_ok = prediction.predict_arrays(p, classes, t)
        got_i, got_ok = prediction.predict_gpu(torch.from_numpy(p).cuda(), classes, t)
        assert np.array_equal(got_i.cpu().numpy(), want_i) and np.array_equal(got_ok.cpu().numpy(), want_ok)
