"""GPU training transforms (`spk_preprocess_rois` + `spk_augment_batch`) against the host
pipeline (sykepic_hip/preprocess.py, the restated Compose of the reference): with the same `random` seed the
batched GPU result is byte-identical to transforming the images one by one on the host."""

import random

import numpy as np
import pytest
import torch

from sykepic_hip import gpu_augment, preprocess as P

pytestmark = pytest.mark.gpu


def _images(rng, n):
    out = []
    for _ in range(n):
        h, w = rng.randint(20, 260), rng.randint(20, 330)
        g = rng.randint(0, 256, (h, w)).astype(np.uint8)
        g[rng.rand(h, w) < 0.45] = rng.randint(120, 210)      # a clear modal grey level
        out.append(np.repeat(g[:, :, None], 3, axis=2))
    return out


PIPELINES = {
    "reference default (flip, translate, zoom, brightness)":
        lambda: [P.Resize(), P.FlipHorizontal(), P.FlipVertical(), P.Translate(), P.Zoom((0.6, 1.4)),
                 P.ChangeBrightness((0.95, 1.1)), P.ToTensor()],
    "all six": lambda: [P.Resize(), P.FlipHorizontal(), P.FlipVertical(), P.Translate(), P.Rotate(10),
                        P.Zoom((0.6, 1.4)), P.ChangeBrightness((0.8, 1.3)), P.ToTensor()],
    "rotate only": lambda: [P.Resize(), P.Rotate(45), P.ToTensor()],
    "zoom only": lambda: [P.Resize(), P.Zoom((0.5, 2.0)), P.ToTensor()],
    "eval": lambda: [P.Resize(), P.ToTensor()],
}


@pytest.mark.parametrize("name", list(PIPELINES))
@pytest.mark.parametrize("border", ["mode", "white"])
def test_gpu_transform_is_byte_identical_to_the_host_pipeline(name, border):
    rng = np.random.RandomState(5)
    imgs = _images(rng, 24)
    t = P.Compose(PIPELINES[name](), (180, 180), border)
    assert gpu_augment.supported(t, 3)
    random.seed(99)
    want = torch.stack([t(im) for im in imgs])                      # [n, 3, H, W] float32 in [0, 1]
    random.seed(99)
    got = gpu_augment.GpuTransform(t, "cuda:0")(imgs)               # [n, H, W, 3] uint8
    want_u8 = (want * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1)
    diff = (got.cpu().int() - want_u8.int()).abs()
    assert int(diff.max()) == 0, f"{name}/{border}: {int((diff > 0).sum())} bytes differ, max {int(diff.max())}"
    assert random.random() == (random.seed(99), [t(im) for im in imgs], random.random())[2]   # same number of draws


def test_unsupported_pipelines_fall_back():
    t = P.Compose([P.Resize(), P.ToTensor(), P.Normalize(P.IMAGENET_MEAN, P.IMAGENET_STD)], (180, 180), "mode")
    assert gpu_augment.supported(t, 3) and not gpu_augment.supported(t, 1)      # Normalize: 3-channel pipelines only
    assert not gpu_augment.supported(P.Compose([P.Resize(), P.Zoom((0.8, 1.2)), P.ToTensor()], (120, 180), "mode"), 3)
    assert not gpu_augment.supported(P.Compose([P.Resize(), P.Normalize(P.IMAGENET_MEAN, P.IMAGENET_STD), P.ToTensor()],
                                               (180, 180), "mode"), 3)          # Normalize anywhere but last
    assert not gpu_augment.supported(P.Compose([P.Resize(), P.ToTensor()], (180, 180), "mode"), 4)
    assert not gpu_augment.supported(P.Compose([P.Resize(), P.ToTensor()], (180, 180), (10, 20, 30)), 3)


def test_loader_feeds_the_training_step(tmp_path):
    """GpuLoader over PNG files: batches of uint8 NHWC on the GPU drive forward_backward directly."""
    from PIL import Image
    from sykepic_hip.net import HipNet
    rng = np.random.RandomState(1)
    paths, labels = [], []
    for i, im in enumerate(_images(rng, 10)):
        p = tmp_path / f"img_{i:02d}.png"
        Image.fromarray(im[..., 0]).save(p)
        paths.append(p)
        labels.append(i % 3)
    t = P.Compose(PIPELINES["reference default (flip, translate, zoom, brightness)"](), (64, 64), "mode")
    loader = gpu_augment.GpuLoader(paths, labels, t, 4, "cuda:0", shuffle=True)
    assert len(loader) == 3 and len(loader.dataset) == 10
    net = HipNet("resnet18", 3, weights=None, head=(8,))
    net.train()
    seen = 0
    for x, y in loader:
        assert x.dtype == torch.uint8 and x.is_cuda and x.shape[1:] == (64, 64, 3)
        net.reset_stats()
        net.forward_backward(x, y)
        seen += len(y)
    assert seen == 10


def test_loader_threads_do_not_change_the_batches(tmp_path):
    """The decode workers and the batch-assembly thread only move work off the training thread: under the same seeds
    the batches (augmentations included) are byte-identical to the single-threaded loader, in the same order, and
    equal to the host pipeline applied image by image.  A grey image stored as RGB takes the GPU path too."""
    import random
    from PIL import Image
    rng = np.random.RandomState(7)
    paths, labels = [], []
    for i, im in enumerate(_images(rng, 23)):
        p = tmp_path / f"img_{i:02d}.png"
        (Image.fromarray(im) if i == 5 else Image.fromarray(im[..., 0])).save(p)    # i == 5: RGB file, equal channels
        paths.append(p)
        labels.append(i % 4)
    t = P.Compose(PIPELINES["reference default (flip, translate, zoom, brightness)"](), (64, 64), "mode")

    def run(workers):
        random.seed(11)
        torch.manual_seed(11)
        loader = gpu_augment.GpuLoader(paths, labels, t, 6, "cuda:0", shuffle=True, workers=workers)
        out = [(x.cpu(), y.clone()) for _ in range(2) for x, y in loader]      # two epochs: fresh order each
        return out, float(torch.rand(1)), random.random()

    (one, t1, r1), (many, t2, r2) = run(1), run(4)
    assert (t1, r1) == (t2, r2)                      # both generators end where the single-threaded loader leaves them
    assert len(one) == len(many) == 8
    assert not torch.equal(one[0][1], one[4][1])     # the second epoch is shuffled anew
    many_first = many[:4]
    for (xa, ya), (xb, yb) in zip(one, many):
        assert torch.equal(ya, yb) and torch.equal(xa, xb)
    # the host pipeline, image by image, in the loader's order and with the same draws
    random.seed(11)
    torch.manual_seed(11)
    order = torch.randperm(len(paths)).tolist()
    from sykepic_hip import pngio
    k = 0
    for xb, _ in many_first:
        for j in range(xb.shape[0]):
            want = t(pngio.read_image(paths[order[k]], 3))              # float CHW in [0, 1]
            want_u8 = (want * 255.0).round().to(torch.uint8).permute(1, 2, 0)
            assert torch.equal(xb[j], want_u8), (k, order[k])
            k += 1
    assert k == len(paths)


def test_gpu_transform_with_imagenet_normalization_and_one_channel():
    """The two pipeline tails that used to fall back to host workers: the ImageNet `Normalize` the reference appends to
    its TRAIN transform (sykepic/train/config.py:55-56) - float32 [n, 3, H, W], bit-equal to ToTensor + Normalize on
    the host - and 1-channel models ([n, H, W, 1] uint8 = the host pipeline's single channel)."""
    rng = np.random.RandomState(3)
    imgs = _images(rng, 10)
    aug = [P.Resize(), P.FlipHorizontal(), P.FlipVertical(), P.Translate(), P.Zoom((0.7, 1.3)), P.ChangeBrightness((0.9, 1.1))]
    t = P.Compose(aug + [P.ToTensor(), P.Normalize(P.IMAGENET_MEAN, P.IMAGENET_STD)], (96, 96), "mode")
    assert gpu_augment.supported(t, 3) and not gpu_augment.supported(t, 1)
    random.seed(5)
    want = torch.stack([t(im) for im in imgs])
    random.seed(5)
    got = gpu_augment.GpuTransform(t, "cuda:0")(imgs)
    assert got.dtype == torch.float32 and tuple(got.shape) == (10, 3, 96, 96)
    assert torch.equal(got.cpu(), want), float((got.cpu() - want).abs().max())
    t1 = P.Compose(aug + [P.ToTensor()], (96, 96), "mode")
    assert gpu_augment.supported(t1, 1)
    random.seed(6)
    want1 = torch.stack([t1(im[..., :1]) for im in imgs])                       # [n, 1, H, W] float in [0, 1]
    random.seed(6)
    got1 = gpu_augment.GpuTransform(t1, "cuda:0", num_chans=1)(imgs)
    assert got1.dtype == torch.uint8 and tuple(got1.shape) == (10, 96, 96, 1)
    assert torch.equal(got1.cpu().permute(0, 3, 1, 2), (want1 * 255.0).round().to(torch.uint8))
