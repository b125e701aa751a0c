"""End-to-end `sykepic train` rate: PNG files on disk -> threaded decode -> GPU resize / augmentation -> ResNet training
step (the reference's unfreeze schedule, compressed: the head alone, then the last stage, then every layer) -> validation.  Prints images/s of whole epochs (wall clock between the
"----- Epoch" banners of train.main, validation included) next to the kernel-only rate bench.py reports.
Usage: python3 tools/e2e_train_rate.py [n_images=12288] [network=resnet50] [size=224] [batch=256]"""
import contextlib
import io
import sys
import time
from collections import namedtuple
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "syke-pic_amd"), str(ROOT / "tests")]
import numpy as np
from PIL import Image

from test_gpu_workflows import INI


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    network = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    tmp = Path("/tmp/e2e_train_rate")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    rng = np.random.RandomState(0)
    classes = 8
    t0 = time.time()
    for ci in range(classes):
        d = tmp / "ds" / f"class_{ci}"
        d.mkdir(parents=True)
        for i in range(n_img // classes):
            h, w = int(rng.randint(40, 160)), int(rng.randint(60, 300))     # IFCB-like ROI sizes
            img = np.clip(rng.normal(170 + 6 * ci, 12, (h, w)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(d / f"D20200101T000000_IFCB114_{i:05d}.png")
    print(f"{n_img} PNGs written in {time.time() - t0:.1f} s", flush=True)
    ini = INI.format(ds=tmp / "ds", models=tmp / "models", network=network)
    ini = ini.replace("shape = 3, 64, 64", f"shape = 3, {size}, {size}")
    ini = ini.replace("batch_size = 16", f"batch_size = {batch}").replace("max_epochs = 8", "max_epochs = 6")
    ini = ini.replace("split = 0.6, 0.2, 0.2", "split = 0.9, 0.05, 0.05").replace("oversample_until = 12", "oversample_until =")
    ini = ini.replace("head = 32, 16", "head = 256, 128").replace("num_workers = 0", "num_workers = 8")
    # the reference's unfreeze schedule, compressed: epochs 1-2 head only, 3 = head + the last two base modules, 4.. = all
    ini = ini.replace("step_1 = 3", "step_1 = 2").replace("step_2 = 5", "step_2 = 3").replace("step_3 = 7", "step_3 = 4")
    (tmp / "train.ini").write_text(ini)


    class Stamp(io.TextIOBase):
        """stdout tee that records the wall-clock time of every epoch banner"""

        def __init__(self, out):
            self.out, self.marks = out, []

        def write(self, s):
            if "----- Epoch" in s or "Model Evaluation" in s or "[STAT] Train Acc" in s:
                self.marks.append((time.time(), s.strip()))
            return self.out.write(s)

        def flush(self):
            self.out.flush()


    # the input pipeline alone (decode workers -> batch thread -> GPU preprocessing), no training step behind it
    import torch
    from sykepic_hip import gpu_augment, preprocess as P
    paths = sorted((tmp / "ds").rglob("*.png"))
    tf = P.Compose([P.Resize(), P.FlipHorizontal(), P.FlipVertical(), P.Translate(), P.Zoom((0.8, 1.2)),
                    P.ChangeBrightness((0.95, 1.1)), P.ToTensor()], (size, size), "mode")
    for workers in (8, 16):
        loader = gpu_augment.GpuLoader(paths, [0] * len(paths), tf, batch, "cuda:0", shuffle=True, workers=workers)
        t0 = time.time()
        n = 0
        for x, y in loader:
            n += len(y)
        torch.cuda.synchronize()
        print(f"input pipeline alone, {workers} decode workers: {n / (time.time() - t0):.0f} images/s", flush=True)

    from sykepic_hip import train
    Args = namedtuple("Args", "config dist collage")
    tee = Stamp(sys.stdout)
    with contextlib.redirect_stdout(tee):
        train.main(Args(str(tmp / "train.ini"), False, None))
    marks = tee.marks
    n_train = int(n_img * 0.9)
    for i, (ta, a) in enumerate(marks):
        if "Epoch" not in a:
            continue
        t_loop = next((tb for tb, b in marks[i + 1:] if "Train Acc" in b), None)
        t_next = next((tb for tb, b in marks[i + 1:] if "Epoch" in b or "Evaluation" in b), None)
        if t_loop is None or t_next is None:
            continue
        print(f"{a}: training loop {t_loop - ta:.2f} s -> {n_train / (t_loop - ta):.0f} images/s; whole epoch (validation, "
              f"plots, checkpoint) {t_next - ta:.2f} s -> {n_train / (t_next - ta):.0f} images/s")


if __name__ == "__main__":
    main()
