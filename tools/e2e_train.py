"""End-to-end `sykepic train` + `sykepic prob` on a small synthetic labelled set (GPU): does it learn?"""
import sys, time, random
from collections import namedtuple
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from PIL import Image
sys.path.insert(0, str(ROOT / "tests"))
from test_gpu_workflows import INI
random.seed(0); np.random.seed(0); torch.manual_seed(0)
tmp = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/e2e_train")
import shutil; shutil.rmtree(tmp, ignore_errors=True)
rng = np.random.RandomState(0)
ds = tmp / "ds"
for ci, name in enumerate(("blob", "bars", "flat", "rings")):
    (ds / name).mkdir(parents=True)
    for i in range(80):
        h, w = rng.randint(40, 100), rng.randint(40, 140)
        img = np.full((h, w), 170 + rng.randint(-15, 15), np.int32)
        yy, xx = np.mgrid[0:h, 0:w]
        if ci == 0:
            cy, cx, r = rng.randint(h // 4, 3 * h // 4), rng.randint(w // 4, 3 * w // 4), rng.randint(6, 14)
            img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = 50
        elif ci == 1:
            img[:, :: rng.randint(5, 9)] = 70
        elif ci == 3:
            cy, cx = h // 2, w // 2
            d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2).astype(int)
            img[(d % 12) < 3] = 90
        img = np.clip(img + rng.randint(-12, 12, (h, w)), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(ds / name / f"{name}_{i:03d}.png")
ini = tmp / "train.ini"
text = INI.format(ds=ds, models=tmp / "models").replace("max_epochs = 8", "max_epochs = 14").replace(
    "oversample_until = 12", "oversample_until = 60").replace("batch_size = 16", "batch_size = 32")
ini.write_text(text)
from sykepic_hip import train
t0 = time.time()
train.main(namedtuple("A", "config collage dist save_images")(str(ini), None, None, None))
print("train.main seconds", round(time.time() - t0, 1))
print((tmp / "models" / "resnet18_1" / "test_report.txt").read_text()[-600:])
