"""PNG input for ``--images`` / ``--image-dir`` and the training set.  The
reference reads with ``cv2.imread`` + BGR2RGB (``sykepic/train/data.py:214-222``);
cv2 is not part of this image, PIL gives the same pixels for 8-bit PNGs."""

import numpy as np
from PIL import Image


def read_image(path, num_chans=3):
    """HxWx3 RGB uint8 (num_chans == 3) or HxWx1 grey uint8."""
    with Image.open(path) as im:
        if num_chans == 3:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
        return np.asarray(im.convert("L"), dtype=np.uint8)[:, :, None]


def decode_png(path, need_mode):
    """One training image for the GPU input pipeline (gpu_augment.GpuLoader), run in a loader worker process:
    (grey HxW uint8, border value) for the greyscale PNGs IFCB writes - also when they are stored as RGB with equal
    channels - or (HxWx3 RGB, None) for a real colour image (the host pipeline takes those).  The grey array equals
    channel 0 of `read_image(path, 3)`; the border value is `preprocess.mode_pixel_value` of it (most common value,
    ties: the lowest).  Only numpy and PIL here: the workers never touch torch or the GPU."""
    with Image.open(path) as im:
        if im.mode == "L":
            g = np.asarray(im, dtype=np.uint8)
        else:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
            if not (np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 0], rgb[..., 2])):
                return rgb, None
            g = np.ascontiguousarray(rgb[..., 0])
    return g, (int(np.argmax(np.bincount(g.reshape(-1), minlength=256))) if need_mode else 0)


class PngDataset:
    """Map-style dataset for the loader's decode processes: index -> decode_png.  Lives here (numpy + PIL only) so
    that a decode process unpickles it without importing the GPU side of the package."""

    def __init__(self, paths, need_mode):
        self.paths, self.need_mode = [str(p) for p in paths], need_mode

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        return decode_png(self.paths[i], self.need_mode)


class IndexedPng(PngDataset):
    """... returning (index, decoded image): the consumer needs the labels of exactly these indices."""

    def __getitem__(self, i):
        return i, decode_png(self.paths[i], self.need_mode)


def identity(batch):
    return batch


def worker_init(worker_id):
    """A decode process that crashes leaves a Python traceback on stderr (the one crash seen so far left only torch's
    "Unexpected segmentation fault encountered in worker")."""
    import faulthandler
    import sys
    faulthandler.enable(file=sys.stderr, all_threads=True)
