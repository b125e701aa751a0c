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
