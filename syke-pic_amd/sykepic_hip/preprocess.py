"""ROI preprocessing: aspect-preserving bilinear resize, centred border of the
image's modal grey level, ``/255`` -> CHW float32 (+ optional ImageNet
normalisation), and the training augmentations.

Host-side mirror of the reference's ``Compose``/``Resize``/... in
``sykepic/train/image.py`` (``Compose.__call__`` :25-56, ``get_new_dims``
:183, ``resize_with_border`` :201, ``mode_pixel_value`` :229) and of
``ToTensor``/``Normalize`` (``sykepic/train/config.py:52-56``).  The reference
delegates the pixel work to OpenCV, which is not available in this image; the
resize below restates OpenCV's published 8-bit INTER_LINEAR algorithm
(11-bit fixed-point coefficients, the same rounding), the warps use plain
bilinear sampling.  Augmentations draw from Python's global ``random`` in the
same order as the reference so a seeded run makes the same decisions.
"""

import math
import random

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def mode_pixel_value(img):
    """Most common value of channel 0 (ties: the lowest), like
    ``cv2.calcHist([img],[0],...)`` + argmax."""
    ch = img[..., 0] if img.ndim == 3 else img
    return int(np.argmax(np.bincount(ch.reshape(-1), minlength=256)))


def get_new_dims(h, w, target_h, target_w):
    if h > w:
        return target_h, int(w * (target_h / float(h)))
    return int(h * (target_w / float(w))), target_w


def _coeffs(src, dst):
    """OpenCV resizeLinear index/weight tables for one axis (8-bit path)."""
    scale = src / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = (d + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = f - s
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    a1 = np.rint(f * 2048.0).astype(np.int64)
    a0 = np.rint((1.0 - f) * 2048.0).astype(np.int64)
    return s, np.minimum(s + 1, src - 1), a0, a1


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h), interpolation=INTER_LINEAR) for uint8."""
    h, w = img.shape[:2]
    if (h, w) == (new_h, new_w):
        return img.copy()
    new_w, new_h = max(int(new_w), 1), max(int(new_h), 1)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[:, :, None]
    if w == 2 * new_w and h == 2 * new_h:
        # OpenCV switches exact 2x down-scaling to its INTER_AREA fast path
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
    else:
        x0, x1, ax0, ax1 = _coeffs(w, new_w)
        y0, y1, ay0, ay1 = _coeffs(h, new_h)
        rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]  # x2048
        r0, r1 = rows[y0], rows[y1]
        out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out


def resize_with_border(img, new_dims, target_dims, border):
    new_h, new_w = new_dims
    target_h, target_w = target_dims
    img = resize_linear_u8(img, new_w, new_h)
    h, w = img.shape[:2]
    pad_h, pad_w = max(target_h - h, 0), max(target_w - w, 0)
    top, left = pad_h // 2, pad_w // 2
    return pad_constant(img, top, pad_h - top, left, pad_w - left, border)


def pad_constant(img, top, bot, left, right, border):
    h, w = img.shape[:2]
    c = img.shape[2] if img.ndim == 3 else 1
    out = np.empty((h + top + bot, w + left + right, c), dtype=np.uint8)
    out[:] = np.asarray(border[:c] if c <= 3 else border, dtype=np.uint8)
    out[top:top + h, left:left + w] = img.reshape(h, w, c)
    return out


def _warp_affine(img, minv, border):
    """dst(x,y) = bilinear src at minv @ (x,y,1); constant border."""
    h, w = img.shape[:2]
    c = img.shape[2]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = minv[0, 0] * xs + minv[0, 1] * ys + minv[0, 2]
    sy = minv[1, 0] * xs + minv[1, 1] * ys + minv[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    bv = np.asarray(border[:c], dtype=np.float64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float64)
        return np.where(ok[..., None], v, bv)

    out = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy)
           + tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


class Resize:
    def __repr__(self):
        return "Resize()"


class FlipHorizontal:
    def __call__(self, img):
        return img[:, ::-1].copy() if random.getrandbits(1) else img

    def __repr__(self):
        return "FlipHorizontal()"


class FlipVertical:
    def __call__(self, img):
        return img[::-1].copy() if random.getrandbits(1) else img

    def __repr__(self):
        return "FlipVertical()"


class Translate:
    def __call__(self, img, limit, border, height=True, width=True):
        x = y = 0
        if height:
            y = random.randint(-limit, limit)
        if width:
            x = random.randint(-limit, limit)
        minv = np.array([[1.0, 0.0, -x], [0.0, 1.0, -y]])
        return _warp_affine(img, minv, border)

    def __repr__(self):
        return "Translate()"


class Zoom:
    """Assumes a square, already resized image (reference quirk Q12: the crop
    branch uses the width for both axes)."""

    def __init__(self, zoom_range):
        self.zoom_range = zoom_range

    def __call__(self, img, border):
        f = round(random.uniform(*self.zoom_range), 2)
        h, w = img.shape[:2]
        zw, zh = int(round(w * f)), int(round(h * f))  # cv2.resize(fx,fy): dsize = round(src*f)
        img = resize_linear_u8(img, zw, zh)
        if f < 1:
            p1 = int((w - zw) / 2)
            p2 = w - zw - p1
            return pad_constant(img, p1, p2, p1, p2, border)
        c1 = (zw - w) / 2
        c2 = int(zw - c1)
        c1 = int(c1)
        return img[c1:c2, c1:c2]

    def __repr__(self):
        return f"Zoom(range={self.zoom_range})"


class Rotate:
    def __init__(self, max_angle):
        self.max_angle = max_angle

    def __call__(self, img, border):
        h, w = img.shape[:2]
        cx, cy = w // 2, h // 2
        angle = random.randint(-self.max_angle, self.max_angle)
        a = math.radians(angle)
        ca, sa = math.cos(a), math.sin(a)
        # forward map of cv2.getRotationMatrix2D(center, angle, 1): [[ca, sa],[-sa, ca]]
        fwd = np.array([[ca, sa, (1 - ca) * cx - sa * cy], [-sa, ca, sa * cx + (1 - ca) * cy]])
        full = np.vstack([fwd, [0, 0, 1]])
        return _warp_affine(img, np.linalg.inv(full)[:2], border)

    def __repr__(self):
        return f"Rotate(max_angle={self.max_angle})"


class ChangeBrightness:
    def __init__(self, brightness_range):
        self.brightness_range = brightness_range

    def __call__(self, img):
        v = random.uniform(*self.brightness_range)
        return (img * v).clip(0, 255).astype(np.uint8)

    def __repr__(self):
        return f"ChangeBrightness(brightness_range={self.brightness_range})"


class ToTensor:
    """HxWxC uint8 -> CxHxW float32 in [0,1]."""

    def __call__(self, img):
        return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float().div_(255.0)

    def __repr__(self):
        return "ToTensor()"


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std

    def __repr__(self):
        return "Normalize()"


class Compose:
    """Transformation pipeline with the reference's special cases: Resize
    gets the aspect-preserving dims and the border colour, Translate moves
    only along the padded axis, Zoom/Rotate get the border colour."""

    def __init__(self, transforms, target_dims, border):
        self.transforms = list(transforms)
        self.target_dims = tuple(target_dims)
        self.border = {"white": (255, 255, 255), "black": (0, 0, 0)}.get(border, border)

    def __call__(self, img):
        if self.border == "mode":
            m = mode_pixel_value(img)
            border = (m, m, m)
        else:
            border = self.border
        h, w = img.shape[:2]
        target_h, target_w = self.target_dims
        new_h, new_w = get_new_dims(h, w, target_h, target_w)
        for t in self.transforms:
            if isinstance(t, Resize):
                if border:
                    img = resize_with_border(img, (new_h, new_w), self.target_dims, border)
                else:
                    img = resize_linear_u8(img, new_w, new_h)
            elif isinstance(t, Translate):
                if h > w:
                    img = t(img, int((target_w - new_w) / 2.5), border, height=False, width=True)
                else:
                    img = t(img, int((target_h - new_h) / 2.5), border, height=True, width=False)
            elif isinstance(t, (Zoom, Rotate)):
                img = t(img, border)
            else:
                img = t(img)
        return img

    def __repr__(self):
        return "Compose(\n" + "\n".join(f"    {t}" for t in self.transforms) + "\n)"
