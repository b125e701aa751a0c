"""Stand-in image primitives shared by tests/golden/make_golden.py (where they are installed as the absent `cv2`
under the REFERENCE's own `sykepic.train.image`) and tests/test_host.py (where they replace the corresponding
primitives of `sykepic_hip.preprocess`).  They are deliberately not OpenCV: cheap, deterministic, shape-correct
operations plus a call log, so that running the two `Compose` pipelines side by side pins everything AROUND the
pixel interpolation - which image gets which size, padding, border colour, translation limit and axis, zoom size /
crop, rotation centre and angle, and the order of the `random` draws - to the reference's code."""
import zlib

import numpy as np


def nn_resize(img, new_w, new_h):
    h, w = img.shape[:2]
    ys = (np.arange(new_h) * h) // max(new_h, 1)
    xs = (np.arange(new_w) * w) // max(new_w, 1)
    return np.ascontiguousarray(img[ys][:, xs])


def pad(img, top, bot, left, right, value):
    v = np.asarray(list(value)[: img.shape[2]] if img.ndim == 3 else [list(value)[0]], dtype=np.uint8)
    out = np.empty((img.shape[0] + top + bot, img.shape[1] + left + right) + img.shape[2:], dtype=np.uint8)
    out[...] = v if img.ndim == 3 else v[0]
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def warp(img, m, border):
    """Shift by the rounded translation column of the FORWARD matrix (rotation part ignored), constant border."""
    m = np.asarray(m, dtype=np.float64)
    dx, dy = int(round(float(m[0, 2]))) % 7 - 3, int(round(float(m[1, 2]))) % 5 - 2
    h, w = img.shape[:2]
    out = pad(np.zeros((0, 0) + img.shape[2:], dtype=np.uint8), 0, h, 0, w, border)
    ys0, ys1 = max(dy, 0), min(h + dy, h)
    xs0, xs1 = max(dx, 0), min(w + dx, w)
    if ys1 > ys0 and xs1 > xs0:
        out[ys0:ys1, xs0:xs1] = img[ys0 - dy:ys1 - dy, xs0 - dx:xs1 - dx]
    return out


def rotation_matrix(center, angle, scale):
    a = np.deg2rad(float(angle))
    al, be = float(scale) * np.cos(a), float(scale) * np.sin(a)
    cx, cy = float(center[0]), float(center[1])
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], dtype=np.float64)


def crc(a):
    a = np.ascontiguousarray(a)
    return int(zlib.crc32(a.tobytes())) & 0xFFFFFFFF


def border_list(b):
    return [int(v) for v in (list(b) if isinstance(b, (list, tuple, np.ndarray)) else [b, b, b])]


def mat_list(m):
    return [[round(float(v), 9) for v in row] for row in np.asarray(m, dtype=np.float64)]


def test_images():
    """(name, HxWx3 uint8) - grey replicated into three channels as IFCB PNGs are read."""
    out = []
    for k, (h, w) in enumerate([(56, 42), (30, 120), (200, 200), (17, 333), (180, 181), (1, 9)]):
        rng = np.random.RandomState(100 + k)
        g = rng.randint(0, 256, (h, w)).astype(np.uint8)
        g[: max(1, h // 3)] = 164 + k          # a dominant grey level: the modal border value
        out.append((f"{h}x{w}", np.repeat(g[:, :, None], 3, axis=2)))
    return out


PIPELINES = [
    # name, target dims, border, [transform names with arguments]
    ("eval_mode_180", (180, 180), "mode", [("Resize",)]),
    ("eval_white_224", (224, 224), "white", [("Resize",)]),
    ("train_mode_180", (180, 180), "mode", [("Resize",), ("FlipHorizontal",), ("FlipVertical",), ("Translate",),
                                          ("Zoom", (0.8, 1.2)), ("Rotate", 20), ("ChangeBrightness", (0.8, 1.2))]),
    ("train_black_224", (224, 224), "black", [("Resize",), ("Translate",), ("Rotate", 45), ("Zoom", (0.5, 1.5)),
                                            ("FlipVertical",)]),
]
