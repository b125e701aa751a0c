"""ROI preprocessing: aspect-preserving bilinear resize, centred border of the
image's modal grey level, ``/255`` -> CHW float32 (+ optional ImageNet
normalisation), and the training augmentations.

Host-side mirror of the reference's ``Compose``/``Resize``/... in
``sykepic/train/image.py`` (``Compose.__call__`` :25-56, ``get_new_dims``
:183, ``resize_with_border`` :201, ``mode_pixel_value`` :229) and of
``ToTensor``/``Normalize`` (``sykepic/train/config.py:52-56``).  The reference
delegates the pixel work to OpenCV (pinned opencv-python-headless 4.5.5.64,
requirements/cpu.txt:146), which cannot be imported in this image (not
installed, no network): the functions below restate OpenCV 4.5.5's published
8-bit algorithms step by step, citing the OpenCV source they follow —
``cv::resize`` INTER_LINEAR (modules/imgproc/src/resize.cpp: float32 source
coordinates, 11-bit fixed-point coefficients, the (b*(S>>4))>>16 vertical
pass, the exact-2x INTER_AREA shortcut) and ``cv::warpAffine`` INTER_LINEAR
(modules/imgproc/src/imgwarp.cpp: matrix inversion, 10-bit fixed-point
coordinates rounded to 1/32 pixel, ``remap`` with the 32x32 table of 15-bit
bilinear weights, constant border taken tap by tap).  Parity with a real cv2
is therefore by construction from the published source, not by a run.
Augmentations draw from Python's global ``random`` in the same order as the
reference so a seeded run makes the same decisions.
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


INTER_RESIZE_COEF_SCALE = 2048      # 1 << INTER_RESIZE_COEF_BITS (11), resize.cpp


def _src_coords(src, dst, scale):
    """resize.cpp, cv::resize (linear branch): ``fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx``
    with scale_x a double and fx a FLOAT.  scale = 1/inv_scale, inv_scale = (double)dst/src when dsize is given
    (hal::resize: ``double scale_x = 1./inv_scale_x``) or the caller's fx when dsize is empty."""
    if scale is None:
        scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return s, (f - s.astype(np.float32)).astype(np.float32)


def _weights(f):
    """``cbuf[0] = 1.f - fx; cbuf[1] = fx; ialpha = saturate_cast<short>(cbuf[k]*INTER_RESIZE_COEF_SCALE)``:
    float32 arithmetic, cvRound (round half to even)."""
    one, sc = np.float32(1.0), np.float32(INTER_RESIZE_COEF_SCALE)
    a0 = np.rint((one - f) * sc).astype(np.int64)
    a1 = np.rint(f * sc).astype(np.int64)
    return np.clip(a0, -32768, 32767), np.clip(a1, -32768, 32767)


def _coeffs_x(src, dst, scale=None):
    """Horizontal tables.  ``if (sx < 0) fx = 0, sx = 0;  if (sx >= ssize.width-1) fx = 0, sx = ssize.width-1``
    (resize.cpp; HResizeLinear then reads S[sx] and S[sx+cn], the latter with weight 0 at the right edge)."""
    s, f = _src_coords(src, dst, scale)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    a0, a1 = _weights(f)
    return s, np.minimum(s + 1, src - 1), a0, a1


def _coeffs_y(src, dst, scale=None):
    """Vertical tables: the weights stay as computed, only the two source rows are clamped
    (resizeGeneric_Invoker: ``sy = clip(sy0 - ksize2 + 1 + k, 0, ssize.height)``)."""
    s, f = _src_coords(src, dst, scale)
    a0, a1 = _weights(f)
    return np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1), a0, a1


def resize_linear_u8(img, new_w, new_h, scale_x=None, scale_y=None):
    """cv2.resize(img, (new_w, new_h), interpolation=INTER_LINEAR) for uint8 (OpenCV 4.5.5 resize.cpp).
    scale_x / scale_y: 1/fx, 1/fy when the reference calls ``cv2.resize(img, None, fx=, fy=)`` (the scale is then
    the caller's, not src/dst)."""
    h, w = img.shape[:2]
    new_w, new_h = max(int(new_w), 1), max(int(new_h), 1)
    sx = (1.0 / (float(new_w) / float(w))) if scale_x is None else float(scale_x)
    sy = (1.0 / (float(new_h) / float(h))) if scale_y is None else float(scale_y)
    if (h, w) == (new_h, new_w) and sx == 1.0 and sy == 1.0:
        return img.copy()
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[:, :, None]
    if sx == 2.0 and sy == 2.0:
        # "if (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation =
        # INTER_AREA": resizeAreaFast, 2x2 boxes, (sum + 2) >> 2
        ys, xs = 2 * np.arange(new_h), 2 * np.arange(new_w)
        y1, x1 = np.minimum(ys + 1, h - 1), np.minimum(xs + 1, w - 1)
        out = (src[ys][:, xs] + src[ys][:, x1] + src[y1][:, xs] + src[y1][:, x1] + 2) >> 2
    else:
        x0, x1, ax0, ax1 = _coeffs_x(w, new_w, sx)
        y0, y1, ay0, ay1 = _coeffs_y(h, new_h, sy)
        rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]  # HResizeLinear: x2048, int
        r0, r1 = rows[y0], rows[y1]
        # VResizeLinear<uchar,int,short>: ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
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


INTER_BITS, INTER_TAB_SIZE = 5, 32           # imgproc.hpp
AB_BITS, AB_SCALE = 10, 1024                 # imgwarp.cpp: AB_BITS = MAX(10, INTER_BITS)
INTER_REMAP_COEF_BITS = 15


def bilinear_tab_i():
    """imgwarp.cpp initInterTab2D(INTER_LINEAR, fixpt): BilinearTab_i[fy*32 + fx] = the four 15-bit weights
    saturate_cast<short>(vy*vx*32768) of taps (0,0) (0,1) (1,0) (1,1).  Every product is a multiple of 32, so the
    weights sum to 32768 exactly — except entry (0,0), whose 32768 saturates to 32767: the `isum != SCALE` repair
    then adds the missing 1 to element [1][1] (for ksize 2 its search window k1,k2 in {1,2} only finds zeros)."""
    fy, fx = np.mgrid[0:INTER_TAB_SIZE, 0:INTER_TAB_SIZE]
    t = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], -1).astype(np.int64) * 32
    t = t.reshape(INTER_TAB_SIZE * INTER_TAB_SIZE, 4)
    t[0] = (32767, 0, 0, 1)
    return t


_BILINEAR_TAB = bilinear_tab_i()


def invert_affine(m):
    """cv::warpAffine without WARP_INVERSE_MAP (imgwarp.cpp): the 2x3 forward matrix, as doubles, inverted with
    exactly these operations."""
    M = [float(v) for v in np.asarray(m, dtype=np.float64).reshape(6)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def warp_affine_u8(img, m_forward, border):
    """cv2.warpAffine(img, M, (w, h), borderValue=border): INTER_LINEAR, BORDER_CONSTANT, uint8 (OpenCV 4.5.5
    imgwarp.cpp: WarpAffineInvoker + remapBilinear<FixedPtCast<int, uchar, 15>>).
      adelta[x] = saturate_cast<int>(M[0]*x*AB_SCALE), bdelta[x] = saturate_cast<int>(M[3]*x*AB_SCALE)
      X0 = saturate_cast<int>((M[1]*y + M[2])*AB_SCALE) + round_delta, round_delta = AB_SCALE/INTER_TAB_SIZE/2 = 16
      X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS);  sx = X >> INTER_BITS;  fx = X & 31   (likewise Y)
      dst = (sum of 4 taps * BilinearTab_i[fy*32 + fx] + (1 << 14)) >> 15, a tap outside the image = borderValue
    (saturate_cast<int> of a double = cvRound: round half to even)."""
    h, w = img.shape[:2]
    c = img.shape[2]
    M = invert_affine(m_forward)
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = np.rint(M[0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(M[3] * xs * AB_SCALE).astype(np.int64)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = np.rint((M[1] * ys + M[2]) * AB_SCALE).astype(np.int64) + round_delta
    Y0 = np.rint((M[4] * ys + M[5]) * AB_SCALE).astype(np.int64) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)      # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    wt = _BILINEAR_TAB[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]   # [h, w, 4]
    cval = np.asarray([int(v) for v in border[:c]], dtype=np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        return np.where(ok[..., None], v, cval)

    acc = (tap(sy, sx) * wt[..., 0:1] + tap(sy, sx + 1) * wt[..., 1:2]
           + tap(sy + 1, sx) * wt[..., 2:3] + tap(sy + 1, sx + 1) * wt[..., 3:4])
    out = (acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8)


def rotation_matrix_2d(center, angle, scale=1.0):
    """cv2.getRotationMatrix2D (imgwarp.cpp): angle in degrees, positive = counter-clockwise."""
    a = angle * (math.pi / 180.0)
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))    # Point2f
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]])


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
        # M = np.float32([[1, 0, x], [0, 1, y]]); cv2.warpAffine(img, M, (w, h), borderValue=border)  (image.py:110-111)
        return warp_affine_u8(img, np.float32([[1, 0, x], [0, 1, y]]), border)

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
        # cv2.resize(img, None, fx=f, fy=f): dsize = saturate_cast<int>(src*f) (round half to even), and the
        # source coordinates use scale = 1/f, not src/dst (resize.cpp)
        zw, zh = int(np.rint(w * f)), int(np.rint(h * f))
        img = resize_linear_u8(img, zw, zh, 1.0 / f, 1.0 / f)
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
        # M = cv2.getRotationMatrix2D(center, angle, 1.0); cv2.warpAffine(img, M, (w, h), borderValue=border)
        return warp_affine_u8(img, rotation_matrix_2d((cx, cy), angle, 1.0), border)

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
