"""Deterministic, library-independent synthetic tensors.

Every value is a pure function of (seed, flat index) through a 64-bit
counter hash (splitmix64 finaliser), so the container that makes the golden
fixtures and the GPU box build bit-identical inputs/weights without shipping
large files and without depending on any RNG implementation.

Shapes follow SURVEY.md §8(d): images take the value set ``k/255`` that
``ToTensor`` can produce (reference ``sykepic/train/config.py:52``), labels
are uniform class ids, weights use Kaiming-uniform-like ranges.
"""

import zlib

import numpy as np

_M64 = (1 << 64) - 1
_GOLD = 0x9E3779B97F4A7C15


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_u32(n, seed):
    """uint32[n]; element i depends only on (seed, i)."""
    with np.errstate(over="ignore"):
        base = np.uint64((int(seed) * _GOLD + 0x1234567) & _M64)
        i = np.arange(int(n), dtype=np.uint64)
        z = _mix(i * np.uint64(_GOLD) + base)
    return (z >> np.uint64(32)).astype(np.uint32)


def uniform(shape, seed, lo=0.0, hi=1.0):
    """float32 uniform in [lo, hi) with 24-bit resolution."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (hash_u32(n, seed) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.float32(lo) + u * np.float32(hi - lo)).reshape(shape)


def key_seed(key, seed):
    return (zlib.crc32(key.encode()) * 2654435761 + int(seed) * 97) & 0x7FFFFFFF


def synth_images(n, c, h, w, seed=0):
    """[n,c,h,w] float32 with values k/255 (k in 0..255), the value set
    ``ToTensor`` emits.  Each image is a graded background with a few
    per-channel rectangles plus +-16 hash noise, built with integer
    arithmetic only so it is bit-reproducible on any host; images differ
    enough from one another that a random-weight CNN separates them."""
    noise = (hash_u32(n * c * h * w, seed) >> np.uint32(27)).astype(np.int32)
    noise = noise.reshape(n, c, h, w) - 16
    img = np.empty((n, c, h, w), np.int32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int32)
    for i in range(n):
        pr = hash_u32(64, int(seed) * 7919 + 1000003 * (i + 1)).astype(np.int64)
        bg = 110 + int(pr[0] % 120)
        sx = int(pr[1] % 65) - 32
        sy = int(pr[2] % 65) - 32
        base = bg + (xx * sx) // w + (yy * sy) // h
        for ch in range(c):
            a = base.copy()
            for b in range(5):
                q = pr[4 + ((ch % 3) * 5 + b) * 4: 8 + ((ch % 3) * 5 + b) * 4]
                bw = 4 + int(q[0] % max(1, w // 2))
                bh = 4 + int(q[1] % max(1, h // 2))
                x0 = int(q[2] % w)
                y0 = int((q[2] >> 12) % h)
                a[y0:y0 + bh, x0:x0 + bw] += int(q[3] % 241) - 120
            img[i, ch] = a
    img = np.clip(img + noise, 0, 255).astype(np.float32)
    return img / np.float32(255.0)


def synth_labels(n, num_classes, seed=1):
    return (hash_u32(n, seed) % np.uint32(num_classes)).astype(np.int64)


def synth_state_dict(param_specs, seed=2, logit_gain=60.0):
    """param_specs: iterable of (key, shape, kind) as produced by
    ``arch.param_specs``. Returns an ordered {key: np.ndarray}.

    kinds: conv_w, bn_w, bn_w_last (last BN of a residual block: small gamma
    so the residual stream keeps O(1) variance through 16 blocks), bn_b,
    bn_mean, bn_var, bn_nbt, fc_w, fc_w_last, fc_b, se_w, se_b.
    """
    out = {}
    for key, shape, kind in param_specs:
        s = key_seed(key, seed)
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            b = float(np.sqrt(6.0 / fan_in))
            v = uniform(shape, s, -b, b)
        elif kind == "bn_w":
            v = uniform(shape, s, 0.9, 1.1)
        elif kind == "bn_w_last":
            v = uniform(shape, s, 0.2, 0.3)
        elif kind in ("bn_b", "bn_mean"):
            v = uniform(shape, s, -0.1, 0.1)
        elif kind == "bn_var":
            v = uniform(shape, s, 0.9, 1.1)
        elif kind == "bn_nbt":
            v = np.zeros(shape, dtype=np.int64)
        elif kind in ("fc_w", "fc_w_last"):
            b = float(1.0 / np.sqrt(shape[1]))
            if kind == "fc_w_last":
                b *= logit_gain
            v = uniform(shape, s, -b, b)
        elif kind == "fc_b":
            v = uniform(shape, s, -0.05, 0.05)
        elif kind == "se_w":   # squeeze-excitation 1x1 convs [out, in, 1, 1]
            b = float(np.sqrt(6.0 / shape[1]))
            v = uniform(shape, s, -b, b)
        elif kind == "se_b":
            v = uniform(shape, s, -0.5, 0.5)
        else:
            raise ValueError(f"unknown param kind {kind!r}")
        out[key] = v
    return out
