"""Training input pipeline on the GPU: resize + border (`spk_preprocess_rois`) and
the random augmentations (`spk_augment_batch`) of the reference's train transform
(sykepic/train/config.py:25-60 -> sykepic/train/image.py:25-56,80-180), applied to a whole batch in device
memory instead of per image with cv2 in DataLoader worker processes.

The random numbers are still drawn on the host from Python's `random`, per image and in the order the host
pipeline (`preprocess.Compose`) draws them, so a seeded run augments byte-for-byte like the host pipeline
(tests compare the two).  Only PNG decoding stays on the CPU."""

import ctypes as C
import math
import random

import numpy as np
import torch

from . import lib, pngio, preprocess as P

_KINDS = (P.Resize, P.FlipHorizontal, P.FlipVertical, P.Translate, P.Zoom, P.Rotate, P.ChangeBrightness, P.ToTensor)


def supported(transform, num_chans):
    """Square 3-channel targets, the reference's augmentation set, constant / modal border, no Normalize."""
    if num_chans != 3 or not isinstance(transform, P.Compose):
        return False
    ts = transform.transforms
    if not ts or not isinstance(ts[0], P.Resize) or not isinstance(ts[-1], P.ToTensor):
        return False
    if any(not isinstance(t, _KINDS) for t in ts) or sum(isinstance(t, P.Resize) for t in ts) != 1:
        return False
    th, tw = transform.target_dims
    if any(isinstance(t, (P.Zoom, P.Rotate)) for t in ts) and th != tw:
        return False
    return transform.border in ("mode", (0, 0, 0), (255, 255, 255)) and (th * tw * 3) % 4 == 0


def draw_ops(transform, dims):
    """Per-sample parameters of every augmentation, drawn exactly as `Compose.__call__` would while processing
    the images one after the other.  dims: [(h, w)] of the decoded images.  -> AugOp array [n_ops][n]."""
    ts = [t for t in transform.transforms if not isinstance(t, (P.Resize, P.ToTensor))]
    n = len(dims)
    ops = (lib.AugOp * (len(ts) * n))()
    th, tw = transform.target_dims
    for i, (h, w) in enumerate(dims):
        new_h, new_w = P.get_new_dims(h, w, th, tw)
        for j, t in enumerate(ts):
            op = ops[j * n + i]
            if isinstance(t, P.FlipHorizontal):
                op.kind, op.i0 = lib.AUG_FLIP_H, random.getrandbits(1)
            elif isinstance(t, P.FlipVertical):
                op.kind, op.i0 = lib.AUG_FLIP_V, random.getrandbits(1)
            elif isinstance(t, P.Translate):
                op.kind = lib.AUG_TRANSLATE
                if h > w:   # only along the padded axis (image.py:39-44)
                    limit = int((tw - new_w) / 2.5)
                    op.i0, op.i1 = random.randint(-limit, limit), 0
                else:
                    limit = int((th - new_h) / 2.5)
                    op.i0, op.i1 = 0, random.randint(-limit, limit)
            elif isinstance(t, P.Zoom):
                f = round(random.uniform(*t.zoom_range), 2)
                op.kind, op.i0 = lib.AUG_ZOOM, int(np.rint(tw * f))
                op.d[0] = f
            elif isinstance(t, P.Rotate):
                angle = random.randint(-t.max_angle, t.max_angle)
                minv = P.invert_affine(P.rotation_matrix_2d((tw // 2, th // 2), angle, 1.0))
                op.kind = lib.AUG_ROTATE
                for k in range(6):
                    op.d[k] = minv[k]
            elif isinstance(t, P.ChangeBrightness):
                op.kind = lib.AUG_BRIGHT
                op.d[0] = random.uniform(*t.brightness_range)
    return ops, len(ts)


class GpuTransform:
    """`Compose` for a list of decoded grey images at once: -> uint8 [n, H, W, 3] on the GPU."""

    def __init__(self, transform, device):
        self.transform, self.device = transform, torch.device(device)
        self.so = lib.load()

    def __call__(self, images):
        """images: HxWx3 / HxW uint8 arrays whose channels are identical (IFCB PNGs are greyscale)."""
        t = self.transform
        th, tw = t.target_dims
        n = len(images)
        grey = [im[..., 0] if im.ndim == 3 else im for im in images]
        dims = [g.shape for g in grey]
        rois = np.zeros(n, dtype=np.dtype([("offset", "<i8"), ("width", "<i4"), ("height", "<i4")]))
        off = 0
        for i, g in enumerate(grey):
            rois[i] = (off, g.shape[1], g.shape[0])
            off += g.size
        blob = np.concatenate([np.ascontiguousarray(g).reshape(-1) for g in grey]) if n else np.zeros(1, np.uint8)
        if t.border == "mode":
            border = np.array([[P.mode_pixel_value(g)] * 4 for g in grey], dtype=np.uint8)
            code = -1
        else:
            border = np.full((n, 4), t.border[0], dtype=np.uint8)
            code = int(t.border[0])
        ops, n_ops = draw_ops(t, dims)
        dev = self.device
        blob_d = torch.from_numpy(blob).to(dev)
        rois_d = torch.from_numpy(rois.view(np.uint8).copy()).to(dev)
        x = torch.empty((n, th, tw, 3), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            lib.check(self.so.spk_preprocess_rois(C.c_void_p(blob_d.data_ptr()), int(blob.size),
                                                  C.c_void_p(rois_d.data_ptr()), n, th, tw, code,
                                                  C.c_void_p(x.data_ptr()), stream))
            if n_ops == 0:
                return x
            ops_d = torch.frombuffer(bytearray(bytes(ops)), dtype=torch.uint8).to(dev)
            border_d = torch.from_numpy(border).to(dev)
            out = torch.empty_like(x)
            tmp = torch.empty_like(x) if n_ops > 1 else x
            lib.check(self.so.spk_augment_batch(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                                C.c_void_p(tmp.data_ptr()), n, th, tw, 3,
                                                C.c_void_p(ops_d.data_ptr()), n_ops,
                                                C.c_void_p(border_d.data_ptr()), stream))
        return out


class GpuLoader:
    """Drop-in for the train/val `DataLoader` of `ModelData.set_data_loaders`: PNG decode on the host, the whole
    transform on the GPU; yields (uint8 [B, H, W, 3] cuda tensor, int64 labels) — `HipNet` takes that layout.

    Decoding is the only per-image host work left, so it is what has to keep up with the training step
    (ResNet-50: ~9.4 k img/s per GPU): PNGs are decoded by a pool of `workers` threads (zlib inflate runs outside
    the GIL) and `prefetch` batches are kept in flight ahead of the consumer, so the GPU step of batch k overlaps
    the decode of batches k+1 ... k+prefetch.  The random draws of the augmentations stay in the consuming thread,
    in batch order (`draw_ops`), so a seeded run is reproducible whatever the worker count."""

    def __init__(self, paths, labels, transform, batch_size, device, shuffle=False, sampler=None, workers=None,
                 prefetch=3):
        import os
        self.paths, self.labels = list(paths), list(labels)
        self.batch_size, self.shuffle, self.sampler = int(batch_size), shuffle, sampler
        self.pipe = GpuTransform(transform, device)
        self.dataset = self.paths  # len(loader.dataset) is used for the [STAT] lines
        if workers is None:
            try:
                workers = len(os.sched_getaffinity(0))
            except AttributeError:  # pragma: no cover
                workers = os.cpu_count() or 1
            workers = max(1, min(16, workers))
        self.workers, self.prefetch = int(workers), max(1, int(prefetch))

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None and hasattr(self.sampler, "__len__") else len(self.paths)
        return (n + self.batch_size - 1) // self.batch_size

    def _to_batch(self, idx, imgs):
        if any(im.ndim == 3 and not (np.array_equal(im[..., 0], im[..., 1]) and np.array_equal(im[..., 0], im[..., 2]))
               for im in imgs):
            # a colour PNG: the host pipeline handles it
            x = torch.stack([self.pipe.transform(im) for im in imgs])
        else:
            x = self.pipe(imgs)
        return x, torch.tensor([int(self.labels[i]) for i in idx], dtype=torch.int64)

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        if self.sampler is not None:
            order = list(iter(self.sampler))
        elif self.shuffle:
            order = torch.randperm(len(self.paths)).tolist()
        else:
            order = list(range(len(self.paths)))
        batches = [order[b:b + self.batch_size] for b in range(0, len(order), self.batch_size)]
        if self.workers <= 1:
            for idx in batches:
                yield self._to_batch(idx, [pngio.read_image(self.paths[i], 3) for i in idx])
            return
        pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="sykepic-png")
        try:
            inflight = deque()
            nxt = 0
            while nxt < len(batches) and len(inflight) < self.prefetch:
                inflight.append((batches[nxt], [pool.submit(pngio.read_image, self.paths[i], 3) for i in batches[nxt]]))
                nxt += 1
            while inflight:
                idx, futs = inflight.popleft()
                imgs = [f.result() for f in futs]
                if nxt < len(batches):
                    inflight.append((batches[nxt], [pool.submit(pngio.read_image, self.paths[i], 3) for i in batches[nxt]]))
                    nxt += 1
                yield self._to_batch(idx, imgs)
        finally:
            pool.shutdown(wait=False, cancel_futures=True)
