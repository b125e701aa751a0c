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
    """Square targets for Zoom / Rotate, the reference's augmentation set, constant / modal border; 3-channel
    pipelines may end in the ImageNet `Normalize` the reference appends to its TRAIN transform
    (sykepic/train/config.py:55-56), 1-channel pipelines may not (the host pipeline decides what that means)."""
    if num_chans not in (1, 3) or not isinstance(transform, P.Compose):
        return False
    ts = list(transform.transforms)
    if ts and isinstance(ts[-1], P.Normalize):
        if num_chans != 3:
            return False
        ts = ts[:-1]
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
    ts = [t for t in transform.transforms if not isinstance(t, (P.Resize, P.ToTensor, P.Normalize))]
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
    """`Compose` for a list of decoded grey images at once: -> uint8 [n, H, W, num_chans] on the GPU, or - when the
    pipeline ends in `Normalize` - the float32 [n, 3, H, W] tensor `ToTensor` + `Normalize` give (same float32
    operations as the host pipeline, through a 3 x 256 table computed on the host)."""

    def __init__(self, transform, device, num_chans=3):
        self.transform, self.device, self.num_chans = transform, torch.device(device), int(num_chans)
        self.so = lib.load()
        last = transform.transforms[-1]
        self.lut = None
        if isinstance(last, P.Normalize):
            # a uint8 pixel has 256 possible values: the 3 x 256 results of ToTensor + Normalize are computed HERE, on
            # the host, with the host pipeline's own float32 operations, and the GPU only looks them up - bit-equal to
            # the host pipeline (torch's GPU float division is not correctly rounded: a few ulp off)
            v = torch.arange(256, dtype=torch.float32).div_(255.0).view(1, 256)
            self.lut = ((v - last.mean.view(-1, 1)) / last.std.view(-1, 1)).to(self.device)

    def _finish(self, x):
        """uint8 [n, H, W, 3] -> what the pipeline's tail (channel count, ToTensor, Normalize) makes of it."""
        if self.num_chans == 1:
            x = x[..., :1].contiguous()
        if self.lut is None:
            return x
        idx = x.permute(0, 3, 1, 2).long()
        return torch.stack([self.lut[c][idx[:, c]] for c in range(idx.shape[1])], dim=1)

    def __call__(self, images, modes=None):
        """images: HxWx3 / HxW uint8 arrays whose channels are identical (IFCB PNGs are greyscale).
        modes: their most common pixel values when the caller has them already (the decode workers do)."""
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
            if modes is None:
                modes = [P.mode_pixel_value(g) for g in grey]
            border = np.repeat(np.asarray(modes, dtype=np.uint8)[:, None], 4, axis=1)
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
                return self._finish(x)
            ops_d = torch.frombuffer(bytearray(bytes(ops)), dtype=torch.uint8).to(dev)
            border_d = torch.from_numpy(border).to(dev)
            out = torch.empty_like(x)
            tmp = torch.empty_like(x) if n_ops > 1 else x
            lib.check(self.so.spk_augment_batch(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                                C.c_void_p(tmp.data_ptr()), n, th, tw, 3,
                                                C.c_void_p(ops_d.data_ptr()), n_ops,
                                                C.c_void_p(border_d.data_ptr()), stream))
        return self._finish(out)


_MP_CTX = None


def decode_context():
    """The multiprocessing context the decode processes are started with: `forkserver`.

    Why not the DataLoader default (`fork`): round 2 recorded one decode worker dying with SIGSEGV one epoch into its
    life ("Unexpected segmentation fault encountered in worker", one run in ~20), a fork()ed child of a process whose
    other threads were inside the HIP runtime (the batch thread forked the workers while the main thread was uploading
    the network's parameters).  The cause could NOT be established from that one record: no traceback was captured,
    and two probes of the plausible mechanisms came back clean on the GPU box (tests/archive/diagnostics/
    fork_dontfork_probe.py: 50 children forked while another thread copies pageable memory to the GPU read every page
    of the buffer; children that finalize an inherited CUDA tensor / event / stream / pinned tensor / HipNet handle
    exit 0).  What is certain is the exposure: a forked child carries the parent's HIP / RCCL state (mapped queues,
    signal pages, runtime locks, atfork handlers), and forking earlier only narrows the window - the second and
    third loader, or a data-parallel rank whose RCCL communicator already exists, still fork from a process with
    live GPU threads.  So the exposure is removed instead: with `forkserver` the workers are children of a helper
    process that was exec'ed fresh and never loads the HIP runtime - nothing of the GPU process is inherited, whenever
    the loader is built.  `start_decode_server()` brings the helper up (train.main calls it first thing, before RCCL /
    HIP exist); the workers need numpy + PIL (and torch's worker loop), all pre-imported in the helper so a worker
    starts in milliseconds.  Should a worker still die, `pngio.worker_init` has armed faulthandler (a Python
    traceback on stderr) and the fallback warning carries every dead worker's exit code."""
    global _MP_CTX
    if _MP_CTX is None:
        import multiprocessing as mp
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(["torch", "torch.utils.data", "sykepic_hip.pngio"])
        _MP_CTX = ctx
    return _MP_CTX


def start_decode_server():
    decode_context()
    from multiprocessing import forkserver
    forkserver.ensure_running()


class _EpochBatches:
    """Batch sampler: a fresh order per epoch - the sampler's, a torch.randperm, or file order - cut into batches."""

    def __init__(self, n, batch_size, shuffle, sampler):
        self.n, self.batch_size, self.shuffle, self.sampler = n, batch_size, shuffle, sampler

    def order(self):
        if self.sampler is not None:
            return list(iter(self.sampler))
        if self.shuffle:
            return torch.randperm(self.n).tolist()
        return list(range(self.n))

    def prepare(self):
        """Draw the next epoch's order.  The loader calls this once per epoch BEFORE it creates the DataLoader iterator:
        that constructor asks the batch sampler twice (persistent workers) and takes a seed from torch's global
        generator in between, which would shift the permutation relative to the single-threaded loader."""
        order = self.order()
        self.pending = [order[b:b + self.batch_size] for b in range(0, len(order), self.batch_size)]

    def __iter__(self):
        if getattr(self, "pending", None) is None:
            self.prepare()
        return iter(self.pending)

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None and hasattr(self.sampler, "__len__") else self.n
        return (n + self.batch_size - 1) // self.batch_size


class GpuLoader:
    """Drop-in for the train/val `DataLoader` of `ModelData.set_data_loaders`: PNG decode on the host, the whole
    transform on the GPU; yields (uint8 [B, H, W, 3] cuda tensor, int64 labels) — `HipNet` takes that layout.

    Decoding is the only per-image host work left, so it is what has to keep up with the training step
    (ResNet-50: ~9.4 k img/s per GPU).  Round 2, measured with `tools/e2e_train_rate.py` (PNG files -> ResNet-50 step,
    batch 256, a 16-core share of the host): decoding on the training thread fed 3.5 k img/s; a pool of decode THREADS
    with prefetch 4.6-5.1 k (PIL and the per-image Python around it serialise on the GIL).  So, like the reference's
    `DataLoader(num_workers=...)` (sykepic/train/data.py:151-160), the images are decoded by `workers` PROCESSES
    (torch's DataLoader machinery: fork at construction, persistent, two batches prefetched per worker), which also take the border
    value (pixel histogram) and the colour check of each image; one thread assembles the batches - concatenation,
    the augmentation draws, the host-to-device copies and the preprocessing / augmentation kernels - up to two
    batches ahead of the training loop, so that work overlaps the training step as well.  The random draws stay in
    ONE thread, in batch order (`draw_ops`), so a seeded run is reproducible whatever the worker count; the workers
    draw nothing."""

    def __init__(self, paths, labels, transform, batch_size, device, shuffle=False, sampler=None, workers=None,
                 prefetch=2, num_chans=3, persistent=True):
        import os
        self.paths, self.labels = list(paths), list(labels)
        self.batch_size, self.shuffle, self.sampler = int(batch_size), shuffle, sampler
        self.dataset = self.paths  # len(loader.dataset) is used for the [STAT] lines
        if workers is None:
            try:
                workers = len(os.sched_getaffinity(0))
            except AttributeError:  # pragma: no cover
                workers = os.cpu_count() or 1
            workers = max(1, min(8, workers))     # 8 decode processes feed ~26 k img/s
        if not persistent:
            workers = min(int(workers), 4)        # validation / test loaders: a small pool that lives for one pass
        self.persistent = bool(persistent)
        self.workers, self.prefetch = int(workers), max(1, int(prefetch))
        self._need_mode = transform.border == "mode"
        self._batches_of = _EpochBatches(len(self.paths), self.batch_size, shuffle, sampler)
        self._dl = None
        self._start_workers()
        self.pipe = GpuTransform(transform, device, num_chans)

    def _start_workers(self):
        """The decode processes (`decode_context`: children of the fork server, not of this process).  A persistent
        pool (the train loader) is brought up now with an empty epoch, so the first training step does not wait for
        it; torch's generator is put back so the epoch orders stay those of the single-threaded loader.  A
        non-persistent pool (validation / test loaders) starts when a pass begins and exits when it ends: three loaders
        no longer hold three idle pools per rank."""
        if self.workers <= 1 or not self.paths:
            return
        from torch.utils.data import DataLoader
        # batch_sampler yields index lists; the "batch" a worker returns is [(index, decoded image)]
        self._dl = DataLoader(pngio.IndexedPng(self.paths, self._need_mode), batch_sampler=self._batches_of,
                              num_workers=self.workers, collate_fn=pngio.identity, persistent_workers=self.persistent,
                              prefetch_factor=self.prefetch, multiprocessing_context=decode_context(),
                              worker_init_fn=pngio.worker_init)
        if not self.persistent:
            return
        self._batches_of.pending = []
        rng_state = torch.get_rng_state()
        # torch installs a SIGCHLD handler (main thread only) that raises "DataLoader worker ... is killed" at an arbitrary
        # point of the MAIN thread when a worker dies - in the middle of a training step, say.  The thread that iterates
        # the DataLoader notices a dead worker by itself (`_batches` reports its exit status), so the previous handler
        # is put back.
        import signal
        import threading
        in_main = threading.current_thread() is threading.main_thread()
        prev = signal.getsignal(signal.SIGCHLD) if in_main else None
        for _ in iter(self._dl):
            pass
        if in_main:
            signal.signal(signal.SIGCHLD, prev if prev is not None else signal.SIG_DFL)
        torch.set_rng_state(rng_state)
        self._batches_of.pending = None

    def __len__(self):
        return len(self._batches_of)

    def _to_batch(self, idx, decoded):
        imgs = [d[0] for d in decoded]
        if any(d[1] is None for d in decoded):
            # a colour PNG in the batch: the host pipeline handles it
            rgb = [im if im.ndim == 3 else np.repeat(im[:, :, None], 3, axis=2) for im in imgs]
            x = torch.stack([self.pipe.transform(im) for im in rgb])
        else:
            x = self.pipe(imgs, [d[1] for d in decoded])
        return x, torch.tensor([int(self.labels[i]) for i in idx], dtype=torch.int64)

    def _batches(self):
        if self.workers <= 1:
            self._batches_of.prepare()
            for idx in self._batches_of:
                yield self._to_batch(idx, [pngio.decode_png(self.paths[i], self._need_mode) for i in idx])
            return
        if self._dl is None:     # (no images at construction)
            return
        self._batches_of.prepare()
        rng_state = torch.get_rng_state()
        done = 0
        it = None
        try:
            it = iter(self._dl)                 # draws a worker seed nobody uses: the decode workers are deterministic
            torch.set_rng_state(rng_state)      # ... so the global generator stays where the single-threaded loader leaves it
            for items in it:
                yield self._to_batch([i for i, _ in items], [d for _, d in items])
                done += 1
        except RuntimeError as e:
            if "DataLoader worker" not in str(e):
                raise
            torch.set_rng_state(rng_state) if done == 0 else None
            # a decode process died (killed from outside, out of memory ...): finish the epoch - and the run - decoding in
            # this thread rather than losing the training run; the batches and their order do not change
            import warnings
            status = []
            try:   # exit code of each decode process: -11 = SIGSEGV, -9 = killed (out of memory, operator) ...
                status = [f"pid {w.pid}: exitcode {w.exitcode}" for w in it._workers if not w.is_alive()]
            except Exception:   # pragma: no cover
                pass
            warnings.warn(f"sykepic_hip: {e} ({'; '.join(status) or 'no exit status'}); decoding in-process from here on")
            self._dl, self.workers = None, 1
            for idx in self._batches_of.pending[done:]:
                yield self._to_batch(idx, [pngio.decode_png(self.paths[i], self._need_mode) for i in idx])

    def __iter__(self):
        """Batches assembled by a background thread, at most two ahead of the consumer."""
        import queue
        import threading
        if self.workers <= 1:
            yield from self._batches()
            return
        q = queue.Queue(maxsize=2)
        end, stop = object(), threading.Event()

        def produce():
            try:
                for item in self._batches():
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(end)
            except BaseException as e:   # surfaces in the consumer
                q.put(e)

        t = threading.Thread(target=produce, name="sykepic-batch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is end:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
