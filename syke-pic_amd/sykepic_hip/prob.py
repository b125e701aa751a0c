"""`sykepic prob`: class probabilities for raw IFCB samples on MI355X.

Host-side mirror of the reference's inference workflow
(``/root/reference/sykepic/compute/probability.py``): same entry points
(``call`` :27, ``main`` :67, ``prepare_model`` :118, ``process_sample`` :133,
``process_images`` :165, ``net_pass`` :180, ``probabilities_to_csv`` :200),
same arguments, same ``<out>/YYYY/MM/DD/<sample>.prob.csv`` files.  What
differs is below the seam: the forward and the base-1.3 softmax run fused in
``libsykepic_hip.so``; ROIs go from the ``.roi`` blob straight to tensors (no
PNG round trip beside the raw data, quirk Q8); no weights are downloaded
before ``best_state.pth`` is loaded (quirk Q7).
"""

import logging
import os
from collections import namedtuple
from configparser import ConfigParser
from pathlib import Path

import numpy as np
import torch

SOFTMAX_EXP = 1.3
FILE_SUFFIX = ".prob"
log = logging.getLogger("prob")

EvalParams = namedtuple(
    "EvalParams", ["batch_size", "num_workers", "classes", "img_shape", "transform", "device"])


def roi_number(path):
    """``…_00002.png`` -> 2 (reference probability.py:190: ``int(Path(path).stem.split("_")[-1])``), without building
    a Path per ROI (20 k ROIs per sample: pathlib alone was 10 % of the sample loop)."""
    name = str(path)
    cut = max(name.rfind("/"), name.rfind("\\")) + 1
    name = name[cut:]
    dot = name.rfind(".")
    if dot > 0:            # Path.stem: a leading dot is not a suffix separator
        name = name[:dot]
    return int(name[name.rfind("_") + 1:])


def _in_process_group():
    import torch.distributed as td
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def net_pass(net, dataloader, device="cuda:0"):
    """[(roi, [p_class...]), ...] sorted by ROI number.

    ``dataloader`` yields ``(x [B,C,H,W] float32 in [0,1], paths)``.  The
    probabilities are softmax(logits * ln 1.3), computed on the GPU."""
    results = []
    net.to(device)
    net.eval()
    pending = []
    for batch in dataloader:
        x, paths = batch[0], batch[1]
        if getattr(net, "_auto_calibration_dir", None) is not None and not _in_process_group():
            auto_calibrate(net, x)
        probs = net.probabilities(x, SOFTMAX_EXP)  # async on the stream
        pending.append((tuple(roi_number(p) for p in paths), probs))
    for rois, probs in pending:  # one device->host copy per batch, after all launches
        results.extend(zip(rois, probs.tolist()))
    return sorted(results)


class ProbRows:
    """The rows of `net_pass` as two arrays (ROI numbers int64 [n], probabilities float32 [n, classes]) instead of
    n Python tuples of 50 Python floats: what the per-sample loop of `sykepic prob` carries between the GPU and the
    CSV file.  Iterating yields the reference's `(roi, [p, ...])` rows."""

    def __init__(self, numbers, probs):
        self.numbers = np.asarray(numbers, dtype=np.int64).reshape(-1)
        self.probs = np.asarray(probs, dtype=np.float32).reshape(self.numbers.size, -1)

    def __len__(self):
        return int(self.numbers.size)

    def __iter__(self):
        return iter(zip(self.numbers.tolist(), self.probs.tolist()))

    def sorted(self):
        order = np.argsort(self.numbers, kind="stable")
        return ProbRows(self.numbers[order], self.probs[order])

    @staticmethod
    def concat(parts):
        parts = [p for p in parts if len(p)]
        if not parts:
            return ProbRows(np.zeros(0, np.int64), np.zeros((0, 0), np.float32))
        return ProbRows(np.concatenate([p.numbers for p in parts]), np.concatenate([p.probs for p in parts]))


class PendingRows:
    """Rows whose probabilities are still being computed: ROI numbers on the host, the probabilities on their way
    into one pinned host buffer (queued behind the sample's last batch, before anything of the next sample).
    `result()` is the only point that waits, and only for that copy's event - callable from another thread."""

    def __init__(self, nums, outs):
        self.nums, self.host, self.event = nums, None, None
        if outs:
            dev = torch.cat(outs)
            self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
            self.host.copy_(dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def result(self):
        if self.host is None:
            return ProbRows(np.zeros(0, np.int64), np.zeros((0, 0), np.float32))
        self.event.synchronize()
        return ProbRows(np.concatenate(self.nums), self.host.numpy()).sorted()


def net_pass_launch(net, batches, device="cuda:0"):
    """`net_pass` for batches of (x, ROI numbers), launch half: every batch is queued on the stream, nothing is
    read back and no per-ROI Python object is built."""
    net.to(device)
    net.eval()
    nums, outs = [], []
    for x, numbers in batches:
        if getattr(net, "_auto_calibration_dir", None) is not None and not _in_process_group():
            auto_calibrate(net, x)
        outs.append(net.probabilities(x, SOFTMAX_EXP))      # async on the stream
        nums.append(np.asarray(numbers, dtype=np.int64))
    return PendingRows(nums, outs)


def net_pass_arrays(net, batches, device="cuda:0"):
    return net_pass_launch(net, batches, device).result()


def _format_rows(rows, n_classes):
    """`"%d," + ",".join(["%.5f"] * n) % row` for every row, as bytes, with vectorised integer arithmetic: the value
    rounded to 5 decimals is rint(p * 1e5); `%.5f` rounds the exact binary value, p * 1e5 carries one more rounding
    (2^-53 relative), so entries within 1e-6 of a half are redone with `%` itself.  None: not representable this way
    (a value outside [0, 10), a ragged row) - the caller formats row by row."""
    p = rows.probs.astype(np.float64)
    if p.ndim != 2 or p.shape[1] != n_classes or p.size == 0:
        return None
    if not np.isfinite(p).all() or p.min() < 0.0 or p.max() >= 9.999994:
        return None
    s = p * 1e5
    v = np.rint(s).astype(np.int64)
    for i, j in np.argwhere(np.abs(s - np.floor(s) - 0.5) < 1e-6):
        v[i, j] = int(("%.5f" % p[i, j]).replace(".", ""))
    buf = np.empty(p.shape + (8,), dtype=np.uint8)
    buf[..., 0] = 48 + v // 100000
    buf[..., 1] = 46
    r = v % 100000
    for k, d in enumerate((10000, 1000, 100, 10, 1)):
        buf[..., 2 + k] = 48 + (r // d) % 10
    buf[..., 7] = 44
    buf[:, -1, 7] = 10
    body = buf.reshape(p.shape[0], -1)
    out = []
    for num, row in zip(rows.numbers.tolist(), body):
        out.append(b"%d," % num)
        out.append(row.tobytes())
    return b"".join(out)


def probabilities_to_csv(probabilities, classes, csv_path):
    csv_path = Path(csv_path)
    csv_path.parent.mkdir(parents=True, exist_ok=True)
    if isinstance(probabilities, ProbRows):
        body = _format_rows(probabilities, len(classes))
        if body is not None:
            csv_path.write_bytes(("roi," + ",".join(classes) + "\n").encode() + body)
            return
    lines = ["roi," + ",".join(classes)]
    row_fmt = "%d," + ",".join(["%.5f"] * len(classes))      # one C-level format per row, not one per value
    for roi, probs in probabilities:
        lines.append(row_fmt % (roi, *probs) if len(probs) == len(classes)
                     else f"{roi}," + ",".join("%.5f" % p for p in probs))
    csv_path.write_text("\n".join(lines) + "\n")


def prepare_model(model_dir, device=None):
    """model directory -> (net, classes, img_shape, eval_transform, device)."""
    from .config import get_img_shape, get_network, get_transforms
    model_dir = Path(model_dir)
    classes = (model_dir / "class_names.txt").read_text().splitlines()
    config = ConfigParser()
    config.read(model_dir / "config.ini")
    img_shape = get_img_shape(config)
    _, eval_transform = get_transforms(config, img_shape)
    device = torch.device(device or "cuda:0")
    net = get_network(config, len(classes), device=device, pretrained_ok=False)
    net.load_state_dict(torch.load(model_dir / "best_state.pth", map_location="cpu"))
    if not use_act_means(net, model_dir):
        arm_auto_calibration(net, model_dir)
    return net, classes, img_shape, eval_transform, device


ACT_MEANS_FILE = "act_means.pth"   # next to best_state.pth: written by `sykepic train` / `sykepic calibrate` / the first `prob` run
AUTO_CALIBRATION_IMAGES = 256
AUTO_CALIBRATION_MIN = 32          # fewer images than this say too little about the channel means: wait for a larger batch


def state_digest(model_dir):
    """sha256 of `best_state.pth` as it lies on disk: ties an `act_means.pth` to the weights it was measured with."""
    import hashlib
    h = hashlib.sha256()
    with open(Path(model_dir) / "best_state.pth", "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def use_act_means(net, model_dir):
    """A model directory that carries activation means OF ITS OWN WEIGHTS runs the calibrated single-pass mode
    (csrc/zero_sum.hip: every conv one fp16 product, as accurate as hi + lo weights).  The sidecar stores the sha256 of
    the `best_state.pth` it was measured with; a file whose digest differs (re-trained or copied weights), one without a
    digest (written before round 5) or one of the wrong size is ignored - rounding against another model's means would
    silently cost the accuracy the mode exists for.  SYKEPIC_CALIBRATED=0 ignores the file."""
    path = Path(model_dir) / ACT_MEANS_FILE
    if os.environ.get("SYKEPIC_CALIBRATED", "1") == "0" or not path.is_file():
        return False
    try:
        means = torch.load(path, map_location="cpu")
        if not isinstance(means, dict) or means.get("state_sha256") != state_digest(model_dir):
            raise ValueError("it was not measured with this best_state.pth")
        net.set_act_means(means["means"])
        net.set_precision("calibrated")
    except Exception as e:  # noqa: BLE001 - a stale or foreign file must not stop classification
        log.warning(f"{path.name} ignored ({e})")
        net.set_act_means(None)
        net.set_precision(split_weights=3)
        return False
    log.info(f"activation means from {path.name}: calibrated single-pass mode")
    return True


def save_act_means(net, model_dir, n_images):
    torch.save({"means": net.act_means(), "images": int(n_images), "network": net.graph.network,
                "state_sha256": state_digest(model_dir)}, Path(model_dir) / ACT_MEANS_FILE)


def drop_act_means(model_dir):
    """Removes a sidecar that can no longer belong to the weights (new training run into the directory)."""
    try:
        (Path(model_dir) / ACT_MEANS_FILE).unlink()
    except FileNotFoundError:
        pass


def arm_auto_calibration(net, model_dir):
    """A model directory as the REFERENCE leaves it (`best_state.pth` only, probability.py:118-130) has no activation
    means.  ResNets then measure them on the first <= 256 ROIs they are about to classify (`auto_calibrate`, called by
    `net_pass` / `net_pass_launch` in front of the first forward), switch to the calibrated single-pass mode for
    everything from that batch on, and leave the sidecar in the directory when it is writable, so that every later run
    starts calibrated and classifies with the same rounded weights.  EfficientNets gain nothing from the mode (their
    error is activation storage: DESIGN.md section 3) and keep the default.  SYKEPIC_AUTO_CALIBRATE=0 switches it off."""
    if os.environ.get("SYKEPIC_AUTO_CALIBRATE", "1") == "0" or os.environ.get("SYKEPIC_CALIBRATED", "1") == "0":
        return
    if not net.graph.network.startswith("resnet"):
        return
    net._auto_calibration_dir = Path(model_dir)


def auto_calibrate(net, x, dist=None):
    """First batch of a `prob` run on an un-calibrated model directory: means from its first <= 256 images (the
    calibration forward: every conv hi + lo, one stream), then the calibrated mode.  `x` None: this rank has no image.
    Single process: called by `net_pass` / `net_pass_launch` in front of the first forward.  Under torch.distributed
    (`dist` given, world > 1) it is a collective that `launch_sample` / `process_images` call on EVERY rank before their
    forward passes, whatever the size of the rank's shard: rank 0 measures on ITS first images and every rank takes
    rank 0's means - the replicas stay one model; if rank 0 has no image the step is retried with the next sample."""
    model_dir = getattr(net, "_auto_calibration_dir", None)
    if model_dir is None:
        return False
    from . import dp
    rank, world = dp.rank_world(dist)
    if world == 1 and (x is None or len(x) < AUTO_CALIBRATION_MIN):
        return False
    n = min(len(x), AUTO_CALIBRATION_IMAGES) if x is not None else 0
    means = None
    if rank == 0 and n >= AUTO_CALIBRATION_MIN:
        try:
            net.calibrate(x[:n])
            means = net.act_means()
        except Exception as e:  # noqa: BLE001 - an optimisation; the other ranks wait in the broadcast below
            if world == 1:
                raise
            log.warning(f"activation means not measured ({e}); default precision mode")
    if world > 1:
        means, n = dp.broadcast_object((means, n), dist)
        if means is not None and rank != 0:
            net.set_act_means(means)
    if means is None:
        return False
    net._auto_calibration_dir = None
    net.set_precision("calibrated")
    log.info(f"no {ACT_MEANS_FILE} in {model_dir}: activation means measured on the first {n} images, calibrated "
             f"single-pass mode from here on")
    if rank == 0:
        try:
            save_act_means(net, model_dir, n)
            log.info(f"{ACT_MEANS_FILE} written: later runs start calibrated")
        except OSError as e:
            log.info(f"{ACT_MEANS_FILE} not written ({e}): the next run measures again")
    return True


def calibrate_model(net, batches, max_images=2048):
    """Accumulates the activation means over `batches` (tensors as `net.forward` takes them, or (x, ...) tuples) until
    `max_images` have been seen.  Returns the number of images used."""
    seen = 0
    for batch in batches:
        x = batch[0] if isinstance(batch, (tuple, list)) else batch
        if seen + len(x) > max_images:
            x = x[:max_images - seen]
        if len(x) == 0:
            break
        net.calibrate(x, reset=seen == 0)
        seen += len(x)
        if seen >= max_images:
            break
    return seen


def _batches(items, transform, batch_size):
    """(x, paths) batches from [(name, HxW or HxWx3 uint8 array)]."""
    for i in range(0, len(items), batch_size):
        chunk = items[i:i + batch_size]
        x = torch.stack([transform(img) for _, img in chunk])
        yield x, [name for name, _ in chunk]


def _skip_existing(csv_path, force, dist):
    """The reference's exists/force rule (probability.py:136-141), decided once by rank 0 for every rank."""
    from . import dp
    rank, _ = dp.rank_world(dist)
    skip = False
    if rank == 0 and csv_path.is_file():
        if force:
            log.warning(f"{csv_path.name} already exists, overwriting")
        else:
            log.warning(f"{csv_path.name} already exists, skipping")
            skip = True
    return dp.broadcast_object(skip, dist)


def _finish(rows, error, what, classes, csv_path, dist):
    """Rows of this rank's shard -> the sample's CSV.  Ranks leave together: if any of them failed on this sample
    all of them raise (the caller logs and goes on to the next sample, as the reference does), so no rank is left
    waiting in the gather."""
    from . import dp
    rank, _ = dp.rank_world(dist)
    if not dp.all_ok(error is None, dist):
        raise error if error is not None else RuntimeError(f"another rank failed on {what}")
    if isinstance(rows, ProbRows):
        parts = dp.gather_parts(rows, dist, dst=0)
        probabilities = ProbRows.concat(parts).sorted() if parts is not None else None
    else:
        probabilities = dp.gather_rows(rows, dist, dst=0)
    if rank == 0:
        probabilities_to_csv(probabilities, classes, csv_path)


def read_sample_ahead(sample_path, params, out_dir, force):
    """What `launch_sample` needs from the disk, done ahead of time on a reader thread: the `.adc` table and the `.roi`
    blob in device memory, copied on a side stream (np.fromfile and the host->device copy release the GIL).  Returns
    (SampleOnGpu | exception | None, event): None = nothing to read ahead (CSV exists and not forced, or the transform has
    no GPU path); an exception is re-raised inside `launch_sample`, where the sample's error handling lives."""
    from . import files, gpu_preprocess
    sample_path = Path(sample_path)
    csv_path = files.sample_csv_path(sample_path, out_dir, suffix=FILE_SUFFIX)
    if (csv_path.is_file() and not force) or not gpu_preprocess.supported(params.transform, params.img_shape[0]):
        return None, None
    try:
        side = _reader_stream(params.device)
        with torch.cuda.stream(side):
            gs = gpu_preprocess.SampleOnGpu(sample_path.with_suffix(".adc"), sample_path.with_suffix(".roi"), params.device)
            ev = torch.cuda.Event()
            ev.record(side)
        return gs, ev
    except Exception as e:  # noqa: BLE001
        return e, None


_READER_STREAMS = {}


def _reader_stream(device):
    key = str(device)
    if key not in _READER_STREAMS:
        _READER_STREAMS[key] = torch.cuda.Stream(device=device)
    return _READER_STREAMS[key]


def launch_sample(sample_path, net, params, out_dir, force=False, dist=None, ahead=None):
    """First half of `process_sample`: parse the sample, put its `.roi` blob on the GPU and queue preprocessing +
    forward for this rank's shard.  Returns None (CSV exists, not forced) or the state `complete_sample` finishes.
    `ahead`: the result of `read_sample_ahead` for this sample, if a reader thread already fetched it."""
    from . import dp, files, ifcb
    sample_path = Path(sample_path)
    sample = sample_path.name
    csv_path = files.sample_csv_path(sample_path, out_dir, suffix=FILE_SUFFIX)
    if _skip_existing(csv_path, force, dist):
        return None
    log.debug(f"Computing probabilities for {sample}")
    rank, world = dp.rank_world(dist)
    rows, error = [], None
    try:
        from . import gpu_preprocess
        if gpu_preprocess.supported(params.transform, params.img_shape[0]):
            # .roi bytes -> GPU -> resized/bordered uint8 batches -> forward: no PNGs, no per-ROI host work
            gs, ev = ahead if ahead is not None else (None, None)
            if isinstance(gs, Exception):
                raise gs
            if gs is None:
                gs = gpu_preprocess.SampleOnGpu(sample_path.with_suffix(".adc"), sample_path.with_suffix(".roi"),
                                                params.device)
            elif ev is not None:
                cur = torch.cuda.current_stream(params.device)
                cur.wait_event(ev)                 # the blob was copied on the reader's stream ...
                gs.blob.record_stream(cur)         # ... and its memory must not be recycled under this stream's kernels
                if gs.rois is not None:
                    gs.rois.record_stream(cur)
            th, tw = params.transform.target_dims
            code = gpu_preprocess.border_code(params.transform)
            lo, hi = dp.shard_range(len(gs), rank, world)

            # --batch-size bounds memory in the reference (default 64, sized for its GPUs); here it is a lower bound:
            # the whole sample already sits in HBM, and 64-image launches leave the card launch-bound (74 k ROI/s
            # against 135 k at 512).  Probabilities do not depend on the batch an image is in (eval-mode network).
            step = max(params.batch_size, int(os.environ.get("SPK_PROB_MIN_BATCH", "512")))

            def gpu_batches():
                for b in range(lo, hi, step):
                    e = min(hi, b + step)
                    yield gs.batch(b, e, th, tw, code), gs.numbers[b:e]
            if world > 1 and getattr(net, "_auto_calibration_dir", None) is not None:
                auto_calibrate(net, gs.batch(lo, min(hi, lo + AUTO_CALIBRATION_IMAGES), th, tw, code) if hi > lo else None, dist)
            rows = net_pass_launch(net, gpu_batches(), params.device)
        else:
            rois = ifcb.read_rois(sample_path.with_suffix(".adc"), sample_path.with_suffix(".roi"))
            lo, hi = dp.shard_range(len(rois), rank, world)
            items = [(f"{sample}_{num:05d}.png", _as_chans(img, params.img_shape[0])) for num, img in rois[lo:hi]]
            if world > 1 and getattr(net, "_auto_calibration_dir", None) is not None:
                auto_calibrate(net, next(_batches(items, params.transform, AUTO_CALIBRATION_IMAGES))[0] if items else None, dist)
            rows = net_pass(net, _batches(items, params.transform, params.batch_size), params.device)
    except Exception as e:  # noqa: BLE001 - re-raised by complete_sample on every rank
        error = e
    return sample, rows, error, csv_path


def complete_sample(state, params, dist=None):
    """Second half: wait for the GPU, gather the ranks' rows, write the CSV (rank 0).  Raises what went wrong in
    either half."""
    sample, rows, error, csv_path = state
    if error is None and isinstance(rows, PendingRows):
        try:
            rows = rows.result()
        except Exception as e:  # noqa: BLE001
            rows, error = [], e
    _finish(rows, error, sample, params.classes, csv_path, dist)
    return sample


def process_sample(sample_path, net, params, out_dir, force=False, dist=None):
    """One IFCB sample -> `<out>/YYYY/MM/DD/<sample>.prob.csv`.  Under `torch.distributed` (one process per GPU)
    the ROI list is split into contiguous shards (`dp.shard_range`), every rank runs `net_pass` on its shard
    with its full weight replica, rank 0 merges the rows by ROI number and writes the file: the batch split of
    SURVEY.md section 8e, no collective on the data path."""
    state = launch_sample(sample_path, net, params, out_dir, force, dist)
    if state is None:
        return Path(sample_path).name
    return complete_sample(state, params, dist)


def _as_chans(gray, num_chans):
    """What cv2.imread + BGR2RGB gives for a greyscale PNG (data.py:214-219)."""
    if num_chans == 1:
        return gray[:, :, None]
    return np.repeat(gray[:, :, None], 3, axis=2)


def process_images(img_paths, net, params, csv_path, force=False, dist=None):
    from . import dp, pngio
    csv_path = Path(csv_path)
    if _skip_existing(csv_path, force, dist):
        return
    rank, world = dp.rank_world(dist)
    img_paths = list(img_paths)
    lo, hi = dp.shard_range(len(img_paths), rank, world)
    rows, error = [], None
    try:
        items = [(str(p), pngio.read_image(p, params.img_shape[0])) for p in img_paths[lo:hi]]
        if world > 1 and getattr(net, "_auto_calibration_dir", None) is not None:
            auto_calibrate(net, next(_batches(items, params.transform, AUTO_CALIBRATION_IMAGES))[0] if items else None, dist)
        rows = net_pass(net, _batches(items, params.transform, params.batch_size), params.device)
    except Exception as e:  # noqa: BLE001 - re-raised below on every rank
        error = e
    _finish(rows, error, csv_path.name, params.classes, csv_path, dist)


def main(sample_paths, model_dir, out_dir, batch_size=64, num_workers=2, force=False,
         progress_bar=True, samples_as_images=False):
    from . import dp
    dist, rank, world, local = dp.init_from_env()   # one process per GPU under torch.distributed.run
    net, classes, img_shape, eval_transform, device = prepare_model(model_dir, f"cuda:{local}")
    params = EvalParams(batch_size, num_workers, classes, img_shape, eval_transform, device)
    progress_bar = progress_bar and rank == 0
    try:
        from tqdm import tqdm
    except ImportError:  # pragma: no cover
        tqdm = None
    if samples_as_images:
        it = sample_paths.items()
        if progress_bar and tqdm:
            it = tqdm(it, desc="Processing samples")
        for sample, img_paths in it:
            process_images(img_paths, net, params, Path(out_dir) / f"{sample}{FILE_SUFFIX}.csv", force, dist)
        return None
    sample_paths = list(sample_paths)
    it = sample_paths
    if progress_bar and tqdm:
        it = tqdm(it, desc="Processing samples")
    # A reader thread fetches sample k+2's files (.adc table, .roi blob -> device memory on a side stream) while sample
    # k+1 is parsed into batches and launched.
    # Two samples in flight: while the GPU runs sample k+1 the previous sample is read back, formatted and written
    # (the reference handles one sample at a time, probability.py:97-114; same files, same per-sample error
    # handling).  Single process: the second half runs on a writer thread (device->host copy, numpy formatting and
    # file I/O all release the GIL), so the main thread goes straight on to parse and launch the next sample.  Under
    # torch.distributed the second half holds collectives (all_ok / gather) and stays on the main thread, in order.
    done = set()
    pending = None          # (sample path, state of launch_sample) or a Future of the writer thread
    pool = None
    from concurrent.futures import ThreadPoolExecutor
    if dist is None:
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="sykepic-csv")

    def guarded(fn, sample_path):
        try:
            return fn()
        except ValueError:
            log.exception(f"Faulty raw data for {Path(sample_path).name}")
        except Exception:
            log.exception(f"Unexpected error for {Path(sample_path).name}")
        return None

    def complete(p):
        return guarded(lambda: complete_sample(p[1], params, dist), p[0])

    def finish(p):
        if p is None:
            return
        name = p.result() if pool is not None else complete(p)
        if name is not None:
            done.add(name)

    reader = None
    paths = list(sample_paths)
    if dist is None and paths:
        reader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="sykepic-read")
    nxt = reader.submit(read_sample_ahead, paths[0], params, out_dir, force) if reader else None
    try:
        for i, sample_path in enumerate(it):
            ahead = nxt.result() if nxt is not None else None
            nxt = reader.submit(read_sample_ahead, paths[i + 1], params, out_dir, force) if reader and i + 1 < len(paths) else None
            state = guarded(lambda: launch_sample(sample_path, net, params, out_dir, force, dist, ahead) or "skip", sample_path)
            finish(pending)
            pending = None
            if state == "skip":
                done.add(Path(sample_path).name)
            elif state is not None:
                pending = pool.submit(complete, (sample_path, state)) if pool is not None else (sample_path, state)
        finish(pending)
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
        if reader is not None:
            reader.shutdown(wait=True)
    return done


def calibrate_call(args):
    """`sykepic calibrate`: measure the activation means of an existing model directory on ROIs of the kind it will
    classify (raw samples or PNG images, chosen as for `prob`) and store them as `act_means.pth` beside
    `best_state.pth`.  From then on `sykepic prob -m <dir>` runs the calibrated single-pass mode."""
    from . import files, ifcb, pngio
    keep = os.environ.get("SYKEPIC_CALIBRATED")
    os.environ["SYKEPIC_CALIBRATED"] = "0"      # measure with the weights alone, whatever file is already there
    try:
        net, classes, img_shape, transform, device = prepare_model(args.model)
    finally:
        if keep is None:
            del os.environ["SYKEPIC_CALIBRATED"]
        else:
            os.environ["SYKEPIC_CALIBRATED"] = keep

    from . import gpu_preprocess
    on_gpu = not (args.image_dir or args.images) and gpu_preprocess.supported(transform, img_shape[0])

    def images():
        if args.image_dir or args.images:
            for p in (sorted(Path(args.image_dir).rglob("*.png")) if args.image_dir else [Path(q) for q in args.images]):
                yield pngio.read_image(p, img_shape[0])
        else:
            for sp in (files.list_sample_paths(args.raw) if args.raw else [Path(q) for q in args.samples]):
                for _, img in ifcb.read_rois(sp.with_suffix(".adc"), sp.with_suffix(".roi")):
                    yield _as_chans(img, img_shape[0])

    def batches():
        if on_gpu:
            # raw samples through the GPU preprocessing `prob` itself uses (.roi blob -> resized uint8 batches): the host
            # transform - a Python restatement of cv2.resize, ~10 ms per ROI - was 21 of the 21.9 s that round 4's
            # `sykepic calibrate` took for 2048 ROIs
            th, tw = transform.target_dims
            code = gpu_preprocess.border_code(transform)
            for sp in (files.list_sample_paths(args.raw) if args.raw else [Path(q) for q in args.samples]):
                gs = gpu_preprocess.SampleOnGpu(sp.with_suffix(".adc"), sp.with_suffix(".roi"), device)
                for b in range(0, len(gs), args.batch_size):
                    yield gs.batch(b, min(len(gs), b + args.batch_size), th, tw, code)
            return
        chunk = []
        for img in images():
            chunk.append(transform(img))
            if len(chunk) == args.batch_size:
                yield torch.stack(chunk)
                chunk = []
        if chunk:
            yield torch.stack(chunk)

    n = calibrate_model(net, batches(), args.num_images)
    if n == 0:
        raise ValueError("no image to calibrate on")
    save_act_means(net, args.model, n)
    log.info(f"activation means of {n} images saved to {Path(args.model) / ACT_MEANS_FILE}")
    return n


def call(args):
    from . import files
    if args.image_dir or args.images:
        as_images = True
        if args.image_dir:
            img_paths = sorted(Path(args.image_dir).rglob("*.png"))
        else:
            img_paths = sorted(Path(p) for p in args.images)
        sample_paths = {}
        for p in img_paths:
            sample_paths.setdefault(p.name.rpartition("_")[0], []).append(p)
        selected = sample_paths
    else:
        as_images = False
        if args.raw:
            sample_paths = files.list_sample_paths(args.raw)
        else:
            sample_paths = [Path(p) for p in args.samples]
        selected = []
        for sp in sample_paths:  # .roi files over 1 GB are skipped (probability.py:45-51)
            if sp.with_suffix(".roi").stat().st_size <= 1e9:
                selected.append(sp)
            else:
                log.warning(f"{sp.name} is over 1G, skipping")
    return main(selected, args.model, args.out, args.batch_size, args.num_workers, args.force,
                progress_bar=True, samples_as_images=as_images)
