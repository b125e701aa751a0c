"""Data parallelism over the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The reference is single-device; this is the one exchange
step the build adds (§8e):
  * training  — synchronous DP: every rank runs forward/backward on its shard
    with LOCAL BatchNorm statistics (exactly what DDP does; differs from one
    device seeing the global batch), then ONE all-reduce (sum) of the flat
    fp32 gradient buffer (24.07 M elements for ResNet-50 + head = 96 MB) and
    identical optimizer steps everywhere (the 1/world mean is folded into the
    optimizer kernel's ``grad_scale``).  A single large message suits xGMI's
    point-to-point mesh: RCCL splits it over all 7 links per GPU.
  * inference — no collective: ranks take contiguous chunks of the ROI list
    and rank 0 concatenates + sorts by ROI id as ``net_pass`` does.
"""

import ctypes
import os

import torch


def init_from_env():
    """(dist, rank, world, local_rank) from the launcher's environment (`torch.distributed.run`: RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_*); (None, 0, 1, 0) for a plain single-process run.  Backend nccl = RCCL over xGMI."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1, 0
    import torch.distributed as dist
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    # SPK_DIST_BACKEND=gloo: rehearsal of the N>1 path on fewer GPUs than ranks (RCCL refuses two ranks on one
    # device); ranks then share the devices round-robin
    backend = os.environ.get("SPK_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank, world, local


def rank_world(dist=None):
    if dist is None or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


class _DevicePtr:
    """Expose a raw device pointer to torch via __cuda_array_interface__."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {
            "shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def shard_range(n_items, rank, world):
    """Contiguous [begin, end) of rank's shard; sizes differ by at most 1."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_indices(indices, rank, world, pad=True):
    """Rank's slice of an (already shuffled) epoch order.  With pad=True the
    order is extended by wrapping so every rank runs the same number of steps
    (a rank that ran out of batches would dead-lock the all-reduce)."""
    indices = list(indices)
    if pad and world > 1 and len(indices) % world:
        indices += indices[: world - len(indices) % world]
    return indices[rank::world]


class GradSync:
    """All-reduce of the library's flat gradient buffer.

    overlap=True (opt-in with SPK_DP_OVERLAP=1 until one RCCL run on >= 2 GPUs has shown gradients bit-equal to the
    plain path - the builder's boxes have one GPU, where the mechanism itself runs over a one-rank RCCL group:
    tests/test_gpu_dp.py): the buffer is reduced in up to three slices
    that the library reports back to front while the backward pass is still running (head + last stage first, the
    stem last; `spk_model_set_grad_ready_callback`): each slice's collective is enqueued on a communication
    stream behind an event, so the 96 MB of ResNet-50 gradients cross xGMI underneath the remaining dgrad / wgrad
    kernels instead of after them; `all_reduce()` then only waits for the collectives still in flight.  xGMI is a
    point-to-point mesh: few large messages (3 x 16-64 MB) keep every link busy, many small buckets would not."""

    def __init__(self, net, dist=None, view=None, overlap=None, buckets=3):
        self.net, self.dist = net, dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        if view is not None:          # CPU/gloo tests pass a host tensor
            self.flat = view
        else:
            ptr, numel = net.grad_buffer()
            self.flat = torch.as_tensor(_DevicePtr(ptr, numel), device=net.device)
        self._works, self._cb, self._comm = [], None, None
        self._covered, self._err = 0, None
        self.waited = 0          # collectives the last all_reduce() waited for (overlapped path)
        if overlap is None:
            overlap = self.world > 1 and view is None and os.environ.get("SPK_DP_OVERLAP", "0") == "1"
        # (an explicit overlap=True installs the hook for a one-rank group too: tests/test_gpu_dp.py drives the whole
        # mechanism - callback, communication stream, RCCL collectives, waits - on the single test GPU that way)
        if overlap and view is None and dist is not None and dist.is_initialized():
            self._install(buckets)

    def _install(self, buckets):
        from . import lib
        self._comm = torch.cuda.Stream(device=self.net.device)

        def ready(_user, bucket, offset, numel):
            # called by the library inside forward_backward, after it made the communication stream wait for
            # the kernels that write flat[offset : offset + numel]
            # ctypes prints and DROPS an exception raised inside a callback: keep it and re-raise from all_reduce(),
            # otherwise a failed collective would leave its slice un-reduced and the replicas would silently diverge
            try:
                if offset < 0 or numel <= 0 or offset + numel > self.flat.numel():
                    raise RuntimeError(f"gradient slice [{offset}, {offset + numel}) outside the buffer")
                with torch.cuda.stream(self._comm):
                    self._works.append(self.dist.all_reduce(self.flat[offset:offset + numel],
                                                            op=self.dist.ReduceOp.SUM, async_op=True))
                self._covered += int(numel)
            except BaseException as e:   # noqa: BLE001 - re-raised on the training thread
                self._err = self._err or e

        self._cb = lib.GRAD_READY_FN(ready)   # keep the ctypes thunk alive as long as the handle may call it
        self.net.set_grad_ready_callback(self._cb, self._comm.cuda_stream, buckets)

    @property
    def overlapped(self):
        """True while the slices are reduced underneath the backward pass (the library calls back per slice)."""
        return self._cb is not None

    def close(self):
        if self._cb is not None:
            self.net.set_grad_ready_callback(None, 0, 0)
            self._cb = None

    def all_reduce(self, optimizer=None):
        if self._cb is not None:
            works, covered, err = self._works, self._covered, self._err
            self._works, self._covered, self._err = [], 0, None
            if err is not None:
                raise RuntimeError("gradient all-reduce failed inside the backward pass") from err
            if covered != self.flat.numel():   # the reported slices must tile the buffer exactly once
                raise RuntimeError(f"overlapped all-reduce covered {covered} of {self.flat.numel()} gradient "
                                   "elements")
            for w in works:     # makes the current stream wait for each collective
                w.wait()
            self.waited = len(works)
        elif self.world > 1:
            self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM)
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world

    def reduce_stats(self, loss_sum, correct, count):
        """Global (sum loss*n, #correct, n) for the [STAT] lines."""
        if self.world == 1:
            return loss_sum, correct, count
        t = torch.tensor([loss_sum, correct, count], dtype=torch.float64, device=self.flat.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return tuple(float(v) for v in t.tolist())


def _comm_device(dist, device):
    """Tensors of a collective live on the GPU for nccl (= RCCL) and on the host for gloo."""
    return torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_state(net, dist=None, src=0):
    """Every replica starts from rank `src`'s parameters AND buffers (what DistributedDataParallel does at
    construction).  Without it each rank keeps its own random initialisation (pretrained weights cannot be
    downloaded here, the head is always random) and the all-reduce averages gradients taken at different
    weights.  One flat float32 broadcast + one int64 broadcast (num_batches_tracked)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    sd = net.state_dict()
    fkeys = [k for k, v in sd.items() if v.dtype != torch.int64]
    ikeys = [k for k, v in sd.items() if v.dtype == torch.int64]
    dev = _comm_device(dist, getattr(net, "device", "cpu"))
    flat = torch.cat([sd[k].detach().float().reshape(-1) for k in fkeys]).to(dev)
    cnt = torch.stack([sd[k].detach().reshape(()) for k in ikeys]).to(dev) if ikeys else None
    dist.broadcast(flat, src)
    if cnt is not None:
        dist.broadcast(cnt, src)
    if dist.get_rank() == src:
        return
    flat = flat.cpu()
    out, off = {}, 0
    for k in fkeys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].reshape(sd[k].shape).clone()
        off += n
    for i, k in enumerate(ikeys):
        out[k] = cnt[i].cpu().clone()
    net.load_state_dict(out)


def sync_buffers(net, dist=None):
    """BatchNorm running statistics are rank-local during an epoch (every rank normalises with the statistics of
    ITS shard, as DDP replicas do); before validation they are averaged so that all ranks evaluate — and rank 0
    checkpoints — the same model.  Returns the number of tensors averaged."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    if hasattr(net, "_specs"):   # HipNet: read only the running statistics out of the library
        sd = {k: net._read_tensor(k, shape, torch.float32) for k, shape, kind in net._specs
              if kind in ("bn_mean", "bn_var")}
    else:                        # any module with a torch state_dict
        sd = {k: v.detach().float() for k, v in net.state_dict().items()
              if k.endswith(("running_mean", "running_var"))}
    keys = list(sd)
    if not keys:
        return 0
    dev = _comm_device(dist, getattr(net, "device", "cpu"))
    flat = torch.cat([sd[k].reshape(-1) for k in keys]).to(dev)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat = (flat / dist.get_world_size()).cpu()
    out, off = {}, 0
    for k in keys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].reshape(sd[k].shape).clone()
        off += n
    net.load_state_dict(out, strict=False)
    return len(keys)


def broadcast_object(obj, dist=None, src=0):
    """Small Python value from rank `src` to everyone (model id, run directory)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src)
    return box[0]


def all_ok(ok, dist=None):
    """True iff `ok` holds on EVERY rank (one tiny all-reduce): lets the replicas leave a loop together instead
    of one of them blocking the others in the next collective."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def gather_parts(part, dist=None, dst=0):
    """Inference, array form: every rank's shard object (prob.ProbRows) on rank `dst` ([part] without a process
    group; None on the other ranks).  Host-side object gather like `gather_rows`."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [part]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(part, out, dst=dst)
    return out


def gather_rows(rows, dist=None, dst=None):
    """Inference: the ranks' [(roi, probs)] lists concatenated and sorted by ROI number, as `net_pass` returns them
    (reference probability.py:195-197).  dst=None: on every rank; dst=r: on rank r only (None elsewhere).  The
    exchange is a host-side object gather: the data path itself has no collective."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(rows)
    world = dist.get_world_size()
    if dst is None:
        out = [None] * world
        dist.all_gather_object(out, rows)
    else:
        out = [None] * world if dist.get_rank() == dst else None
        dist.gather_object(rows, out, dst=dst)
        if out is None:
            return None
    return sorted(r for part in out for r in part)
