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

import torch


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
    """All-reduce of the library's flat gradient buffer."""

    def __init__(self, net, dist=None, view=None):
        self.net, self.dist = net, dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        if view is not None:          # CPU/gloo tests pass a host tensor
            self.flat = view
        else:
            ptr, numel = net.grad_buffer()
            self.flat = torch.as_tensor(_DevicePtr(ptr, numel), device=net.device)

    def all_reduce(self, optimizer=None):
        if self.world > 1:
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


def gather_rows(rows, dist=None):
    """Inference: concatenate every rank's [(roi, probs)] on all ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(rows)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rows)
    return sorted(r for part in out for r in part)
