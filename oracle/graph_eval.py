"""ORACLE (test infrastructure): torch fp32 interpreter of the layer graph in
``sykepic_hip.arch`` — gives every intermediate activation so the HIP kernels
can be checked layer by layer.  Its logits are themselves checked against
``refnet.RefNet`` (which the goldens pin to the reference) in the tests."""

import torch
import torch.nn.functional as F

from sykepic_hip import arch


def run(graph, state, x, train=False, eps=1e-5):
    """state: {key: torch tensor}.  Returns {tensor id: activation}."""
    acts = {0: x}
    for op in graph.ops:
        a = acts[op.src]
        if op.kind == arch.OP_CONV:
            y = F.conv2d(a, state[op.name + ".weight"], None, op.stride, op.pad)
            y = F.batch_norm(y, state[op.bn + ".running_mean"].clone(), state[op.bn + ".running_var"].clone(),
                             state[op.bn + ".weight"], state[op.bn + ".bias"], train, 0.1, eps)
            if op.res >= 0:
                y = y + acts[op.res]
            if int(op.relu) == arch.ACT_SILU:
                y = F.silu(y)
            elif op.relu:
                y = F.relu(y)
        elif op.kind == arch.OP_DWCONV:   # EfficientNet: depthwise conv + BN + SiLU
            y = F.conv2d(a, state[op.name + ".weight"], None, op.stride, op.pad, groups=op.cin)
            y = F.batch_norm(y, state[op.bn + ".running_mean"].clone(), state[op.bn + ".running_var"].clone(),
                             state[op.bn + ".weight"], state[op.bn + ".bias"], train, 0.1, eps)
            y = F.silu(y) if int(op.relu) == arch.ACT_SILU else (F.relu(y) if op.relu else y)
        elif op.kind == arch.OP_SE:       # squeeze-excitation gate
            s = a.mean((2, 3), keepdim=True)
            s = F.silu(F.conv2d(s, state[op.name + ".fc1.weight"], state[op.name + ".fc1.bias"]))
            s = torch.sigmoid(F.conv2d(s, state[op.name + ".fc2.weight"], state[op.name + ".fc2.bias"]))
            y = a * s
        elif op.kind == arch.OP_MAXPOOL:
            y = F.max_pool2d(a, op.k, op.stride, op.pad)
        elif op.kind == arch.OP_GAVGPOOL:
            y = a.mean((2, 3))
        elif op.kind == arch.OP_LINEAR:
            y = F.linear(a, state[op.name + ".weight"], state[op.name + ".bias"])
        else:
            y = a
        acts[op.dst] = y
    return acts


def _bf16_st(t):
    """Round to bf16 with a straight-through gradient."""
    return t + (t.detach().bfloat16().float() - t.detach())


def run_train_bf16(graph, state, x, eps=1e-5):
    """Train-mode forward that rounds to bf16 exactly where the HIP training
    path does (conv operands, raw conv output, post-BN activation), everything
    else fp32.  With the same roundings the ReLU masks agree with the GPU, so
    autograd through this graph checks the backward KERNELS to ~1e-2 instead
    of being dominated by mask flips of a pure-fp32 forward.
    state: {key: tensor (requires_grad for parameters)}.  Returns acts."""
    acts = {0: _bf16_st(x * 255.0) / 255.0}   # (csrc/spk_common.h SPK_INPUT_SCALE: pixel values x 255 are exact in bf16)
    for op in graph.ops:
        a = acts[op.src]
        if op.kind == arch.OP_CONV:
            w = state[op.name + ".weight"]
            # (the 7x7 stem's weights are packed as bf16(w / 255): its input holds pixel values x 255)
            wb = _bf16_st(w / INPUT_SCALE) * INPUT_SCALE if op.src == 0 and op.k == 7 else _bf16_st(w)
            y32 = F.conv2d(a, wb, None, op.stride, op.pad)
            mean = y32.mean((0, 2, 3), keepdim=True)
            var = y32.var((0, 2, 3), unbiased=False, keepdim=True)
            yr = _bf16_st(y32)
            g = state[op.bn + ".weight"].view(1, -1, 1, 1)
            b = state[op.bn + ".bias"].view(1, -1, 1, 1)
            v = (yr - mean) * torch.rsqrt(var + eps) * g + b
            if op.res >= 0:
                v = v + acts[op.res]
            if op.relu:
                v = F.relu(v)
            v = _bf16_st(v)
        elif op.kind == arch.OP_MAXPOOL:
            v = F.max_pool2d(a, op.k, op.stride, op.pad)
        elif op.kind == arch.OP_GAVGPOOL:
            v = a.mean((2, 3))
        elif op.kind == arch.OP_LINEAR:
            v = F.linear(a, state[op.name + ".weight"], state[op.name + ".bias"])
        else:
            v = a
        acts[op.dst] = v
    return acts


def _act(v, op):
    a = int(op.relu)
    return F.silu(v) if a == arch.ACT_SILU else (F.relu(v) if a else v)


def run_train_forced(graph, state, x, forced, eps=1e-5, row_scale=None):
    """Train-mode forward in which every activation is overwritten (straight
    through) with the value the GPU produced (`forced[id]`).  Each layer's
    local derivative is then evaluated at exactly the GPU's operating point —
    same ReLU masks, same BatchNorm statistics — so autograd through this
    graph isolates the backward kernels from forward rounding chaos.
    EfficientNet graphs (torchvision MBConv: depthwise conv, squeeze-excitation, SiLU) are covered too;
    `row_scale[dst]` is the per-image StochasticDepth factor (0 or 1/(1-p), "row" mode) of the block that writes
    tensor `dst`, as drawn by the implementation under test (None: no row dropped)."""
    def force(v, t):
        return v + (forced[t].to(v.dtype) - v).detach() if t in forced else v

    acts = {0: _bf16_st(x * 255.0) / 255.0}   # (csrc/spk_common.h SPK_INPUT_SCALE: pixel values x 255 are exact in bf16)
    for op in graph.ops:
        a = acts[op.src]
        if op.kind in (arch.OP_CONV, arch.OP_DWCONV):
            groups = op.cin if op.kind == arch.OP_DWCONV else 1
            y32 = F.conv2d(a, _bf16_st(state[op.name + ".weight"]), None, op.stride, op.pad, groups=groups)
            v = F.batch_norm(y32, None, None, state[op.bn + ".weight"], state[op.bn + ".bias"], True, 0.1, eps)
            if op.res >= 0:
                if row_scale is not None and op.dst in row_scale:
                    v = v * row_scale[op.dst].view(-1, 1, 1, 1)
                v = v + acts[op.res]
            v = _act(v, op)
        elif op.kind == arch.OP_SE:
            s = a.mean((2, 3), keepdim=True)
            s = F.silu(F.conv2d(s, state[op.name + ".fc1.weight"], state[op.name + ".fc1.bias"]))
            s = torch.sigmoid(F.conv2d(s, state[op.name + ".fc2.weight"], state[op.name + ".fc2.bias"]))
            v = a * s
        elif op.kind == arch.OP_MAXPOOL:
            v = F.max_pool2d(a, op.k, op.stride, op.pad)
        elif op.kind == arch.OP_GAVGPOOL:
            v = a.mean((2, 3))
        elif op.kind == arch.OP_LINEAR:
            v = F.linear(a, state[op.name + ".weight"], state[op.name + ".bias"])
        else:
            v = a
        acts[op.dst] = force(v, op.dst)
    return acts


INPUT_SCALE = 255.0   # the HIP path stores pixel values x 255 in 16 bits (exact for the k / 255 values ToTensor gives)


def _r16_input(x):
    """the input as the HIP path sees it: bf16(x * 255) / 255 (csrc/spk_common.h SPK_INPUT_SCALE)"""
    return (x * INPUT_SCALE).bfloat16().float() / INPUT_SCALE


def _r16_stem_w(w):
    """the 7x7 stem's training weights as packed: bf16(w / 255) (the input holds pixel values x 255), in input units"""
    return (w / INPUT_SCALE).bfloat16().float() * INPUT_SCALE


def _r16(t):
    return t.bfloat16().float()


def train_step_bf16(graph, state, x, y, eps=1e-5, round_grads=True, forced=None, first_op=0):
    """One training step (forward, mean cross-entropy, backward) with every tensor rounded to bf16 exactly where
    the HIP path stores one: conv operands, the raw conv output, the post-BatchNorm activation, and — unlike
    autograd through `run_train_bf16` — every GRADIENT tensor too (the gradient of each activation, the gradient
    w.r.t. each conv output, accumulation into a tensor with two consumers rounds after each add, in the
    executor's reverse layer order).  Statistics, parameter gradients and the head stay float32.

    Purpose: it splits the distance between the GPU's gradients and fp32 autograd into
      |GPU - this|        what the kernels add (expected ~1e-3: accumulation order, rare 1-ulp flips), and
      |this - autograd|   what bf16 gradient STORAGE costs for any implementation with these rounding points.
    The chain rule below is the textbook Conv2d / BatchNorm2d(train) / ReLU / MaxPool2d / AdaptiveAvgPool2d /
    Linear backward (what torch autograd evaluates for the reference at sykepic/train/train.py:242).

    round_grads=False keeps every gradient tensor in float32 on the SAME forward (= exact backpropagation through
    the rounded forward with straight-through roundings): the difference between the two settings is purely the
    cost of storing gradients in bf16.

    forced: {tensor id: activation the GPU produced}.  Each layer then starts from the GPU's own input (its own
    output still decides its ReLU mask and BatchNorm statistics), so a 1-ulp difference cannot grow through the
    following layers: random-weight nets with train-mode BatchNorm over a handful of samples amplify one flipped
    bf16 ulp into 10-50 % of a gradient tensor (measured: two float32 evaluation orders of this very function),
    which says nothing about a kernel.

    first_op > 0 (needs `forced`): only ops[first_op:] are evaluated, from the GPU's activations — a truncated
    backward over the tail of the network (cheap enough at the benched batch size); x may then be None.

    state: {key: float32 tensor}.  Returns dict(logits, loss, acts, grads {state key: tensor},
    act_grads {tensor id: tensor})."""
    rg = _r16 if round_grads else (lambda t: t)
    acts = {0: _r16_input(x)} if x is not None else {}
    if first_op > 0:
        for op in graph.ops[first_op:]:
            for t in (op.src, op.res):
                if t >= 0 and t in forced:
                    acts.setdefault(t, forced[t].float())
    saved, own = {}, {}
    for i, op in enumerate(graph.ops):
        if i < first_op:
            continue
        a = acts[op.src]
        if op.kind == arch.OP_CONV:
            wb = _r16_stem_w(state[op.name + ".weight"]) if op.src == 0 and op.k == 7 else _r16(state[op.name + ".weight"])
            y32 = F.conv2d(a, wb, None, op.stride, op.pad)
            mean = y32.double().mean((0, 2, 3))
            var = y32.double().var((0, 2, 3), unbiased=False)
            invstd = (1.0 / torch.sqrt(var + eps)).float()
            mean = mean.float()
            yr = _r16(y32)
            g, b = state[op.bn + ".weight"], state[op.bn + ".bias"]
            scale = g * invstd
            shift = b - mean * scale
            v = yr * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            if op.res >= 0:
                v = v + acts[op.res]
            pos = None
            if op.relu:
                pos = v > 0
                v = F.relu(v)
            saved[i] = (a, wb, yr, mean, invstd, pos)
            v = _r16(v)
        elif op.kind == arch.OP_MAXPOOL:
            v, idx = F.max_pool2d(a, op.k, op.stride, op.pad, return_indices=True)
            saved[i] = (a.shape, idx)
        elif op.kind == arch.OP_GAVGPOOL:
            v = a.mean((2, 3))
            saved[i] = a.shape
        elif op.kind == arch.OP_LINEAR:
            v = F.linear(a, state[op.name + ".weight"], state[op.name + ".bias"])
        elif op.kind == arch.OP_DROPOUT and forced is not None and op.dst in forced and op.p > 0:
            # nn.Dropout(p) in train mode: the kept set is whatever the GPU drew (its generator is its own, as
            # torch's is); the arithmetic y = x * keep / (1 - p) and its backward are checked on that set
            keep = (forced[op.dst] != 0) | (a == 0)
            v = a * keep / (1.0 - op.p) if op.p < 1 else torch.zeros_like(a)
            saved[i] = keep
        else:
            v = a
        own[op.dst] = v   # this layer's own output (before forcing): forward check at the GPU's operating point
        acts[op.dst] = forced[op.dst].to(v.dtype) if forced is not None and op.dst in forced else v
    last = graph.ops[-1].dst
    logits = acts[last]
    n = logits.shape[0]
    logp = F.log_softmax(logits, 1)
    loss = -logp[torch.arange(n), y].mean()
    G = {last: (logp.exp() - F.one_hot(y, logits.shape[1]).float()) / n}
    grads = {}

    # Since round 4 the HIP path takes a layer's BatchNorm-backward SUMS from the data-gradient kernel that completes the
    # gradient of its output - when that kernel is a stride-1 conv dgrad and the last writer (no consumer of the tensor
    # comes earlier in the graph) - i.e. from the float32 values before they are rounded for storage (csrc/conv_igemm.hip,
    # spk_set_bnb); everything else (dy, the shortcut gradient) still reads the stored bf16 tensor.
    G32 = {}
    pre_round_sums = set()
    for t in {o.dst for o in graph.ops if o.kind == arch.OP_CONV}:
        users = [j for j, o in enumerate(graph.ops) if o.src == t or (o.kind == arch.OP_CONV and o.res == t)]
        if users and graph.ops[min(users)].kind == arch.OP_CONV and graph.ops[min(users)].src == t \
                and graph.ops[min(users)].stride == 1 and graph.ops[min(users)].k == 1:   # (default SPK_BNB_FUSE=3: 1x1 consumers)
            pre_round_sums.add(t)

    def put(t, val, dgrad=False):  # first consumer writes, later ones add and round again
        full = val if t not in G else G[t] + val
        if dgrad:
            G32[t] = full
        else:
            G32.pop(t, None)
        G[t] = rg(full)

    for i in range(len(graph.ops) - 1, first_op - 1, -1):
        op = graph.ops[i]
        if op.dst not in G:
            continue
        gy = G[op.dst]
        if op.kind == arch.OP_LINEAR:
            xin = acts[op.src]
            grads[op.name + ".weight"] = gy.t() @ xin
            grads[op.name + ".bias"] = gy.sum(0)
            G[op.src] = gy @ state[op.name + ".weight"]
        elif op.kind == arch.OP_GAVGPOOL:
            shp = saved[i]
            inv = torch.tensor(1.0 / (shp[2] * shp[3]), dtype=torch.float32)
            G[op.src] = rg((gy * inv).view(shp[0], shp[1], 1, 1).expand(shp).contiguous())
        elif op.kind == arch.OP_MAXPOOL:
            shp, idx = saved[i]
            gx = torch.zeros(shp[0], shp[1], shp[2] * shp[3])
            gx.scatter_add_(2, idx.flatten(2), gy.flatten(2))
            G[op.src] = rg(gx.view(shp))
        elif op.kind == arch.OP_CONV:
            a, wb, yr, mean, invstd, pos = saved[i]
            dz = gy * pos if pos is not None else gy
            if op.res >= 0:
                put(op.res, dz)
            m = dz.shape[0] * dz.shape[2] * dz.shape[3]
            xhat = (yr - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
            dzs = dz
            if round_grads and op.dst in pre_round_sums and op.dst in G32:   # sums made by the dgrad epilogue
                dzs = G32[op.dst] * pos if pos is not None else G32[op.dst]
            s1 = dzs.double().sum((0, 2, 3))
            s2 = (dzs * xhat).double().sum((0, 2, 3))
            grads[op.bn + ".bias"] = s1.float()
            grads[op.bn + ".weight"] = s2.float()
            c1, c2 = (s1 / m).float().view(1, -1, 1, 1), (s2 / m).float().view(1, -1, 1, 1)
            k3 = (state[op.bn + ".weight"] * invstd).view(1, -1, 1, 1)
            dy = rg(k3 * (dz - c1 - xhat * c2))
            if op.src != 0:
                put(op.src, torch.nn.grad.conv2d_input(a.shape, wb, dy, op.stride, op.pad), dgrad=True)
            grads[op.name + ".weight"] = torch.nn.grad.conv2d_weight(a, wb.shape, dy, op.stride, op.pad)
        elif op.kind == arch.OP_DROPOUT and i in saved:
            G[op.src] = gy * saved[i] / (1.0 - op.p) if op.p < 1 else torch.zeros_like(gy)
        else:
            G[op.src] = gy
    return {"logits": logits, "loss": loss, "acts": acts, "own": own, "grads": grads, "act_grads": G}
