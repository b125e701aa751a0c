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
    acts = {0: _bf16_st(x)}
    for op in graph.ops:
        a = acts[op.src]
        if op.kind == arch.OP_CONV:
            y32 = F.conv2d(a, _bf16_st(state[op.name + ".weight"]), None, op.stride, op.pad)
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


def run_train_forced(graph, state, x, forced, eps=1e-5):
    """Train-mode forward in which every activation is overwritten (straight
    through) with the value the GPU produced (`forced[id]`).  Each layer's
    local derivative is then evaluated at exactly the GPU's operating point —
    same ReLU masks, same BatchNorm statistics — so autograd through this
    graph isolates the backward kernels from forward rounding chaos."""
    def force(v, t):
        return v + (forced[t].to(v.dtype) - v).detach() if t in forced else v

    acts = {0: _bf16_st(x)}
    for op in graph.ops:
        a = acts[op.src]
        if op.kind == arch.OP_CONV:
            y32 = F.conv2d(a, _bf16_st(state[op.name + ".weight"]), None, op.stride, op.pad)
            v = F.batch_norm(y32, None, None, state[op.bn + ".weight"], state[op.bn + ".bias"], True, 0.1, eps)
            if op.res >= 0:
                v = v + acts[op.res]
            if op.relu:
                v = F.relu(v)
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
