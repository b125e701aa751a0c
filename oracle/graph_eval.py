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
            if op.relu:
                y = F.relu(y)
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
