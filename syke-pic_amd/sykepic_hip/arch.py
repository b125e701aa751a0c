"""Layer graph of the CNN the reference builds in ``TorchVisionNet``.

The reference takes a torchvision backbone, drops its last child and bolts a
stack of ``Linear`` layers (no activations) on top
(``/root/reference/sykepic/train/network.py:48-63``).  torchvision is a
third-party dependency that is not vendored in the reference, so the backbone
topology is restated here from its published definition (ResNet v1.5: stride
on the 3x3 of a Bottleneck; children ``conv1,bn1,relu,maxpool,layer1..4,
avgpool,fc``).  The graph is consumed by the C-ABI library
(``include/sykepic_hip.h``: ``spk_layer_desc``) and also defines the
``state_dict`` key layout ``base.<child>.…`` / ``head.<i>.…`` that
``best_state.pth`` uses.
"""

from dataclasses import dataclass, field

OP_CONV, OP_MAXPOOL, OP_GAVGPOOL, OP_LINEAR, OP_DROPOUT = 1, 2, 3, 4, 5

_RESNETS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}


def supported_networks():
    return sorted(_RESNETS)


@dataclass
class Op:
    kind: int
    name: str = ""        # state_dict prefix of the weight-bearing module
    bn: str = ""          # state_dict prefix of the BatchNorm that follows a conv
    cin: int = 0
    cout: int = 0
    k: int = 1
    stride: int = 1
    pad: int = 0
    relu: bool = False
    src: int = 0          # input activation id
    dst: int = 0          # output activation id
    res: int = -1         # residual activation id added before the ReLU
    child: int = -1       # index of the owning child of ``base`` (-1: head)
    last_bn: bool = False  # last BN of a residual block (synthetic init only)
    p: float = 0.0        # dropout probability


@dataclass
class Graph:
    network: str
    in_chans: int
    num_classes: int
    feat: int
    ops: list = field(default_factory=list)
    n_base_children: int = 9
    head_modules: list = field(default_factory=list)  # [("linear", i, in, out) | ("dropout", i, p)]


def build_graph(network, num_classes, head=(256, 128), dropout=(), in_chans=3):
    """Mirror of ``TorchVisionNet.__init__`` for the ResNet family."""
    if network not in _RESNETS:
        raise ValueError(
            f"network {network!r} has no MI355X path yet; supported: {supported_networks()}"
        )
    kind, depths = _RESNETS[network]
    expansion = 1 if kind == "basic" else 4
    g = Graph(network, in_chans, num_classes, 512 * expansion)
    ops = g.ops
    t = [0]

    def new_t():
        t[0] += 1
        return t[0]

    cur = 0
    # children 0..3: conv1, bn1, relu, maxpool
    d = new_t()
    ops.append(Op(OP_CONV, "base.0", "base.1", in_chans, 64, 7, 2, 3, True, cur, d, -1, 0))
    cur = d
    d = new_t()
    ops.append(Op(OP_MAXPOOL, "", "", 64, 64, 3, 2, 1, False, cur, d, -1, 3))
    cur = d
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), depths)):
        child = 4 + li
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 0) else 1
            pre = f"base.{child}.{b}"
            outp = planes * expansion
            need_ds = stride != 1 or inplanes != outp
            ident = cur
            if kind == "basic":
                a = new_t()
                ops.append(Op(OP_CONV, f"{pre}.conv1", f"{pre}.bn1", inplanes, planes, 3,
                              stride, 1, True, cur, a, -1, child))
                if need_ds:
                    ident = new_t()
                    ops.append(Op(OP_CONV, f"{pre}.downsample.0", f"{pre}.downsample.1",
                                  inplanes, outp, 1, stride, 0, False, cur, ident, -1, child))
                o = new_t()
                ops.append(Op(OP_CONV, f"{pre}.conv2", f"{pre}.bn2", planes, planes, 3, 1, 1,
                              True, a, o, ident, child, True))
            else:
                a = new_t()
                ops.append(Op(OP_CONV, f"{pre}.conv1", f"{pre}.bn1", inplanes, planes, 1, 1, 0,
                              True, cur, a, -1, child))
                b2 = new_t()
                ops.append(Op(OP_CONV, f"{pre}.conv2", f"{pre}.bn2", planes, planes, 3, stride,
                              1, True, a, b2, -1, child))
                if need_ds:
                    ident = new_t()
                    ops.append(Op(OP_CONV, f"{pre}.downsample.0", f"{pre}.downsample.1",
                                  inplanes, outp, 1, stride, 0, False, cur, ident, -1, child))
                o = new_t()
                ops.append(Op(OP_CONV, f"{pre}.conv3", f"{pre}.bn3", planes, outp, 1, 1, 0,
                              True, b2, o, ident, child, True))
            cur = o
            inplanes = outp
    d = new_t()
    ops.append(Op(OP_GAVGPOOL, "", "", inplanes, inplanes, 0, 1, 0, False, cur, d, -1, 8))
    cur = d

    # Head: Linear(feat->h0) ... Linear(h[-1]->num_classes), no activations;
    # Dropout inserted by *list index* exactly as list.insert does
    # (reference network.py:56-61).
    widths = [inplanes] + [int(h) for h in head] + [num_classes]
    mods = [("linear", widths[i], widths[i + 1]) for i in range(len(widths) - 1)]
    for idx, p in dropout:
        mods.insert(int(idx), ("dropout", float(p)))
    n_lin = sum(1 for m in mods if m[0] == "linear")
    seen = 0
    for i, m in enumerate(mods):
        if m[0] == "linear":
            seen += 1
            d = new_t()
            ops.append(Op(OP_LINEAR, f"head.{i}", "", m[1], m[2], 1, 1, 0, False, cur, d, -1, -1,
                          seen == n_lin))
            g.head_modules.append(("linear", i, m[1], m[2]))
            cur = d
        else:
            d = new_t()
            ops.append(Op(OP_DROPOUT, f"head.{i}", "", 0, 0, 1, 1, 0, False, cur, d, -1, -1,
                          False, m[1]))
            g.head_modules.append(("dropout", i, m[1]))
            cur = d
    return g


def param_specs(g):
    """[(state_dict key, shape, kind)] in torch's ``state_dict()`` order."""
    specs = []

    def bn(prefix, c, last):
        specs.append((f"{prefix}.weight", (c,), "bn_w_last" if last else "bn_w"))
        specs.append((f"{prefix}.bias", (c,), "bn_b"))
        specs.append((f"{prefix}.running_mean", (c,), "bn_mean"))
        specs.append((f"{prefix}.running_var", (c,), "bn_var"))
        specs.append((f"{prefix}.num_batches_tracked", (), "bn_nbt"))

    # torch orders a block's modules conv1,bn1,conv2,bn2,(conv3,bn3),downsample;
    # the graph runs the downsample branch earlier, so re-sort per block.
    def order_key(op):
        if op.kind != OP_CONV or op.child < 4:
            return 0
        return 1 if ".downsample." in op.name else 0

    convs = [op for op in g.ops if op.kind == OP_CONV]
    blocks = {}
    ordered = []
    for op in convs:
        if op.child < 4:
            ordered.append(op)
            continue
        blk = op.name.rsplit(".", 1)[0] if ".downsample." not in op.name else op.name.rsplit(".", 2)[0]
        if blk not in blocks:
            blocks[blk] = []
            ordered.append(blk)
        blocks[blk].append(op)
    for item in ordered:
        group = [item] if isinstance(item, Op) else sorted(blocks[item], key=order_key)
        for op in group:
            specs.append((f"{op.name}.weight", (op.cout, op.cin, op.k, op.k), "conv_w"))
            bn(op.bn, op.cout, op.last_bn)
    lin = [op for op in g.ops if op.kind == OP_LINEAR]
    for op in lin:
        specs.append((f"{op.name}.weight", (op.cout, op.cin), "fc_w_last" if op.last_bn else "fc_w"))
        specs.append((f"{op.name}.bias", (op.cout,), "fc_b"))
    return specs


def conv_flops_per_image(g, h, w):
    """Algorithmic FLOPs (1 MAC = 2 FLOP) of conv + head per image; used by
    bench.py for the MFMA roofline."""
    dims = {0: (h, w)}
    total = 0
    for op in g.ops:
        ih, iw = dims.get(op.src, (1, 1))
        if op.kind in (OP_CONV, OP_MAXPOOL):
            oh = (ih + 2 * op.pad - op.k) // op.stride + 1
            ow = (iw + 2 * op.pad - op.k) // op.stride + 1
            dims[op.dst] = (oh, ow)
            if op.kind == OP_CONV:
                total += 2 * oh * ow * op.cout * op.cin * op.k * op.k
        elif op.kind == OP_GAVGPOOL:
            dims[op.dst] = (1, 1)
        elif op.kind == OP_LINEAR:
            dims[op.dst] = (1, 1)
            total += 2 * op.cin * op.cout
        else:
            dims[op.dst] = dims.get(op.src, (1, 1))
    return total
