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

OP_CONV, OP_MAXPOOL, OP_GAVGPOOL, OP_LINEAR, OP_DROPOUT, OP_DWCONV, OP_SE = 1, 2, 3, 4, 5, 6, 7
ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2   # Op.relu (bool for the ResNets: True == ACT_RELU)

_RESNETS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}


# torchvision efficientnet_b0..b4: (width multiplier, depth multiplier).
# (b5-b7 use BatchNorm eps 1e-3 and are not built.)
_EFFNETS = {
    "efficientnet_b0": (1.0, 1.0),
    "efficientnet_b1": (1.0, 1.1),
    "efficientnet_b2": (1.1, 1.2),
    "efficientnet_b3": (1.2, 1.4),
    "efficientnet_b4": (1.4, 1.8),
    "efficientnet_b5": (1.6, 2.2),
    "efficientnet_b6": (1.8, 2.6),
    "efficientnet_b7": (2.0, 3.1),
}
# expand ratio, kernel, stride, input channels, output channels, layers (B0 baseline; Tan & Le 2019, table 1)
_MBCONV = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
           (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))


# BatchNorm2d(eps, momentum) of the backbone: torch's defaults, except torchvision's efficientnet_b5..b7
# (norm_layer = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01))
def bn_params(network):
    return (1e-3, 0.01) if network in ("efficientnet_b5", "efficientnet_b6", "efficientnet_b7") else (1e-5, 0.1)


def supported_networks():
    return sorted(_RESNETS) + sorted(_EFFNETS)


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


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
    p: float = 0.0        # dropout probability; on a block-closing conv with a shortcut: stochastic-depth probability


@dataclass
class Graph:
    network: str
    in_chans: int
    num_classes: int
    feat: int
    ops: list = field(default_factory=list)
    n_base_children: int = 9
    head_modules: list = field(default_factory=list)  # [("linear", i, in, out) | ("dropout", i, p)]


def _add_head(g, ops, new_t, cur, feat, num_classes, head, dropout):
    """Linear(feat->h0) ... Linear(h[-1]->num_classes), no activations; Dropout
    inserted by *list index* exactly as list.insert does (reference network.py:56-61)."""
    widths = [feat] + [int(h) for h in head] + [num_classes]
    mods = [("linear", widths[i], widths[i + 1]) for i in range(len(widths) - 1)]
    for idx, p in dropout:
        mods.insert(int(idx), ("dropout", float(p)))
    n_lin = sum(1 for m in mods if m[0] == "linear")
    seen = 0
    for i, m in enumerate(mods):
        d = new_t()
        if m[0] == "linear":
            seen += 1
            ops.append(Op(OP_LINEAR, f"head.{i}", "", m[1], m[2], 1, 1, 0, False, cur, d, -1, -1, seen == n_lin))
            g.head_modules.append(("linear", i, m[1], m[2]))
        else:
            ops.append(Op(OP_DROPOUT, f"head.{i}", "", 0, 0, 1, 1, 0, False, cur, d, -1, -1, False, m[1]))
            g.head_modules.append(("dropout", i, m[1]))
        cur = d
    return cur


def _build_efficientnet(network, num_classes, head, dropout, in_chans, stochastic_depth=0.2):
    """torchvision EfficientNet: children [features, avgpool, classifier]; the reference keeps
    [features, avgpool] as ``base`` and reads ``in_features`` off the classifier's Linear
    (network.py:50-55).  state_dict keys: base.0.<i>... for features[i]."""
    wm, dm = _EFFNETS[network]
    ch = lambda c: _make_divisible(c * wm)
    import math
    g = Graph(network, in_chans, num_classes, 4 * ch(320), n_base_children=2)
    ops = g.ops
    t = [0]

    def new_t():
        t[0] += 1
        return t[0]

    cur = new_t()
    ops.append(Op(OP_CONV, "base.0.0.0", "base.0.0.1", in_chans, ch(32), 3, 2, 1, ACT_SILU, 0, cur, -1, 0))
    # torchvision: sd_prob = stochastic_depth_prob * block index / number of blocks (train mode only: "row" mode
    # StochasticDepth on the block's branch before the shortcut add)
    total_blocks = sum(int(math.ceil(n * dm)) for (_, _, _, _, _, n) in _MBCONV)
    block_id = 0
    for si, (ratio, k, stride, cin, cout, n) in enumerate(_MBCONV):
        cin, cout, n = ch(cin), ch(cout), int(math.ceil(n * dm))
        for b in range(n):
            bi, bs = (cin, stride) if b == 0 else (cout, 1)
            pre = f"base.0.{si + 1}.{b}.block"
            exp = _make_divisible(bi * ratio)
            x_in, j = cur, 0
            if exp != bi:
                d = new_t()
                ops.append(Op(OP_CONV, f"{pre}.{j}.0", f"{pre}.{j}.1", bi, exp, 1, 1, 0, ACT_SILU, cur, d, -1, 0))
                cur, j = d, j + 1
            d = new_t()
            ops.append(Op(OP_DWCONV, f"{pre}.{j}.0", f"{pre}.{j}.1", exp, exp, k, bs, (k - 1) // 2, ACT_SILU, cur, d,
                          -1, 0))
            cur, j = d, j + 1
            d = new_t()
            ops.append(Op(OP_SE, f"{pre}.{j}", "", exp, exp, max(1, bi // 4), 1, 0, ACT_NONE, cur, d, -1, 0))
            cur, j = d, j + 1
            d = new_t()
            res = x_in if (bs == 1 and bi == cout) else -1
            sd = stochastic_depth * block_id / total_blocks if res >= 0 else 0.0
            ops.append(Op(OP_CONV, f"{pre}.{j}.0", f"{pre}.{j}.1", exp, cout, 1, 1, 0, ACT_NONE, cur, d, res, 0,
                          res >= 0, sd))
            cur = d
            block_id += 1
    d = new_t()
    ops.append(Op(OP_CONV, "base.0.8.0", "base.0.8.1", ch(320), 4 * ch(320), 1, 1, 0, ACT_SILU, cur, d, -1, 0))
    cur = d
    d = new_t()
    ops.append(Op(OP_GAVGPOOL, "", "", g.feat, g.feat, 0, 1, 0, False, cur, d, -1, 1))
    _add_head(g, ops, new_t, d, g.feat, num_classes, head, dropout)
    return g


def build_graph(network, num_classes, head=(256, 128), dropout=(), in_chans=3, stochastic_depth=0.2):
    """Mirror of ``TorchVisionNet.__init__`` (ResNet and EfficientNet-B0..B4 families).  ``stochastic_depth`` is
    torchvision's ``stochastic_depth_prob`` (0.2 for every EfficientNet variant; 0 turns the train-mode row dropping off)."""
    if network in _EFFNETS:
        return _build_efficientnet(network, num_classes, list(head), list(dropout), in_chans, stochastic_depth)
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

    if g.network in _EFFNETS:   # graph order == module order == state_dict order
        for op in g.ops:
            if op.kind in (OP_CONV, OP_DWCONV):
                cin1 = 1 if op.kind == OP_DWCONV else op.cin
                specs.append((f"{op.name}.weight", (op.cout, cin1, op.k, op.k), "conv_w"))
                bn(op.bn, op.cout, op.last_bn)
            elif op.kind == OP_SE:
                specs.append((f"{op.name}.fc1.weight", (op.k, op.cin, 1, 1), "se_w"))
                specs.append((f"{op.name}.fc1.bias", (op.k,), "se_b"))
                specs.append((f"{op.name}.fc2.weight", (op.cout, op.k, 1, 1), "se_w"))
                specs.append((f"{op.name}.fc2.bias", (op.cout,), "se_b"))
            elif op.kind == OP_LINEAR:
                specs.append((f"{op.name}.weight", (op.cout, op.cin), "fc_w_last" if op.last_bn else "fc_w"))
                specs.append((f"{op.name}.bias", (op.cout,), "fc_b"))
        return specs
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
        if op.kind in (OP_CONV, OP_MAXPOOL, OP_DWCONV):
            oh = (ih + 2 * op.pad - op.k) // op.stride + 1
            ow = (iw + 2 * op.pad - op.k) // op.stride + 1
            dims[op.dst] = (oh, ow)
            if op.kind == OP_CONV:
                total += 2 * oh * ow * op.cout * op.cin * op.k * op.k
            elif op.kind == OP_DWCONV:
                total += 2 * oh * ow * op.cout * op.k * op.k
        elif op.kind == OP_GAVGPOOL:
            dims[op.dst] = (1, 1)
        elif op.kind == OP_LINEAR:
            dims[op.dst] = (1, 1)
            total += 2 * op.cin * op.cout
        else:
            dims[op.dst] = dims.get(op.src, (1, 1))
    return total


_RESNET_CHILDREN = {"conv1": 0, "bn1": 1, "layer1": 4, "layer2": 5, "layer3": 6, "layer4": 7}


def backbone_key(network, key):
    """state_dict key of a torchvision backbone checkpoint (``resnet50-*.pth``: ``conv1.weight``,
    ``layer1.0.conv1.weight``, ``fc.weight`` ...; EfficientNet: ``features.1.0.block...``,
    ``classifier.1.weight``) -> key of the same tensor under ``TorchVisionNet.base`` (``base.<child>...``),
    or None for the classifier the reference drops (network.py:49-55)."""
    first, _, rest = key.partition(".")
    if network in _EFFNETS:
        return f"base.0.{rest}" if first == "features" else None
    if first in _RESNET_CHILDREN:
        return f"base.{_RESNET_CHILDREN[first]}.{rest}"
    return None
