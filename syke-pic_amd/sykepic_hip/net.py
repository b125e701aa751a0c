"""HipNet: the object ``get_network`` returns on an MI355X.

It answers every call the reference makes on its ``TorchVisionNet``
(``/root/reference/sykepic/train/network.py:11-72``; call sites listed in
SURVEY.md §8b): ``net.to``, ``net.eval``, ``net.train``, ``net(x)``,
``net.state_dict``, ``net.load_state_dict``, ``net.parameters``, ``net.base``
(iterable, sliceable, ``.children()``, ``.parameters()``), ``net.head``
(printable) — but all arithmetic runs in ``libsykepic_hip.so`` (hand-written
gfx950 kernels) through the C-ABI of ``include/sykepic_hip.h``.  torch is
only the container for device memory.  No CPU fallback.
"""

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import arch, lib

SOFTMAX_EXP = 1.3  # reference sykepic/compute/probability.py:18


class HipParam:
    """Stand-in for an ``nn.Parameter`` living inside the library."""

    def __init__(self, net, key, shape, is_bn):
        self._net, self.key, self.shape, self.is_bn = net, key, tuple(shape), is_bn
        self._requires_grad = True

    @property
    def requires_grad(self):
        return self._requires_grad

    @requires_grad.setter
    def requires_grad(self, flag):
        self._requires_grad = bool(flag)
        self._net._set_requires_grad(self.key, self._requires_grad)

    def numel(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def data(self):
        return self._net._read_tensor(self.key, self.shape, torch.float32)

    @property
    def grad(self):
        return self._net._read_grad(self.key, self.shape)

    def __repr__(self):
        return f"HipParam({self.key}, shape={self.shape}, requires_grad={self._requires_grad})"


class HipLeaf:
    """A leaf module (Conv2d / BatchNorm2d / Linear / parameter-free op)."""

    def __init__(self, name, kind, params, text):
        self.name, self.kind, self._params, self.text = name, kind, params, text
        self.training = True

    @property
    def is_bn(self):
        return self.kind == "bn"

    def children(self):
        return iter(())

    def parameters(self):
        return iter(self._params)

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def leaves(self):
        yield self

    def __repr__(self):
        return self.text


class HipSequential:
    """Ordered container of leaves / containers (``nn.Sequential`` view)."""

    def __init__(self, items, label="Sequential"):
        self._items, self._label = list(items), label
        self.training = True

    def children(self):
        return iter(self._items)

    def __iter__(self):
        return iter(self._items)

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return HipSequential(self._items[i], self._label)
        return self._items[i]

    def leaves(self):
        for it in self._items:
            yield from it.leaves()

    def parameters(self):
        for leaf in self.leaves():
            yield from leaf.parameters()

    def train(self, mode=True):
        self.training = bool(mode)
        for it in self._items:
            it.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def __repr__(self):
        body = "\n".join(f"  ({i}): " + repr(it).replace("\n", "\n  ") for i, it in enumerate(self._items))
        return f"{self._label}(\n{body}\n)"


class HipNet:
    def __init__(self, name, num_classes, weights="DEFAULT", head=(256, 128), dropout=(),
                 last_activation=None, device=None, init=True, stochastic_depth=0.2):
        # `weights` names pretrained torchvision weights; there is nothing to
        # download from here and a best_state.pth / training run overwrites
        # every tensor anyway (quirk Q7) — only None/"" vs. other is recorded.
        self.name, self.num_classes = name, int(num_classes)
        self.weights = weights
        self.last_activation = last_activation
        self.graph = arch.build_graph(name, num_classes, list(head), list(dropout), stochastic_depth=stochastic_depth)
        self._specs = arch.param_specs(self.graph)
        self._lib = lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipNet needs a ROCm GPU (MI355X); there is no CPU path")
        self.device = torch.device(device if device is not None else "cuda:0")
        idx = self.device.index or 0
        descs = (lib.LayerDesc * len(self.graph.ops))()
        for d, op in zip(descs, self.graph.ops):
            d.kind, d.cin, d.cout, d.k, d.stride, d.pad = op.kind, op.cin, op.cout, op.k, op.stride, op.pad
            d.relu, d.src, d.dst, d.res, d.child, d.p = int(op.relu), op.src, op.dst, op.res, op.child, op.p
            d.name, d.bn = op.name.encode(), op.bn.encode()
        handle = C.c_void_p()
        lib.check(self._lib.spk_model_create(descs, len(descs), self.graph.in_chans, self.num_classes,
                                             idx, C.byref(handle)))
        self._h = handle
        eps, momentum = arch.bn_params(name)
        if (eps, momentum) != (1e-5, 0.1):
            lib.check(self._lib.spk_model_set_bn(self._h, eps, momentum))
        self.training = True
        self._params = OrderedDict()
        self._build_views()
        self._stats = None
        # Random start: the seed is drawn from torch's global generator NOW (so `torch.manual_seed` before the
        # constructor reproduces the network, as for the reference), the tensors themselves are generated at
        # first use — a `load_state_dict` of a full checkpoint in between makes them unnecessary.
        self._init_seed = int(torch.empty((), dtype=torch.int64).random_().item()) if init else None
        if init:
            self._load_pretrained(weights)

    def _ensure_init(self):
        if getattr(self, "_init_seed", None) is not None:
            seed, self._init_seed = self._init_seed, None
            self.reset_parameters(seed)

    def _load_pretrained(self, weights):
        """``weights`` names torchvision's pretrained ImageNet weights in the reference (network.py:48).
        Nothing can be downloaded here; a path to a local torchvision checkpoint (e.g. ``resnet50-11ad3fa6.pth``)
        is loaded into ``base``, anything else keeps the random start (with a warning)."""
        import logging
        import os
        if not weights:
            return
        if isinstance(weights, (str, os.PathLike)) and os.path.isfile(weights):
            ckpt = torch.load(weights, map_location="cpu")
            mapped = {}
            for k, v in ckpt.items():
                nk = arch.backbone_key(self.name, k)
                if nk is not None:
                    mapped[nk] = v
            missing = [k for k, _, _ in self._specs if k.startswith("base.") and k not in mapped]
            if missing:
                raise RuntimeError(f"{weights}: not a {self.name} backbone checkpoint (missing {missing[:3]} ...)")
            self.load_state_dict(mapped, strict=False)
            return
        logging.getLogger("sykepic_hip").warning(
            "pretrained weights %r cannot be downloaded here: %s starts from a random initialisation "
            "(give the path of a local torchvision checkpoint as `weights` to load one)", weights, self.name)

    def reset_parameters(self, seed=None):
        """Random initialisation with the distributions torchvision / torch.nn give a freshly constructed
        ``TorchVisionNet`` (reference network.py:48 with ``weights=None``): convolutions
        ``kaiming_normal_(mode="fan_out", nonlinearity="relu")``, BatchNorm weight 1 / bias 0 / mean 0 / var 1,
        squeeze-excitation biases 0, head ``Linear`` layers torch's default (``kaiming_uniform_(a=sqrt(5))``
        = U(+-1/sqrt(fan_in)) for weight and bias).  Drawn from torch's global CPU generator, so
        ``torch.manual_seed`` makes it reproducible as it does for the reference.  Pretrained ImageNet
        weights (``weights="DEFAULT"``) cannot be downloaded here: the same random start is used and a
        checkpoint (`load_state_dict`) overwrites it (quirk Q7)."""
        self._init_seed = None
        gen = None
        if seed is not None:
            gen = torch.Generator()
            gen.manual_seed(seed)
        sd = {}
        for key, shape, kind in self._specs:
            if kind in ("conv_w", "se_w"):
                fan_out = shape[0] * shape[2] * shape[3]
                sd[key] = torch.randn(shape, generator=gen) * (2.0 / fan_out) ** 0.5
            elif kind in ("bn_w", "bn_w_last", "bn_var"):
                sd[key] = torch.ones(shape)
            elif kind in ("bn_b", "bn_mean", "se_b"):
                sd[key] = torch.zeros(shape)
            elif kind == "bn_nbt":
                sd[key] = torch.zeros(shape, dtype=torch.int64)
            elif kind in ("fc_w", "fc_w_last"):
                bound = 1.0 / shape[1] ** 0.5
                sd[key] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
                fan_in = shape[1]
            elif kind == "fc_b":
                sd[key] = (torch.rand(shape, generator=gen) * 2 - 1) * (1.0 / fan_in ** 0.5)
            else:
                raise ValueError(f"no initialiser for parameter kind {kind!r}")
        self.load_state_dict(sd)

    def set_precision(self, split_weights=3, precise_residual=False, bf16=False):
        """Eval-path precision knobs (see include/sykepic_hip.h).  split_weights = 5 ("calibrated"): every conv as ONE
        fp16 product with zero-sum rounded weights; needs `calibrate` / `set_act_means` first."""
        if split_weights == "calibrated":
            split_weights = 5
        lib.check(self._lib.spk_model_set_infer_dtype(self._h, int(bool(bf16))))
        lib.check(self._lib.spk_model_set_precision(self._h, int(split_weights), int(bool(precise_residual))))
        return self

    # ---- calibrated single-pass mode (csrc/zero_sum.hip) ----
    def calibrate(self, x, reset=True):
        """Per-channel means of every conv's input on the representative batch `x` (as `forward` takes it), measured
        with the most accurate mode; further calls with reset=False accumulate.  Does not change the precision mode:
        `set_precision("calibrated")` (or `zero_sum(True)` for the un-split convs of another mode) uses the means."""
        self._ensure_init()
        x, n, h, w, layout, dtype = self._prep(x)
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            lib.check(self._lib.spk_model_calibrate_act_means(self._h, C.c_void_p(x.data_ptr()), n, h, w, layout, dtype,
                                                              int(bool(reset))))
        return self

    def act_means(self):
        """The calibrated means as one flat float32 tensor (what `act_means.pth` stores next to `best_state.pth`)."""
        out = torch.empty(int(self._lib.spk_model_act_means_size(self._h)), dtype=torch.float32)
        lib.check(self._lib.spk_model_get_act_means(self._h, C.c_void_p(out.data_ptr()), out.numel()))
        return out

    def set_act_means(self, means):
        """Restore stored means (None forgets them)."""
        if means is None:
            lib.check(self._lib.spk_model_set_act_means(self._h, None, 0))
            return self
        t = torch.as_tensor(means, dtype=torch.float32).contiguous().cpu()
        lib.check(self._lib.spk_model_set_act_means(self._h, C.c_void_p(t.data_ptr()), t.numel()))
        return self

    def zero_sum(self, on=True):
        """Zero-sum round the un-split convs of the CURRENT split mode against the calibrated means (diagnostics)."""
        lib.check(self._lib.spk_model_set_zero_sum(self._h, int(bool(on))))
        return self

    def num_fp8_blocks(self):
        """MBConv blocks that qualify for the e4m3 path (those with an expand conv)."""
        return int(self._lib.spk_model_num_fp8_blocks(self._h))

    def set_fp8(self, on=True, calibration_batch=None, blocks=None):
        """fp8 (e4m3) eval mode of the EfficientNet MBConv blocks (BASELINE config 5; include/sykepic_hip.h).
        `calibration_batch`: a representative image batch (as `forward` takes it) whose activation ranges set the
        tensor scales; required before the first fp8 forward.  `blocks`: one flag per qualifying block (graph order),
        True = e4m3, False = keep that block fp16; "all" = every block; None = the library default (blocks with a
        shortcut only)."""
        self._ensure_init()
        if isinstance(blocks, str) and blocks == "all":
            blocks = [True] * self.num_fp8_blocks()
        if blocks is not None:
            flags = (C.c_ubyte * len(blocks))(*[1 if b else 0 for b in blocks])
            lib.check(self._lib.spk_model_set_fp8_blocks(self._h, flags, len(blocks)))
        lib.check(self._lib.spk_model_set_fp8(self._h, int(bool(on))))
        if on and calibration_batch is not None:
            x, n, h, w, layout, dtype = self._prep(calibration_batch)
            with torch.cuda.device(self.device):
                lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
                lib.check(self._lib.spk_model_calibrate_fp8(self._h, C.c_void_p(x.data_ptr()), n, h, w, layout, dtype))
        return self

    def set_seed(self, seed):
        """Seed of the Dropout masks drawn by training steps (reproducible runs)."""
        lib.check(self._lib.spk_model_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))
        return self

    def conv_ops(self):
        """[(op index, conv name)] of the graph's convolutions, in execution order."""
        return [(i, op.name) for i, op in enumerate(self.graph.ops) if op.kind == arch.OP_CONV]

    def set_split_ops(self, names):
        """Carry exactly the named convs as hi+lo fp16 halves in the eval path
        (spk_model_set_split_ops); `set_precision` switches back to a mode."""
        names = set(names)
        known = {n for _, n in self.conv_ops()}
        if names - known:
            raise KeyError(f"not convolutions of this network: {sorted(names - known)}")
        flags = bytes(1 if (op.kind == arch.OP_CONV and op.name in names) else 0 for op in self.graph.ops)
        lib.check(self._lib.spk_model_set_infer_dtype(self._h, 0))
        lib.check(self._lib.spk_model_set_split_ops(self._h, flags, len(flags)))
        return self

    # ---- lifetime ----
    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.spk_model_destroy(h)
            except Exception:
                pass

    # ---- module views for freeze / LRWarmup ----
    def _build_views(self):
        g = self.graph
        shapes = {k: s for k, s, _ in self._specs}

        def P(key, is_bn):
            p = HipParam(self, key, shapes[key], is_bn)
            self._params[key] = p
            return p

        def conv_leaf(op):
            return HipLeaf(op.name, "conv", [P(op.name + ".weight", False)],
                           f"Conv2d({op.cin}, {op.cout}, kernel_size=({op.k}, {op.k}), "
                           f"stride=({op.stride}, {op.stride}), padding=({op.pad}, {op.pad}), bias=False)")

        def bn_leaf(op):
            return HipLeaf(op.bn, "bn", [P(op.bn + ".weight", True), P(op.bn + ".bias", True)],
                           f"BatchNorm2d({op.cout}, eps=1e-05, momentum=0.1)")

        if g.n_base_children == 2:   # EfficientNet: base = [features, avgpool]; flat leaf list under features
            leaves = []
            for op in g.ops:
                if op.kind in (arch.OP_CONV, arch.OP_DWCONV):
                    grp = f", groups={op.cin}" if op.kind == arch.OP_DWCONV else ""
                    leaves.append(HipLeaf(op.name, "conv", [P(op.name + ".weight", False)],
                                          f"Conv2d({op.cin}, {op.cout}, kernel_size=({op.k}, {op.k}), "
                                          f"stride=({op.stride}, {op.stride}), padding=({op.pad}, {op.pad}){grp}, bias=False)"))
                    leaves.append(bn_leaf(op))
                elif op.kind == arch.OP_SE:
                    for fc, (a, b) in (("fc1", (op.cin, op.k)), ("fc2", (op.k, op.cout))):
                        leaves.append(HipLeaf(f"{op.name}.{fc}", "conv",
                                              [P(f"{op.name}.{fc}.weight", False), P(f"{op.name}.{fc}.bias", False)],
                                              f"Conv2d({a}, {b}, kernel_size=(1, 1), stride=(1, 1))"))
            self.base = HipSequential([HipSequential(leaves),
                                       HipLeaf("base.1", "avgpool", [], "AdaptiveAvgPool2d(output_size=1)")])
            HipNet._finish_views(self, g, P)
            return
        children = [None] * g.n_base_children
        convs = [op for op in g.ops if op.kind == arch.OP_CONV]
        stem = convs[0]
        children[0], children[1] = conv_leaf(stem), bn_leaf(stem)
        children[2] = HipLeaf("base.2", "relu", [], "ReLU(inplace=True)")
        children[3] = HipLeaf("base.3", "maxpool", [], "MaxPool2d(kernel_size=3, stride=2, padding=1)")
        for child in range(4, 8):
            blocks = OrderedDict()
            for op in convs:
                if op.child != child:
                    continue
                blk = op.name.split(".downsample.")[0] if ".downsample." in op.name else op.name.rsplit(".", 1)[0]
                blocks.setdefault(blk, []).append(op)
            items = []
            for blk, ops in blocks.items():
                main = [o for o in ops if ".downsample." not in o.name]
                ds = [o for o in ops if ".downsample." in o.name]
                leaves = []
                for o in main:
                    leaves += [conv_leaf(o), bn_leaf(o)]
                leaves.append(HipLeaf(blk + ".relu", "relu", [], "ReLU(inplace=True)"))
                for o in ds:
                    leaves.append(HipSequential([conv_leaf(o), bn_leaf(o)]))
                items.append(HipSequential(leaves, "Block"))
            children[child] = HipSequential(items)
        children[8] = HipLeaf("base.8", "avgpool", [], "AdaptiveAvgPool2d(output_size=1)")
        self.base = HipSequential(children)
        HipNet._finish_views(self, g, P)

    def _finish_views(self, g, P):
        head_items = []
        for mod in g.head_modules:
            if mod[0] == "linear":
                _, i, fin, fout = mod
                head_items.append(HipLeaf(f"head.{i}", "linear",
                                          [P(f"head.{i}.weight", False), P(f"head.{i}.bias", False)],
                                          f"Linear(in_features={fin}, out_features={fout}, bias=True)"))
            else:
                head_items.append(HipLeaf(f"head.{mod[1]}", "dropout", [], f"Dropout(p={mod[2]}, inplace=False)"))
        self.head = HipSequential(head_items)
        # state_dict order for parameters()
        ordered = OrderedDict()
        for k, _, kind in self._specs:
            if k in self._params:
                ordered[k] = self._params[k]
        self._params = ordered

    # ---- nn.Module surface ----
    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("HipNet lives on the GPU; the CPU path is the reference itself")
        if (dev.index or 0) != (self.device.index or 0):
            raise RuntimeError("HipNet cannot migrate between GPUs; create it on the target device")
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        self.base.train(mode)
        self.head.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return iter(self._params.values())

    def named_parameters(self):
        return iter(self._params.items())

    def children(self):
        return iter((self.base, self.head))

    # ---- tensors in / out ----
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _prep(self, x):
        if not isinstance(x, torch.Tensor):
            raise TypeError("expected a torch.Tensor batch")
        if x.device != self.device:
            x = x.to(self.device, non_blocking=True)
        if x.dtype == torch.uint8:
            # NHWC uint8 image batch (GPU preprocessing path)
            if x.dim() != 4 or x.shape[3] != self.graph.in_chans:
                raise ValueError(f"uint8 batch must be [N,H,W,{self.graph.in_chans}], got {tuple(x.shape)}")
            x = x.contiguous()
            return x, x.shape[0], x.shape[1], x.shape[2], lib.LAYOUT_NHWC, lib.DTYPE_U8
        if x.dim() != 4 or x.shape[1] != self.graph.in_chans:
            raise ValueError(f"batch must be [N,{self.graph.in_chans},H,W], got {tuple(x.shape)}")
        x = x.float().contiguous()
        return x, x.shape[0], x.shape[2], x.shape[3], lib.LAYOUT_NCHW, lib.DTYPE_F32

    def forward(self, x, softmax_base=0.0):
        """Eval-mode forward.  softmax_base <= 0: raw logits."""
        self._ensure_init()
        x, n, h, w, layout, dtype = self._prep(x)
        out = torch.empty((n, self.num_classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            lib.check(self._lib.spk_forward_infer(self._h, C.c_void_p(x.data_ptr()), n, h, w, layout,
                                                  dtype, float(softmax_base), C.c_void_p(out.data_ptr())))
        return out

    def __call__(self, x):
        if self.training:
            raise RuntimeError(
                "HipNet in train mode has no separate forward: the fused "
                "forward+loss+backward is net.forward_backward(x, y) (see sykepic_hip.train)")
        out = self.forward(x)
        if self.last_activation:
            out = getattr(torch.nn.functional, self.last_activation)(out, dim=1)
        return out

    def probabilities(self, x, base=SOFTMAX_EXP):
        """softmax(net(x) * ln(base)) fused on the GPU (net_pass body)."""
        return self.forward(x, softmax_base=base)

    def _stats_buf(self):
        if self._stats is None:
            self._stats = torch.zeros(2, dtype=torch.float32, device=self.device)
        return self._stats

    def reset_stats(self):
        self._stats_buf().zero_()

    def read_stats(self):
        """(sum of loss*n, number correct) accumulated since reset_stats();
        one device sync per call instead of the reference's two per step."""
        s = self._stats_buf().tolist()
        return float(s[0]), float(s[1])

    def eval_step(self, x, y, want_logits=False):
        self._ensure_init()
        x, n, h, w, layout, dtype = self._prep(x)
        y = y.to(self.device, dtype=torch.int64).contiguous()
        logits = torch.empty((n, self.num_classes), dtype=torch.float32, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            lib.check(self._lib.spk_eval_step(
                self._h, C.c_void_p(x.data_ptr()), n, h, w, layout, dtype, C.c_void_p(y.data_ptr()),
                C.c_void_p(self._stats_buf().data_ptr()),
                C.c_void_p(logits.data_ptr()) if want_logits else None))
        return logits

    def forward_backward(self, x, y, want_logits=False):
        """zero_grad + train-mode forward + CrossEntropyLoss + backward
        (reference sykepic/train/train.py:239-242) in one library call."""
        self._ensure_init()
        x, n, h, w, layout, dtype = self._prep(x)
        y = y.to(self.device, dtype=torch.int64).contiguous()
        logits = torch.empty((n, self.num_classes), dtype=torch.float32, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            try:
                lib.check(self._lib.spk_train_forward_backward(
                    self._h, C.c_void_p(x.data_ptr()), n, h, w, layout, dtype, C.c_void_p(y.data_ptr()),
                    C.c_void_p(self._stats_buf().data_ptr()),
                    C.c_void_p(logits.data_ptr()) if want_logits else None))
            except RuntimeError as e:
                if "Expected more than 1 value per channel" in str(e):   # torch raises ValueError for this
                    raise ValueError(str(e).split(": ", 1)[-1]) from None
                raise
        return logits

    def optim_step(self, desc):
        self._ensure_init()
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            lib.check(self._lib.spk_optim_step(self._h, C.byref(desc)))

    def set_grad_ready_callback(self, cb, comm_stream, buckets):
        """Install (cb = a lib.GRAD_READY_FN) or remove (cb = None) the gradient-slice callback of the training
        step (data-parallel overlap; include/sykepic_hip.h)."""
        self._ensure_init()
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_grad_ready_callback(
                self._h, C.cast(cb, C.c_void_p) if cb is not None else None, None,
                C.c_void_p(int(comm_stream)) if comm_stream else None, int(buckets)))

    def grad_buffer(self):
        """(device pointer, numel) of the flat fp32 gradient buffer."""
        ptr, n = C.c_void_p(), C.c_int64()
        lib.check(self._lib.spk_model_grad_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    # ---- state_dict ----
    def _set_requires_grad(self, key, flag):
        lib.check(self._lib.spk_model_set_requires_grad(self._h, key.encode(), int(flag)))

    def set_param_group(self, key, group):
        lib.check(self._lib.spk_model_set_param_group(self._h, key.encode(), int(group)))

    def _read_tensor(self, key, shape, dtype):
        self._ensure_init()
        if dtype == torch.int64:
            buf = np.zeros((), dtype=np.int64)
        else:
            buf = np.empty(shape, dtype=np.float32)
        lib.check(self._lib.spk_model_read_param(self._h, key.encode(), C.c_void_p(buf.ctypes.data),
                                                 int(buf.size)))
        return torch.from_numpy(buf) if buf.shape else torch.tensor(int(buf), dtype=torch.int64)

    def _read_grad(self, key, shape):
        buf = np.empty(shape, dtype=np.float32)
        lib.check(self._lib.spk_model_read_grad(self._h, key.encode(), C.c_void_p(buf.ctypes.data),
                                                int(buf.size)))
        return torch.from_numpy(buf)

    def state_dict(self):
        """CPU tensors, NCHW fp32 (+ int64 counters): the on-disk contract of
        best_state.pth (reference train.py:300)."""
        self._ensure_init()
        sd = OrderedDict()
        for key, shape, kind in self._specs:
            sd[key] = self._read_tensor(key, shape, torch.int64 if kind == "bn_nbt" else torch.float32)
        return sd

    def load_state_dict(self, state, strict=True):
        want = {k: (s, kind) for k, s, kind in self._specs}
        missing = [k for k in want if k not in state]
        unexpected = [k for k in state if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for HipNet: missing keys {missing}, "
                               f"unexpected keys {unexpected}")
        if missing:
            self._ensure_init()     # a partial load keeps the random start of the other tensors
        else:
            self._init_seed = None  # a full checkpoint replaces the pending random start
        for k, v in state.items():
            if k not in want:
                continue
            shape, kind = want[k]
            t = v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(v)
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(t.shape)} "
                                   f"from checkpoint, the shape in current model is {tuple(shape)}")
            a = np.ascontiguousarray(t.numpy().astype(np.int64 if kind == "bn_nbt" else np.float32))
            lib.check(self._lib.spk_model_load_param(self._h, k.encode(), C.c_void_p(a.ctypes.data),
                                                     int(a.size)))
        return self

    def read_activation(self, tensor_id, n, shape):
        """Test hook: activation `tensor_id` of the last forward as a CPU
        float32 tensor of `shape` ([n,c,h,w] or [n,c])."""
        buf = np.empty(shape, dtype=np.float32)
        lib.check(self._lib.spk_model_read_activation(self._h, int(tensor_id), int(n),
                                                      C.c_void_p(buf.ctypes.data), int(buf.size)))
        return torch.from_numpy(buf)

    def read_activation_grad(self, tensor_id, n, shape):
        buf = np.empty(shape, dtype=np.float32)
        lib.check(self._lib.spk_model_read_activation_grad(self._h, int(tensor_id), int(n),
                                                           C.c_void_p(buf.ctypes.data), int(buf.size)))
        return torch.from_numpy(buf)

    def profile_train(self, x, y, iters=3):
        """[(phase, ms per step, algorithmic FLOPs, launches per step)]."""
        self._ensure_init()
        x, n, h, w, layout, dtype = self._prep(x)
        y = y.to(self.device, dtype=torch.int64).contiguous()
        recs = (lib.LayerTime * 32)()
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            cnt = self._lib.spk_model_profile_train(self._h, C.c_void_p(x.data_ptr()), n, h, w, layout, dtype,
                                                    C.c_void_p(y.data_ptr()),
                                                    C.c_void_p(self._stats_buf().data_ptr()), iters, recs, 32)
        if cnt < 0:
            lib.check(cnt)
        return [(r.name.decode(), r.ms, r.flops, r.bytes) for r in recs[:cnt]]

    def profile_layers(self, x, iters=5):
        self._ensure_init()
        x, n, h, w, layout, dtype = self._prep(x)
        cap = len(self.graph.ops) + 4
        recs = (lib.LayerTime * cap)()
        with torch.cuda.device(self.device):
            lib.check(self._lib.spk_model_set_stream(self._h, self._stream()))
            cnt = self._lib.spk_model_profile_infer(self._h, C.c_void_p(x.data_ptr()), n, h, w, layout,
                                                    dtype, iters, recs, cap)
        if cnt < 0:
            lib.check(cnt)
        return [(r.name.decode(), r.ms, r.flops, r.bytes) for r in recs[:cnt]]
