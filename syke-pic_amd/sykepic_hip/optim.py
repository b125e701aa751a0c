"""Optimizer adapter: looks like ``getattr(torch.optim, name)([...])`` to the
code around the hot loop (``param_groups`` list of dicts with ``params`` and
``lr``, ``zero_grad()``, ``step()``, printable — reference
``sykepic/train/train.py:131-138``, ``network.py:101-128``) while the update
itself is one fused multi-tensor kernel launch inside ``libsykepic_hip.so``.
"""

from . import lib

# torch.optim's own defaults: the reference constructs `getattr(optim, name)(groups)` with nothing but `lr`
# (sykepic/train/train.py:131-138), so these are the hyper-parameters a reference run uses.
_DEFAULTS = {
    "SGD": dict(momentum=0.0, weight_decay=0.0),
    "Adam": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0),
    "AdamW": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01),
    "RMSprop": dict(alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0),
    "Adagrad": dict(lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10),
    "Adamax": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0),
    "NAdam": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, momentum_decay=4e-3),
    "RAdam": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0),
    "Adadelta": dict(rho=0.9, eps=1e-6, weight_decay=0.0),
    "ASGD": dict(lambd=1e-4, alpha=0.75, t0=1e6, weight_decay=0.0),
    "Rprop": dict(etas=(0.5, 1.2), step_sizes=(1e-6, 50.0)),
}
_KINDS = {"SGD": lib.OPT_SGD, "Adam": lib.OPT_ADAM, "AdamW": lib.OPT_ADAMW, "RMSprop": lib.OPT_RMSPROP,
          "Adagrad": lib.OPT_ADAGRAD, "Adamax": lib.OPT_ADAMAX, "NAdam": lib.OPT_NADAM, "RAdam": lib.OPT_RADAM,
          "Adadelta": lib.OPT_ADADELTA, "ASGD": lib.OPT_ASGD, "Rprop": lib.OPT_RPROP}


class HipOptimizer:
    def __init__(self, net, name, param_groups, **kw):
        if name not in _DEFAULTS:
            raise ValueError(f"optimizer {name!r} has no MI355X kernel (supported: {sorted(_DEFAULTS)}; LBFGS needs a "
                             "closure the reference's loop does not pass, SparseAdam needs sparse gradients)")
        self.net, self.name = net, name
        self.defaults = dict(_DEFAULTS[name])
        self.defaults.update(kw)
        if len(param_groups) > 3:
            raise ValueError("at most 3 param groups (the reference uses exactly 3)")
        self.param_groups = [dict(g) for g in param_groups]
        while len(self.param_groups) < 3:
            self.param_groups.append({"params": [], "lr": 0.0})
        for g in self.param_groups:
            g["params"] = list(g["params"])
        self.grad_scale = 1.0
        self.sync_groups()

    def sync_groups(self):
        """Push the group membership to the library (after LRWarmup edits)."""
        member = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.key in member:
                    raise ValueError("some parameters appear in more than one parameter group")
                member[p.key] = gi
        for p in self.net.parameters():
            self.net.set_param_group(p.key, member.get(p.key, -1))

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten by every forward_backward
        return None

    def step(self):
        d = lib.OptimDesc()
        d.kind = _KINDS[self.name]
        for i in range(3):
            d.lr[i] = float(self.param_groups[i]["lr"])
        b1, b2 = self.defaults.get("betas", (0.9, 0.999))
        d.beta1, d.beta2 = float(b1), float(b2)
        d.eps = float(self.defaults.get("eps", 1e-8))
        d.weight_decay = float(self.defaults.get("weight_decay", 0.0))
        d.momentum = float(self.defaults.get("momentum", 0.0))
        d.grad_scale = float(self.grad_scale)
        d.alpha = float(self.defaults.get("alpha", self.defaults.get("rho", 0.99)))
        d.momentum_decay = float(self.defaults.get("momentum_decay", 4e-3))
        d.lr_decay = float(self.defaults.get("lr_decay", 0.0))
        d.initial_accumulator_value = float(self.defaults.get("initial_accumulator_value", 0.0))
        if self.name == "ASGD":      # lambd travels in lr_decay, the power alpha in alpha (include/sykepic_hip.h)
            d.lr_decay = float(self.defaults["lambd"])
            d.alpha = float(self.defaults["alpha"])
            if float(self.defaults["t0"]) < 1e5:
                raise ValueError("ASGD: the averaged parameters (t0) are not kept on the MI355X path")
        elif self.name == "Rprop":   # etas in beta1 / beta2, step-size bounds in eps / alpha
            d.beta1, d.beta2 = (float(v) for v in self.defaults["etas"])
            d.eps, d.alpha = (float(v) for v in self.defaults["step_sizes"])
        self.net.optim_step(d)

    def __repr__(self):
        lines = [f"{self.name} ("]
        for i, g in enumerate(self.param_groups):
            lines.append(f"Parameter Group {i}")
            for k in sorted(self.defaults):
                lines.append(f"    {k}: {self.defaults[k]}")
            lines.append(f"    lr: {g['lr']}")
            lines.append(f"    params: {len(g['params'])} tensors")
        lines.append(")")
        return "\n".join(lines)
