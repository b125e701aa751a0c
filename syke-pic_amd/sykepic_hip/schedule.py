"""Freezing policy and learning-rate schedule around the training hot loop.

Same behaviour as the reference's ``sykepic/train/network.py``: ``freeze``
(:149) / ``recursive_freeze`` (:160) keep every BatchNorm trainable and
freeze the rest of the base; ``filter_params`` (:175) yields trainable non-BN
parameters; ``LRWarmup`` (:75-130) lowers the head lr at ``step_1`` and
unfreezes ``base[-2:]`` / ``base[:-2]`` into param groups 1 / 2 at
``step_2`` / ``step_3``.  They operate on the module views of ``HipNet``.
``ReduceLROnPlateau`` reproduces torch's scheduler as the reference calls it
(``train.py:159-161``, quirk Q3: ``verbose`` lands in ``threshold``).
"""


def make_trainable(module):
    for p in module.parameters():
        p.requires_grad = True
    module.train()


def make_untrainable(module):
    for p in module.parameters():
        p.requires_grad = False
    module.eval()


def recursive_freeze(module):
    kids = list(module.children())
    if kids:
        for k in kids:
            recursive_freeze(k)
    elif getattr(module, "is_bn", False):
        make_trainable(module)
    else:
        make_untrainable(module)


def freeze(module, n=None):
    """Freeze the children of `module` up to index n (all when n is None);
    BatchNorm layers stay trainable."""
    kids = list(module.children())
    upto = int(n) if n else len(kids)
    for k in kids[:upto]:
        recursive_freeze(k)
    for k in kids[upto:]:
        make_trainable(k)


def filter_params(module):
    """Trainable parameters of non-BatchNorm leaves."""
    kids = list(module.children())
    if kids:
        for k in kids:
            yield from filter_params(k)
    elif not getattr(module, "is_bn", False):
        for p in module.parameters():
            if p.requires_grad:
                yield p


class LRWarmup:
    def __init__(self, net, optimizer, factor_1=0.1, factor_2=0.5, step_1=5, step_2=15, step_3=30,
                 verbose=True):
        self.net, self.optimizer = net, optimizer
        self.factor_1, self.factor_2 = factor_1, factor_2
        self.step_1, self.step_2, self.step_3 = step_1, step_2, step_3
        self.verbose = verbose

    def _unfreeze_into(self, part, group, base_lr):
        make_trainable(part)
        g = self.optimizer.param_groups
        g[group]["params"] = list(filter_params(part))
        g[group]["lr"] = base_lr * self.factor_1

    def __call__(self, epoch):
        g = self.optimizer.param_groups
        if epoch == self.step_1:
            g[0]["lr"] *= self.factor_1
            done = 1
        elif epoch == self.step_2:
            self._unfreeze_into(self.net.base[-2:], 1, g[0]["lr"])
            g[0]["lr"] *= self.factor_2
            done = 2
        elif epoch == self.step_3:
            self._unfreeze_into(self.net.base[:-2], 2, g[1]["lr"])
            g[0]["lr"] *= self.factor_2
            done = 3
        else:
            return
        if hasattr(self.optimizer, "sync_groups"):
            self.optimizer.sync_groups()
        if self.verbose:
            print(f"[INFO] LRWarmup step {done} completed:\n{self.optimizer}")


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau semantics for
    (optimizer, mode, factor, patience, threshold) with threshold_mode="rel",
    cooldown 0, min_lr 0, eps 1e-8."""

    def __init__(self, optimizer, mode="min", factor=0.1, patience=10, threshold=1e-4):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.optimizer, self.mode, self.factor, self.patience = optimizer, mode, factor, patience
        self.threshold = float(threshold)  # True -> 1.0 under quirk Q3
        self.best = float("inf") if mode == "min" else -float("inf")
        self.num_bad_epochs = 0
        self.eps = 1e-8

    def _better(self, a):
        if self.mode == "min":
            return a < self.best * (1.0 - self.threshold)
        return a > self.best * (self.threshold + 1.0)

    def step(self, metric):
        current = float(metric)
        if self._better(current):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for group in self.optimizer.param_groups:
                old = float(group["lr"])
                new = max(old * self.factor, 0.0)
                if old - new > self.eps:
                    group["lr"] = new
            self.num_bad_epochs = 0
