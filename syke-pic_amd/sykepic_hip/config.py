"""``train.ini`` / model-dir ``config.ini``: one declarative table of every key the hot path reads.

The drop-in contract is the FILE FORMAT of the reference (``sykepic/train/config.py:20-77`` and the ini block of
``sykepic/train/train.py:17-125``): section and key names, value syntax, which keys may be absent.  Here that
contract is data - ``KEYS[section][key] = (parser, default)`` - and ``Settings(config).section.key`` parses a value
when it is first asked for, so a key that the chosen options never reach (``max_rotation`` without ``rotate``,
``[lr_warmup]`` factors with ``use = no``) may be missing, exactly as with the reference's statement-by-statement
reads; a missing REQUIRED key raises configparser's own ``NoOptionError`` / ``NoSectionError``.
"""

from configparser import NoOptionError, RawConfigParser
from pathlib import Path

from . import preprocess as P

REQUIRED = object()


def _csv(item):
    return lambda s: tuple(item(v) for v in s.split(","))


def _optional(item):
    return lambda s: item(s) if s else None


def _names(s):
    return [v.strip() for v in s.split(",")]


def _flag(s):
    try:
        return RawConfigParser.BOOLEAN_STATES[s.lower()]
    except KeyError:
        raise ValueError(f"Not a boolean: {s}") from None


def _index_prob_pairs(s):
    """``dropout = 1,0.5;2,0.3`` -> [(1, 0.5), (2, 0.3)] (list index in the head, probability)."""
    return [(int(i), float(p)) for i, p in (pair.split(",") for pair in s.split(";"))] if s else []


KEYS = {
    "dataset": {
        "path": (Path, REQUIRED), "split": (_csv(float), REQUIRED), "min_N": (_optional(int), REQUIRED),
        "max_N": (_optional(int), REQUIRED), "exclude": (_names, REQUIRED), "random_seed": (int, REQUIRED),
        "oversample_until": (_optional(int), ""), "oversample_with_decay": (_optional(float), ""),
        "external_test": (str, ""),
    },
    "image": {
        "shape": (_csv(int), REQUIRED), "batch_size": (int, REQUIRED), "num_workers": (int, REQUIRED),
        "augmentations": (_names, REQUIRED), "border": (str, REQUIRED), "max_rotation": (int, REQUIRED),
        "zoom_range": (_csv(float), REQUIRED), "brightness_range": (_csv(float), REQUIRED),
        "imagenet_normalization": (_flag, REQUIRED),
    },
    "model": {
        "network": (str, REQUIRED), "id": (str, REQUIRED), "path": (Path, REQUIRED), "exist_ok": (_flag, REQUIRED),
        # legacy model directories have no `weights` key (quirk Q7): "DEFAULT", and nothing is downloaded here
        "weights": (lambda s: s or None, "DEFAULT"),
        "head": (_csv(int), REQUIRED), "dropout": (_index_prob_pairs, REQUIRED),
    },
    "train": {
        "gpu": (_flag, REQUIRED), "max_epochs": (int, REQUIRED), "early_stop_patience": (int, REQUIRED),
        "learning_rate": (float, REQUIRED), "optimizer": (str, REQUIRED),
    },
    "lr_warmup": {
        "use": (_flag, REQUIRED), "factor_1": (float, REQUIRED), "factor_2": (float, REQUIRED), "step_1": (int, REQUIRED),
        "step_2": (int, REQUIRED), "step_3": (int, REQUIRED), "verbose": (_flag, REQUIRED),
    },
    "lr_reduction": {
        "use": (_flag, REQUIRED), "factor": (float, REQUIRED), "patience": (int, REQUIRED), "verbose": (_flag, REQUIRED),
    },
}


class _Section:
    def __init__(self, config, name):
        self._config, self._name = config, name

    def __getattr__(self, key):
        try:
            parser, default = KEYS[self._name][key]
        except KeyError:
            raise AttributeError(f"[{self._name}] has no key {key!r} in the settings table") from None
        if default is REQUIRED:
            raw = self._config.get(self._name, key)
        else:
            try:
                raw = self._config.get(self._name, key)
            except NoOptionError:
                raw = default
        return parser(raw)


class Settings:
    """``Settings(config).image.batch_size`` -> parsed value of ``[image] batch_size`` (see KEYS)."""

    def __init__(self, config):
        for name in KEYS:
            setattr(self, name, _Section(config, name))


# augmentation name in `[image] augmentations` -> the transforms it adds to the TRAIN pipeline (in this order)
AUGMENTATIONS = {
    "flip": lambda im: [P.FlipHorizontal(), P.FlipVertical()],
    "translate": lambda im: [P.Translate()],
    "rotate": lambda im: [P.Rotate(im.max_rotation)],
    "zoom": lambda im: [P.Zoom(im.zoom_range)],
    "brightness": lambda im: [P.ChangeBrightness(im.brightness_range)],
}


def get_img_shape(config):
    return Settings(config).image.shape


def get_transforms(config, img_shape):
    """(train, eval) ``Compose`` pipelines (reference config.py:25-60).  Unknown augmentation names are ignored and,
    as in the reference, only the TRAIN pipeline gets the ImageNet normalisation (:55-56)."""
    im = Settings(config).image
    extra = [t for name in im.augmentations for t in AUGMENTATIONS.get(name, lambda _: [])(im)]
    tail = [P.Normalize(P.IMAGENET_MEAN, P.IMAGENET_STD)] if im.imagenet_normalization else []
    size, border = img_shape[1:], im.border
    return (P.Compose([P.Resize()] + extra + [P.ToTensor()] + tail, size, border),
            P.Compose([P.Resize(), P.ToTensor()], size, border))


def get_network(config, num_classes, device=None, pretrained_ok=True):
    """The only constructor call site of the network (reference config.py:63-77).
    Returns the MI355X-native ``HipNet``; raises if no GPU / library."""
    from .net import HipNet
    mo = Settings(config).model
    # pretrained_ok=False: the caller loads a checkpoint next (prob.prepare_model) - skip the random initialisation
    return HipNet(mo.network, num_classes, mo.weights, list(mo.head), mo.dropout, device=device, init=pretrained_ok)
