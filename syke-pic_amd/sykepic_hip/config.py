"""``train.ini`` / model-dir ``config.ini`` -> image shape, transform pipelines
and the network.  Same keys and defaults as the reference's
``sykepic/train/config.py`` (``get_img_shape`` :20, ``get_transforms`` :25,
``get_network`` :63); legacy files without a ``weights`` key are accepted."""

from configparser import NoOptionError

from . import preprocess as P


def get_img_shape(config):
    return tuple(int(i) for i in config.get("image", "shape").split(","))


def get_transforms(config, img_shape):
    augmentations = [a.strip() for a in config.get("image", "augmentations").split(",")]
    border = config.get("image", "border")
    train_t, eval_t = [P.Resize()], [P.Resize()]
    for aug in augmentations:
        if aug == "flip":
            train_t += [P.FlipHorizontal(), P.FlipVertical()]
        if aug == "translate":
            train_t.append(P.Translate())
        if aug == "rotate":
            train_t.append(P.Rotate(config.getint("image", "max_rotation")))
        if aug == "zoom":
            train_t.append(P.Zoom(tuple(float(i) for i in config.get("image", "zoom_range").split(","))))
        if aug == "brightness":
            train_t.append(P.ChangeBrightness(
                tuple(float(i) for i in config.get("image", "brightness_range").split(","))))
    train_t.append(P.ToTensor())
    eval_t.append(P.ToTensor())
    if config.getboolean("image", "imagenet_normalization"):
        # as in the reference, only the TRAIN pipeline is normalised (config.py:55-56)
        train_t.append(P.Normalize(P.IMAGENET_MEAN, P.IMAGENET_STD))
    return P.Compose(train_t, img_shape[1:], border), P.Compose(eval_t, img_shape[1:], border)


def get_network(config, num_classes, device=None, pretrained_ok=True):
    """The only constructor call site of the network (reference config.py:63-77).
    Returns the MI355X-native ``HipNet``; raises if no GPU / library."""
    from .net import HipNet
    network = config.get("model", "network")
    try:
        weights = config.get("model", "weights") or None
    except NoOptionError:
        weights = "DEFAULT"  # legacy model configs (quirk Q7): nothing is downloaded here
    head = [int(i) for i in config.get("model", "head").split(",")]
    dropout = []
    if config.get("model", "dropout"):
        for drop in config.get("model", "dropout").split(";"):
            idx, p = drop.split(",")
            dropout.append((int(idx), float(p)))
    # pretrained_ok=False: the caller loads a checkpoint next (prob.prepare_model) - skip the random initialisation
    return HipNet(network, num_classes, weights, head, dropout, device=device, init=pretrained_ok)
