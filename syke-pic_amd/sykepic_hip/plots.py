"""Cosmetic plots the training workflow writes (train-stat curves, class
distribution, batch collage).  Out of the hot path (reference
``sykepic/analyze/plot.py``); kept minimal so ``sykepic train`` leaves the same
files behind."""

import numpy as np


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def plot_stats(train_accs, train_losses, val_accs, val_losses, title=None, outfile=None, first_epoch=1,
               epoch_step=1):
    plt = _plt()
    fig, (ax1, ax2) = plt.subplots(2, 1, sharex=True, dpi=100, figsize=(12, 8.4))
    ticks = np.arange(0, len(train_accs), epoch_step)
    ax2.set_xticks(ticks)
    ax2.set_xticklabels(ticks + first_epoch)
    ax2.set_xlabel("Epoch")
    if title:
        ax1.set_title(title)
    for ax, tr, va, name in ((ax1, train_accs, val_accs, "Accuracy"), (ax2, train_losses, val_losses, "Loss")):
        ax.plot(tr, label="Training", lw=2)
        ax.plot(va, label="Validation", lw=2)
        ax.legend(loc="upper left")
        ax.set_ylabel(name)
    fig.tight_layout()
    if outfile:
        fig.savefig(outfile)
    plt.close(fig)


def dataset_distribution(data, save=None, size=(8.4, 12)):
    plt = _plt()
    items = sorted(sorted(data.distribution.items()), key=lambda kv: kv[1][0])
    fig, ax = plt.subplots(figsize=size)
    ax.barh([k for k, _ in items], [v[0] for _, v in items])
    ax.set_xlabel("images")
    fig.tight_layout()
    if save:
        fig.savefig(save)
    plt.close(fig)


def view_batch(loader, height, width, outfile=None):
    plt = _plt()
    batch = next(iter(loader))[0]
    fig, axes = plt.subplots(height, width, figsize=(width * 1.5, height * 1.5))
    for ax, img in zip(np.atleast_1d(axes).ravel(), batch):
        ax.imshow(img.permute(1, 2, 0).clamp(0, 1).numpy())
        ax.axis("off")
    fig.tight_layout()
    if outfile:
        fig.savefig(outfile)
    plt.close(fig)
