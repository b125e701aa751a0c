"""Labelled data for `sykepic train`: per-class split, label ids, optional
oversampling, data loaders.  Behavioural mirror of the reference's
``sykepic/train/data.py`` (``ModelData`` :17-192, ``ImageDataset`` :195-231,
``list_files`` :234, ``auto_id`` :278, ``oversample`` :297,
``combined_shuffle`` :320, ``extra_eval_dataloader`` :329): same seeded
``random`` call sequence, so a given ``random_seed`` yields the same split;
class ids are the alphabetical order of class directory names (what
sklearn's LabelEncoder produces).  PNGs are read with PIL instead of cv2."""

import os
import random
from itertools import groupby
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import pngio


class LabelIndex:
    """The part of sklearn's LabelEncoder the workflow uses."""

    def fit(self, labels):
        self.classes_ = np.array(sorted(set(labels)))
        self._ids = {c: i for i, c in enumerate(self.classes_)}
        return self

    def transform(self, labels):
        try:
            return np.array([self._ids[label] for label in labels], dtype=np.int64)
        except KeyError as e:
            raise ValueError(f"y contains previously unseen labels: {e}") from None

    def inverse_transform(self, ids):
        return self.classes_[np.asarray(ids, dtype=np.int64)]


def list_files(root_dir, extension, min_N=None, max_N=None, exclude=(), random_seed=24):
    if not isinstance(extension, list):
        extension = [extension]
    for dirpath, _, filenames in os.walk(root_dir):
        dirpath = Path(dirpath)
        if dirpath.name in exclude:
            continue
        if min_N and len(filenames) < min_N:
            continue
        if max_N and len(filenames) > max_N:
            random.seed(random_seed)
            random.shuffle(filenames)
            filenames = filenames[:max_N]
        for filename in filenames:
            path = dirpath / filename
            if path.suffix in extension:
                yield path.resolve()


def auto_id(name, directory):
    best = 0
    directory = Path(directory)
    if directory.is_dir():
        for path in directory.glob(f"{name}_*"):
            if path.is_dir():
                best = max(best, int(path.name.split("_")[-1]))
    return best + 1


def oversample(x, y, until=None, decay=None):
    if not until and not decay:
        raise ValueError("Must provide either 'until' or 'decay'")
    if not until:
        until = int((1 + 1 * decay ** len(x)) * len(x))
    over_x, over_y, i = [], [], 0
    while len(x) + len(over_x) < until:
        over_x.append(x[i])
        over_y.append(y[i])
        i = (i + 1) % len(x)
    return over_x, over_y


def combined_shuffle(list1, list2, random_seed=24):
    random.seed(random_seed)
    pairs = list(zip(list1, list2))
    random.shuffle(pairs)
    return zip(*pairs)


class ImageDataset(Dataset):
    """1- or 3-channel PNG dataset; yields (tensor, label) or (tensor, path)."""

    def __init__(self, paths, labels=None, transform=None, num_chans=3, num_classes=None):
        self.paths, self.labels, self.transform = paths, labels, transform
        self.num_chans, self.num_classes = num_chans, num_classes

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        path = self.paths[idx]
        img = pngio.read_image(path, self.num_chans)
        if self.transform:
            img = self.transform(img)
        if self.labels is not None:
            return img, int(self.labels[idx])
        return img, str(path)


class ModelData:
    def __init__(self, dataset, split, min_N, max_N, exclude=(), random_seed=24):
        self.dataset = Path(dataset)
        self.split, self.min_N, self.max_N = split, min_N, max_N
        self.exclude, self.random_seed = list(exclude), random_seed
        self.oversampled = False
        self._init_paths()
        self._init_labels()

    def _init_paths(self):
        three = len(self.split) == 3
        train_split, val_split = self.split[0], self.split[1]
        self.train_x, self.val_x = [], []
        self.test_x = [] if three else None
        self.distribution = {}
        for class_dir in self.dataset.iterdir():
            paths = sorted(list_files(class_dir, ".png", self.min_N, self.max_N, self.exclude, self.random_seed))
            if not paths:
                continue
            random.seed(self.random_seed)
            random.shuffle(paths)
            train_stop = int(round(len(paths) * train_split))
            val_stop = train_stop + int(round(len(paths) * val_split))
            train = paths[:train_stop]
            msg = (f"'{class_dir.name}' doesn't have enough samples ({len(paths)})."
                   " Consider using another min_N or split value.")
            if not three:
                val = paths[train_stop:]
                assert train and val, msg
                self.distribution[class_dir.name] = [len(paths), len(train), len(val)]
            else:
                val, test = paths[train_stop:val_stop], paths[val_stop:]
                assert train and val and test, msg
                self.distribution[class_dir.name] = [len(paths), len(train), len(val), len(test)]
                self.test_x.extend(test)
                random.seed(self.random_seed)
                random.shuffle(self.test_x)
            self.train_x.extend(train)
            self.val_x.extend(val)
        random.seed(self.random_seed)
        random.shuffle(self.train_x)
        random.seed(self.random_seed)
        random.shuffle(self.val_x)

    def _init_labels(self):
        train_labels = [p.parent.name for p in self.train_x]
        self.le = LabelIndex().fit(train_labels)
        self.train_y = list(self.le.transform(train_labels))
        self.val_y = list(self.le.transform([p.parent.name for p in self.val_x]))
        if self.test_x:
            self.test_y = list(self.le.transform([p.parent.name for p in self.test_x]))

    def save(self, out_dir):
        out_dir = Path(out_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
        header = "class,total,train,validation" + (",test" if self.test_x else "")
        if self.oversampled:
            header += ",oversampled"
        rows = sorted(sorted(self.distribution.items()), key=lambda kv: kv[1][0], reverse=True)
        body = "".join(f"\n{name}," + ",".join(str(i) for i in vals) for name, vals in rows)
        (out_dir / "class_distribution.csv").write_text(header + body)
        (out_dir / "class_names.txt").write_text("\n".join(self.le.classes_))

    def oversample(self, until, decay):
        pairs = sorted(zip(self.train_x, self.train_y), key=lambda xy: xy[1])
        self.over_x, self.over_y = [], []
        for key, group in groupby(pairs, lambda xy: xy[1]):
            x, y = (list(t) for t in zip(*group))
            ox, oy = oversample(x, y, until, decay)
            name = self.le.inverse_transform([key])[0]
            self.distribution[name].append(len(ox))
            self.distribution[name][1] += len(ox)
            self.over_x.extend(ox)
            self.over_y.extend(oy)
        self.oversampled = True

    def set_data_loaders(self, batch_size, num_workers, train_transform, eval_transform, num_chans=3,
                         rank=0, world=1, device=None):
        """device: run the transforms (resize, border, augmentations) on that GPU for whole batches
        (`gpu_augment.GpuLoader`) where the pipeline allows it; None: per-image host transforms in DataLoader
        workers, as the reference (data.py:165-183)."""
        self.batch_size, self.num_workers = batch_size, num_workers
        self.train_transform, self.eval_transform, self.num_chans = train_transform, eval_transform, num_chans
        if self.oversampled:
            train_x, train_y = combined_shuffle(self.train_x + self.over_x, self.train_y + self.over_y,
                                                self.random_seed)
            train_x, train_y = list(train_x), list(train_y)
        else:
            train_x, train_y = self.train_x, self.train_y
        n_cls = len(self.le.classes_)
        train_data = ImageDataset(train_x, train_y, train_transform, num_chans, n_cls)
        val_x, val_y = self.val_x, self.val_y
        if world > 1:   # each rank validates a contiguous shard; train_net sums the counters over the ranks
            from .dp import shard_range
            b, e = shard_range(len(val_x), rank, world)
            val_x, val_y = val_x[b:e], val_y[b:e]
        val_data = ImageDataset(val_x, val_y, eval_transform, num_chans, n_cls)
        sampler = ShardedShuffle(len(train_data), rank, world, self.random_seed) if world > 1 else None
        if device is not None:
            from . import gpu_augment
            if gpu_augment.supported(train_transform, num_chans) and gpu_augment.supported(eval_transform, num_chans):
                G = gpu_augment.GpuLoader
                kw = {"num_chans": num_chans, "workers": num_workers if num_workers and num_workers > 0 else None}
                self.train_loader = G(train_x, train_y, train_transform, batch_size, device, shuffle=sampler is None,
                                      sampler=sampler, **kw)
                # validation / test passes are short and rare: their decode pools live for one pass only
                self.val_loader = G(val_x, val_y, eval_transform, batch_size, device, persistent=False, **kw)
                if self.test_x:
                    self.test_loader = G(self.test_x, self.test_y, eval_transform, batch_size, device, persistent=False,
                                         **kw)
                return
        self.train_loader = DataLoader(train_data, batch_size, shuffle=sampler is None, sampler=sampler,
                                       num_workers=num_workers)
        self.val_loader = DataLoader(val_data, batch_size, num_workers=num_workers)
        if self.test_x:
            test_data = ImageDataset(self.test_x, self.test_y, eval_transform, num_chans, n_cls)
            self.test_loader = DataLoader(test_data, batch_size, num_workers=num_workers)


class ShardedShuffle(torch.utils.data.Sampler):
    """``shuffle=True`` semantics under data parallelism: every epoch one
    global permutation (same on all ranks), padded by wrap-around to a
    multiple of the world size, rank r takes elements r, r+world, ..."""

    def __init__(self, n, rank, world, seed=0):
        self.n, self.rank, self.world, self.seed, self.epoch = n, rank, world, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        from .dp import shard_indices
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + self.epoch)
        order = torch.randperm(self.n, generator=g).tolist()
        self.epoch += 1
        return iter(shard_indices(order, self.rank, self.world))

    def __len__(self):
        return (self.n + self.world - 1) // self.world


def extra_eval_dataloader(data_dir, model_data, exclude=(), random_seed=24):
    x = sorted(list_files(data_dir, ".png", exclude=exclude))
    random.seed(random_seed)
    random.shuffle(x)
    y = list(model_data.le.transform([p.parent.name for p in x]))
    ds = ImageDataset(x, y, model_data.eval_transform, num_chans=3, num_classes=len(model_data.le.classes_))
    return DataLoader(ds, model_data.batch_size, num_workers=model_data.num_workers)
