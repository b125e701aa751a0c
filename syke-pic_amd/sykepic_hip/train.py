"""`sykepic train`: the training workflow on MI355X.

Host-side mirror of the reference's ``sykepic/train/train.py`` (``main`` :17,
``train_net`` :201, ``test_net`` :323): same ``train.ini`` keys, same model
directory artefacts (``config.ini``, ``class_names.txt``,
``class_distribution.csv``, ``best_state.pth``, ``test_report*.txt``), same
``[STAT]``/``[INFO]`` lines, same checkpoint-on-val-accuracy and
early-stop-on-val-loss rules (quirk Q4), same swallowed exceptions (Q5).
The five-line hot loop (zero_grad/forward/loss/backward/step, :239-243)
becomes ``net.forward_backward(x, y)`` + ``optimizer.step()``: fused kernels
in libsykepic_hip.so; loss and accuracy accumulate on the GPU and are read
once per epoch instead of twice per step.

Data parallelism: launched under ``torch.distributed.run`` (one process per
GPU) every rank trains on its shard, gradients are all-reduced over RCCL and
rank 0 alone prints, checkpoints and writes reports.
"""

import os
import shutil
from configparser import ConfigParser
from pathlib import Path

import torch

from . import data, schedule
from .config import Settings, get_network, get_transforms
from . import dp
from .dp import GradSync
from .optim import HipOptimizer


def _dist():
    return dp.init_from_env()


def _check_split(split):
    """`[dataset] split` = fractions for train, validation and (optionally) test; the reference's two complaints."""
    if sum(split) != 1.0:
        raise ValueError(f"Dataset split does not add up to 1.0. Got {sum(split)}")
    if len(split) < 2:
        raise ValueError("Dataset split needs to cover at least train and validation")
    return len(split) == 3


def _png_path(name):
    out = Path(name)
    return out if out.suffix else out.with_suffix(".png")


def main(args):
    config = ConfigParser()
    config.read(args.config)
    cfg = Settings(config)          # config.KEYS: every key of train.ini, parsed on first use
    if not os.environ.get("SYKEPIC_HOST_TRANSFORMS"):
        # the helper process the PNG decode workers are forked from: up before RCCL / HIP exist in this process
        from . import gpu_augment
        gpu_augment.start_decode_server()
    dist, rank, world, local = _dist()
    chief = rank == 0

    ds, im, mo, tr = cfg.dataset, cfg.image, cfg.model, cfg.train
    split, seed = ds.split, ds.random_seed
    test_split = _check_split(split)
    model_data = data.ModelData(ds.path, split, ds.min_N, ds.max_N, ds.exclude, seed)

    # side outputs of the CLI that end the run before any training (`--save-images` only copies)
    if getattr(args, "save_images", None):
        parts = {"train": model_data.train_x, "val": model_data.val_x, "test": model_data.test_x if test_split else []}
        for name, paths in parts.items():
            if paths:
                (Path(args.save_images) / name).mkdir(exist_ok=True, parents=True)
                for p in paths:
                    shutil.copy(p, Path(args.save_images) / name / p.name)
    if getattr(args, "dist", None):
        from . import plots
        plots.dataset_distribution(model_data, _png_path(args.dist))
        print(f"[INFO] Distribution plot saved to {_png_path(args.dist)}")
        return

    if (until := ds.oversample_until) is not None:
        model_data.oversample(until, None)
    elif (decay := ds.oversample_with_decay) is not None:
        model_data.oversample(None, decay)

    img_shape = im.shape
    train_transform, eval_transform = get_transforms(config, img_shape)
    if getattr(args, "collage", None):
        from . import plots
        rows, cols = int(args.collage[0]), int(args.collage[1])
        model_data.set_data_loaders(rows * cols, im.num_workers, train_transform, eval_transform, img_shape[0])
        plots.view_batch(model_data.train_loader, rows, cols, _png_path(args.collage[2]))
        print(f"[INFO] Image collage saved to {_png_path(args.collage[2])}")
        return
    # transforms run batched on the GPU where the pipeline allows it (SYKEPIC_HOST_TRANSFORMS=1: host workers)
    gpu_tf = None if os.environ.get("SYKEPIC_HOST_TRANSFORMS") else torch.device("cuda", local)
    model_data.set_data_loaders(im.batch_size, im.num_workers, train_transform, eval_transform, img_shape[0],
                                rank=rank, world=world, device=gpu_tf)
    num_classes = len(model_data.le.classes_)
    external_test = ds.external_test
    if external_test:
        extra_loader = data.extra_eval_dataloader(external_test, model_data, exclude=["Unclassified"])

    # model directory <path>/<network>[_<id>]; id "auto" = next free number, picked by rank 0 for the whole job
    # (another rank could already see the directory rank 0 has just made)
    model_id = mo.id
    if model_id == "auto":
        model_id = dp.broadcast_object(data.auto_id(mo.network, mo.path) if chief else None, dist)
    model_dir = mo.path / (mo.network + (f"_{model_id}" if model_id else ""))
    if chief:
        model_dir.mkdir(parents=True, exist_ok=mo.exist_ok)
        from . import prob as _prob
        _prob.drop_act_means(model_dir)   # exist_ok = yes: means of an earlier run do not belong to the weights about to be written
        model_data.save(model_dir)
        shutil.copy(args.config, model_dir / "config.ini")
    if dist is not None:
        dist.barrier()

    if not tr.gpu:
        raise RuntimeError("this build trains on the MI355X only; for `gpu = no` use the reference itself")
    device = torch.device("cuda", local)
    net = get_network(config, num_classes, device=device)
    # Dropout masks are a function of (seed, step, layer): give every rank its own stream, otherwise the j-th sample of
    # every shard draws the same mask at every step.  A run stays deterministic for a given random_seed and world size.
    net.set_seed(seed * world + rank)
    dp.broadcast_state(net, dist)   # every replica starts from rank 0's (random / pretrained) tensors
    schedule.freeze(net.base)
    initial = [p for p in net.parameters() if p.requires_grad]
    optimizer = HipOptimizer(net, tr.optimizer, [{"params": initial, "lr": tr.learning_rate},
                                                 {"params": [], "lr": 0.0}, {"params": [], "lr": 0.0}])
    if chief:
        print("---- Network Head ----")
        print(net.head)

    lr_warmup = lr_scheduler = None
    if (wu := cfg.lr_warmup).use:
        lr_warmup = schedule.LRWarmup(net, optimizer, wu.factor_1, wu.factor_2, wu.step_1, wu.step_2, wu.step_3,
                                      wu.verbose and chief)
    if (red := cfg.lr_reduction).use:
        # quirk Q3: the reference passes `verbose` positionally into `threshold`
        lr_scheduler = schedule.ReduceLROnPlateau(optimizer, "min", red.factor, red.patience, red.verbose)

    best_state = train_net(net, model_data.train_loader, model_data.val_loader, optimizer, None, tr.max_epochs,
                           tr.early_stop_patience, model_dir, device, lr_scheduler, lr_warmup, dist=dist)
    if dist is not None:
        dist.barrier()
    net.load_state_dict(torch.load(best_state, map_location="cpu"))
    if chief:
        # activation means of the best model on (this rank's share of) the validation images, stored beside
        # best_state.pth: `sykepic prob` then runs the calibrated single-pass mode (prob.use_act_means)
        from . import prob
        try:
            n_cal = prob.calibrate_model(net, model_data.val_loader, 2048)
            if n_cal:
                prob.save_act_means(net, model_dir, n_cal)
                print(f"[INFO] Activation means of {n_cal} validation images saved to {prob.ACT_MEANS_FILE}")
        except Exception as e:  # noqa: BLE001 - an optimisation of later inference, never a reason to lose the run
            prob.drop_act_means(model_dir)     # whatever is there now belongs to other weights
            print(f"[INFO] No activation means stored ({e})")
    tests = ([(None, model_data.test_loader)] if test_split else []) + \
            ([(Path(external_test).name, extra_loader)] if external_test else [])
    for name, loader in tests if chief else []:
        report = test_net(net, loader, model_data.le.classes_, device, test_name=name)
        print(report)
        (model_dir / (f"test_report_{name}.txt" if name else "test_report.txt")).write_text(report)


def train_net(net, train_dataloader, val_dataloader, optimizer, loss_fn, max_epochs, early_stop_patience,
              model_dir, device, lr_scheduler=None, lr_warmup=None, dist=None):
    """loss_fn is accepted for signature compatibility; CrossEntropyLoss is
    fused into the training step (the only loss the reference constructs)."""
    net = net.to(device)
    chief = dist is None or dist.get_rank() == 0
    parallel = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    sync = GradSync(net, dist)
    max_val_acc, min_val_loss, no_improvement = 0, 0, 0
    train_accs, train_losses, val_accs, val_losses = [], [], [], []
    best_state = Path(model_dir) / "best_state.pth"
    try:
        from tqdm import tqdm
    except ImportError:  # pragma: no cover
        def tqdm(x, **kw):
            return x
    try:
        for epoch in range(1, max_epochs + 1):
            if chief:
                print(f"\n----- Epoch {epoch} -----")
            if lr_warmup:
                lr_warmup(epoch)
            net.train()
            net.reset_stats()
            seen = 0
            for batch in (tqdm(train_dataloader) if chief else train_dataloader):
                x, y = batch[0], batch[1]
                optimizer.zero_grad()
                net.forward_backward(x, y)
                sync.all_reduce(optimizer)
                optimizer.step()
                seen += len(y)
            loss_sum, correct = net.read_stats()          # one device sync per epoch
            loss_sum, correct, seen = sync.reduce_stats(loss_sum, correct, seen)
            train_acc, train_loss = correct / seen, loss_sum / seen
            train_accs.append(train_acc)
            train_losses.append(train_loss)
            if chief:
                print(f"[STAT] Train Acc: {train_acc:.3f}, Train Loss: {train_loss:.3f}")

            # validation: every rank holds the same model (running statistics averaged), evaluates ITS shard of
            # the validation set, and the counters are summed: checkpoint, early-stop and learning-rate decisions
            # below are then identical on every rank by construction
            dp.sync_buffers(net, dist)
            net.eval()
            net.reset_stats()
            seen = 0
            for batch in val_dataloader:
                net.eval_step(batch[0], batch[1])
                seen += len(batch[1])
            loss_sum, correct = net.read_stats()
            loss_sum, correct, seen = sync.reduce_stats(loss_sum, correct, seen)
            val_acc, val_loss = correct / seen, loss_sum / seen
            val_accs.append(val_acc)
            val_losses.append(val_loss)
            if chief:
                print(f"[STAT] Val Acc: {val_acc:.3f}, Val Loss: {val_loss:.3f}")
                _plot(train_accs, train_losses, val_accs, val_losses, Path(model_dir), epoch)
            if val_acc > max_val_acc:
                max_val_acc = val_acc
                if chief:
                    print("[INFO] Increased accuracy, saving model state")
                    torch.save(net.state_dict(), best_state)
            if val_loss < min_val_loss or epoch == 1:
                no_improvement = 0
                min_val_loss = val_loss
            else:
                no_improvement += 1
                if chief:
                    print(f"[INFO] No reduction in loss for {no_improvement} epochs")
            if no_improvement >= early_stop_patience:
                if chief:
                    print("[INFO] Stopping early")
                break
            if lr_scheduler and (not lr_warmup or epoch > lr_warmup.step_3):
                lr_scheduler.step(val_loss)
    except KeyboardInterrupt:
        print("[INFO] Stopping early")
        sync.close()
    except Exception as e:  # the reference swallows every error here (quirk Q5)
        print(f"[ERROR] {e}")
        if parallel:
            # ... but one replica leaving the loop would leave the others blocked in the next all-reduce:
            # under data parallelism the error ends the rank (non-zero exit; the launcher stops the job)
            raise
    sync.close()
    return best_state


def _plot(train_accs, train_losses, val_accs, val_losses, model_dir, epoch):
    try:
        from . import plots
        plots.plot_stats(train_accs, train_losses, val_accs, val_losses, outfile=model_dir / "train_stats.png",
                         first_epoch=1, epoch_step=3)
        if epoch >= 11:
            plots.plot_stats(train_accs[10:], train_losses[10:], val_accs[10:], val_losses[10:],
                             outfile=model_dir / "train_stats_zoomed.png", first_epoch=11, epoch_step=2)
    except Exception:  # plotting is cosmetic (out of scope)
        pass


def test_net(net, dataloader, classes, device, test_name=None):
    net = net.to(device)
    net.eval()
    print(f"\n----- Model Evaluation ({test_name}) -----" if test_name else "\n----- Model Evaluation -----")
    true_labels, predicted = [], []
    pending = []
    for batch in dataloader:
        x, y = batch[0], batch[1]
        pending.append((net(x).argmax(1), y))
    for preds, y in pending:
        predicted.extend(preds.tolist())
        true_labels.extend(torch.as_tensor(y).tolist())
    acc = sum(int(a == b) for a, b in zip(true_labels, predicted)) / max(1, len(true_labels))
    print(f"[STAT] Test Accuracy: {acc:.3f}\n")
    from sklearn.metrics import classification_report
    return classification_report(true_labels, predicted, target_names=list(classes), zero_division=0)
