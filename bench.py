#!/usr/bin/env python3
"""bench.py — IFCB images/sec of the CNN classification hot path on MI355X.

One "step" = one pass of the hot path over one per-GPU batch of synthetic
224x224x3 ROIs through ResNet-50 (+ the reference's 256,128,50 head):
  infer: net_pass body (forward + base-1.3 softmax), reference
         sykepic/compute/probability.py:184-194
  train: zero_grad/forward/CE/backward/step, reference sykepic/train/train.py:239-243
Inputs are resident in HBM before the timed region.  Weak scaling: every rank
runs the same per-GPU batch (inference needs no collective; training
all-reduces the flat gradient buffer over RCCL).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernels
(the conv kernels of the forward: conv_bneck / conv_pw / conv_pwr / conv_c3 / conv_igemm / conv_stem): algorithmic conv FLOPs / their
HIP-event time.  The
`cpu_baseline` leg times the oracle (torch fp32 CPU restatement of the
reference path) on this box's host cores on a bounded sample.
"""

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--network", default="resnet50")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--classes", type=int, default=50)
    ap.add_argument("--mode", default="both", choices=["both", "infer", "train"])
    ap.add_argument("--precision", default=None, choices=["calibrated", "mixed", "precise", "balanced", "fast", "bf16", "fp8"],
                    help="calibrated: every conv as ONE fp16 product, weights zero-sum rounded against "
                         "per-channel activation means measured on 32 calibration images that are not in the timed batch "
                         "(csrc/zero_sum.hip; as accurate as `precise`: tests/test_gpu_calibrated.py); "
                         "mixed (what a model directory without act_means.pth runs): fp16 + hi/lo split weights on every conv except the 3x3 convs "
                         "inside a residual block (max |dp| 6.8e-4 over 3 nets x 512 images, tests/archive/diagnostics/split_rules.py: "
                         "passes the 1e-3 parity tolerance); precise: split on every conv (5.9e-4); balanced: split "
                         "only the layers that write the residual trunk (1.3e-3 worst case); fast: plain fp16 "
                         "(1.5e-3); bf16 (5e-3); fp8: e4m3 inside the EfficientNet MBConv blocks (BASELINE config 5; "
                         "EfficientNets only, not a parity mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--layers-out", default="", help="write the per-layer table (JSON) here")
    ap.add_argument("--no-kernel-profile", action="store_true",
                    help="skip the per-layer / per-phase event profile behind `roofline` (for rocprofv3 traces: the last steps "
                         "of the trace are then timed steps, not the single-stream profiling passes)")
    args = ap.parse_args()
    if args.precision is None:
        # the fastest mode that holds the parity tolerance: ResNets - calibrated single pass (tests/test_gpu_calibrated.py);
        # EfficientNets - `mixed`, which splits none of their convs (the weight rounding does not show beside the fp16
        # rounding of their activations: tests/archive/diagnostics/effnet_calibrated.py)
        args.precision = "mixed" if args.network.startswith("efficientnet") else "calibrated"
    return args


# the sources the ResNet inference conv kernels (the roofline's dominant kernel) are built from AND the executor that
# decides which of them run (shortcut fusion, pw / c3 routing, the split rule: all of these change the HBM bytes per
# launch): the PMC traffic figure is re-taken when any of these changes; tools/pmc_traffic.py carries the same list
TRAFFIC_SOURCES = ("conv_igemm.hip", "conv_pw.hip", "conv_pwr.hip", "conv_c3.hip", "conv_bneck.hip", "conv_stem.hip", "spk_common.h", "model.hip",
                   "model.h")
CONV_KERNELS = "conv_bneck_kernel + conv_pw_kernel + conv_pwr_kernel + conv_c3_kernel + conv_igemm_kernel + conv_stem_kernel"


def kernel_source_sha(sources=None):
    """sha256 over those sources: a committed PMC traffic figure is only quoted for the kernels it was taken on."""
    h = hashlib.sha256()
    for name in sources or TRAFFIC_SOURCES:
        f = ROOT / "syke-pic_amd" / "csrc" / name
        h.update(name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def train_traffic(args):
    """HBM bytes per conv launch of the training step from the committed PMC passes (tools/pmc_traffic.py train), quoted
    only for the kernel sources it was taken on."""
    cands = sorted((ROOT / "profiles").glob("r*_pmc_traffic_train_bf16.json"))
    if not cands or args.batch != 256 or args.network != "resnet50" or args.size != 224:
        return None
    rec = json.loads(cands[-1].read_text())
    return round(rec["traffic_bytes_per_launch"]) if rec.get("kernel_src_sha") == kernel_source_sha(TRAIN_SOURCES) else None


TRAIN_SOURCES = ("conv_igemm.hip", "conv_wgrad.hip", "conv_stem.hip", "spk_common.h", "train.hip", "train_kernels.hip",
                 "ordered_reduce.h", "model.h")


def cpu_model():
    try:
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup
    CPU quota (a GPU box hands out a share of a large host; using every
    visible core would oversubscribe it)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if q > 0:
                    cores = min(cores, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(cores, 64))


PARITY_N = 32          # images of the timed batch that the oracle classifies too (the cpu_baseline leg's batch)
PARITY_TOL = 1e-3      # north_star: probabilities within 1e-3 of the CPU reference
TRAIN_LOSS_TOL = 2e-2  # bf16 training forward vs the fp32 oracle: relative loss error (tests/test_gpu_train.py bounds)
TRAIN_LOGIT_TOL = 8e-2  # ... relative L2 of the logits


def cpu_baseline(network, classes, size, mode, budget_s, x=None, y=None):
    """The reference's CPU path (torch fp32 NCHW kernels driven by the
    net_pass / train-step logic) timed on this host; oracle = its restatement.
    `x`, `y`: the first PARITY_N images / labels of the timed batch (host tensors).  Returns (record, reference outputs of
    the first step on those images): probabilities for infer, (loss, logits) of the un-trained net for train."""
    from oracle import refnet
    from sykepic_hip import arch, synth
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {network} {mode} on {cores} host threads", file=sys.stderr, flush=True)
    g = arch.build_graph(network, classes)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = refnet.load_numpy_state(refnet.RefNet(network, classes), sd)
    bs = PARITY_N
    if x is None:
        x = torch.from_numpy(synth.synth_images(bs, 3, size, size, seed=0))
        y = torch.from_numpy(synth.synth_labels(bs, classes, seed=1))
    if mode == "train":
        net.train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        step = lambda: refnet.train_step(net, opt, x, y)  # noqa: E731
    else:
        step = lambda: refnet.probabilities(net, x)  # noqa: E731
    first = step()  # warm-up; its outputs are the parity reference
    if mode == "train":
        first = (first[0], first[2])
    t0 = time.perf_counter()
    iters = 0
    while True:
        step()
        iters += 1
        if time.perf_counter() - t0 > budget_s or iters >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": round(bs * iters / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(),
            "sample": f"{iters} x batch {bs} of {network} {mode} fp32 NCHW on host CPU "
                      f"({iters * bs} images, {dt:.1f} s)"}, first


def infer_parity(p_gpu, p_ref):
    """max |p_gpu - p_cpu| and top-1 agreement of the TIMED precision mode, recorded in the same
    run.  p_gpu: rows of the timed batch's forward (two streams, as timed); p_ref: the oracle on the same images."""
    p_gpu, p_ref = p_gpu.double().cpu().numpy(), p_ref.double().numpy()
    top = np.sort(p_ref, axis=1)
    decided = top[:, -1] - top[:, -2] > 2 * PARITY_TOL      # a margin the tolerance cannot flip
    agree = p_gpu.argmax(1) == p_ref.argmax(1)
    err = float(np.abs(p_gpu - p_ref).max())
    return {"max_abs_dp": float(f"{err:.3e}"), "top1_agree": round(float(agree.mean()), 4),
            "top1_agree_decided": round(float(agree[decided].mean()), 4) if decided.any() else None,
            "n": int(p_ref.shape[0]), "n_decided": int(decided.sum()), "tol": PARITY_TOL,
            "ok": bool(err <= PARITY_TOL and agree[decided].all())}


def train_parity(loss_gpu, logits_gpu, ref):
    loss_ref, logits_ref = ref
    rel_loss = abs(loss_gpu - loss_ref) / max(abs(loss_ref), 1e-12)
    lg, lr_ = logits_gpu.double().cpu(), logits_ref.double()
    rel_logits = float((lg - lr_).norm() / lr_.norm())
    return {"loss_gpu": round(loss_gpu, 5), "loss_cpu": round(loss_ref, 5), "loss_rel_err": float(f"{rel_loss:.3e}"),
            "logits_rel_l2": float(f"{rel_logits:.3e}"), "n": int(logits_ref.shape[0]),
            "tol": {"loss_rel_err": TRAIN_LOSS_TOL, "logits_rel_l2": TRAIN_LOGIT_TOL},
            "ok": bool(rel_loss <= TRAIN_LOSS_TOL and rel_logits <= TRAIN_LOGIT_TOL)}


def device_id_string(dev):
    """What tells two GPUs apart in the line: the device UUID where torch exposes it, else PCI bus id."""
    pr = torch.cuda.get_device_properties(dev)
    uuid = getattr(pr, "uuid", None)
    if uuid is not None:
        return str(uuid)
    return f"{pr.name}/pci{getattr(pr, 'pci_bus_id', '?')}:{getattr(pr, 'pci_device_id', '?')}"


def dist_record(dist, dev, rank, world):
    """Who took part: the line itself shows whether RCCL saw N ranks on N different devices."""
    seen = [None] * world
    dist.all_gather_object(seen, (rank, device_id_string(dev) if dev.type == "cuda" else "cpu"))
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
            "ranks_seen": [list(t) for t in sorted(seen)],
            "distinct_devices": len({d for _, d in seen})}


def replicas_equal(net, dist):
    """True iff the sha256 of every rank's parameters is the same (one small object all-gather).  BatchNorm running
    statistics are left out: they are rank-local between two `dp.sync_buffers` calls by design (local statistics)."""
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            h.update(v.detach().cpu().contiguous().numpy().tobytes())
    digests = [None] * dist.get_world_size()
    dist.all_gather_object(digests, h.hexdigest())
    return len(set(digests)) == 1


def run_mode(mode, args, net, x, y, dist, dev, rank, world):
    """Times K steps of one mode; returns the result dict on rank 0."""
    parity_gpu = None
    if mode == "train":
        from sykepic_hip.dp import GradSync
        from sykepic_hip.optim import HipOptimizer
        net.train()
        for p in net.parameters():   # post-step_3 state: everything unfrozen (the most expensive phase)
            p.requires_grad = True
        opt = HipOptimizer(net, "Adam", [
            {"params": [p for p in net.parameters()], "lr": 1e-4}, {"params": [], "lr": 0.0},
            {"params": [], "lr": 0.0}])
        sync = GradSync(net, dist) if world > 1 else None
        ar_marks = []   # (before, after) events around the gradient exchange of every step
        if world == 1 and not args.no_cpu_baseline and rank == 0:
            # parity of the training forward in the timed precision (bf16), BEFORE any optimizer step: loss and logits of
            # the first PARITY_N images against the oracle's first step on them (compared below, next to cpu_baseline)
            net.reset_stats()
            lg = net.forward_backward(x[:PARITY_N], y[:PARITY_N], want_logits=True)
            parity_gpu = (net.read_stats()[0] / PARITY_N, lg.float().cpu())
            net.reset_stats()

        def step():
            net.forward_backward(x, y)
            if sync:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                cur = torch.cuda.current_stream(dev)   # the stream the library enqueues on (HipNet._stream) and the one
                e0.record(cur)                         # the collective's wait is put on
                sync.all_reduce(opt)
                e1.record(cur)
                ar_marks.append((e0, e1))
            opt.step()
        dtype = "bf16"
    else:
        net.eval()

        def step():
            return net.probabilities(x)
        dtype = "bf16" if args.precision == "bf16" else ("fp8" if args.precision == "fp8" else "f16")

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # per-step times from events on the launch stream (the library enqueues on torch's current stream); the
    # headline stays the wall time of the K steps between the two fences
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    if mode == "infer" and world == 1 and not args.no_cpu_baseline:
        parity_gpu = step()[:PARITY_N].float().cpu()     # one more forward of the timed configuration (two streams)
    uncal_ms = None
    if mode == "infer" and args.precision == "calibrated" and world == 1 and rank == 0 and not args.no_kernel_profile:
        # what a model directory runs until it has activation means (`prob` measures them on its first batch): the default
        # hi + lo split mode, timed like the headline.  (Here, not at the end: the LAST forward of the process stays one of the
        # timed configuration - the PMC tools take their per-step figures from it.)
        uncal_ms = uncalibrated_ms(net, x, args)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3)  # noqa: E731
    step_ms = {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "min": round(per_step[0], 3)}
    dist_rec = None
    if dist is not None:
        dist_rec = dist_record(dist, dev, rank, world)
        if mode == "train":
            ar = sorted(a.elapsed_time(b) for a, b in ar_marks[-args.steps:])
            dist_rec.update({
                "allreduce_bytes": int(sync.flat.numel() * sync.flat.element_size()),
                # time the launch stream spends in the exchange (overlapped: only the wait for slices still in flight)
                "allreduce_ms": round(ar[len(ar) // 2], 3) if ar else None,
                "overlap": sync.overlapped,
                # every replica must hold bit-identical parameters after the timed steps
                "params_equal": replicas_equal(net, dist)})
    if mode == "train" and sync:
        sync.close()   # the per-phase profile below runs on rank 0 alone: no collectives from here on
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None

    if args.no_kernel_profile:
        total_images = args.batch * world * args.steps
        return {"metric": f"IFCB images/sec, {args.network} {args.size}x{args.size} {mode} step",
                "value": round(total_images / dt, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "step_ms": step_ms,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                "config": {"workload": f"{args.network}_{mode}_b{args.batch}x{world}_{args.size}x{args.size}x3_"
                                       f"{args.classes}cls_head256-128", "precision": args.precision if mode == "infer" else "bf16"},
                "roofline": None}
    # ---- roofline of the dominant kernel, HIP events on the launch stream ----
    layers_out = args.layers_out and (args.layers_out if mode == args.mode or args.mode == "both" and mode == "infer"
                                      else args.layers_out + "." + mode)
    if mode == "train":
        phases = net.profile_train(x, y, iters=3)
        conv = [(n_, ms, fl) for n_, ms, fl, _ in phases if fl > 0]
        conv_ms = sum(ms for _, ms, _ in conv)
        conv_fl = sum(fl for _, _, fl in conv)
        all_ms = sum(ms for _, ms, _, _ in phases)
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_igemm_kernel (fwd, dgrad) + conv_wgrad_kernel",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "traffic": None if args.network.startswith("efficientnet") else train_traffic(args),
                "launches": int(sum(l for n_, _, fl, l in phases if fl > 0)),
                "conv_ms_per_step": round(conv_ms, 3), "all_kernels_ms_per_step": round(all_ms, 3),
                "phases_ms": {n_: round(ms, 3) for n_, ms, _, _ in phases}}
        table = [{"phase": n_, "ms": round(ms, 4), "gflop": round(fl / 1e9, 2), "launches": l,
                  "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 and fl > 0 else None}
                 for n_, ms, fl, l in phases]
    else:
        layers = net.profile_layers(x, iters=5)
        conv = [(n, ms, fl, by) for n, ms, fl, by in layers if fl > 0 and not n.startswith("head.")]
        conv_ms = sum(ms for _, ms, _, _ in conv)
        conv_fl = sum(fl for _, _, fl, _ in conv)
        all_ms = sum(ms for _, ms, _, _ in layers)
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": CONV_KERNELS, "achieved": round(achieved, 2),
                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "traffic": None,
                "launches": len(conv), "avg_launch_us": round(conv_ms * 1e3 / len(conv), 2),
                "algorithmic_bytes_per_launch": round(sum(by for _, _, _, by in conv) / len(conv)),
                "conv_ms_per_step": round(conv_ms, 3), "all_kernels_ms_per_step": round(all_ms, 3),
                "hbm_GBs_algorithmic": round(sum(by for _, _, _, by in layers) / (all_ms * 1e-3) / 1e9, 1)}
        # The timed step runs the two halves of the batch on two streams (kernels of the two chains overlap), so a
        # kernel's duration inside it is not its own: achieved / frac above are per-kernel figures from a single-stream
        # pass (spk_model_profile_infer; what rocprofv3 shows under SPK_EVAL_STREAMS=1).  step_*: the same algorithmic
        # conv FLOPs over the conv share of the TIMED step.
        step_conv_ms = dt / args.steps * 1e3 * conv_ms / max(all_ms, 1e-9)
        roof["step_conv_ms"] = round(step_conv_ms, 3)
        roof["step_achieved"] = round(conv_fl / (step_conv_ms * 1e-3) / 1e12, 2)
        roof["step_frac"] = round(roof["step_achieved"] / MFMA_PEAK_TFLOPS, 4)
        roof["streams"] = int(os.environ.get("SPK_EVAL_STREAMS", "2")) if args.batch >= 64 else 1
        if args.network.startswith("efficientnet"):
            streams = roof["streams"]
            # EfficientNet: 2/3 of the time is depthwise / squeeze-excitation / padded 1x1 passes that move bytes:
            # the bound is HBM, over all kernels of the forward
            all_by = sum(by for _, _, _, by in layers)
            gbs = all_by / (all_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "all forward kernels (conv_igemm 1x1, dwconv, se_*, stem3x3)",
                    "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                    "traffic": None, "launches": len(layers), "avg_launch_us": round(all_ms * 1e3 / len(layers), 2),
                    "algorithmic_bytes_per_launch": round(all_by / len(layers)),
                    "all_kernels_ms_per_step": round(all_ms, 3),
                    "streams": streams,
                    "conv1x1_tflops": round(sum(fl for n, _, fl, _ in conv if ".block." in n or n.endswith("8.0")) / 1e12 /
                                            max(sum(ms for n, ms, _, _ in conv) * 1e-3, 1e-9), 1)}
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure comes from the
        # committed rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py) and is quoted ONLY if it
        # was taken on the kernel sources of this build (sha over csrc/), else null
        cands = sorted((ROOT / "profiles").glob(f"r*_pmc_traffic_infer_{args.precision}.json"))
        if cands and args.batch == 256 and args.network == "resnet50" and args.size == 224:
            rec = json.loads(cands[-1].read_text())
            if rec.get("kernel_src_sha") == kernel_source_sha():
                roof["traffic"] = round(rec["traffic_bytes_per_launch"])
                roof["traffic_source"] = str(cands[-1].relative_to(ROOT))
                roof["traffic_taken_at"] = rec.get("taken_at")
            else:
                roof["traffic_source"] = f"{cands[-1].name} is stale (taken on other kernel sources): not quoted"
        table = [{"layer": n, "ms": round(ms, 4), "gflop": round(fl / 1e9, 3), "mbytes": round(by / 1e6, 2),
                  "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None,
                  "gbs": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
                 for n, ms, fl, by in layers]
    if layers_out:
        Path(layers_out).parent.mkdir(parents=True, exist_ok=True)
        Path(layers_out).write_text(json.dumps(table, indent=1))

    total_images = args.batch * world * args.steps
    out = {
        "metric": f"IFCB images/sec, {args.network} {args.size}x{args.size} {mode} step",
        "value": round(total_images / dt, 1), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "step_ms": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": f"{args.network}_{mode}_b{args.batch}x{world}_"
                               f"{args.size}x{args.size}x3_{args.classes}cls_head256-128",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "precision": args.precision if mode == "infer" else "bf16",
                   "parallelism": f"dp{world}"},
        "roofline": roof,
    }
    if dist_rec is not None:
        out["dist"] = dist_rec
    if uncal_ms is not None:
        out["uncalibrated_ms_per_step"] = uncal_ms
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], ref = cpu_baseline(args.network, args.classes, args.size, mode,
                                                args.cpu_seconds if mode == "infer" else args.cpu_seconds / 2,
                                                x[:PARITY_N].float().cpu(), y[:PARITY_N].cpu())
        out["parity"] = infer_parity(parity_gpu, ref) if mode == "infer" else train_parity(*parity_gpu, ref)
    return out


def uncalibrated_ms(net, x, args):
    net.set_precision(split_weights=3)
    try:
        for _ in range(max(args.warmup, 3)):      # (the first forward of a precision mode tunes its kernels on one stream)
            net.probabilities(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net.probabilities(x)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / args.steps * 1e3, 3)
    finally:
        net.set_precision("calibrated")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks (one process per GPU) before anything touches the GPU and relay
        # rank 0's JSON line — a bare `python bench.py --gpus 8` must not print a 1-GPU number
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): refusing to report a number "
              f"for the wrong GPU count", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("SPK_DIST_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from sykepic_hip import arch, synth
    from sykepic_hip.net import HipNet

    g = arch.build_graph(args.network, args.classes)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = HipNet(args.network, args.classes, weights=None, device=dev)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    if args.precision == "fp8":
        if not args.network.startswith("efficientnet"):
            print("[bench] --precision fp8 is the EfficientNet mode (BASELINE config 5)", file=sys.stderr)
            sys.exit(2)
        net.set_precision(split_weights=3)
    elif args.precision == "bf16":
        net.set_precision(split_weights=False, bf16=True)
    elif args.precision == "calibrated":
        # model preparation, outside the timed region: what `sykepic train` stores as act_means.pth next to best_state.pth
        net.eval()
        # (the SAME images on every rank: the replicas of a multi-GPU line are one model)
        net.calibrate(torch.from_numpy(synth.synth_images(32, 3, args.size, args.size, seed=9000)).to(dev))
        net.set_precision("calibrated")
    else:
        net.set_precision(split_weights={"mixed": 3, "precise": 1, "balanced": 2, "fast": 0}[args.precision])
    # the job's batch is world x per-GPU batch images (weak scaling); rank r holds the contiguous shard
    # dp.shard_range gives it — the same split `sykepic prob` makes of a sample's ROI list (prob.process_sample).
    # Image i of the global batch is generated from seed i // per-GPU batch, so a rank builds only its own shard.
    from sykepic_hip import dp
    lo, hi = dp.shard_range(args.batch * world, rank, world)
    assert (lo, hi) == (rank * args.batch, (rank + 1) * args.batch)
    x = torch.from_numpy(synth.synth_images(args.batch, 3, args.size, args.size, seed=rank)).to(dev)
    y = torch.from_numpy(synth.synth_labels(args.batch, args.classes, seed=1000 + rank)).to(dev)

    if args.precision == "fp8":   # activation ranges from the bench batch itself (a deployment calibrates on real ROIs)
        net.eval()
        net.set_fp8(True, calibration_batch=x[:min(args.batch, 64)])
    # headline = the inference step (net_pass body); the training step of the
    # same model/config rides along under "train" (BASELINE metric names both)
    modes = ["infer", "train"] if args.mode == "both" else [args.mode]
    if args.network.startswith("efficientnet") and args.mode == "both":
        modes = ["infer"]   # BASELINE config 5 is an inference config; `--mode train` measures the MBConv training step
    results = []
    failed = parity_failed = False
    for m in modes:
        if rank == 0:
            print(f"[bench] {m}: {args.warmup} warm-up + {args.steps} timed steps", file=sys.stderr, flush=True)
        if m == "train" and results and args.mode == "both":
            # the training step rides along under "train": a failure there must not take the headline with it
            try:
                results.append(run_mode(m, args, net, x, y, dist, dev, rank, world))
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] train leg failed: {exc!r}", file=sys.stderr, flush=True)
                results.append({"metric": f"IFCB images/sec, {args.network} {args.size}x{args.size} train step", "value": None,
                                "error": repr(exc)})
                failed = True
            continue
        results.append(run_mode(m, args, net, x, y, dist, dev, rank, world))
    if rank == 0:
        out = results[0]
        if len(results) > 1:
            t = results[1]
            out["train"] = {k: t[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline",
                                              "dist", "error", "parity") if k in t}
            if "cpu_baseline" in t:
                out["train"]["cpu_baseline"] = t["cpu_baseline"]
        print(json.dumps(out), flush=True)
        for r in results:
            if isinstance(r, dict) and r.get("parity") and not r["parity"]["ok"]:
                print(f"[bench] PARITY BROKEN: {r['parity']}", file=sys.stderr, flush=True)
                parity_failed = True
    if dist is not None:
        dist.destroy_process_group()
    if failed:      # the headline line is out; the run still counts as failed
        sys.exit(3)
    if parity_failed:   # a fast number whose results differ from the reference's is not a number
        sys.exit(4)


if __name__ == "__main__":
    main()
