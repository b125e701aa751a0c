"""The N>1 launch path on the single test GPU: two ranks (both on cuda:0,
backend gloo — RCCL refuses two ranks on one device) run bench.py exactly as
the driver launches it; plus the zero-copy view of the library's gradient
buffer that the all-reduce operates on."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from sykepic_hip import arch, synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_grad_buffer_view_aliases_library_memory():
    from sykepic_hip.dp import GradSync
    from sykepic_hip.net import HipNet
    g = arch.build_graph("resnet18", 10)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
    net = HipNet("resnet18", 10, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = torch.from_numpy(synth.synth_images(8, 3, 64, 64, seed=10)).cuda()
    y = torch.from_numpy(synth.synth_labels(8, 10, seed=11)).cuda()
    net.train()
    net.forward_backward(x, y)
    sync = GradSync(net, None)
    n_params = sum(p.numel() for p in net.parameters())
    assert sync.flat.is_cuda and sync.flat.dtype == torch.float32 and sync.flat.numel() >= n_params
    before = net._read_grad("head.2.bias", (10,)).clone()
    assert float(before.abs().sum()) > 0
    sync.flat.mul_(2.0)                      # what an all-reduce over 2 equal ranks would do
    torch.cuda.synchronize()
    assert torch.allclose(net._read_grad("head.2.bias", (10,)), 2 * before)


def test_gradient_slices_are_reported_back_to_front_and_complete():
    """The overlap hook of the data-parallel step (spk_model_set_grad_ready_callback): the library reports the
    flat gradient buffer in three contiguous slices, head + last stage first, covering it exactly once, and each
    slice is final when the communication stream runs the callee's work: doubling every slice on that stream (what
    a 2-rank sum of equal gradients does) must give exactly 2 x the gradients of a step without the hook."""
    from sykepic_hip import lib
    from sykepic_hip.dp import _DevicePtr
    from sykepic_hip.net import HipNet
    g = arch.build_graph("resnet18", 10)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
    net = HipNet("resnet18", 10, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = torch.from_numpy(synth.synth_images(8, 3, 64, 64, seed=10)).cuda()
    y = torch.from_numpy(synth.synth_labels(8, 10, seed=11)).cuda()
    net.train()
    net.forward_backward(x, y)
    ptr, numel = net.grad_buffer()
    flat = torch.as_tensor(_DevicePtr(ptr, numel), device=net.device)
    torch.cuda.synchronize()
    base = flat.clone()
    comm = torch.cuda.Stream()
    seen = []

    def ready(_user, bucket, offset, n):
        seen.append((bucket, offset, n))
        with torch.cuda.stream(comm):
            flat[offset:offset + n].mul_(2.0)

    cb = lib.GRAD_READY_FN(ready)
    net.set_grad_ready_callback(cb, comm.cuda_stream, 3)
    net.forward_backward(x, y)
    torch.cuda.current_stream().wait_stream(comm)
    torch.cuda.synchronize()
    net.set_grad_ready_callback(None, 0, 0)
    assert [b for b, _, _ in seen] == [0, 1, 2]
    spans = sorted((o, o + n) for _, o, n in seen)
    assert spans[0][0] == 0 and spans[-1][1] == numel and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert seen[0][1] > seen[1][1] > seen[2][1] == 0            # back to front
    head_off = seen[0][1]
    assert head_off < numel - 10 * 128 - 128                    # the head lies inside the first slice
    assert torch.equal(flat, 2 * base)
    net.forward_backward(x, y)                                   # hook removed: plain gradients again
    torch.cuda.synchronize()
    assert torch.equal(flat, base)


def test_bench_two_ranks_like_the_driver(tmp_path):
    env = dict(os.environ, SPK_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--batch", "16", "--size", "96", "--network", "resnet18",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1                     # rank 0 only
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["global_batch"] == 32
    assert res["value"] > 0 and res["train"]["value"] > 0 and res["config"]["parallelism"] == "dp2"
    # the line describes the job it measured: who took part, what was exchanged, and that the replicas still agree
    d, t = res["dist"], res["train"]["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and [r for r, _ in d["ranks_seen"]] == [0, 1]
    assert t["world_size"] == 2 and t["params_equal"] is True and t["overlap"] is False
    # ResNet-18 + head = 11,347,186 fp32 gradients in 66 tensors, each padded to a multiple of 64 floats in the flat buffer
    assert 4 * 11_347_186 <= t["allreduce_bytes"] <= 4 * (11_347_186 + 64 * 66) and t["allreduce_ms"] > 0


def test_bench_two_ranks_efficientnet_training(tmp_path):
    """The MBConv training step under data parallelism (2 ranks over gloo on the one GPU): squeeze-excitation and
    depthwise gradients travel in the same flat buffer the all-reduce sums."""
    env = dict(os.environ, SPK_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--batch", "8", "--size", "64", "--network", "efficientnet_b0",
           "--mode", "train", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 16 and res["config"]["parallelism"] == "dp2"
    assert res["value"] > 0 and "train step" in res["metric"]


def _synthetic_sample(raw_dir, n_rois=150, seed=5):
    """.adc + .roi of a synthetic IFCB sample (column 16/17/18 = width/height/start byte, one empty trigger)."""
    rng = np.random.default_rng(seed)
    raw_dir.mkdir(parents=True, exist_ok=True)
    name = "D20200101T000000_IFCB999"
    blob, lines, start = [], [], 0
    for i in range(n_rois):
        w, h = (0, 0) if i == 40 else (int(rng.integers(24, 120)), int(rng.integers(24, 90)))
        cols = ["0"] * 24
        cols[15], cols[16], cols[17] = str(w), str(h), str(start)
        lines.append(",".join(cols))
        if w * h:
            bg = int(rng.integers(150, 220))
            img = np.clip(rng.normal(bg, 6, (h, w)), 0, 255).astype(np.uint8)
            img[h // 4: h // 2, w // 4: w // 2] = int(rng.integers(20, 120))
            blob.append(img.reshape(-1))
            start += w * h
    (raw_dir / f"{name}.adc").write_text("\n".join(lines) + "\n")
    (raw_dir / f"{name}.roi").write_bytes(np.concatenate(blob).tobytes())
    return name


def test_prob_cli_two_ranks_equals_one_rank(tmp_path, golden_dir):
    """`sykepic prob` launched as the multi-GPU job (torch.distributed.run, one process per rank; here 2 ranks on
    the one test GPU over gloo): every rank preprocesses and classifies ITS contiguous shard of the sample's ROI
    list with a full weight replica, rank 0 merges the rows by ROI id and writes the CSV.  The file must equal the
    single-process file byte for byte (same kernels, same per-image arithmetic: a row does not depend on which
    images share its batch)."""
    import shutil
    model = tmp_path / "model"
    model.mkdir()
    shutil.copy(golden_dir / "ref_data" / "class_names.txt", model / "class_names.txt")
    shutil.copy(golden_dir / "ref_data" / "config.ini", model / "config.ini")
    g = arch.build_graph("resnet18", 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, model / "best_state.pth")
    name = _synthetic_sample(tmp_path / "raw")
    env = dict(os.environ, PYTHONPATH=f"{ROOT / 'syke-pic_amd'}:{os.environ.get('PYTHONPATH', '')}", MASTER_ADDR="127.0.0.1")
    base = ["-m", "sykepic_hip", "prob", "-r", str(tmp_path / "raw"), "-m", str(model), "-b", "32"]
    one = subprocess.run([sys.executable] + base + ["-o", str(tmp_path / "out1")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + base + ["-o", str(tmp_path / "out2")],
                         env=dict(env, SPK_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    f1 = list((tmp_path / "out1").rglob(f"{name}.prob.csv"))
    f2 = list((tmp_path / "out2").rglob(f"{name}.prob.csv"))
    assert len(f1) == len(f2) == 1
    rows = f1[0].read_text().splitlines()
    assert len(rows) == 1 + 149 and rows[1].startswith("1,") and rows[-1].startswith("150,")
    assert f1[0].read_bytes() == f2[0].read_bytes()


def test_tune_cache_persists_and_is_reused(tmp_path):
    """SPK_TUNE_CACHE: the first process times the candidates and appends its winners; a second process loads the
    file and tunes nothing (no `[spk tune]` lines), also for a ragged tail batch (nearest tuned batch size within
    a factor of two) — and both produce bit-identical probabilities."""
    cache = tmp_path / "tune.txt"
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path[:0] = [{str(ROOT)!r}, {str(ROOT / 'syke-pic_amd')!r}]\n"
        "from sykepic_hip import arch, synth\n"
        "from sykepic_hip.net import HipNet\n"
        "g = arch.build_graph('resnet18', 10)\n"
        "sd = synth.synth_state_dict(arch.param_specs(g), seed=2)\n"
        "net = HipNet('resnet18', 10, weights=None)\n"
        "net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})\n"
        "net.eval()\n"
        "x = torch.from_numpy(synth.synth_images(48, 3, 64, 64, seed=0)).cuda()\n"
        "p = net.probabilities(x).cpu(); q = net.probabilities(x[:37]).cpu()\n"
        "np.save(sys.argv[1], np.concatenate([p.numpy(), q.numpy()]))\n")
    env = dict(os.environ, SPK_TUNE_CACHE=str(cache), SPK_TUNE_LOG="1")
    first = subprocess.run([sys.executable, "-c", code, str(tmp_path / "a.npy")], env=env, capture_output=True,
                           text=True, timeout=600)
    assert first.returncode == 0, first.stderr[-2000:]
    assert "[spk tune]" in first.stderr and cache.is_file()
    lines = cache.read_text().splitlines()
    assert lines and all(ln.startswith(("conv ", "wgrad ", "pw1x1 ", "c3 ")) for ln in lines)
    assert any(ln.startswith("pw1x1 ") for ln in lines)          # the 1x1 downsample convs run on conv_pw.hip
    second = subprocess.run([sys.executable, "-c", code, str(tmp_path / "b.npy")], env=env, capture_output=True,
                            text=True, timeout=600)
    assert second.returncode == 0, second.stderr[-2000:]
    assert not any(t in second.stderr for t in ("[spk tune]", "[spk tune 1x1]", "[spk tune 3x3]"))
    assert cache.read_text().splitlines() == lines              # nothing re-tuned, nothing appended
    assert np.array_equal(np.load(tmp_path / "a.npy"), np.load(tmp_path / "b.npy"))


_ONE_RANK_RCCL = r"""
import sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/syke-pic_amd"]
from sykepic_hip import arch, synth
from sykepic_hip.dp import GradSync
from sykepic_hip.net import HipNet
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], world_size=1, rank=0)
g = arch.build_graph("resnet18", 10)
sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
net = HipNet("resnet18", 10, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
x = torch.from_numpy(synth.synth_images(8, 3, 64, 64, seed=10)).cuda()
y = torch.from_numpy(synth.synth_labels(8, 10, seed=11)).cuda()
net.train()
net.forward_backward(x, y)
plain = GradSync(net, dist, overlap=False)
plain.all_reduce()
torch.cuda.synchronize()
base = plain.flat.clone()
sync = GradSync(net, dist, overlap=True)
assert sync.overlapped
for _ in range(3):
    net.forward_backward(x, y)
    sync.all_reduce()
    torch.cuda.synchronize()
    assert sync.waited == 3, sync.waited
    assert torch.equal(sync.flat, base), float((sync.flat - base).abs().max())
sync.close()
assert not sync.overlapped
net.forward_backward(x, y)
sync.all_reduce()
torch.cuda.synchronize()
assert torch.equal(sync.flat, base)
dist.destroy_process_group()
print("ONE_RANK_RCCL_OK", int(base.numel()))
"""


def test_overlapped_all_reduce_runs_over_rccl_with_one_rank(tmp_path):
    """The overlapped gradient exchange (dp.GradSync(overlap=True): the library reports three slices of the flat
    gradient buffer back to front while the backward pass runs, each slice's all-reduce is enqueued on a communication
    stream behind an event, all_reduce() waits for them) executed for real over RCCL - with the one rank this box
    allows (two ranks on one device are refused; the 2-rank tests use gloo and cannot take this path, which needs device
    tensors).  A one-rank sum is the identity: the gradients must come back bit for bit, every step, with exactly three
    collectives waited for, and the plain path must work again after close().  What one rank cannot show is the
    cross-rank sum itself: that is the driver's 2 / 4 / 8-GPU run (`params_equal` in the bench line)."""
    script = tmp_path / "one_rank.py"
    script.write_text(_ONE_RANK_RCCL)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), str(ROOT), str(_free_port())], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "ONE_RANK_RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
