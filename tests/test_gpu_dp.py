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


def test_bench_two_ranks_like_the_driver(tmp_path):
    env = dict(os.environ, SPK_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--batch", "16", "--size", "96", "--network", "resnet18",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1                     # rank 0 only
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["global_batch"] == 32
    assert res["value"] > 0 and res["train"]["value"] > 0 and res["config"]["parallelism"] == "dp2"
