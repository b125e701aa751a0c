"""Data-parallel plumbing on CPU: world_size-2 `gloo` processes (the GPU path
uses the same code with backend nccl = RCCL).  The flat-gradient all-reduce
must equal the sum of the per-shard oracle gradients and, with the optimizer's
1/world grad_scale, their mean; inference sharding must reassemble the
ROI-sorted rows of a single-process net_pass."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, ROOT / "syke-pic_amd"):
        sys.path.insert(0, str(p))
    import torch.distributed as dist
    from oracle import refnet
    from sykepic_hip import arch, dp, synth
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    classes, n, hw = 10, 8, 32
    g = arch.build_graph("resnet18", classes)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=5, logit_gain=2.0)
    net = refnet.load_numpy_state(refnet.RefNet("resnet18", classes), sd)
    net.train()
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10))
    y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
    b, e = dp.shard_range(n, rank, world)
    loss = torch.nn.functional.cross_entropy(net(x[b:e]), y[b:e])
    loss.backward()
    flat = torch.cat([p.grad.flatten() for p in net.parameters()])
    local = flat.clone()
    sync = dp.GradSync(None, dist, view=flat)

    class Opt:
        grad_scale = 1.0
    opt = Opt()
    sync.all_reduce(opt)
    stats = sync.reduce_stats(float(loss) * (e - b), 1.0, e - b)
    rows = dp.gather_rows([(100 * rank + i, [float(rank)]) for i in range(3)], dist)
    torch.save({"local": local, "reduced": flat, "scale": opt.grad_scale, "stats": stats, "rows": rows,
                "range": (b, e)}, Path(out_dir) / f"r{rank}.pt")
    dist.destroy_process_group()


def test_gradient_allreduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    want = r[0]["local"] + r[1]["local"]
    for i in range(world):
        assert torch.allclose(r[i]["reduced"], want, rtol=1e-6, atol=1e-7)
        assert r[i]["scale"] == 0.5
        assert r[i]["stats"][2] == 8 and r[i]["stats"][1] == 2.0
    assert r[0]["range"] == (0, 4) and r[1]["range"] == (4, 8)
    assert r[0]["rows"] == r[1]["rows"] == sorted(r[0]["rows"])
    assert [k for k, _ in r[0]["rows"]] == [0, 1, 2, 100, 101, 102]


def test_sharding_helpers():
    sys.path.insert(0, str(ROOT / "syke-pic_amd"))
    from sykepic_hip import dp
    from sykepic_hip.data import ShardedShuffle
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            cover = []
            for rank in range(world):
                b, e = dp.shard_range(n, rank, world)
                cover.extend(range(b, e))
                assert 0 <= e - b <= n // world + 1
            assert cover == list(range(n))
    order = list(range(10))
    parts = [dp.shard_indices(order, r, 4) for r in range(4)]
    assert len({len(p) for p in parts}) == 1                 # same number of steps on every rank
    assert set(sum(parts, [])) == set(order)
    s0, s1 = ShardedShuffle(11, 0, 2, seed=3), ShardedShuffle(11, 1, 2, seed=3)
    a, b = list(s0), list(s1)
    assert len(a) == len(b) == 6 and set(a) | set(b) == set(range(11))
    assert list(s0) != a                                      # reshuffled next epoch
