"""Diagnostic (not a test): GPU vs oracle.graph_eval.train_step_bf16 (teacher-forced) error by depth."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np
import torch
from oracle import graph_eval, refnet
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


for network, hw, n in (("resnet18", 64, 8), ("resnet50", 96, 16)):
    classes = 10
    g = arch.build_graph(network, classes)
    specs = arch.param_specs(g)
    sd = synth.synth_state_dict(specs, seed=5, logit_gain=2.0)
    ref = refnet.load_numpy_state(refnet.RefNet(network, classes), sd)
    net = HipNet(network, classes, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=10))
    y = torch.from_numpy(synth.synth_labels(n, classes, seed=11))
    net.train()
    net.forward_backward(x.cuda(), y.cuda())
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    shapes = {t: tuple(v.shape) for t, v in graph_eval.run(g, state, x, train=True).items()}
    forced = {op.dst: net.read_activation(op.dst, n, shapes[op.dst]) for op in g.ops}
    emu = graph_eval.train_step_bf16(g, state, x, y, forced=forced)
    emu32 = graph_eval.train_step_bf16(g, state, x, y, forced=forced, round_grads=False)
    print(f"== {network} {hw} x{n}: op, act-grad err (GPU vs emu), act-grad (emu vs fp32-grad emu), weight-grad err GPU vs emu, "
          f"bn.weight err, bn.bias err, fwd own err")
    for op in reversed(g.ops):
        t = op.src
        line = f"{op.name or 'op'+str(op.kind):28s}"
        if t != 0 and t in emu["act_grads"]:
            ga = net.read_activation_grad(t, n, shapes[t])
            line += f" dX {rel(ga, emu['act_grads'][t]):.2e} ({rel(emu['act_grads'][t], emu32['act_grads'][t]):.2e})"
        for suffix, key in ((".weight", op.name + ".weight"), ("bn.w", op.bn + ".weight"), ("bn.b", op.bn + ".bias")):
            if key in emu["grads"]:
                line += f" {suffix} {rel(net._read_grad(key, tuple(emu['grads'][key].shape)), emu['grads'][key]):.2e}"
        line += f" fwd {rel(forced[op.dst], emu['own'][op.dst]):.1e}"
        print(line)
