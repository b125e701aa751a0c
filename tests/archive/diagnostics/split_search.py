"""Which convolutions need hi+lo split fp16 weights to keep |dp| < 1e-3?  (GPU diagnostic)

For every conv i the logit error of "all split except i" against the fp32 oracle gives the
error energy c_i that layer's weight rounding adds (errors of different layers add in
quadrature); profile_layers under both modes gives the time t_i its lo-product costs.
A greedy knapsack then drops the split where it buys the most time per unit of error
energy, and the chosen mask is verified on fresh images.

    python tools/split_search.py [n_images] [budget_rms_ratio]
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np
import torch

from oracle import refnet
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet

torch.set_num_threads(16)
network, hw = "resnet50", 224
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 96
gold = np.load(ROOT / "tests/golden/net_pass.npz")
g = arch.build_graph(network, 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
last = [k for k in sd if k.startswith("head.") and k.endswith(".bias")][-1]
sd[last] = sd[last] + gold["resnet50_224_bias_adj"]
ref = refnet.load_numpy_state(refnet.RefNet(network, 50), sd)
net = HipNet(network, 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
net.eval()


def images(seed0, n):
    return torch.cat([torch.from_numpy(synth.synth_images(32, 3, hw, hw, seed=seed0 + i)) for i in range(n // 32)])


def oracle(x):
    lg = torch.cat([refnet.probabilities(ref, x[i:i + 32], base=0) for i in range(0, len(x), 32)]).numpy()
    pr = torch.softmax(torch.from_numpy(lg) * float(np.log(1.3)), 1).numpy()
    return lg, pr


def measure(x, lg_ref, pr_ref):
    xg = x.cuda()
    lg = torch.cat([net.forward(xg[i:i + 32]) for i in range(0, len(xg), 32)]).float().cpu().numpy()
    pr = torch.cat([net.probabilities(xg[i:i + 32]) for i in range(0, len(xg), 32)]).cpu().numpy()
    return float(np.sqrt(np.mean((lg - lg_ref) ** 2))), float(np.abs(pr - pr_ref).max())


x = images(300, n_img)
lg_ref, pr_ref = oracle(x)
convs = [n for _, n in net.conv_ops()]
net.set_split_ops(convs)
rms_full, dp_full = measure(x, lg_ref, pr_ref)
net.set_split_ops([])
rms_none, dp_none = measure(x, lg_ref, pr_ref)
print(f"all split: logit rms {rms_full:.3e} max|dp| {dp_full:.2e};  none: {rms_none:.3e} {dp_none:.2e}", flush=True)

energy = {}
for name in convs:
    net.set_split_ops([c for c in convs if c != name])
    r, _ = measure(x, lg_ref, pr_ref)
    energy[name] = max(r * r - rms_full * rms_full, 0.0)
print("sum of single-layer energies", sum(energy.values()) ** 0.5, "vs none-all", (rms_none**2 - rms_full**2) ** 0.5)

xb = torch.from_numpy(synth.synth_images(256, 3, hw, hw, seed=1)).cuda()
net.set_split_ops(convs)
t_full = {n: ms for n, ms, _, _ in net.profile_layers(xb, iters=5)}
net.set_split_ops([])
t_none = {n: ms for n, ms, _, _ in net.profile_layers(xb, iters=5)}
saved = {n: max(t_full.get(n, 0) - t_none.get(n, 0), 0.0) for n in convs}
print(f"conv time all split {sum(t_full[n] for n in convs):.3f} ms, none {sum(t_none[n] for n in convs):.3f} ms")

rows = sorted(convs, key=lambda n: energy[n] / max(saved[n], 1e-6))
for n in rows:
    print(f"  {n:34s} energy {energy[n]:.3e}  saves {saved[n]*1e3:7.1f} us")

# greedy: un-split in order of error energy per saved microsecond while the predicted logit rms
# stays below `ratio` x the rms of the all-split mode ... calibrated on max|dp| afterwards
results = []
for ratio in (1.15, 1.3, 1.5, 1.75, 2.0, 2.5):
    budget = (ratio * rms_full) ** 2 - rms_full ** 2
    acc, unsplit = 0.0, []
    for n in rows:
        if saved[n] <= 0:
            continue
        if acc + energy[n] <= budget:
            acc += energy[n]
            unsplit.append(n)
    keep = [c for c in convs if c not in unsplit]
    net.set_split_ops(keep)
    r, dp = measure(x, lg_ref, pr_ref)
    results.append((ratio, keep, r, dp, sum(saved[n] for n in unsplit)))
    print(f"ratio {ratio}: {len(keep)} of {len(convs)} convs split, rms {r:.3e} max|dp| {dp:.2e}, "
          f"predicted saving {sum(saved[n] for n in unsplit):.3f} ms", flush=True)

# verify on fresh images
xv = images(900, 128)
lgv, prv = oracle(xv)
out = []
for ratio, keep, r, dp, sv in results:
    net.set_split_ops(keep)
    rv, dpv = measure(xv, lgv, prv)
    print(f"verify ratio {ratio}: rms {rv:.3e} max|dp| {dpv:.2e}")
    out.append({"ratio": ratio, "split": keep, "search_rms": r, "search_max_dp": dp, "verify_rms": rv,
                "verify_max_dp": dpv, "saved_ms": sv})
for mode, kw in (("all", convs), ("none", [])):
    net.set_split_ops(kw)
    print("verify", mode, measure(xv, lgv, prv))
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out/split_search.json").write_text(json.dumps(
    {"energy": energy, "saved_ms": saved, "rms_full": rms_full, "rms_none": rms_none, "results": out}, indent=1))
