"""EfficientNet: logit error vs precision knobs (GPU diagnostic)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd"), str(ROOT / "tests")]
import numpy as np, torch
from oracle import refnet
from sykepic_hip import synth
from sykepic_hip.net import HipNet
from effnet_util import calibrated_state
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b4"
gold = np.load(ROOT / "tests/golden/net_pass_effnet.npz")
g, sd, ref = calibrated_state(name, 224, gold)
net = HipNet(name, 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
x = torch.cat([torch.from_numpy(synth.synth_images(16, 3, 224, 224, seed=21 + i)) for i in range(4)])
z = torch.cat([refnet.probabilities(ref, x[i:i+16], base=0) for i in range(0, 64, 16)]).numpy()
base = 1.3 ** 0.25
pr = torch.softmax(torch.from_numpy(z) * float(np.log(base)), 1).numpy()
for sw in (3, 1, 0):
    for pres in (False, True):
        net.set_precision(split_weights=sw, precise_residual=pres)
        zg = torch.cat([net.forward(x[i:i+16].cuda()) for i in range(0, 64, 16)]).cpu().numpy()
        pg = torch.softmax(torch.from_numpy(zg) * float(np.log(base)), 1).numpy()
        print(f"{name} split {sw} precise_res {pres}: rel logit rms {np.sqrt(np.mean((zg-z)**2))/z.std():.2e}  max|dp| {np.abs(pg-pr).max():.2e}", flush=True)
net.set_precision(split_weights=3)
zg = torch.cat([net.forward(x[i:i+16].cuda()) for i in range(0, 64, 16)]).cpu().numpy()
per = np.sqrt(np.mean((zg - z) ** 2, 1)) / z.std()
print("per-image rel err:", np.round(per * 1e3, 1).tolist())
worst = int(per.argmax())
print("worst image", worst, per[worst])
from oracle import graph_eval
tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
xw = x[worst:worst + 1]
acts = graph_eval.run(g, tsd, xw)
net.forward(xw.cuda())
prev = 0
for op in g.ops:
    want = acts[op.dst]
    got = net.read_activation(op.dst, 1, tuple(want.shape))
    d = got - want
    rel = float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    flag = " <<<" if rel > 1.5 * prev and rel > 2e-3 else ""
    print(f"{op.dst:3d} {op.name or op.kind!s:26s} max|x| {float(want.abs().max()):8.2f} rel-rms {rel:.2e} max err {float(d.abs().max()):.2e}{flag}")
    prev = rel
