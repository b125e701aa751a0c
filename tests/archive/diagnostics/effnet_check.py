"""EfficientNet eval path vs the fp32 oracle (GPU diagnostic)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from oracle import refnet
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
torch.set_num_threads(16)
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b4"
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 224
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
g = arch.build_graph(name, 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
ref = refnet.load_numpy_state(refnet.RefNet(name, 50), sd)
net = HipNet(name, 50, weights=None)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
x = torch.from_numpy(synth.synth_images(n, 3, hw, hw, seed=3))
z = refnet.probabilities(ref, x, base=0).numpy()
pr = refnet.probabilities(ref, x).numpy()
for mode in (3, 1, 0):
    net.set_precision(split_weights=mode)
    zg = net.forward(x.cuda()).cpu().numpy()
    pg = net.probabilities(x.cuda()).cpu().numpy()
    print(f"{name} mode {mode}: logits ref range {z.min():.2f}..{z.max():.2f}  rms err {np.sqrt(np.mean((zg - z) ** 2)):.3e}  "
          f"max|dp| {np.abs(pg - pr).max():.2e}  top1 same {(pg.argmax(1) == pr.argmax(1)).mean():.2f}  pmax {pr.max(1).min():.3f}..{pr.max(1).max():.3f}")
net.set_precision()
xb = torch.from_numpy(synth.synth_images(128, 3, hw, hw, seed=5)).cuda()
for _ in range(2): net.probabilities(xb)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): net.probabilities(xb)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"batch 128: {dt*1e3:.2f} ms  {128/dt:.0f} img/s")
rows = net.profile_layers(xb, iters=3)
rows.sort(key=lambda r: -r[1])
for nme, ms, fl, by in rows[:12]:
    print(f"  {nme:34s} {ms*1e3:8.1f} us  {fl/ms/1e9 if ms else 0:8.1f} GF/s  {by/ms/1e6 if ms else 0:8.1f} GB/s")
print("total ms", sum(r[1] for r in rows))
g2 = {op.name: op for op in g.ops}
agg = {}
for nme, ms, fl, by in rows:
    op = g2.get(nme)
    kind = "other"
    if op is not None:
        kind = {arch.OP_CONV: "conv1x1" if op.k == 1 else "stem", arch.OP_DWCONV: f"dw{op.k}", arch.OP_SE: "se",
                arch.OP_LINEAR: "head"}.get(op.kind, "other")
    a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by
for k, (c, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:10s} x{c:3d} {ms:7.3f} ms  {fl/ms/1e9 if ms else 0:9.1f} GF/s  {by/ms/1e6 if ms else 0:8.1f} GB/s")
