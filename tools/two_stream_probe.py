"""Probe: do two half batches on two streams (two handles) beat one full batch?  usage: two_stream_probe.py [network] [batch] [fp8]
(the ResNet eval path does this inside one handle since round 3; this tool measures other networks.)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet

network = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
fp8 = len(sys.argv) > 3 and sys.argv[3] == "fp8"
x = torch.from_numpy(synth.synth_images(batch, 3, 224, 224, seed=0)).cuda()

def make():
    g = arch.build_graph(network, 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = HipNet(network, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net.eval()
    if fp8:
        net.set_fp8(True, calibration_batch=x[:16])
    return net

def timed(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3

nets = [make(), make()]
print(f"{network} batch {batch}{' fp8' if fp8 else ''}: one stream {timed(lambda: nets[0].probabilities(x)):.3f} ms")
streams = [torch.cuda.Stream() for _ in range(2)]
xs = list(x.chunk(2))
def run():
    for net, st, xi in zip(nets, streams, xs):
        with torch.cuda.stream(st):
            net.probabilities(xi)
print(f"  two handles, two streams x {batch // 2}: {timed(run):.3f} ms")
