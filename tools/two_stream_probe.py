"""Probe: does running two half batches on two streams (two handles) beat one full batch?  """
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet

def make():
    g = arch.build_graph("resnet50", 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = HipNet("resnet50", 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return net.eval()

x = torch.from_numpy(synth.synth_images(256, 3, 224, 224, seed=0)).cuda()
n1, n2 = make(), make()
def timed(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
print("one stream, batch 256: %.3f ms" % timed(lambda: n1.probabilities(x)))
for parts in (2, 4):
    nets = [n1, n2] + [make() for _ in range(parts - 2)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = list(x.chunk(parts))
    def run():
        for net, st, xi in zip(nets, streams, xs):
            with torch.cuda.stream(st):
                net.probabilities(xi)
    print("%d streams x batch %d: %.3f ms" % (parts, 256 // parts, timed(run)))
    print("same %d parts, one stream: %.3f ms" % (parts, timed(lambda: [net.probabilities(xi) for net, xi in zip(nets, xs)])))
