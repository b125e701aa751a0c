import sys
sys.path[:0] = ["/root/repo", "/root/repo/syke-pic_amd"]
import numpy as np, torch, time
from sykepic_hip import arch, synth
from sykepic_hip.net import HipNet
for name, n in (("resnet50", 1024), ("efficientnet_b4", 1024)):
    g = arch.build_graph(name, 50)
    sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
    net = HipNet(name, 50, weights=None)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); net.eval()
    x = torch.from_numpy(synth.synth_images(64, 3, 224, 224, seed=1)).cuda().repeat(n // 64, 1, 1, 1)
    ref = net.probabilities(x[:64])
    out = net.probabilities(x)
    torch.cuda.synchronize(); t0 = time.time()
    out = net.probabilities(x); torch.cuda.synchronize(); dt = time.time() - t0
    d = (out.view(n // 64, 64, -1) - ref[None]).abs().max().item()
    print(name, "batch", n, "max diff vs batch-64 run", d, f"{n/dt:.0f} img/s", "mem GB", torch.cuda.max_memory_allocated() / 1e9)
    del net
