"""Is the fp16 rounding of the INPUT pixels (k/255 is not an fp16 number) what is left of the error on trained nets?
Compares the HIP eval path with the fp32 oracle fed (a) the exact images, (b) the images rounded to fp16."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd"), str(ROOT / "tests")]
import numpy as np, torch
from oracle import refnet
import test_gpu_trained as T
for network, steps, lr, seed in (("resnet18", 300, 1e-3, 11), ("efficientnet_b0", 400, 2e-3, 12)):
    net, acc = T.train_hip(network, steps, lr, seed)
    ref = refnet.RefNet(network, T.CLASSES, head=(64, 32))
    ref.load_state_dict(net.state_dict()); ref.eval()
    x, y = T.labelled_images(256, 77)
    net.set_precision(split_weights=1)
    p = net.probabilities(x.cuda()).cpu().numpy().astype(np.float64)
    for tag, xin in (("exact input", x), ("fp16-rounded input", x.half().float())):
        pr = refnet.probabilities(ref, xin).numpy().astype(np.float64)
        d = np.abs(p - pr).max(1)
        print(f"{network} vs oracle on {tag:20s}: max {d.max():.2e} p90 {np.percentile(d, 90):.2e} median {np.median(d):.2e}", flush=True)
