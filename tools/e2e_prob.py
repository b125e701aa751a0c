"""End-to-end `sykepic prob` throughput on a synthetic IFCB sample (disk ->
.roi blob -> GPU preprocessing -> ResNet forward -> CSV)."""
import sys, time, shutil, tempfile
from collections import namedtuple
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "syke-pic_amd")]
import numpy as np, torch
from sykepic_hip import arch, synth, prob
n_roi = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
network = sys.argv[2] if len(sys.argv) > 2 else "resnet18"
tmp = Path(tempfile.mkdtemp())
rng = np.random.RandomState(0)
adc, blobs, off = [], [], 0
for i in range(n_roi):
    h, w = int(rng.randint(20, 120)), int(rng.randint(30, 300))
    cols = ["0"] * 24; cols[15], cols[16], cols[17] = str(w), str(h), str(off)
    adc.append(",".join(cols)); blobs.append(rng.randint(0, 256, h * w).astype(np.uint8)); off += h * w
raw = tmp / "raw"; raw.mkdir()
(raw / "D20200101T000000_IFCB114.adc").write_text("\n".join(adc) + "\n")
np.concatenate(blobs).tofile(raw / "D20200101T000000_IFCB114.roi")
raw3 = tmp / "raw3"; raw3.mkdir()
import os
for k in range(3):
    for ext in ("adc", "roi"):
        os.link(raw / f"D20200101T000000_IFCB114.{ext}", raw3 / f"D20200101T00000{k + 1}_IFCB114.{ext}")
raw9 = tmp / "raw9"; raw9.mkdir()
for k in range(9):
    for ext in ("adc", "roi"):
        os.link(raw / f"D20200101T000000_IFCB114.{ext}", raw9 / f"D20200101T0001{k:02d}_IFCB114.{ext}")
model = tmp / "model"; model.mkdir()
g = arch.build_graph(network, 50)
sd = synth.synth_state_dict(arch.param_specs(g), seed=2)
torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, model / "best_state.pth")
(model / "class_names.txt").write_text("\n".join(f"class_{i}" for i in range(50)))
cfg = (ROOT / "tests/golden/ref_data/config.ini").read_text().replace("network = resnet18", f"network = {network}")
(model / "config.ini").write_text(cfg)
Args = namedtuple("Args", "raw samples image_dir images model out batch_size num_workers force")
import os
if os.environ.get("E2E_CALIBRATE"):
    # what `sykepic calibrate -m MODEL -r RAW -n 2048` does: act_means.pth beside best_state.pth, after which `prob` runs
    # the calibrated single-pass mode
    CArgs = namedtuple("CArgs", "raw samples image_dir images model batch_size num_images")
    t0 = time.perf_counter()
    n_cal = prob.calibrate_call(CArgs(str(raw), None, None, None, str(model), 256, 2048))
    print(f"calibrated on {n_cal} ROIs in {time.perf_counter() - t0:.2f} s", flush=True)
for bs in (64, 512):
    out = tmp / f"out{bs}"
    prof = None
    if os.environ.get("E2E_PROFILE") and bs == 512:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    prob.call(Args(str(raw), None, None, None, str(model), out, bs, 2, True))
    torch.cuda.synchronize()
    dt_first = time.perf_counter() - t0        # includes loading the model and tuning this batch size's kernels
    t0 = time.perf_counter()
    prob.call(Args(str(raw3), None, None, None, str(model), out, bs, 2, True))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3        # three more samples in one call: what a further sample of a run costs
    t0 = time.perf_counter()
    prob.call(Args(str(raw9), None, None, None, str(model), out, bs, 2, True))
    torch.cuda.synchronize()
    dt9 = (time.perf_counter() - t0) / 9       # nine samples in one call: the model load is 1/9 per sample
    if prof:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
    print(f"{network} 180x180, {n_roi} ROIs ({off/1e6:.0f} MB .roi), batch {bs}: first call {dt_first:.2f} s = {n_roi/dt_first:.0f} ROI/s "
          f"(model load + kernel tuning included); three more samples in one call: {dt:.2f} s each = {n_roi/dt:.0f} ROI/s; "
          f"nine samples in one call: {dt9:.3f} s each = {n_roi/dt9:.0f} ROI/s", flush=True)
shutil.rmtree(tmp)
