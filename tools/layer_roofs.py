"""Per-layer roof table from a `bench.py --layers-out` file: for every launch of the forward the roof that bounds it - HBM
(8 TB/s peak) or MFMA (2.5 PFLOP/s dense fp16 peak) - and the fraction of that roof it reaches (algorithmic bytes / FLOPs over
the HIP-event time on one stream).  usage: layer_roofs.py layers.json > table.txt"""
import json
import sys

HBM, MFMA = 8.0e12, 2.5e15
rows = json.load(open(sys.argv[1]))
tot = sum(r["ms"] for r in rows)
print(f"{'layer':66s} {'us':>7s} {'GFLOP':>7s} {'MB':>7s}  roof  frac of roof   (TFLOP/s, GB/s)")
acc = {"hbm": 0.0, "mfma": 0.0}
for r in rows:
    t = r["ms"] * 1e-3
    if t <= 0 or (r["gflop"] == 0 and r["mbytes"] == 0):
        continue
    t_hbm, t_mfma = r["mbytes"] * 1e6 / HBM, r["gflop"] * 1e9 / MFMA
    roof = "hbm " if t_hbm >= t_mfma else "mfma"
    frac = max(t_hbm, t_mfma) / t
    acc[roof.strip()] += r["ms"]
    print(f"{r['layer'][:66]:66s} {r['ms'] * 1e3:7.1f} {r['gflop']:7.1f} {r['mbytes']:7.1f}  {roof}  {frac:5.2f}          "
          f"({r['tflops'] or 0:6.1f}, {r['gbs'] or 0:6.1f})")
fl, by = sum(r["gflop"] for r in rows), sum(r["mbytes"] for r in rows)
print(f"\nall launches on one stream: {tot:.3f} ms; {fl / 1e3:.3f} TFLOP = {fl / tot:.0f} TFLOP/s = {fl / tot / 2500:.3f} of the MFMA peak; "
      f"{by / 1e3:.2f} GB algorithmic = {by / tot / 1e3:.2f} TB/s = {by / tot / 1e3 / 8:.3f} of the HBM peak; "
      f"time in HBM-roofed launches {acc['hbm']:.3f} ms, in MFMA-roofed launches {acc['mfma']:.3f} ms")
