"""HBM traffic of the convolution kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the
same bench command), corrected as MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB, FETCH_SIZE
reports exactly 1/2 of a wide coalesced read stream (checked on to_nhwc4_kernel, whose byte counts are known),
WRITE_SIZE is exact.

usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [infer|train]
Takes the dispatches of ONE steady-state step (between the last two to_nhwc4_kernel launches: every step starts with
the layout conversion of its input batch), so autotuning launches are excluded."""
import csv
import datetime
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (TRAFFIC_SOURCES / TRAIN_SOURCES: the figure is quoted only for the sources it was taken on)

SOURCES = {"infer": bench.TRAFFIC_SOURCES, "train": bench.TRAIN_SOURCES}
KERNELS = {"infer": ("conv_igemm_kernel", "conv_pw_kernel", "conv_pwr_kernel", "conv_c3_kernel", "conv_stem_kernel", "conv_bneck_kernel", "conv_btail_kernel"),
           "train": ("conv_igemm_kernel", "conv_wgrad_kernel", "conv_stem_kernel")}


def kernel_source_sha(mode):
    h = hashlib.sha256()
    for name in SOURCES[mode]:
        f = Path(__file__).resolve().parent.parent / "syke-pic_amd" / "csrc" / name
        h.update(name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _git_head():
    try:
        return subprocess.check_output(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], text=True,
                                       stderr=subprocess.DEVNULL).strip()
    except Exception:   # noqa: BLE001  (the GPU box snapshot has no .git: pass SPK_COMMIT)
        return None


def one_pass(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "to_nhwc4" in r["Kernel_Name"]]
    return rows[marks[-2]:marks[-1]]


def main():
    fetch, write, out = sys.argv[1:4]
    mode = sys.argv[4] if len(sys.argv) > 4 else "infer"
    f, w = one_pass(fetch), one_pass(write)
    is_conv = lambda r: any(k in r["Kernel_Name"] for k in KERNELS[mode])   # noqa: E731
    cal_f = [float(r["Counter_Value"]) for r in f if "to_nhwc4" in r["Kernel_Name"]]
    cal_w = [float(r["Counter_Value"]) for r in w if "to_nhwc4" in r["Kernel_Name"]]
    conv_f = [float(r["Counter_Value"]) for r in f if is_conv(r)]
    conv_w = [float(r["Counter_Value"]) for r in w if is_conv(r)]
    per_kernel = {}
    for rows, key, mul in ((f, "fetch_bytes", 2048), (w, "write_bytes", 1024)):
        for r in rows:
            for k in KERNELS[mode]:
                if k in r["Kernel_Name"]:
                    d = per_kernel.setdefault(k, {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
                    d[key] += float(r["Counter_Value"]) * mul
                    d["launches"] += key == "fetch_bytes"
    all_f = sum(float(r["Counter_Value"]) for r in f)
    all_w = sum(float(r["Counter_Value"]) for r in w)
    res = {
        "kernel": " + ".join(KERNELS[mode]),
        "mode": mode,
        "kernel_src_sha": kernel_source_sha(mode),
        "taken_at": {"date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"),
                     "commit": os.environ.get("SPK_COMMIT") or _git_head(),
                     "eval_streams": int(os.environ.get("SPK_EVAL_STREAMS", "2")),
                     "wgrad_stream": int(os.environ.get("SPK_WGRAD_STREAM", "1"))},
        "launches_per_step": len(conv_f),
        "fetch_bytes_per_step": sum(conv_f) * 1024 * 2,
        "write_bytes_per_step": sum(conv_w) * 1024,
        "per_kernel": per_kernel,
        "calibration": {"to_nhwc4 FETCH_SIZE KiB (raw)": cal_f, "to_nhwc4 WRITE_SIZE KiB": cal_w,
                        "note": "batch 256 x 3 x 224 x 224 fp32 in = 154,140,672 B; NHWC4 16-bit out = 102,760,448 B"},
        "all_kernels_bytes_per_step": all_f * 2048 + all_w * 1024,
    }
    res["traffic_bytes_per_launch"] = (res["fetch_bytes_per_step"] + res["write_bytes_per_step"]) / max(1, len(conv_f))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
