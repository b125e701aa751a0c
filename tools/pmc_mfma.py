"""MFMA utilisation and wave-cycle split of the conv kernels from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY) of
`bench.py --mode infer`: the conv dispatches of ONE steady-state forward (between the last two to_nhwc4 launches).
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; MfmaUtil = MFMA busy cycles / (GUI_ACTIVE/8 x 256 CUs x 4 SIMDs).

usage: pmc_mfma.py counter_collection.csv out.json"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
disp = {}
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
marks = [i for i in ids if "to_nhwc4" in disp[i]["name"]]
seg = [i for i in ids if marks[-2] <= i < marks[-1]]
agg, n = collections.defaultdict(float), 0
for i in seg:
    if any(k in disp[i]["name"] for k in ("conv_igemm_kernel", "conv_pw_kernel", "conv_pwr_kernel", "conv_c3_kernel", "conv_stem", "conv_bneck_kernel", "conv_btail_kernel")):
        n += 1
        for k, v in disp[i].items():
            if k != "name":
                agg[k] += v
gui = agg["GRBM_GUI_ACTIVE"] / 8.0
out = {
    "kernels": "conv_bneck_kernel + conv_pw_kernel + conv_pwr_kernel + conv_c3_kernel + conv_igemm_kernel + conv_stem_kernel, one forward", "launches": n, "counters": dict(agg),
    "gpu_active_cycles": gui,
    "mfma_util": agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4),
    "wave_cycles_split": {"wait_any (s_waitcnt / barrier)": agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"],
                          "wait_inst_any (issue stall)": agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"],
                          "active_inst_any": agg["SQ_ACTIVE_INST_ANY"] / agg["SQ_WAVE_CYCLES"]},
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("launches", "mfma_util", "wave_cycles_split")}, indent=1))
