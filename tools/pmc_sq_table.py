"""Table of the per-kernel SQ counters collected by tools/pmc_sq.sh (three rocprofv3 --pmc passes).
Usage: python3 tools/pmc_sq_table.py <tag> [out.txt]   (reads gpurun_out/<tag>_pmc_sq_set{1,2,3}.csv)"""
import collections
import csv
import re
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
    return d, cnt


def main():
    tag = sys.argv[1]
    base = f"gpurun_out/{tag}_pmc_sq_set"
    a_all, n_all = load(base + "1.csv")
    b_all, _ = load(base + "2.csv")
    c_all, _ = load(base + "3.csv")
    rows = []
    for k, a in a_all.items():
        if a.get("SQ_WAVE_CYCLES", 0) <= 0:
            continue
        rows.append((a.get("GRBM_GUI_ACTIVE", 0), k, n_all[k], a, b_all.get(k, {}), c_all.get(k, {})))
    rows.sort(reverse=True)
    out = ["kernel | launches | GPU-active cycles per launch | MFMA pipes busy (SQ_VALU_MFMA_BUSY_CYCLES / (GPU-active cycles "
           "per XCD x 1024 SIMDs); - when not collected) | wave cycles: waiting (s_waitcnt/barrier) / issue stall / "
           "issuing | issuing split: VALU / LDS / VMEM | LDS bank-conflict cycles per LDS instruction | per launch: VALU, "
           "LDS, VMEM-read, VMEM-write instructions"]
    for g, k, n, a, b, c in rows[:40]:
        wc = a["SQ_WAVE_CYCLES"]
        f = lambda d, x: d.get(x, 0) / wc
        mf = "%.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8.0 * 1024.0)) if a.get("SQ_VALU_MFMA_BUSY_CYCLES") and g > 0 else "-"
        out.append("%-60s %5d %9.0f | %5s | %.2f %.2f %.2f | %.2f %.2f %.2f | %5.2f | %d %d %d %d" % (
            k[:60], n, g / n, mf, f(a, "SQ_WAIT_ANY"), f(a, "SQ_WAIT_INST_ANY"), f(a, "SQ_ACTIVE_INST_ANY"),
            f(b, "SQ_ACTIVE_INST_VALU"), f(b, "SQ_ACTIVE_INST_LDS"), f(b, "SQ_ACTIVE_INST_VMEM"),
            c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_INSTS_LDS", 1.0)),
            c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n, c.get("SQ_INSTS_VMEM_RD", 0) / n,
            c.get("SQ_INSTS_VMEM_WR", 0) / n))
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
