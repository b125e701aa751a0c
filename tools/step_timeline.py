"""Per-stream timeline of ONE steady-state step from a `rocprofv3 --kernel-trace --output-format csv` trace:
busy time per queue, union, gaps on the main queue, and per-kernel totals per queue.
usage: step_timeline.py kernel_trace.csv [top=25]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# every step starts with the layout conversion of its input batch - one launch, or one per half batch (two eval streams:
# launches less than 100 us apart open the same step)
marks = []
for i, r in enumerate(rows):
    if "to_nhwc4" in r["Kernel_Name"]:
        if marks and int(r["Start_Timestamp"]) - int(rows[marks[-1]]["Start_Timestamp"]) < 100_000 and i - marks[-1] < 4:
            continue
        marks.append(i)
a, b = marks[-3], marks[-2]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])   # noqa: E731
for q, rs in byq.items():
    print(f"queue {q}: {len(rs)} kernels, busy {sum(map(dur, rs)) / 1e6:.3f} ms, first start "
          f"{(int(rs[0]['Start_Timestamp']) - t0) / 1e6:.3f}, last end {(max(int(r['End_Timestamp']) for r in rs) - t0) / 1e6:.3f}")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
u, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
print(f"union busy {(u + ce - cs) / 1e6:.3f} ms")
mainq = max(byq, key=lambda q: len(byq[q]))
rs = byq[mainq]
gaps = [int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"]) for i in range(len(rs) - 1)]
print(f"main queue gaps: sum {sum(g for g in gaps if g > 0) / 1e6:.3f} ms, median {statistics.median(gaps) / 1e3:.1f} us")
for g, i in sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]:
    print(f"  {g / 1e3:7.1f} us after {rs[i]['Kernel_Name'][:50]} before {rs[i + 1]['Kernel_Name'][:50]}")
for q, rs in byq.items():
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][0] += dur(r)
        agg[k][1] += 1
    print(f"--- queue {q}")
    for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"  {d / 1e6:7.3f} ms {n:4d} x {d / n / 1e3:7.1f} us  {k[:90]}")
