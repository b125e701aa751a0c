"""Kernels of ONE steady-state step in launch order, from a `rocprofv3 --kernel-trace --output-format csv` trace:
start offset, duration, grid, queue, short kernel name.  usage: step_sequence.py kernel_trace.csv [min_us=0]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = []
for i, r in enumerate(rows):   # (two eval streams: one conversion per half batch, less than 100 us apart, open one step)
    if "to_nhwc4" in r["Kernel_Name"]:
        if marks and int(r["Start_Timestamp"]) - int(rows[marks[-1]]["Start_Timestamp"]) < 100_000 and i - marks[-1] < 4:
            continue
        marks.append(i)
a, b = marks[-3], marks[-2]
t0 = int(rows[a]["Start_Timestamp"])
queues = {}
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < min_us:
        continue
    q = queues.setdefault(r["Queue_Id"], len(queues))
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    short = name.split("(")[0]
    grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:7.1f} us  q{q}  grid {grid:>9}  {short[:110]}")
