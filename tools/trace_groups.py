#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv by consecutive groups of N dispatches of one kernel (for tools/bench_ops.py runs)."""
import csv
import sys

path, pattern, group = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if pattern in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for g in range(0, len(rows), group):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[g:g + group])
    if d:
        print(f"group {g // group}: n={len(d)} min {d[0]:.2f} us  med {d[len(d) // 2]:.2f} us  p90 {d[int(len(d) * 0.9)]:.2f} us  grid {rows[g]['Grid_Size_X']}")
