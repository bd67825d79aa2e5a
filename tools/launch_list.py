#!/usr/bin/env python3
"""Every launch of one PIV forward from a rocprofv3 kernel_trace.csv, in start order: start (us from the forward's first launch),
duration, gap to the previous launch's end, queue, grid / workgroup, kernel.

  python tools/launch_list.py gpurun_out/prof_dir [forward_index]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(trace)) if "pivlfn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "prep_images_kernel" in r["Kernel_Name"]]
i0 = starts[which]
i1 = starts[which + 1] if which + 1 < len(starts) and which != -1 else len(rows)
fw = rows[i0:i1]
t0 = int(fw[0]["Start_Timestamp"])
prev_end = t0
for r in fw:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("pivlfn::")[-1].split("(")[0]
    grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size_X", "?")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>2}  grid {grid:>16} wg {wg:>4}  lds {r.get('LDS_Block_Size', '?'):>6}  {name}")
    prev_end = max(prev_end, e)
