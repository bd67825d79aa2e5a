#!/usr/bin/env python3
"""Per-level wall time of one PIV forward from a rocprofv3 kernel_trace.csv: segments end at every reg_tail_kernel (the last
kernel of a level); the first segment (up to the level-6 warp+correlation) is the pyramid + NetC.

  python tools/level_timeline.py gpurun_out/prof_dir [forward_index]"""
import csv
import glob
import os
import sys
from collections import defaultdict

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
print(f"forward #{which}: {len(fw)} launches, wall {(max(int(r['End_Timestamp']) for r in fw) - t0) / 1e6:.3f} ms")
seg, name, level = [], "pyramid + NetC", 6
segs = []
for r in fw:
    k = r["Kernel_Name"]
    short = k.split("pivlfn::")[-1].split("(")[0]
    if name == "pyramid + NetC" and "warp_corr" in k:
        segs.append((name, seg))
        seg, name = [], f"level {level}"
    seg.append((short, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
    if "reg_tail_kernel" in k:
        segs.append((name, seg))
        level -= 1
        seg, name = [], f"level {level}"
if seg:
    segs.append((name + " (rest)", seg))
for name, seg in segs:
    if not seg:
        continue
    wall = (max(e for _, _, e, _ in seg) - min(s for _, s, _, _ in seg)) / 1e3
    busy = sum(e - s for _, s, e, _ in seg) / 1e3
    by = defaultdict(lambda: [0, 0.0])
    for n, s, e, _ in seg:
        by[n][0] += 1
        by[n][1] += (e - s) / 1e3
    top = sorted(by.items(), key=lambda kv: -kv[1][1])[:6]
    print(f"{name:16s} wall {wall:9.1f} us  kernel-sum {busy:9.1f} us  launches {len(seg):3d} | " +
          "; ".join(f"{n} x{c} {t:.0f}" for n, (c, t) in top))
