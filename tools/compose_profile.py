#!/usr/bin/env python3
"""profiles/rNN_a_kernel_stats.md from what tools/profile_round.sh left under gpurun_out/rNN: the stats table, the level-3 / level-1
warp+correlation launches of the trace, the per-level timeline and the un-profiled bench line.   python tools/compose_profile.py r05"""
import csv
import glob
import subprocess
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
R = f"gpurun_out/{rnd}"
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
stats = open(f"{R}/kernel_stats_fp32.md").read()
tl = open(f"{R}/timeline_fp32.txt").read()
bench = open(f"{R}/bench_default.json").read().strip().splitlines()[-1]
trace = glob.glob(f"{R}/prof_fp32/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
d3 = [dur(r) for r in rows if "warp_corr_v7_kernel<true>" in r["Kernel_Name"] and r.get("Grid_Size_X") == "262144"]
d1 = [x for x in (dur(r) for r in rows if "warp_corr_v6_kernel<true, 2>" in r["Kernel_Name"]) if x > 60]
l3 = (f"Level-3 warp+correlation launches (`warp_corr_v7_kernel<true>`, grid 256 x 1024): n={len(d3)} avg {sum(d3) / len(d3):.2f} us "
      f"min {min(d3):.2f} max {max(d3):.2f} -> {24707072 / (sum(d3) / len(d3)) / 1e3:.0f} GB/s algorithmic")
l1 = (f"Level-1 warp+correlation launches (`warp_corr_v6_kernel<true, 2>`, 395 MB): n={len(d1)} avg {sum(d1) / len(d1):.2f} us "
      f"-> {395313152 / (sum(d1) / len(d1)) / 1e3:.0f} GB/s algorithmic")
print(f"""# Round {rnd[1:].lstrip('0')}: kernel statistics of the default bench (fp32: the fp32 matrix instruction + Winograd F(2x2,3x3))

Command (tools/profile_round.sh, on the MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/{rnd}/prof_fp32 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-arithmetic --lean`  
Sources: commit {commit}.  `--lean` leaves out the roofline_conv / arithmetic legs so that the table holds the forward's launches only.

{stats}
{l3}  
{l1}

## Per-level timeline of one forward (tools/level_timeline.py)

```
{tl}```

## The un-profiled bench line of the same build and box (python3 bench.py)

```
{bench}
```""")
