#!/usr/bin/env python3
"""profiles/rNN_a_kernel_stats.md from what tools/profile_round.sh left under gpurun_out/rNN: the stats table, the level-3 / level-1
warp+correlation launches of the trace, the per-level timeline and the un-profiled bench line.   python tools/compose_profile.py r05"""
import csv
import glob
import subprocess
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
WARMUP = int(sys.argv[2]) if len(sys.argv) > 2 else 3        # the --warmup of the profiled bench command: those forwards are left out
R = f"gpurun_out/{rnd}"
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
stats = open(f"{R}/kernel_stats_fp32.md").read()
tl = open(f"{R}/timeline_fp32.txt").read()
bench = open(f"{R}/bench_default.json").read().strip().splitlines()[-1]
trace = glob.glob(f"{R}/prof_fp32/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# a forward launches warp+correlation for levels 6, 5, 4, 3, 2, 1 in that order: launch k of the process -> level 6 - k % 6; the
# kernel each pick must be is checked (the launch policy or the forward changing fails loudly instead of mislabelling a row)
wc = sorted((r for r in rows if "warp_corr" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
assert len(wc) % 6 == 0, f"{len(wc)} warp+correlation launches: not a whole number of forwards"
lvl = lambda L: [r for k, r in enumerate(wc) if 6 - k % 6 == L][WARMUP:]        # the warm-up forwards of the bench command are left out
assert all("warp_corr_v7_kernel<true>" in r["Kernel_Name"] for r in lvl(3)), "level 3 is not the one-tile-per-CU kernel"
assert all("warp_corr_v6_kernel<true, 2>" in r["Kernel_Name"] for r in lvl(1)), "level 1 is not the persistent kernel"
d3 = [dur(r) for r in lvl(3)]
d1 = [dur(r) for r in lvl(1)]
l3 = (f"Level-3 warp+correlation launches (`warp_corr_v7_kernel<true>`, grid 256 x 1024): n={len(d3)} avg {sum(d3) / len(d3):.2f} us "
      f"min {min(d3):.2f} max {max(d3):.2f} -> {24707072 / (sum(d3) / len(d3)) / 1e3:.0f} GB/s algorithmic")
l1 = (f"Level-1 warp+correlation launches (`warp_corr_v6_kernel<true, 2>`, 395 MB): n={len(d1)} avg {sum(d1) / len(d1):.2f} us "
      f"-> {395313152 / (sum(d1) / len(d1)) / 1e3:.0f} GB/s algorithmic")
print(f"""# Round {rnd[1:].lstrip('0')}: kernel statistics of the default bench (fp32: Winograd F(2x2,3x3) with exactly split operands on the bf16 matrix cores / on the fp32 matrix instruction, direct convolution on the fp32 instruction)

Command (tools/profile_round.sh, on the MI355X box): `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/{rnd}/prof_fp32 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-arithmetic --lean`  
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
