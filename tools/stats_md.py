#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --output-format csv` directory into the markdown summary kept under profiles/: per-kernel time
over the TIMED forwards only.  A forward starts at its `prep_images_kernel` launch; everything in front of the first one (the weight
uploads of pivlfn_create: `__amd_rocclr_copyBuffer`, fills) and the first WARMUP forwards are left out -- round 5's table divided the
whole process's kernel time by the forward count and listed create-time copies as 0.18 ms / forward.

  python tools/stats_md.py gpurun_out/prof_dir WARMUP [--l3 'warp_corr_v7_kernel<true>'] > profiles/rNN_x.md
"""
import csv
import glob
import os
import sys


def main():
    d, warmup = sys.argv[1], int(sys.argv[2])
    l3 = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--l3" else "warp_corr_v7_kernel<true>"
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "prep_images_kernel" in r["Kernel_Name"]]
    assert len(starts) > warmup, f"{len(starts)} forwards in the trace, {warmup} to drop"
    timed = rows[starts[warmup]:]
    forwards = len(starts) - warmup
    agg = {}
    for r in timed:
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    total = sum(a[1] for a in agg.values())
    wall = (int(timed[-1]["End_Timestamp"]) - int(timed[0]["Start_Timestamp"])) / forwards
    print(f"Kernel time per forward: {total / forwards / 1e6:.2f} ms (sum over all streams); wall clock per forward {wall / 1e6:.2f} ms; {forwards} timed forwards, "
          f"{warmup} warm-up forwards and {starts[0]} launches in front of the first forward (weight uploads) left out.\n")
    print("| kernel | calls / forward | avg us | ms / forward | % |\n|---|---|---|---|---|")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / total < 0.0005:
            continue
        print(f"| `{name[:90]}` | {n / forwards:.1f} | {t / n / 1e3:.1f} | {t / forwards / 1e6:.3f} | {100 * t / total:.1f} |")
    tr = [r for r in timed if l3 in r["Kernel_Name"] and r["Grid_Size_X"] == str(256 * 1024)]
    if tr:
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
        print(f"\nLevel-3 warp+correlation launches (`{l3}`, grid 256 x 1024): n={len(du)} avg {sum(du) / len(du):.2f} us "
              f"min {min(du):.2f} max {max(du):.2f} -> {24707072 / (sum(du) / len(du)) / 1e3:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
