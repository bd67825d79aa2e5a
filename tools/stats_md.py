#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the markdown summary kept under profiles/.

  python tools/stats_md.py gpurun_out/prof_dir FORWARDS [--l3 warp_corr_v4_kernel<true>] > profiles/rNN_x.md
"""
import csv
import glob
import os
import sys


def main():
    d, forwards = sys.argv[1], int(sys.argv[2])
    l3 = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--l3" else "warp_corr_v4_kernel<true>"
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"Sum of kernel time per forward: {total / forwards / 1e6:.2f} ms over {forwards} forwards.\n")
    print("| kernel | calls | avg us | ms / forward | % |\n|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        t = float(r["TotalDurationNs"])
        if t / total < 0.0005:
            continue
        print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {t / forwards / 1e6:.3f} | {100 * t / total:.1f} |")
    tr = [r for r in csv.DictReader(open(trace)) if l3 in r["Kernel_Name"] and r["Grid_Size_X"] == str(256 * 1024)]
    if tr:
        tr.sort(key=lambda r: int(r["Start_Timestamp"]))
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
        print(f"\nLevel-3 warp+correlation launches (`{l3}`, grid 256 x 1024): n={len(du)} avg {sum(du) / len(du):.2f} us "
              f"min {min(du):.2f} max {max(du):.2f} -> {24707072 / (sum(du) / len(du)) / 1e3:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
