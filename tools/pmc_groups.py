#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per consecutive group of N dispatches of one kernel."""
import csv
import sys
from collections import defaultdict

path, pattern, group = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if pattern in r["Kernel_Name"]]
by_disp = defaultdict(dict)
order = []
for r in rows:
    d = int(r["Dispatch_Id"])
    if d not in by_disp:
        order.append(d)
    by_disp[d][r["Counter_Name"]] = float(r["Counter_Value"])
order.sort()
for g in range(0, len(order), group):
    ds = order[g:g + group]
    names = sorted(by_disp[ds[0]].keys())
    print(f"group {g // group} (n={len(ds)}): " + "  ".join(f"{n}={sum(by_disp[d].get(n, 0) for d in ds) / len(ds):.0f}" for n in names))
