#!/usr/bin/env python3
"""Per-launch HBM traffic of the level-L warp+correlation launch from the rocprofv3 --pmc passes of tools/pmc_l3.sh.

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts a 128-byte request of a 16-byte-per-lane read as 64
bytes (MI355X_MICROARCH.md, section HBM): it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  The JSON carries
the hash of the kernel sources it was measured on: bench.py quotes `traffic` only when that hash matches the working tree."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir, level = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # 8: bench.py's standalone batch-8 level-3 launches (`roofline_batch8`)
C, n, s = {1: (64, 1024, 2), 2: (64, 512, 2), 3: (64, 256, 2)}[level]
tiles = (n // s // 8) ** 2 * batch


# what each figure's launch must be (fails loudly when the launch policy, the forward or the profiled process changes)
EXPECT = {(3, 1): "warp_corr_v7_kernel<true>", (1, 1): "warp_corr_v6_kernel<true, 2>", (2, 1): "warp_corr_v6_kernel<true, 2>", (3, 8): "warp_corr_v6_kernel<true, 2>"}


def pick_dispatches(rows):
    """Dispatch ids of the level's launches among the warp+correlation rows of one profiled process, by position -- a lean bench
    run consists of forwards only and a forward launches warp+correlation for levels 6, 5, 4, 3, 2, 1 in that order (dispatch k of
    the process -> level 6 - k % 6); the batch-8 figures come from a process that launches nothing else (tools/wc_standalone.py),
    its first two launches dropped as warm-up -- and checked against the kernel the launch policy picks for that shape."""
    by_disp = {}
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
    ids = sorted(by_disp)
    if batch == 8:
        pick = ids[2:]
    else:
        assert len(ids) % 6 == 0, f"{len(ids)} warp+correlation dispatches: not a whole number of forwards"
        pick = [d for k, d in enumerate(ids) if 6 - k % 6 == level][1:]          # the first forward is the warm-up
    want = EXPECT.get((level, batch))
    if want is None:
        sys.exit(f"pmc_summary.py: no expected kernel is recorded for level {level} at batch {batch} (known: {sorted(EXPECT)}): add it to EXPECT "
                 "after checking which kernel the launch policy picks for that shape")
    for d in pick:
        name = by_disp[d][0]["Kernel_Name"]
        assert want in name, f"dispatch {d} picked for level {level} batch {batch} is {name!r}, expected {want}"
    return by_disp, pick


def trace_duration_us():
    """Average duration of the same launches in the plain --kernel-trace pass (no counters)."""
    rows = []
    sub = "b8_trace" if batch == 8 else "net_trace"
    for path in glob.glob(os.path.join(out_dir, sub, "**", "*kernel_trace.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(path)) if "warp_corr" in r["Kernel_Name"]]
    if not rows:
        return None, 0
    by_disp, pick = pick_dispatches(rows)
    d = [(int(by_disp[k][0]["End_Timestamp"]) - int(by_disp[k][0]["Start_Timestamp"])) * 1e-3 for k in pick]
    return sum(d) / len(d), len(d)


def counters(tag):
    """Counter values of the level's launches in the --pmc pass `tag`."""
    rows = []
    sub = ("b8_" if batch == 8 else "net_") + tag
    for path in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(path)) if "warp_corr" in r["Kernel_Name"]]
    by_disp, pick = pick_dispatches(rows)
    vals = {}
    for d in pick:
        for r in by_disp[d]:
            vals.setdefault(r["Counter_Name"], {}).setdefault(d, 0.0)
            vals[r["Counter_Name"]][d] += float(r["Counter_Value"])
            vals["_kernel"] = r["Kernel_Name"]
    return vals


res = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
    res.update(counters(tag))
mean = lambda d: sum(d.values()) / max(1, len(d))        # noqa: E731
fetch_kb, write_kb = mean(res.get("FETCH_SIZE", {})), mean(res.get("WRITE_SIZE", {}))
h = hashlib.sha256()
for rel in ("piv_liteflownet-pytorch_amd/csrc/warp_corr.hip", "piv_liteflownet-pytorch_amd/csrc/common.h"):
    h.update(open(os.path.join(ROOT, rel), "rb").read())
no = n // s
alg = 4 * (C * no * no + C * n * n + 2 * n * n + 49 * no * no) * batch
trace_us, trace_n = trace_duration_us()
print(json.dumps({
    "kernel": f"{res.get('_kernel', '?')} (level {level} of PIV 1024x1024 B={batch}: C={C}, stride {s}, {tiles} tiles)",
    "launches_averaged": len(res.get("FETCH_SIZE", {})),
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane reads; MI355X_MICROARCH.md section HBM); WRITE_SIZE exact",
    "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
    "algorithmic_bytes_per_launch": alg,
    "TCC_HIT_sum": mean(res.get("TCC_HIT_sum", {})), "TCC_MISS_sum": mean(res.get("TCC_MISS_sum", {})),
    "rocprofv3_kernel_trace_avg_us": None if trace_us is None else round(trace_us, 2), "kernel_trace_launches_averaged": trace_n,
    "roofline_frac_from_kernel_trace": None if trace_us is None else round(alg / (trace_us * 1e-6) / 8e12, 4),
    "kernel_source_sha256_16": h.hexdigest()[:16],
    "passes": "three separate rocprofv3 --pmc runs (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum) of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-level 0 --no-arithmetic --lean` (batch 1, launches picked by position in the forward) or of `tools/wc_standalone.py --level 3 --batch 8` (batch 8), tools/pmc_l3.sh",
}, indent=1))
