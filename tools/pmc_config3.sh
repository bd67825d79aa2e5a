#!/bin/bash
# BASELINE config #3 (batch 32 x 512x512): counter passes on the level-3 warp+correlation launch of that workload (128x128
# features, stride 2, 64 channels, 2048 tiles), each counter group in its own rocprofv3 --pmc run.  From the repo root on the
# GPU box:  bash tools/pmc_config3.sh  -> gpurun_out/pmc_config3/r05_pmc_config3.json   (needs the tools build for bench_ops.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_config3
mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -- python3 tools/bench_ops.py warp_corr --size 512 --batch 32 --levels 3 --variants 0 > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, hashlib, json, os, sys
out = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", ".")
vals, kern, dur = {}, "?", []
for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
    for p in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "warp_corr" not in r["Kernel_Name"]:
                continue
            kern = r["Kernel_Name"].split("(")[0]
            vals.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            vals[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for p in glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True):
        dur += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(p)) if "warp_corr" in r["Kernel_Name"]]
mean = lambda d: sum(d.values()) / max(1, len(d))
f, w = mean(vals.get("FETCH_SIZE", {})), mean(vals.get("WRITE_SIZE", {}))
h = hashlib.sha256()
for rel in ("piv_liteflownet-pytorch_amd/csrc/warp_corr.hip", "piv_liteflownet-pytorch_amd/csrc/common.h"):
    h.update(open(os.path.join(root, rel), "rb").read())
alg = 4 * (64 * 64 * 64 + 64 * 128 * 128 + 2 * 128 * 128 + 49 * 64 * 64) * 32
dur.sort()
print(json.dumps({"workload": "BASELINE config #3: level-3 warp+correlation of batch 32 x 512x512 (C=64, stride 2, 2048 tiles), standalone launches (tools/bench_ops.py warp_corr --size 512 --batch 32 --levels 3)",
                  "kernel": kern, "launches_averaged": len(vals.get("FETCH_SIZE", {})), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                  "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane reads); WRITE_SIZE exact",
                  "hbm_bytes_per_launch": int(round((2 * f + w) * 1024)), "algorithmic_bytes_per_launch": alg,
                  "TCC_HIT_sum": mean(vals.get("TCC_HIT_sum", {})), "TCC_MISS_sum": mean(vals.get("TCC_MISS_sum", {})),
                  "launch_us_under_the_profiler_median": dur[len(dur) // 2] if dur else None,
                  "kernel_source_sha256_16": h.hexdigest()[:16]}, indent=1))
PY
