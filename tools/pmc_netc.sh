#!/bin/bash
# Counters of NetC's dedicated kernels inside the forward (conv_c3k7<true>: conv1 + the level-1 1 x 1 layers; conv_s2c32: the stride-2
# layers from 32 channels): matrix-pipe busy cycles, waits, HBM bytes.  Separate rocprofv3 --pmc passes (with --kernel-trace for the per-dispatch rows; none of the hip / hsa / memory-copy / marker domains).
#   bash tools/pmc_netc.sh -> gpurun_out/pmc_netc/
set -e
OUT=$PWD/gpurun_out/pmc_netc
mkdir -p "$OUT"
export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-level 0 --no-arithmetic --lean > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
agg = {}
for p in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0]
        if "conv_c3k7" not in k and "conv_s2c32" not in k:
            continue
        agg.setdefault(k, {}).setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        agg[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
res = {}
for k, c in agg.items():
    m = {n: sum(v.values()) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 8.0) / 8.0
    m["launch_cycles"] = cyc
    m["mfma_busy_fraction_of_simd_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, cyc * 1024)
    if "FETCH_SIZE" in m:
        m["hbm_read_MB_x2_correction"] = m["FETCH_SIZE"] * 2 * 1024 / 1e6
    if "WRITE_SIZE" in m:
        m["hbm_write_MB"] = m["WRITE_SIZE"] * 1024 / 1e6
    res[k] = m
print(json.dumps(res, indent=1))
PY
