#!/bin/bash
# Counters of the separable distance convolutions (conv_col7 / conv_row7, 1024 x 1024): matrix-pipe busy cycles, wave waits,
# instruction counts.  One rocprofv3 --pmc pass (with --kernel-trace for the per-dispatch rows; none of the hip / hsa / memory-copy / marker domains).  bash tools/pmc_dist.sh -> gpurun_out/pmc_dist/
set -e
OUT=$PWD/gpurun_out/pmc_dist
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_MFMA} GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq" -- python3 tools/bench_dist.py --sizes 1024 --knobs 0 > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
agg = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "conv_col7" not in k and "conv_row7" not in k:
        continue
    agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 8.0) / 8.0
    m["launch_cycles"] = cyc
    m["mfma_busy_fraction_of_simd_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, cyc * 1024)
    res[k] = m
print(json.dumps(res, indent=1))
PY
