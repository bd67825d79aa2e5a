#!/bin/bash
# Counters of the Winograd conv kernel (128 -> 128, 3 x 3, 1024 x 1024) beside the direct fp32 kernel: matrix-pipe busy cycles, LDS
# bank conflicts, wave wait cycles.  One rocprofv3 --pmc pass (no other trace domains).  bash tools/pmc_wino.sh -> gpurun_out/pmc_wino/
set -e
OUT=$PWD/gpurun_out/pmc_wino
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY} GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq" -- python3 tools/bench_wino.py --levels 1 --layers ${LAYERS:-128x128} --rounds 2 --n 3 ${ARGS:-} > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(out, "sq", "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
agg = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "conv" not in k:
        continue
    agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 8.0) / 8.0
    m["launch_cycles"] = cyc
    m["mfma_busy_fraction_of_simd_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, cyc * 1024)
    m["lds_conflict_fraction_of_lds_cycles"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1.0))
    m["lds_active_fraction_of_cu_cycles"] = m.get("SQ_LDS_IDX_ACTIVE", 0.0) / max(1.0, cyc * 256)
    m["launches"] = len(c.get("GRBM_GUI_ACTIVE", []))
    res[k] = m
print(json.dumps({"workload": "3x3 layer at 1024 x 1024, batch 1 (tools/bench_wino.py): direct fp32 kernel and Winograd kernel", "kernels": res}, indent=1))
PY
