#!/bin/bash
# Matrix-core busy counters of the dominant conv kernel (128 -> 128, 3 x 3, 1024 x 1024) under each fp32-grade arithmetic, one
# rocprofv3 --pmc pass (no other trace domains).  Run on the GPU box from the repo root:  bash tools/pmc_conv.sh  -> gpurun_out/pmc_conv/
set -e
OUT=$PWD/gpurun_out/pmc_conv
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python3 tools/bench_ops.py conv --variants 0,2000000,3000000 --filter "L1 R.2" > "$OUT/mfma.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(out, "mfma", "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
agg = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "conv" not in k or "reduce" in k:
        continue
    agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (32 cycles per 32x32x16 fp16 MFMA, 64 per 32x32x2 fp32 one);
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launch = GRBM_GUI_ACTIVE / 8
    cyc = m.get("GRBM_GUI_ACTIVE", 8.0) / 8.0
    m["launch_cycles"] = cyc
    m["mfma_busy_fraction_of_simd_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, cyc * 1024)
    m["launches"] = len(c.get("GRBM_GUI_ACTIVE", []))
    res[k] = m
print(json.dumps({"workload": "128 -> 128, 3 x 3, 1024 x 1024, batch 1 (tools/bench_ops.py conv --variants 0,2000000,3000000 --filter 'L1 R.2'): the fp32 "
                              "instruction kernel, the six-term and the three-term split kernels", "kernels": res}, indent=1))
PY
