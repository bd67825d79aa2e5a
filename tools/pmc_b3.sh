#!/bin/bash
# Counters of the split-operand Winograd kernel (csrc/conv_wino_b3.hip) beside the fp32-instruction Winograd kernel on the 128 -> 128
# 3 x 3 layer at 1024 x 1024: HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes), matrix-pipe busy cycles and the clock
# (GRBM_GUI_ACTIVE / duration).  Each counter group in its own rocprofv3 --pmc run, never combined with other trace domains.
# From the repo root on the GPU box:  bash tools/pmc_b3.sh  -> gpurun_out/pmc_b3/r06_pmc_b3.json
set -e
OUT=$PWD/gpurun_out/pmc_b3
mkdir -p "$OUT"
export TMPDIR=/tmp
LAYERS=${LAYERS:-128x128}
# PMC_GROUPS: other counter groups, separated by ';' (e.g. "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"); PMC_JSON: output name
IFS=';' read -ra GROUPS_ <<< "${PMC_GROUPS:-FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES}"
export PMC_JSON=${PMC_JSON:-r06_pmc_b3.json}
for grp in "${GROUPS_[@]}"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -- python3 tools/bench_b3.py --layers $LAYERS --rounds 1 --n 3 > "$OUT/$tag.log" 2>&1 || { echo "pass $tag failed"; tail -3 "$OUT/$tag.log"; }
done
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/bench_b3.py --layers $LAYERS --rounds 2 --n 5 > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
agg = {}
for p in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv_wino" not in k:
            continue
        agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = {}
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv_wino" in k:
            dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, c in agg.items():
    d = sorted(dur.get(k, [0]))
    med = d[len(d) // 2]
    e = {"median_us": round(med, 1), "launches_timed": len(d)}
    for name, v in c.items():
        e[name] = sum(v) / len(v)
    if "FETCH_SIZE" in e:
        e["hbm_read_MB"] = round(e["FETCH_SIZE"] * 2 * 1024 / 1e6, 1)       # FETCH_SIZE is in KB; x 2: the gfx950 correction of MI355X_MICROARCH.md
    if "WRITE_SIZE" in e:
        e["hbm_write_MB"] = round(e["WRITE_SIZE"] * 1024 / 1e6, 1)
    if "GRBM_GUI_ACTIVE" in e and med:
        e["clock_GHz"] = round(e["GRBM_GUI_ACTIVE"] / 8 / (med * 1e3), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        e["mfma_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)      # busy cycles summed over 1024 SIMDs
    if "SQ_INSTS_VALU" in e and "SQ_INSTS_MFMA" in e:
        e["valu_per_mfma"] = round((e["SQ_INSTS_VALU"] - e["SQ_INSTS_MFMA"]) / e["SQ_INSTS_MFMA"], 3)
    res[k] = e
json.dump(res, open(os.path.join(out, os.environ.get("PMC_JSON", "r06_pmc_b3.json")), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
