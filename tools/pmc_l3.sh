#!/bin/bash
# Counter passes for the warp+correlation launches of one 1024x1024 PIV forward (level 3: the roofline kernel; level 1: the
# 395 MB launch beyond the Infinity Cache), each counter group in its own rocprofv3 --pmc run (never combined with other
# trace domains).  Run on the GPU box from the repo root:  bash tools/pmc_l3.sh   -> gpurun_out/pmc_l3/*.json
set -e
OUT=$PWD/gpurun_out/pmc_l3
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-level 0 --no-arithmetic"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$tag" -- $CMD > "$OUT/$tag.log" 2>&1
done
python3 tools/pmc_summary.py "$OUT" 3 > "$OUT/r03_pmc_l3_warp_corr.json"
python3 tools/pmc_summary.py "$OUT" 1 > "$OUT/r03_pmc_l1_warp_corr.json"
python3 tools/pmc_summary.py "$OUT" 3 8 > "$OUT/r03_pmc_l3b8_warp_corr.json"
cat "$OUT/r03_pmc_l3_warp_corr.json" "$OUT/r03_pmc_l1_warp_corr.json" "$OUT/r03_pmc_l3b8_warp_corr.json"
