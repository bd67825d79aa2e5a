#!/bin/bash
# Counter passes for the warp+correlation launches of one 1024x1024 PIV forward (level 3: the roofline kernel; level 1: the
# 395 MB launch beyond the Infinity Cache) and for the batch-8 level-3 shape, each counter group in its own rocprofv3 --pmc run
# (never combined with other trace domains).  Run on the GPU box from the repo root:  bash tools/pmc_l3.sh   -> gpurun_out/pmc_l3/*.json
# The forward's six warp+correlation dispatches come in the order of the levels (6, 5, 4, 3, 2, 1): a lean bench run has nothing but
# forwards, so dispatch k of the process belongs to level 6 - k % 6.  Batch 8: standalone launches of tools/wc_standalone.py.
set -e
R=${PMC_ROUND:-r05}
OUT=$PWD/gpurun_out/pmc_l3
mkdir -p "$OUT"
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/net_$tag" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-level 0 --no-arithmetic --lean > "$OUT/net_$tag.log" 2>&1
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/b8_$tag" -- python3 tools/wc_standalone.py --level 3 --batch 8 --launches 12 > "$OUT/b8_$tag.log" 2>&1
done
# kernel durations without counters: a plain --kernel-trace pass of the same two commands (the --pmc passes serialise the dispatches)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/net_trace" -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-level 0 --no-arithmetic --lean > "$OUT/net_trace.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/b8_trace" -- python3 tools/wc_standalone.py --level 3 --batch 8 --launches 40 > "$OUT/b8_trace.log" 2>&1
python3 tools/pmc_summary.py "$OUT" 3 > "$OUT/${R}_pmc_l3_warp_corr.json"
python3 tools/pmc_summary.py "$OUT" 1 > "$OUT/${R}_pmc_l1_warp_corr.json"
python3 tools/pmc_summary.py "$OUT" 3 8 > "$OUT/${R}_pmc_l3b8_warp_corr.json"
cat "$OUT/${R}_pmc_l3_warp_corr.json" "$OUT/${R}_pmc_l1_warp_corr.json" "$OUT/${R}_pmc_l3b8_warp_corr.json"
