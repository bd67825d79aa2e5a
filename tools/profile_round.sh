#!/bin/bash
# Everything profiles/r03_* is made from, in one go on the GPU box (from the repo root):  bash tools/profile_round.sh
# (rocprofv3 runs with the program itself after `--`; counters in their own --pmc passes, never combined with other trace domains)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r03
mkdir -p $R
# 1. kernel trace + stats of the default bench (fp32: the instruction + Winograd), per-level timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_fp32 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-arithmetic --lean > $R/prof_fp32.log 2>&1
python3 tools/stats_md.py $R/prof_fp32 13 > $R/kernel_stats_fp32.md
python3 tools/level_timeline.py $R/prof_fp32 > $R/timeline_fp32.txt
echo "stats done"
# 2. warp+correlation counter passes (level 3, level 1, batch-8 level 3)
bash tools/pmc_l3.sh > $R/pmc_l3.log 2>&1
cp gpurun_out/pmc_l3/r03_pmc_*.json $R/
echo "pmc_l3 done"
# 3. config #3 (32 x 512^2) counter passes on the level-3 warp+correlation
bash tools/pmc_config3.sh > $R/r03_pmc_config3.json 2> $R/pmc_config3.err || echo "pmc_config3 failed"
echo "config3 done"
# 4. Winograd kernel counters: busy / waits, then instruction mix
bash tools/pmc_wino.sh > $R/r03_pmc_wino_busy.json 2> $R/pmc_wino_busy.err
PMC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM" bash tools/pmc_wino.sh > $R/r03_pmc_wino_insts.json 2> $R/pmc_wino_insts.err
echo "pmc_wino done"
# 5. micro-benchmark: vector instructions beside the fp32 MFMA
tools/micro/mfma_valu_mix > $R/r03_mfma_valu_mix.log 2>&1 || true
# 6. the bench line itself (un-profiled), with the CPU baseline and the arithmetic table
python3 bench.py > $R/bench_default.json 2> $R/bench_default.err
echo "all done"
